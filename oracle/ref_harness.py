"""TEST INFRASTRUCTURE ONLY — loader for the *real* reference implementation.

This module imports CyberAgentAILab/layout-dm from ``/root/reference`` (read-only,
present only in the build container, NOT on the GPU box) with in-memory stubs for
the third-party modules the image lacks (hydra, omegaconf, torch_geometric,
torchvision, seaborn, prdc, pytorch_fid).  It is used by
``oracle/make_golden.py`` to generate the fixtures under ``tests/golden/`` and by
``tests/test_oracle_vs_reference.py`` (skipped when the reference is absent) to
pin ``oracle/restatement.py`` against the reference itself.

Nothing in the product path (``layout_dm_amd/``) may import this file.

Recipe follows SURVEY.md App. H.  Reference symbols used:
  trainer/models/categorical_diffusion/constrained.py:27  ConstrainedMaskAndReplaceDiffusion
  trainer/helpers/layout_tokenizer.py:196                  LayoutSequenceTokenizer
  trainer/models/base_model.py:108-116                     _init_weights
  trainer/models/common/util.py:36-44                      shrink
"""
from __future__ import annotations

import copy
import dataclasses
import importlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("LAYOUTDM_REFERENCE", "/root/reference")
_TRAINER_PATH = os.path.join(REFERENCE_ROOT, "src", "trainer")
# oracle/_ref/: the same package byte-compiled by oracle/build_ref.py (git-ignored build output that travels to the GPU
# box with the snapshot; no source text) — only the timing of the real reference (bench.py cpu_baseline) uses it
_SNAPSHOT_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref")


def reference_available() -> bool:
    """The reference TREE is present (build container): fixtures can be regenerated, sources inspected."""
    return os.path.isdir(os.path.join(_TRAINER_PATH, "trainer"))


def snapshot_available() -> bool:
    return os.path.exists(os.path.join(_SNAPSHOT_PATH, "trainer", "__init__.pyc"))


def reference_importable() -> bool:
    """The reference's `trainer` package can be imported: from the tree, or from the byte-compiled oracle/_ref/."""
    return reference_available() or snapshot_available()


# --------------------------------------------------------------------------- stubs
class DictConfig(dict):
    """Attribute-dict standing in for omegaconf.DictConfig."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __deepcopy__(self, memo):
        return DictConfig({k: copy.deepcopy(v, memo) for k, v in self.items()})


def to_cfg(obj):
    if dataclasses.is_dataclass(obj):
        if isinstance(obj, type):
            obj = obj()
        obj = dataclasses.asdict(obj)
    if isinstance(obj, dict):
        return DictConfig({k: to_cfg(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return [to_cfg(v) for v in obj]
    return obj


class _OmegaConf:
    @staticmethod
    def structured(obj):
        return to_cfg(obj)

    @staticmethod
    def create(obj=None):
        return to_cfg(obj or {})

    @staticmethod
    def to_container(obj, **_):
        return obj


def _instantiate(cfg, *args, **kwargs):
    """Minimal hydra.utils.instantiate: import _target_, recurse, call."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    cfg.pop("_partial_", None)
    mod, name = target.rsplit(".", 1)
    fn = getattr(importlib.import_module(mod), name)
    kw = {}
    for k, v in cfg.items():
        if isinstance(v, dict) and "_target_" in v:
            v = _instantiate(v)
        kw[k] = v
    kw.update(kwargs)
    return fn(*args, **kw)


class _Dummy:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return None


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__path__ = []  # behave as a package

    def _getattr(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return type(attr, (_Dummy,), {})

    m.__getattr__ = _getattr
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(sys.modules[parent], child, m)
    return m


# --- minimal, faithful stand-ins for the two torch_geometric.utils functions the relational constraint losses
# call (trainer/models/clg/const.py:5,12,29,76,121,155).  Semantics of PyG 2.x: nodes / edges are grouped by graph
# in order; duplicate (b, i, j) entries accumulate.
def to_dense_batch(x, batch=None, fill_value=0.0, max_num_nodes=None, batch_size=None):
    import torch

    if batch is None:
        batch = x.new_zeros(x.size(0), dtype=torch.long)
    B = int(batch.max()) + 1 if batch_size is None else batch_size
    num = torch.zeros(B, dtype=torch.long, device=x.device).scatter_add_(0, batch, torch.ones_like(batch))
    cum = torch.cat([num.new_zeros(1), num.cumsum(0)])
    N = int(num.max()) if max_num_nodes is None else max_num_nodes
    local = torch.arange(batch.numel(), device=x.device) - cum[batch]
    out = x.new_full((B, N) + tuple(x.shape[1:]), fill_value)
    mask = torch.zeros(B, N, dtype=torch.bool, device=x.device)
    out[batch, local] = x
    mask[batch, local] = True
    return out, mask


def to_dense_adj(edge_index, batch=None, edge_attr=None, max_num_nodes=None, batch_size=None):
    import torch

    if batch is None:
        n = int(edge_index.max()) + 1 if edge_index.numel() > 0 else 0
        batch = edge_index.new_zeros(n)
    B = int(batch.max()) + 1 if batch_size is None else batch_size
    num = torch.zeros(B, dtype=torch.long, device=batch.device).scatter_add_(0, batch, torch.ones_like(batch))
    cum = torch.cat([num.new_zeros(1), num.cumsum(0)])
    N = int(num.max()) if max_num_nodes is None else max_num_nodes
    eb = batch[edge_index[0]]
    i = edge_index[0] - cum[eb]
    j = edge_index[1] - cum[eb]
    if edge_attr is None:
        edge_attr = torch.ones(edge_index.size(1), device=edge_index.device)
    adj = edge_attr.new_zeros((B, N, N) + tuple(edge_attr.shape[1:]))
    flat = adj.view((B * N * N,) + tuple(edge_attr.shape[1:]))
    flat.index_add_(0, eb * N * N + i * N + j, edge_attr)
    return adj


class GraphBatch:
    """The attributes of a torch_geometric DataBatch that logit_adjustment.update / clg/const.py read
    (cond["batch_w_canvas"], helpers/task.py:112-114): y (labels, canvas = 0, others label+1: data/util.py:111-125),
    edge_index (2,E) global node ids, edge_attr (E,) relation bitmasks (data/util.py:128-177), batch (node -> graph)."""

    def __init__(self, y, edge_index, edge_attr, batch):
        self.y, self.edge_index, self.edge_attr, self.batch = y, edge_index, edge_attr, batch

    def to(self, *_a, **_k):
        return self


class Data:
    """One layout as the reference's datasets hand it to a transform (torch_geometric.data.Data): x (n,4) boxes, y (n,)
    labels, attr dict (datasets/rico.py: name, width, height, filtered, has_canvas_element, NoiseAdded)."""

    def __init__(self, x, y, attr=None):
        self.x, self.y = x, y
        self.attr = attr if attr is not None else {"has_canvas_element": False, "filtered": False, "NoiseAdded": False}
        self.edge_index, self.edge_attr = None, None


class DataBatch(GraphBatch):
    """What torch_geometric.loader.DataLoader's collate makes of a list of Data, reduced to the fields the reference's
    sampling path reads (helpers/task.py:27-151 get_cond, data/util.py:270-286 sparse_to_dense, clg/const.py): nodes and
    edges concatenated graph by graph, edge_index shifted by the node offsets, `batch` = node -> graph, dict attributes
    collated key by key (bools -> a bool tensor, as PyG's collate does for numbers)."""

    def __init__(self, datas):
        import torch

        off, eis, eas = 0, [], []
        for d in datas:
            if d.edge_index is not None:
                eis.append(d.edge_index.view(2, -1) + off)
                eas.append(d.edge_attr)
            off += d.x.size(0)
        ei = torch.cat(eis, dim=1) if eis else torch.zeros((2, 0), dtype=torch.long)
        ea = torch.cat(eas) if eas else torch.zeros(0, dtype=torch.long)
        super().__init__(torch.cat([d.y for d in datas]), ei, ea,
                         torch.cat([torch.full((d.x.size(0),), i, dtype=torch.long) for i, d in enumerate(datas)]))
        self.x = torch.cat([d.x for d in datas])
        self.attr = {}
        for k in datas[0].attr:
            vals = [d.attr[k].item() if hasattr(d.attr[k], "item") else d.attr[k] for d in datas]
            self.attr[k] = torch.tensor(vals) if all(isinstance(v, (bool, int, float)) for v in vals) else vals
        self.num_graphs = len(datas)


def synth_layout_batch(n_category: int, B: int, seed: int, relation: bool = False, edge_ratio: float = 0.1,
                       n_lo: int = 3, n_hi: int = 12, transform_seed: int = 0):
    """B random layouts collated like one batch of the reference's test DataLoader (test.py:169-180); with relation=True
    every layout goes through the reference's OWN test-time transforms first — AddCanvasElement + AddRelationConstraints
    (seed, edge_ratio as test.py:152-158; data/util.py:111-177)."""
    import torch

    install_stubs()
    g = torch.Generator().manual_seed(seed)
    tf = []
    if relation:
        from trainer.data.util import AddCanvasElement, AddRelationConstraints

        tf = [AddCanvasElement(), AddRelationConstraints(seed=transform_seed, edge_ratio=edge_ratio)]
    datas = []
    for _ in range(B):
        n = int(torch.randint(n_lo, n_hi + 1, (1,), generator=g))
        box = torch.rand(n, 4, generator=g) * torch.tensor([0.8, 0.8, 0.5, 0.5]) + torch.tensor([0.1, 0.1, 0.05, 0.05])
        lab = torch.randint(0, n_category, (n,), generator=g)
        d = Data(box, lab, {"has_canvas_element": torch.tensor([False]), "filtered": False, "NoiseAdded": False})
        for t in tf:
            d = t(d)
        datas.append(d)
    return DataBatch(datas)


_INSTALLED = False


def install_stubs():
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_importable():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT} (and no oracle/_ref/: python -m oracle.build_ref)")
    sys.dont_write_bytecode = True  # /root/reference must stay pristine
    path = _TRAINER_PATH if reference_available() else _SNAPSHOT_PATH
    if path not in sys.path:
        sys.path.insert(0, path)

    def need(name):
        try:
            importlib.import_module(name)
            return False
        except Exception:
            return True

    if need("omegaconf"):
        _stub("omegaconf", DictConfig=DictConfig, OmegaConf=_OmegaConf)
    if need("hydra"):
        _stub("hydra")
        _stub("hydra.utils", instantiate=_instantiate)
        _stub("hydra.core")
        _stub("hydra.core.config_store")
    if need("torch_geometric"):
        _stub("torch_geometric")
        _stub("torch_geometric.utils", to_dense_batch=to_dense_batch, to_dense_adj=to_dense_adj)
        for sub in ("data", "loader"):
            _stub(f"torch_geometric.{sub}")
        for sub in ("collate", "dataset", "makedirs", "separate"):
            _stub(f"torch_geometric.data.{sub}")
    if need("torchvision"):
        _stub("torchvision")
        _stub("torchvision.transforms")
        _stub("torchvision.utils")
    for name in ("seaborn", "prdc", "pytorch_fid"):
        if need(name):
            _stub(name)
    if "pytorch_fid.fid_score" not in sys.modules and need("pytorch_fid.fid_score"):
        _stub("pytorch_fid.fid_score")
    _INSTALLED = True


# --------------------------------------------------------------------------- the `test` entry point's own imports
class SynthLayoutDataset:
    """Stands in for the reference's processed-dataset classes (datasets/base.py:19-34 load <dir>/<name>/processed/<split>.pt,
    absent offline): `n` random layouts; __getitem__ applies the transform like torch_geometric's Dataset does."""
    n, seed = 6, 0

    def __init__(self, dir=None, split="test", max_seq_length=25, transform=None, n_category=25):
        import torch

        g = torch.Generator().manual_seed(self.seed)
        self.transform, self.items = transform, []
        for _ in range(self.n):
            k = int(torch.randint(3, 10, (1,), generator=g))
            box = torch.rand(k, 4, generator=g) * torch.tensor([0.8, 0.8, 0.5, 0.5]) + torch.tensor([0.1, 0.1, 0.05, 0.05])
            self.items.append((box, torch.randint(0, n_category, (k,), generator=g)))
        self.colors = [(0, 0, 0)] * n_category

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        import torch

        box, lab = self.items[i]
        d = Data(box.clone(), lab.clone(), {"has_canvas_element": torch.tensor([False]), "filtered": False, "NoiseAdded": False})
        return self.transform(d) if self.transform is not None else d


class _Loader:
    """torch_geometric.loader.DataLoader(dataset, batch_size, shuffle=False) as test.py:176-180 uses it."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **_k):
        assert not shuffle
        self.dataset, self.batch_size = dataset, batch_size

    def __len__(self):
        return -(-len(self.dataset) // self.batch_size)

    def __iter__(self):
        for i in range(0, len(self.dataset), self.batch_size):
            yield DataBatch([self.dataset[j] for j in range(i, min(i + self.batch_size, len(self.dataset)))])


def install_entry_stubs():
    """What `import trainer.test` and its main() need beyond install_stubs(): hydra.main (key=value overrides of the
    structured TestConfig), the ConfigStore, OmegaConf.load / set_struct, instantiate with _partial_, the PyG DataLoader,
    torchvision's Compose.  Only where the real packages are absent (this image)."""
    import functools

    install_stubs()
    store = {}

    class ConfigStore:
        @staticmethod
        def instance():
            return ConfigStore

        @staticmethod
        def store(name, node, **_k):
            store[name] = node

    def hydra_main(version_base=None, config_name=None, config_path=None):
        def deco(fn):
            @functools.wraps(fn)
            def run():
                node = store[config_name]
                fields = {f.name: f for f in dataclasses.fields(node)}
                kw = {}
                for arg in sys.argv[1:]:
                    k, v = arg.split("=", 1)
                    k = k.lstrip("+")
                    typ = fields[k].type
                    if typ in (bool, "bool"):
                        v = v.lower() in ("true", "1")
                    elif typ in (int, "int"):
                        v = int(v)
                    elif typ in (float, "float"):
                        v = float(v)
                    kw[k] = v
                return fn(to_cfg(node(**kw)))
            return run
        return deco

    def instantiate(cfg, *args, **kwargs):
        if isinstance(cfg, dict) and cfg.get("_partial_"):
            c = dict(cfg)
            c.pop("_partial_")
            target = c.pop("_target_")
            mod, name = target.rsplit(".", 1)
            return functools.partial(getattr(importlib.import_module(mod), name), **c)
        return _instantiate(cfg, *args, **kwargs)

    def load(file_obj):
        import yaml

        return to_cfg(yaml.safe_load(file_obj))

    hy = sys.modules["hydra"]
    if getattr(hy, "__file__", None) is None:       # (our stub, not the real package)
        hy.main = hydra_main
        sys.modules["hydra.utils"].instantiate = instantiate
        sys.modules["hydra.core.config_store"].ConfigStore = ConfigStore
    oc = sys.modules["omegaconf"]
    if getattr(oc, "__file__", None) is None:
        _OmegaConf.load = staticmethod(load)
        _OmegaConf.set_struct = staticmethod(lambda cfg, flag: None)
    tg = sys.modules["torch_geometric.loader"]
    if getattr(tg, "__file__", None) is None:
        tg.DataLoader = _Loader
    tv = sys.modules["torchvision.transforms"]
    if getattr(tv, "__file__", None) is None:
        class Compose:
            def __init__(self, ts):
                self.ts = ts

            def __call__(self, d):
                for t in self.ts:
                    d = t(d)
                return d
        tv.Compose = Compose


# --------------------------------------------------------------------------- configs
DATASETS = {
    "rico25": "trainer.datasets.rico.Rico25Dataset",
    "publaynet": "trainer.datasets.publaynet.PubLayNetDataset",
}


def make_cfgs(dataset: str = "rico25", n_step: int = 100):
    data_cfg = DictConfig(
        num_bin_bboxes=32,
        pad_until_max=True,
        shared_bbox_vocab="x-y-w-h",
        bbox_quantization="linear",
        special_tokens=["pad", "mask"],
        var_order="c-x-y-w-h",
    )
    dataset_cfg = DictConfig(_target_=DATASETS[dataset], max_seq_length=25)
    backbone_cfg = DictConfig(
        _target_="trainer.models.transformer_utils.TransformerEncoder",
        encoder_layer=DictConfig(
            _target_="trainer.models.transformer_utils.Block",
            d_model=512,
            nhead=8,
            dim_feedforward=2048,
            dropout=0.0,
            batch_first=True,
            norm_first=True,
            timestep_type="adalayernorm",
            diffusion_step=n_step,
        ),
        num_layers=4,
    )
    return data_cfg, dataset_cfg, backbone_cfg


def build_reference_model(dataset: str = "rico25", seed: int = 0, perturb: bool = False,
                          q_type: str = "constrained", n_step: int = 100):
    """Instantiate the reference diffusion module with the reference's own init.

    perturb=True additionally randomises every bias / LayerNorm affine parameter so
    that no term of the forward pass is trivially 0 or 1 (used for golden vectors).
    Returns (module, tokenizer).
    """
    install_stubs()
    import torch
    from trainer.helpers.layout_tokenizer import LayoutSequenceTokenizer
    from trainer.models.base_model import BaseModel
    from trainer.models.categorical_diffusion.constrained import (
        ConstrainedMaskAndReplaceDiffusion,
    )
    from trainer.models.common.util import shrink

    data_cfg, dataset_cfg, backbone_cfg = make_cfgs(dataset, n_step)
    torch.manual_seed(seed)
    tok = LayoutSequenceTokenizer(data_cfg, dataset_cfg)
    cls = ConstrainedMaskAndReplaceDiffusion
    if q_type == "vanilla":  # Q_TYPES, models/layoutdm.py:20-23
        from trainer.models.categorical_diffusion.vanilla import VanillaMaskAndReplaceDiffusion as cls
    m = cls(
        backbone_cfg=shrink(backbone_cfg, 29 / 32),
        num_classes=tok.N_total,
        max_token_length=tok.max_token_length,
        num_timesteps=n_step,  # BASELINE config 5 samples with T = 200: base.py:310-311 needs a T >= 200 model
        pos_emb="elem_attr",
        transformer_type="flattened",
        auxiliary_loss_weight=0.1,
        tokenizer=tok,
    )
    m.apply(lambda mod: BaseModel._init_weights(None, mod))
    if perturb:
        g = torch.Generator().manual_seed(seed + 1234)
        with torch.no_grad():
            for name, p in m.named_parameters():
                if p.ndim == 1:
                    p.add_(0.1 * torch.randn(p.shape, generator=g))
    m.eval()
    return m, tok


def sampling_cfg(name="deterministic", **kw):
    cfg = DictConfig(name=name, temperature=1.0, num_timesteps=100)
    cfg.update(kw)
    return cfg


def state_dict_layoutdm_keys(m):
    """Reference checkpoint key layout: LayoutDM wraps the diffusion module in
    CustomDataParallel => every key is prefixed 'model.module.' (layoutdm.py:52)."""
    return {f"model.module.{k}": v.detach().clone() for k, v in m.state_dict().items()}
