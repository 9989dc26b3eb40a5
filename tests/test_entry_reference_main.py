"""CPU, build container only: the reference's OWN `trainer.test.main()` driven through `layout_dm_amd.test_entry.main` with
the model class swapped (Seam 1) — the branch of test_entry.py that runs when hydra and the
reference package are importable (VERDICT r4 missing #2 / next #3).

What is real here: the reference's main() (test.py:57-283: job_dir / config.yaml / best_model.pt handling, sampling-config
aggregation, result pickles), its LayoutSequenceTokenizer, its `get_cond` on batches collated from its own test-time
transforms, its `load_model`, its relation-violation metric; and the drop-in `layout_dm_amd.layoutdm.LayoutDM` with
everything above the C-ABI (cond plumbing, refinement prior from `seq_orig`, `batch_w_canvas` -> CSR, decode plan).
What is faked: hydra / omegaconf / torch_geometric (oracle/ref_harness.install_entry_stubs — absent from this image), the
processed dataset files (a synthetic dataset class of the same name), and the ENGINE below the C-ABI (no GPU here): a
recorder that fills every [MASK] with a valid token of its attribute.  The GPU half of the same claim is
tests/test_getcond_gpu.py (reference-produced cond dicts through the real engine against the reference's tokens).
"""
import os
import pickle
import sys

import numpy as np
import pytest
import torch
import yaml

from oracle import ref_harness as rh
from oracle import spec as SP

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")

SPEC = SP.RICO25
CALLS = []


class FakeEngine:
    """layout_dm_amd.binding.Engine's surface as diffusion.py / relation.py / layoutdm.py use it, on the CPU."""

    def __init__(self, *, n_category, n_bin=32, max_elem=25, n_attr=5, n_step=100, precision="exact", max_batch=512,
                 q_type="constrained", **_k):
        self.S, self.C, self.T = max_elem * n_attr, n_category + 4 * n_bin + 2, n_step
        self.n_attr, self.n_bin, self.n_category, self.q_type = n_attr, n_bin, n_category, q_type
        self.pad_id, self.mask_id = self.C - 2, self.C - 1
        self.max_batch, self.batch_round, self.precision = max_batch, 256, precision
        self.device = torch.device("cpu")
        self.loaded = None

    def load_state_dict(self, sd):
        self.loaded = sorted(sd)

    def _tok(self, t):
        return torch.as_tensor(t).to(torch.int32).contiguous()

    def make_relation(self, graph, centres, canvas_bins, relation_lambda, num_update, n_graph_total):
        from layout_dm_amd.relation import graph_to_csr

        off, src, dst, ea = graph_to_csr(graph, int(n_graph_total))
        plan = {"off": off, "src": src, "dst": dst, "attr": ea, "centres": np.asarray(centres), "canvas_bins": list(canvas_bins),
                "lambda": relation_lambda, "num_update": num_update}
        return plan, []

    def sample_loop(self, tokens, t_model, t_post, sampling_cfg, cond=None, seed=0, first_layout=0, intermediates=False,
                    use_graph=True, lc_keep=None, relation=None):
        tokens = self._tok(tokens)
        B = tokens.shape[0]
        CALLS.append({"B": B, "n_steps": len(t_model), "sampler": dict(sampling_cfg)["name"],
                      "cond": None if cond is None else {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in cond.items()},
                      "relation": relation[0] if relation else None, "start": tokens.clone()})
        out = tokens.clone()
        for a in range(self.n_attr):   # every [MASK] becomes a valid token of its attribute's sub-vocabulary
            lo = 0 if a == 0 else self.n_category + (a - 1) * self.n_bin
            col = out[:, a::self.n_attr]
            fill = (lo + (torch.arange(col.shape[1])[None] + first_layout + seed) % (self.n_category if a == 0 else self.n_bin)).int()
            out[:, a::self.n_attr] = torch.where(col == self.mask_id, fill.expand_as(col), col)
        if cond is not None and cond.get("mask") is not None:
            m = torch.as_tensor(cond["mask"]).bool()
            out[m] = self._tok(cond["seq"])[m]
        tokens.copy_(out)
        return tokens, None

    def decode(self, tokens, centres=None):
        E = self.S // self.n_attr
        t = tokens.long().view(-1, E, self.n_attr)
        label = t[..., 0]
        mask = label < self.n_category
        bins = (t[..., 1:] - self.n_category - torch.arange(4) * self.n_bin).clamp(0, self.n_bin - 1).float()
        bbox = torch.stack([bins[..., 0] / self.n_bin, bins[..., 1] / self.n_bin, (bins[..., 2] + 1) / self.n_bin,
                            (bins[..., 3] + 1) / self.n_bin], dim=-1)
        return {"bbox": bbox * mask[..., None], "label": label * mask, "mask": mask}

    def close(self):
        pass


TRAIN_CFG = {
    "model": {"_target_": "trainer.models.layoutdm.LayoutDM", "_partial_": True, "q_type": "constrained", "precision": "exact"},
    "backbone": {"_target_": "trainer.models.transformer_utils.TransformerEncoder",
                 "encoder_layer": {"_target_": "trainer.models.transformer_utils.Block", "d_model": 512, "nhead": 8,
                                   "dim_feedforward": 2048, "dropout": 0.0, "batch_first": True, "norm_first": True,
                                   "timestep_type": "adalayernorm", "diffusion_step": 100},
                 "num_layers": 4},
    "data": {"num_bin_bboxes": 32, "pad_until_max": True, "shared_bbox_vocab": "x-y-w-h", "bbox_quantization": "linear",
             "special_tokens": ["pad", "mask"], "var_order": "c-x-y-w-h", "transforms": ["RandomOrder"]},
    "dataset": {"_target_": "trainer.datasets.rico.Rico25Dataset", "_partial_": True, "dir": "???", "max_seq_length": 25},
}


@pytest.fixture()
def job(tmp_path, monkeypatch):
    rh.install_entry_stubs()
    import trainer.datasets.rico as rico
    import trainer.test as ref_test

    import layout_dm_amd.diffusion as D

    job_dir, result_dir = tmp_path / "job", tmp_path / "results"
    job_dir.mkdir()
    result_dir.mkdir()
    (job_dir / "config.yaml").write_text(yaml.safe_dump(TRAIN_CFG))
    torch.save({"model.module.transformer.cat_emb.weight": torch.zeros(2, 2)}, job_dir / "best_model.pt")
    synth_ds = type("Rico25Dataset", (rh.SynthLayoutDataset,), {"labels": rico.Rico25Dataset.labels})
    monkeypatch.setattr(rico, "Rico25Dataset", synth_ds)
    monkeypatch.setattr(ref_test, "save_image", lambda *a, **k: None)
    monkeypatch.setattr(D, "Engine", FakeEngine)
    monkeypatch.setattr(sys, "argv", list(sys.argv))
    monkeypatch.chdir(tmp_path)
    del CALLS[:]
    return str(job_dir), str(result_dir)


@pytest.mark.parametrize("cond", ["unconditional", "c", "cwh", "partial", "refinement", "relation"])
def test_reference_main_runs_with_the_dropin_class(job, cond, capsys):
    from layout_dm_amd import test_entry as TE

    job_dir, result_dir = job
    argv = [f"cond={cond}", f"job_dir={job_dir}", f"result_dir={result_dir}", "max_batch_size=4", "num_uncond_samples=6",
            "sampling=random"]
    TE.main(argv)
    import trainer.models.layoutdm as ref_layoutdm

    from layout_dm_amd.layoutdm import LayoutDM

    assert ref_layoutdm.LayoutDM is LayoutDM                        # the seam: hydra's _target_ now resolves to the drop-in
    printed = capsys.readouterr().out
    assert "ms per sample" in printed                                # test.py:257-258
    # two batches either way: 4 + 2 layouts
    assert [c["B"] for c in CALLS] == [4, 2] and all(c["n_steps"] == 100 and c["sampler"] == "random" for c in CALLS)
    out_dirs = os.listdir(result_dir)
    assert len(out_dirs) == 1 and out_dirs[0].startswith(cond + "_")
    data = pickle.load(open(os.path.join(result_dir, out_dirs[0], "seed_0.pkl"), "rb"))
    assert len(data["results"]) == 6 and data["test_cfg"]["cond"] == cond
    for bbox, label in data["results"]:
        assert bbox.shape == (len(label), 4) and len(label) >= 1
    if cond == "unconditional":
        assert all(c["cond"] is None and (c["start"] == SPEC.mask_id).all() for c in CALLS)
        return
    # the cond of every call is what the reference's get_cond built from the collated batch (task.py:27-151)
    A = SPEC.n_attr
    for c in CALLS:
        cd = c["cond"]
        seq, mask = cd["seq"].long(), cd["mask"].bool()
        assert cd["type"] == cond and torch.equal(c["start"].long(), seq)
        n_elem = (seq[:, ::A] != SPEC.pad_id).sum(1) if cond != "partial" else None
        if cond in ("c", "relation", "refinement"):
            assert (mask[:, ::A]).all() and (seq[:, 1::A][seq[:, ::A] != SPEC.pad_id] == SPEC.mask_id).all()
        if cond == "refinement":       # seq_orig -> the additive prior (task.py:204-224), (B, C, S), zero on the fixed tokens
            wl = cd["weak_logits"]
            assert wl.shape == (c["B"], SPEC.n_class, SPEC.seq_len) and float(wl.abs().max()) == 3.0
        if cond == "relation":         # batch_w_canvas -> CSR: per-layout edge lists with local node ids (canvas = 0)
            plan = c["relation"]
            assert plan is not None and plan["num_update"] == 3 and plan["lambda"] == 3e6
            off = plan["off"]
            assert len(off) == c["B"] + 1 and int(off[-1]) == len(plan["src"]) > 0
            for b in range(c["B"]):
                s, d = plan["src"][off[b]:off[b + 1]], plan["dst"][off[b]:off[b + 1]]
                if len(s):
                    assert int(max(s.max(), d.max())) <= int(n_elem[b]) and int(min(s.min(), d.min())) >= 0
    if cond in ("partial", "refinement"):
        assert len(data["inputs"]) == 6                               # test.py:215-228 decodes cond["seq"] / ["seq_orig"]
