"""ctypes binding of libldm_hip.so (include/ldm_hip.h) — the only way Python reaches the HIP path.

PyTorch is used for device memory and streams only: tensors are handed over as raw
`data_ptr()`s together with the current HIP stream handle.  There is no CPU / eager fallback:
if the shared library or a GPU is missing every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libldm_hip.so")
ABI_VERSION = 5

PREC_EXACT_F32, PREC_FAST_F16, PREC_SPLIT_F16, PREC_MIXED_F16, PREC_HYBRID_F16 = 0, 1, 2, 3, 4
PRECISIONS = {"exact": PREC_EXACT_F32, "fast": PREC_FAST_F16, "split": PREC_SPLIT_F16, "mixed": PREC_MIXED_F16, "hybrid": PREC_HYBRID_F16,
              "f32": PREC_EXACT_F32, "f16": PREC_FAST_F16, "f16x3": PREC_SPLIT_F16, "f16x2": PREC_MIXED_F16}
SAMPLERS = {"deterministic": 0, "random": 1, "top_p": 2, "top_k": 3, "gumbel": 4}

EXPORTS = (
    "ldm_create", "ldm_destroy", "ldm_last_error", "ldm_load_weight", "ldm_finalize_weights",
    "ldm_denoise_logits", "ldm_posterior", "ldm_sample_tokens", "ldm_sample_step", "ldm_sample_loop",
    "ldm_decode_layouts", "ldm_relation_update", "ldm_set_tie_report", "ldm_get_tie_flags",
    "ldm_last_loop_ms", "ldm_set_profiling", "ldm_profile_count", "ldm_profile_get", "ldm_profile_reset",
    "ldm_abi_version", "ldm_get_layout", "ldm_describe",
    # FID feature extractor (bound in layout_dm_amd/fid.py)
    "ldm_fid_create", "ldm_fid_destroy", "ldm_fid_last_error", "ldm_fid_load_weight", "ldm_fid_finalize",
    "ldm_fid_features", "ldm_prdc", "ldm_layout_metrics",
)


class LdmConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "n_category", "n_bin", "max_elem", "n_attr", "d_model", "n_head", "d_ff",
        "n_layer", "n_step", "precision", "max_batch", "chunk", "q_type", "lanes")]


Q_TYPES = {"constrained": 0, "vanilla": 1}  # models/layoutdm.py:20-23


class LdmRelation(C.Structure):  # include/ldm_hip.h: ldm_relation
    _fields_ = [("d_edge_offsets", C.c_void_p), ("d_edge_src", C.c_void_p), ("d_edge_dst", C.c_void_p),
                ("d_edge_attr", C.c_void_p), ("d_centres", C.c_void_p), ("canvas_bins", C.c_int32 * 4),
                ("relation_lambda", C.c_float), ("num_update", C.c_int32), ("n_graph_total", C.c_int32)]


class LdmSampler(C.Structure):
    _fields_ = [("kind", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32)]


class LdmCond(C.Structure):
    _fields_ = [("d_cond_seq", C.c_void_p), ("d_strong_mask", C.c_void_p), ("d_weak_logits", C.c_void_p),
                ("pad_disable", C.c_int32)]


_lib = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libldm_hip.so and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("LDM_HIP_LIB") or LIB_PATH  # LDM_HIP_LIB: dev A/B of two builds on one GPU box
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: the MI355X HIP extension is not built (run `python -m layout_dm_amd.build` "
            "or __graft_entry__.build()). There is no CPU fallback.")
    if path is None and not os.environ.get("LDM_HIP_LIB"):
        # build provenance: the prebuilt library must come from THIS tree's sources (a snapshot pushed to a GPU box carries both)
        from . import build as _build

        want, have = _build.source_digest(), _build.built_digest(p)
        if have != want:
            #  one builder at a time: under torchrun every rank arrives here at once, and the ranks that wait find the
            # library rebuilt when they get the lock; the link goes to a temporary name and is renamed into place (build.build)
            import fcntl
            import sys

            os.makedirs(os.path.join(os.path.dirname(p), "build"), exist_ok=True)
            with open(os.path.join(os.path.dirname(p), "build", ".lock"), "w") as lock:
                fcntl.flock(lock, fcntl.LOCK_EX)
                try:
                    if _build.built_digest(p) != want:
                        print(f"layout_dm_amd: {os.path.basename(p)} is stale against this tree's sources - rebuilding with hipcc "
                              "(a minute or two; python -m layout_dm_amd.build does the same ahead of time)", file=sys.stderr, flush=True)
                        _build.build(verbose=False)
                except Exception as e:  # no hipcc on this machine
                    raise RuntimeError(f"{p} was built from other sources (library {have}, tree {want}) and could not be rebuilt: {e}") from e
                finally:
                    fcntl.flock(lock, fcntl.LOCK_UN)
    lib = C.CDLL(p)
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    lib.ldm_abi_version.restype = C.c_int
    lib.ldm_create.argtypes = [C.POINTER(LdmConfig), i32, C.POINTER(vp)]
    lib.ldm_destroy.argtypes = [vp]
    lib.ldm_destroy.restype = None
    lib.ldm_last_error.argtypes = [vp]
    lib.ldm_last_error.restype = C.c_char_p
    lib.ldm_load_weight.argtypes = [vp, C.c_char_p, vp, C.POINTER(C.c_int64), i32]
    lib.ldm_finalize_weights.argtypes = [vp]
    lib.ldm_denoise_logits.argtypes = [vp, vp, i32, i32, vp, vp]
    lib.ldm_posterior.argtypes = [vp, vp, vp, i32, i32, C.POINTER(LdmCond), vp, vp]
    lib.ldm_sample_tokens.argtypes = [vp, vp, C.POINTER(LdmCond), C.POINTER(LdmSampler), u64, u64, i32, i32, vp, vp]
    lib.ldm_sample_step.argtypes = [vp, vp, vp, i32, i32, C.POINTER(LdmCond), C.POINTER(LdmRelation),
                                    C.POINTER(LdmSampler), u64, u64, i32, i32, vp]
    lib.ldm_sample_loop.argtypes = [vp, vp, C.POINTER(LdmCond), C.POINTER(LdmRelation), C.POINTER(C.c_int32),
                                    C.POINTER(C.c_int32), i32, C.POINTER(LdmSampler), u64, u64, i32, vp, i32, vp]
    lib.ldm_decode_layouts.argtypes = [vp, vp, i32, vp, i32, vp, vp, vp, vp]
    lib.ldm_relation_update.argtypes = [vp, vp, vp, C.POINTER(LdmRelation), i32, i32, vp]
    lib.ldm_set_tie_report.argtypes = [vp, C.c_float, C.c_float]
    lib.ldm_get_tie_flags.argtypes = [vp, vp, i32, i32, vp]
    lib.ldm_last_loop_ms.argtypes = [vp, C.POINTER(C.c_float)]
    lib.ldm_set_profiling.argtypes = [vp, i32]
    lib.ldm_profile_count.argtypes = [vp]
    lib.ldm_profile_get.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ldm_profile_reset.argtypes = [vp]
    lib.ldm_get_layout.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ldm_describe.argtypes = [vp, C.c_char_p, i32]
    lib.ldm_layout_metrics.argtypes = [vp, vp, i32, i32, vp, vp]
    for name in EXPORTS:
        if name not in ("ldm_destroy", "ldm_last_error") and not name.startswith("ldm_fid_"):
            getattr(lib, name).restype = C.c_int
    if lib.ldm_abi_version() != ABI_VERSION:
        raise RuntimeError("libldm_hip.so ABI version mismatch — rebuild the extension")
    if path is None:
        _lib = lib
    return lib


def make_sampler(cfg) -> LdmSampler:
    """sampling_cfg (DictConfig / dict / attr object with the reference's field names,
    helpers/sampling.py:13-59) -> ldm_sampler."""
    get = (lambda k, d=None: cfg.get(k, d)) if hasattr(cfg, "get") else (lambda k, d=None: getattr(cfg, k, d))
    name = get("name")
    if name not in SAMPLERS:
        raise NotImplementedError(f"sampling '{name}'")  # sampling.py:117-118
    return LdmSampler(SAMPLERS[name], float(get("temperature", 1.0)), float(get("top_p", 1.0) or 1.0),
                      int(get("top_k", 1) or 1))


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class Engine:
    """One libldm_hip handle = one model replica on one GPU."""

    def __init__(self, *, n_category: int, n_bin: int = 32, max_elem: int = 25, n_attr: int = 5,
                 d_model: int = 464, n_head: int = 8, d_ff: int = 1856, n_layer: int = 4, n_step: int = 100,
                 precision="exact", max_batch: int = 512, chunk: int = 0, device: Optional[int] = None,
                 q_type: str = "constrained", lanes: int = 0):
        if q_type not in Q_TYPES:
            raise NotImplementedError(f"q_type={q_type}: one of {sorted(Q_TYPES)}")
        if not torch.cuda.is_available():
            raise RuntimeError("layout_dm_amd needs a ROCm GPU (MI355X); there is no CPU path")
        self.lib = load_library()
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        prec = PRECISIONS[precision] if isinstance(precision, str) else int(precision)
        self.cfg = LdmConfig(ABI_VERSION, n_category, n_bin, max_elem, n_attr, d_model, n_head, d_ff, n_layer,
                             n_step, prec, max_batch, chunk, Q_TYPES[q_type], lanes)
        self.q_type = q_type
        self.S = max_elem * n_attr
        self.n_attr, self.n_bin, self.n_category = n_attr, n_bin, n_category
        self.C = n_category + 4 * n_bin + 2
        self.T = n_step
        self.pad_id, self.mask_id = self.C - 2, self.C - 1
        self.max_batch = max_batch
        h = C.c_void_p()
        rc = self.lib.ldm_create(C.byref(self.cfg), self.device_index, C.byref(h))
        if rc != 0:
            raise RuntimeError(f"ldm_create failed ({rc}): {self.lib.ldm_last_error(None).decode()}")
        self._h = h
        self._keep = {}  # tensors referenced by cached graphs must stay alive
        ck, ln = C.c_int(), C.c_int()
        self._check(self.lib.ldm_get_layout(self._h, C.byref(ck), C.byref(ln)), "ldm_get_layout")
        self.chunk, self.lanes = int(ck.value), int(ln.value)

    def close(self):
        if getattr(self, "_h", None):
            self.lib.ldm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self.lib.ldm_last_error(self._h).decode()}")

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict: Dict[str, "torch.Tensor | np.ndarray"]):
        """Reference checkpoint (keys of SURVEY App. C, with or without 'model.module.')."""
        for k, v in state_dict.items():
            if isinstance(v, torch.Tensor):
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            self._check(self.lib.ldm_load_weight(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                        f"ldm_load_weight({k})")
        self._check(self.lib.ldm_finalize_weights(self._h), "ldm_finalize_weights")
        self._round = None

    @property
    def batch_round(self) -> int:
        """Layouts per "round" of a sampling call (ldm_describe `round`): the one-launch loop runs one workgroup per layout
        on the device's compute units (256 on the MI355X), the per-step path `chunk` layouts on each of its lanes — a call
        costs whole rounds, so B = round + 1 costs what B = 2 * round costs (bench.py `batch_shapes`)."""
        if getattr(self, "_round", None) is None:
            self._round = max(1, int(self.describe().get("round", "256")))
        return self._round

    # ------------------------------------------------------------------ helpers
    def _tok(self, t: torch.Tensor) -> torch.Tensor:
        if t.device != self.device or t.dtype != torch.int32 or not t.is_contiguous():
            t = t.to(device=self.device, dtype=torch.int32).contiguous()
        return t

    def make_cond(self, cond: Optional[dict], B: int):
        """cond dict of the reference (base.py:243-284 consumers) -> (ldm_cond, keep-alive list)."""
        if not cond:
            return None, []
        keep = []
        lc = LdmCond()
        seq = self._tok(torch.as_tensor(cond["seq"]))
        assert seq.shape == (B, self.S), f"cond['seq'] must be ({B},{self.S})"
        keep.append(seq)
        lc.d_cond_seq = seq.data_ptr()
        if "mask" in cond and cond["mask"] is not None:
            m = torch.as_tensor(cond["mask"]).to(device=self.device, dtype=torch.uint8).contiguous()
            keep.append(m)
            lc.d_strong_mask = m.data_ptr()
        if cond.get("type") == "refinement":
            wl = torch.as_tensor(cond["weak_logits"]).to(device=self.device, dtype=torch.float32).contiguous()
            assert wl.shape == (B, self.C, self.S)
            # weak_mask is by construction ~mask broadcast over classes (helpers/task.py:216)
            keep.append(wl)
            lc.d_weak_logits = wl.data_ptr()
        lc.pad_disable = 1 if cond.get("type") in ("c", "cwh", "refinement", "relation") else 0
        return lc, keep

    # ------------------------------------------------------------------ parity hooks
    def denoise_logits(self, tokens: torch.Tensor, t: int) -> torch.Tensor:
        tokens = self._tok(tokens)
        B = tokens.shape[0]
        out = torch.empty((B, self.S, self.C), dtype=torch.float32, device=self.device)
        self._check(self.lib.ldm_denoise_logits(self._h, tokens.data_ptr(), int(t), B, out.data_ptr(),
                                                _stream_ptr(self.device)), "ldm_denoise_logits")
        return out

    def posterior(self, logits: torch.Tensor, tokens: torch.Tensor, t_post: int, cond: Optional[dict] = None):
        tokens = self._tok(tokens)
        B = tokens.shape[0]
        logits = logits.to(device=self.device, dtype=torch.float32).contiguous()
        lc, keep = self.make_cond(cond, B)
        out = torch.empty((B, self.C, self.S), dtype=torch.float32, device=self.device)
        self._check(self.lib.ldm_posterior(self._h, logits.data_ptr(), tokens.data_ptr(), int(t_post), B,
                                           C.byref(lc) if lc else None, out.data_ptr(), _stream_ptr(self.device)),
                    "ldm_posterior")
        torch.cuda.current_stream(self.device).synchronize() if keep else None
        return out

    def sample_tokens(self, logp: torch.Tensor, sampling_cfg, seed: int = 0, first_layout: int = 0, step: int = 0,
                      cond: Optional[dict] = None):
        """helpers/sampling.py:81-130 on (B,C,S) log-probabilities.  cond (optional): {"seq", "type"} — the [PAD]
        disabling of base.py:272-284 for cond types c / cwh / refinement / relation."""
        logp = logp.to(device=self.device, dtype=torch.float32).contiguous()
        B = logp.shape[0]
        s = make_sampler(sampling_cfg)
        lc, keep = self.make_cond({"seq": cond["seq"], "type": cond.get("type")}, B) if cond else (None, [])
        out = torch.empty((B, self.S), dtype=torch.int32, device=self.device)
        self._check(self.lib.ldm_sample_tokens(self._h, logp.data_ptr(), C.byref(lc) if lc else None, C.byref(s), seed,
                                               first_layout, step, B, out.data_ptr(), _stream_ptr(self.device)),
                    "ldm_sample_tokens")
        if keep:
            torch.cuda.current_stream(self.device).synchronize()
        return out

    # ------------------------------------------------------------------ hot path
    def sample_step(self, tokens: torch.Tensor, t_model: int, sampling_cfg, t_post: Optional[int] = None,
                    cond: Optional[dict] = None, seed: int = 0, first_layout: int = 0, step: int = 0, relation=None):
        """relation: (LdmRelation, keep-alives) from make_relation for cond["type"] == "relation"."""
        tokens = self._tok(tokens)
        B = tokens.shape[0]
        s = make_sampler(sampling_cfg)
        lc, keep = self.make_cond(cond, B)
        out = torch.empty_like(tokens)
        self._check(self.lib.ldm_sample_step(self._h, tokens.data_ptr(), out.data_ptr(), int(t_model),
                                             int(t_model if t_post is None else t_post),
                                             C.byref(lc) if lc else None, C.byref(relation[0]) if relation else None,
                                             C.byref(s), seed, first_layout, step, B,
                                             _stream_ptr(self.device)), "ldm_sample_step")
        if keep:
            torch.cuda.current_stream(self.device).synchronize()
        return out

    def sample_loop(self, tokens: torch.Tensor, t_model: Sequence[int], t_post: Sequence[int], sampling_cfg,
                    cond: Optional[dict] = None, seed: int = 0, first_layout: int = 0,
                    intermediates: bool = False, use_graph: bool = True, lc_keep=None, relation=None):
        """In-place T-step loop on `tokens` (B,S) int32 cuda. Returns (tokens, intermediates|None).
        relation: (LdmRelation, keep-alives) from make_relation — the whole cond=relation loop (posterior ->
        logit adjustment -> [PAD] disable -> draw per step) then runs inside the same launch sequence / hipGraph."""
        tokens = self._tok(tokens)
        B = tokens.shape[0]
        n = len(t_model)
        s = make_sampler(sampling_cfg)
        if lc_keep is not None:
            lc, keep = lc_keep
        else:
            lc, keep = self.make_cond(cond, B)
        inter = torch.empty((n, B, self.S), dtype=torch.int32, device=self.device) if intermediates else None
        tm = (C.c_int32 * n)(*[int(x) for x in t_model])
        tp = (C.c_int32 * n)(*[int(x) for x in t_post])
        self._check(self.lib.ldm_sample_loop(self._h, tokens.data_ptr(), C.byref(lc) if lc else None,
                                             C.byref(relation[0]) if relation else None, tm, tp, n,
                                             C.byref(s), seed, first_layout, B,
                                             inter.data_ptr() if inter is not None else None,
                                             1 if use_graph else 0, _stream_ptr(self.device)), "ldm_sample_loop")
        if keep and lc_keep is None:
            torch.cuda.current_stream(self.device).synchronize()
        return tokens, inter

    # ------------------------------------------------------------------ near-tie report (deterministic decoding)
    def set_tie_report(self, tie_rel: float, tie_abs: float = 0.0):
        """Deterministic steps / loops mark, per (step, layout), whether some token was decided with a lead over the
        runner-up below max(tie_rel * max |logit of the token|, tie_abs) (include/ldm_hip.h); 0, 0 disables."""
        self._check(self.lib.ldm_set_tie_report(self._h, float(tie_rel), float(tie_abs)), "ldm_set_tie_report")

    def tie_flags(self, n_steps: int, B: int) -> torch.Tensor:
        """(n_steps, B) uint8 flags of the most recent deterministic call."""
        out = torch.empty((n_steps, B), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ldm_get_tie_flags(self._h, out.data_ptr(), int(n_steps), int(B), _stream_ptr(self.device)),
                    "ldm_get_tie_flags")
        return out

    # ------------------------------------------------------------------ cond=relation
    def make_relation(self, graph, centres, canvas_bins, relation_lambda: float, num_update: int, n_graph_total: int):
        """Device-side description of cond["batch_w_canvas"] for ldm_relation_update.  graph: object / dict with
        y (nodes,), edge_index (2,E) global node ids, edge_attr (E,), batch (nodes,) — torch_geometric DataBatch
        fields (helpers/task.py:112-114).  Returns (LdmRelation, keep-alive tensors)."""
        from .relation import graph_to_csr

        B = int(n_graph_total)
        off, src_l, dst_l, ea = graph_to_csr(graph, B)
        dev = self.device
        keep = [off.to(dev), src_l.to(dev), dst_l.to(dev), ea.to(dev),
                torch.as_tensor(centres, dtype=torch.float64).float().reshape(4, self.n_bin).contiguous().to(dev)]
        rel = LdmRelation(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(),
                          keep[4].data_ptr(), (C.c_int32 * 4)(*[int(x) for x in canvas_bins]),
                          float(relation_lambda), int(num_update), B)
        return rel, keep

    def relation_update(self, logp: torch.Tensor, cond_seq: torch.Tensor, rel, t: int, layout_offset: int = 0):
        """In-place logit adjustment of `logp` (B,C,S) float32 cuda (logit_adjustment.update, l.88-126)."""
        assert logp.is_cuda and logp.dtype == torch.float32 and logp.is_contiguous()
        lr, keep = rel
        cond_seq = self._tok(cond_seq)
        assert layout_offset == 0, "relation graphs are indexed from the first layout of the call"
        self._check(self.lib.ldm_relation_update(self._h, logp.data_ptr(), cond_seq.data_ptr(), C.byref(lr), int(t),
                                                 logp.shape[0], _stream_ptr(self.device)), "ldm_relation_update")
        torch.cuda.current_stream(self.device).synchronize()  # cond_seq copy / keep-alives may be temporaries
        return logp

    # ------------------------------------------------------------------ result packaging
    def decode(self, tokens: torch.Tensor, centres: Optional[torch.Tensor] = None):
        """ids (B,S) -> {"bbox" (B,E,4), "label" (B,E) int64, "mask" (B,E) bool} on the device
        (LayoutSequenceTokenizer.decode + BboxTokenizer.decode, layout_tokenizer.py:255-266 /
        bbox_tokenizer.py:117-168).  centres: None = linear bins (float32 boxes); (4,n_bin) float64
        cluster centres in x,y,w,h order = kmeans/percentile (float64 boxes, like the reference)."""
        tokens = self._tok(tokens)
        B = tokens.shape[0]
        E = self.S // self.n_attr
        f64 = centres is not None
        if f64:
            centres = torch.as_tensor(centres, dtype=torch.float64).to(self.device).contiguous()
            assert centres.numel() == 4 * self.n_bin, "centres must be (4, n_bin)"
        bbox = torch.empty((B, E, 4), dtype=torch.float64 if f64 else torch.float32, device=self.device)
        label = torch.empty((B, E), dtype=torch.int64, device=self.device)
        mask = torch.empty((B, E), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ldm_decode_layouts(self._h, tokens.data_ptr(), B, centres.data_ptr() if f64 else None,
                                                1 if f64 else 0, bbox.data_ptr(), label.data_ptr(),
                                                mask.data_ptr(), _stream_ptr(self.device)), "ldm_decode_layouts")
        if f64:
            torch.cuda.current_stream(self.device).synchronize()  # `centres` must outlive the kernel
        return {"bbox": bbox, "label": label, "mask": mask.bool()}

    # ------------------------------------------------------------------ introspection
    def describe(self) -> Dict[str, str]:
        """What this handle runs (ldm_describe): numerics mode, kernel family, loop structure, chunk / lanes, near-tie
        thresholds and the development knobs the LIBRARY honoured (LDM_DEV=1 only) — not what os.environ says."""
        cap = 1024
        while True:
            buf = C.create_string_buffer(cap)
            n = self.lib.ldm_describe(self._h, buf, cap)
            if n < 0:
                raise RuntimeError("ldm_describe failed")
            if n < cap:
                break
            cap = n + 1   # (returns the length needed: the honoured-knob list is unbounded)
        out = {}
        for kv in buf.value.decode().split(";"):
            k, _, v = kv.partition("=")
            if k == "knobs":
                out[k] = buf.value.decode().split("knobs=", 1)[1]
                break
            out[k] = v
        return out

    def last_loop_ms(self) -> float:
        ms = C.c_float()
        self._check(self.lib.ldm_last_loop_ms(self._h, C.byref(ms)), "ldm_last_loop_ms")
        return float(ms.value)

    def set_profiling(self, on: bool):
        self._check(self.lib.ldm_set_profiling(self._h, 1 if on else 0), "ldm_set_profiling")

    def profile(self, reset: bool = False):
        n = self.lib.ldm_profile_count(self._h)
        rows = []
        for i in range(n):
            name, ms, cnt, fl, by = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            self.lib.ldm_profile_get(self._h, i, C.byref(name), C.byref(ms), C.byref(cnt), C.byref(fl), C.byref(by))
            rows.append({"name": name.value.decode(), "ms": ms.value, "launches": cnt.value, "flops": fl.value,
                         "bytes": by.value})
        if reset:
            self.lib.ldm_profile_reset(self._h)
        return rows
