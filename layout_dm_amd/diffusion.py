"""Host-side mirror of the reference's diffusion class for the sampling path.

`HipMaskAndReplaceDiffusion` exposes the public surface the reference's callers use on
`ConstrainedMaskAndReplaceDiffusion` (trainer/models/categorical_diffusion/constrained.py:27,
base.py:293-371): construct from the model geometry, `load_state_dict` with the reference's
checkpoint keys, `.sample(batch_size, cond, sampling_cfg, get_intermediate_results)` returning a
CPU LongTensor (B,S).  Everything numeric runs in libldm_hip.so; this file only does what
`sample()` does on the host in the reference: the timestep list, skip-step bookkeeping, cond
duplication and dtype plumbing.
"""
from __future__ import annotations

import itertools
import logging
from typing import Dict, List, Optional, Union

import torch

from .binding import Engine

_seed_counter = itertools.count()
# the reference's entry point logs at INFO (trainer/test.py:38); `precision="auto"` says here which engine a checkpoint got and why
logger = logging.getLogger("layout_dm_amd")
# what each engine delivers on one MI355X (Rico25-shaped sequences, T = 100, batch 512; bench.py measures it on the box at hand)
THROUGHPUT_CLASS = {
    "fast": "~3 900 layouts/s (fp16 operands, one launch per sampling call)",
    "fast_verified": "~3 400 - 3 900 layouts/s (fp16 engine; greedy decoding re-checked by the reference-precision engine)",
    "hybrid": "~2 200 layouts/s (attention path: hi + lo fp16 activations x fp16 weights; FFN and head in plain fp16; per-step launches)",
    "hybrid_verified": "~2 200 layouts/s (attention path hi + lo x fp16, FFN / head plain fp16; greedy decoding re-checked by the reference-precision engine)",
    "mixed": "~1 500 layouts/s (hi + lo fp16 activations x fp16 weights: two matrix passes per weight product, per-step launches)",
    "mixed_verified": "~1 500 layouts/s (hi + lo fp16 activations x fp16 weights; greedy decoding re-checked by the reference-precision engine)",
    "split": "~1 150 - 1 200 layouts/s (reference precision on the fp16 matrix pipe, per-step launches)",
    "exact": "~420 layouts/s (fp32 MFMA)",
}


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    if hasattr(cfg, "get"):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


def batch_cuts(B: int, max_batch: int, round_: int) -> List[int]:
    """Sizes of the calls a batch of B layouts is cut into when it exceeds max_batch.  The reference caps a call at 512
    (test.py splits 1000 into 512 + 488); here a call costs whole ROUNDS of `round_` layouts (Engine.batch_round: one
    workgroup per layout per compute unit), so the cuts are the largest multiple of the round that fits max_batch — 1000
    layouts under max_batch = 600 are 512 + 488 (4 rounds), not 600 + 400 (5)."""
    if B <= max_batch:
        return [B] if B > 0 else []
    step = (max_batch // round_) * round_ or max_batch
    return [step] * (B // step) + ([B % step] if B % step else [])


def timestep_schedule(num_timesteps: int, num_timesteps_eval: int, time_difference: float = 0.0):
    """(t_model list, t_post list) exactly as base.py:310-315 + 218-240 derive them:
    diffusion_list = [int(i*T/T_eval)], skip_step = delta_t - 1, noise_t = clamp(t - int(T*td)),
    posterior evaluated at noise_t - skip_step when noise_t > skip_step."""
    assert num_timesteps_eval <= num_timesteps  # base.py:311
    t_model, t_post = [], []
    prev = num_timesteps
    for i in range(num_timesteps_eval - 1, -1, -1):
        t = int(i * num_timesteps / num_timesteps_eval)
        delta = prev - t
        if delta <= 0:
            raise NotImplementedError  # base.py:361-362
        skip = delta - 1
        noise_t = t
        if time_difference > 0.0:
            noise_t = min(max(t - int(num_timesteps * time_difference), 0), num_timesteps - 1)
        if skip > 0 and noise_t > skip:
            noise_t = noise_t - skip
        t_model.append(t)
        t_post.append(noise_t)
        prev = t
    return t_model, t_post


class HipMaskAndReplaceDiffusion:
    """MI355X implementation behind the reference's Seam-2."""

    def __init__(self, *, n_category: int, n_bin: int = 32, max_elem: int = 25, n_attr: int = 5,
                 d_model: int = 464, n_head: int = 8, d_ff: int = 1856, n_layer: int = 4,
                 num_timesteps: int = 100, precision: str = "exact", max_batch: int = 512, chunk: int = 0,
                 device: Optional[int] = None, use_graph: bool = True, q_type: str = "constrained", lanes: int = 0,
                 verifier: str = "split"):
        # q_type: Q_TYPES of models/layoutdm.py:20-23 — "constrained" (constrained.py) or "vanilla" (vanilla.py)
        # precision "fast_verified": the fp16 engine for every sampler, plus an exact (fp32) engine of the same weights
        # that re-decides the near-tie layouts of DETERMINISTIC decoding (layout_dm_amd/verified.py): greedy tokens are
        # then the exact mode's, i.e. the reference's, bit for bit
        # precision "auto" (r04): both engines are built; load_state_dict measures the fp16 engine's logits error against the
        # fp32 engine ON THE CHECKPOINT (verified.measure_fast_error: probe states over the whole timestep range) and keeps
        # the fast engine only if it is inside the north star's 1e-3 relative tolerance — otherwise every call runs exact.
        # (The fp16 mode's error is a property of the weights: 3e-4 on the reference's init, ~1e-3 at sigma = 0.06,
        # percents once attention rows saturate: DESIGN.md section 3.5.)
        # r06: a rung between the two — "mixed" (LDM_PREC_MIXED_F16: the split mode with fp16-only WEIGHTS, two matrix passes per weight
        # product instead of three, +19 % over split; its logits error is 2e-4 on a fitted checkpoint whose fp16 error is 1.2e-3).  When the
        # fp16 engine is outside the tolerance, auto builds a mixed engine, measures IT the same way, and keeps it — again with verified
        # greedy decoding — if it is inside; only then the reference-precision engine runs every call.  "mixed" / "mixed_verified" select it
        # unconditionally.  In front of it sits "hybrid" (LDM_PREC_HYBRID_F16: mixed with the FFN and the head in plain fp16 — the fp16 error
        # lives on the attention-score path; 2.8e-4 on the fitted checkpoint), so auto's ladder is fast -> hybrid -> mixed -> reference precision,
        # each rung built on first need and held to the same measurement.
        self.verified = None
        self._v_fast = None                      # VerifiedGreedy(fp16 engine, verifier)
        self._v_rung: Dict[str, object] = {}     # rung -> VerifiedGreedy(its engine, verifier), built on first need
        self._rung_error: Dict[str, Dict[str, float]] = {}   # rung -> what auto measured on the checkpoint loaded last
        self.verifier = verifier
        self.auto = precision == "auto"
        self.auto_tolerance = 1e-3
        self.verifier_tolerance = 5e-4     # split vs a small fp32-MFMA probe engine: half the north star's tolerance (two valid fp32-level
                                           # engines differ by 4.2e-4 on the "wide" point, where the reference's own f32 noise floor is 1.2e-4)
        self.verifier_check: Dict[str, float] = {}
        self.selection_report: Dict[str, object] = {}
        self.selected_precision = None if self.auto else precision
        self._mk = mk = lambda prec, mb=max_batch: Engine(n_category=n_category, n_bin=n_bin, max_elem=max_elem, n_attr=n_attr,
                                 d_model=d_model, n_head=n_head, d_ff=d_ff, n_layer=n_layer, n_step=num_timesteps,
                                 precision=prec, max_batch=mb, chunk=chunk, device=device, q_type=q_type,
                                 lanes=lanes)
        if precision in ("fast_verified", "mixed_verified", "hybrid_verified", "auto"):
            from .verified import VerifiedGreedy

            # the reference-precision engine behind fast_verified / mixed_verified / auto: "split" (fp16 x 3 on the fp16 matrix
            # pipe: the exact mode's logits error at 1.7 x its speed, r04) or "exact" (fp32 MFMA)
            assert verifier in ("split", "exact")
            self.engine = mk(precision[:-len("_verified")] if precision in ("mixed_verified", "hybrid_verified") else "fast")
            self.verified = self._v_fast = VerifiedGreedy(self.engine, mk(verifier))
        else:
            self.engine = mk(precision)
        self.q_type = q_type
        self.num_timesteps = num_timesteps
        self.num_classes = self.engine.C
        self.max_token_length = self.engine.S
        self.precision = precision
        self.use_graph = use_graph
        self.mask_id = self.engine.mask_id
        self.pad_id = self.engine.pad_id

    # -- nn.Module-ish surface used by the reference's callers --------------------------------
    @property
    def device(self) -> torch.device:
        return self.engine.device

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def _check_verifier(self, state_dict) -> None:
        """auto / fast_verified treat the verifier engine as ground truth: check THAT once per checkpoint.  The
        split engine rounds every operand to an fp16 hi part (overflow above 65 504) and an unscaled fp16 lo part, so it is
        compared with a small fp32-MFMA engine of the same weights on two probe states; if its logits are non-finite or
        differ by more than `verifier_tolerance` the fp32-MFMA engine becomes the verifier, and if that one is non-finite
        too the checkpoint is refused here, at load time — not at the first sampling call."""
        from .verified import measure_fast_error, probe_states

        v = self.verified
        n = max(1, min(4, int(v.exact.max_batch)))   # (the probe batch must fit an engine built with max_batch < 4)
        if self.verifier != "split":
            states = probe_states(v.exact, n_layouts=n, ts=[v.exact.T // 2])
            lg = v.exact.denoise_logits(*states[0])
            if not bool(torch.isfinite(lg).all()):
                raise FloatingPointError("reference-precision engine: non-finite logits on this checkpoint")
            self.verifier_check = {"verifier": self.verifier, "finite": True}
            return
        small = self._mk("exact", n)
        try:
            small.load_state_dict(state_dict)
            states = probe_states(small, n_layouts=n, ts=[small.T - 1, small.T // 2, 1])
            c = measure_fast_error(v.exact, small, states)
            lg_ok = bool(torch.isfinite(small.denoise_logits(*states[0])).all())
        finally:
            small.close()
        self.verifier_check = {"verifier": "split", "err_rel_vs_fp32_mfma": c["err_rel"], "finite": c["finite"],
                               "tolerance": self.verifier_tolerance}
        if c["finite"] and c["err_rel"] <= self.verifier_tolerance:
            return
        if not lg_ok:
            raise FloatingPointError("reference-precision engines: non-finite logits on this checkpoint (fp32 MFMA too)")
        import warnings

        warnings.warn(f"split verifier outside {self.verifier_tolerance:g} of the fp32-MFMA engine on this checkpoint "
                      f"({c['err_rel']:.3g}, finite={c['finite']}): verifying with the fp32-MFMA engine instead",
                      RuntimeWarning)
        old = v.exact
        v.exact = self._mk("exact")
        v.exact.load_state_dict(state_dict)
        old.close()
        self.verifier = "exact"
        self.verifier_check["verifier"] = "exact (fallback)"

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        if self.verified is not None:
            self.verified = self._v_fast                        # (auto may have switched to the mixed / the reference-precision engine before)
            self.engine = self.verified.fast
        self.engine.load_state_dict(state_dict)
        if self.verified is not None:
            self.verified.exact.load_state_dict(state_dict)
            self._check_verifier(state_dict)
            try:
                cal = self.verified.calibrate()                 # tie_abs <- 6 x safety x measured |logits error|
            except FloatingPointError:
                if not self.auto:
                    raise
                # the fp16 engine overflows on this checkpoint (the verifier was checked finite above): auto's answer
                cal = dict(self.verified.calibration, err_rel=float("inf"), finite=False)
                self.verified.calibration = cal
            if self.auto:
                self._rung_error = {}
                ok = cal["err_rel"] <= self.auto_tolerance       # (NaN / inf compare False)
                self.selected_precision = "fast_verified" if ok else self.verifier
                if not ok:
                    for rung in self.RUNGS:
                        if self._try_rung(rung, state_dict):
                            self.selected_precision = rung + "_verified"
                            break
                    else:
                        self.engine = self.verified.exact
        self._report_selection()
        return self

    RUNGS = ("hybrid", "mixed")   # auto's ladder between the fp16 engine and the reference-precision engine, fastest first

    def _try_rung(self, rung: str, state_dict) -> bool:
        """One rung of auto's ladder: the engine (built on first need; unavailable where the library has no such kernels for the model's
        geometry) measured against the verifier exactly like the fp16 engine was; True = it is inside the tolerance and now the engine."""
        from .verified import VerifiedGreedy

        if rung not in self._v_rung:
            try:
                self._v_rung[rung] = VerifiedGreedy(self._mk(rung), self._v_fast.exact)
            except RuntimeError as e:                            # (ldm_create: geometry without these kernels)
                self._rung_error[rung] = {"unavailable": str(e)}
                return False
        vm = self._v_rung[rung]
        vm.exact = self._v_fast.exact                            # (_check_verifier may have replaced the verifier engine)
        vm.fast.load_state_dict(state_dict)
        try:
            cal = vm.calibrate()
        except FloatingPointError:
            self._rung_error[rung] = {"err_rel": float("inf"), "finite": False}
            return False
        self._rung_error[rung] = {"err_rel": cal["err_rel"], "err_abs": cal["err_abs"], "finite": True, "tie_abs": cal["tie_abs"]}
        if not cal["err_rel"] <= self.auto_tolerance:
            return False
        self.verified = vm
        self.engine = vm.fast
        return True

    def close(self) -> None:
        """Release every engine this object built (the handles own GBs of workspace)."""
        seen = []
        for v in [self._v_fast] + list(self._v_rung.values()):
            if v is not None:
                seen += [v.fast, v.exact]
        done = []
        for e in seen + [self.engine]:
            if e is not None and not any(e is d for d in done):
                done.append(e)
                e.close()

    def _report_selection(self) -> None:
        """One INFO record per loaded checkpoint: which engine runs, the fp16 engine's measured logits error,
        the tolerance it was held against and the throughput class to expect — a user must be able to tell why a job runs at
        1 200 instead of 3 900 layouts/s.  The same dictionary is `selection_report` (python -m layout_dm_amd.check_checkpoint)."""
        sel = self.selected_precision
        rep = {"precision_requested": self.precision, "engine_selected": sel,
               "expected_throughput": THROUGHPUT_CLASS.get(sel, "?")}
        if self.verified is not None:
            explicit = self.precision in ("mixed_verified", "hybrid_verified")
            cal = self._v_fast.calibration if not explicit else {}
            rep.update({"fast_logits_err_rel": cal.get("err_rel"), "fast_logits_err_abs": cal.get("err_abs"),
                        "tolerance": self.auto_tolerance, "verifier": self.verifier, "verifier_check": dict(self.verifier_check),
                        "tie_abs": self.verified.calibration.get("tie_abs")})
            if explicit:
                rep[self.precision[:-len("_verified")] + "_logits_err_rel"] = self.verified.calibration.get("err_rel")
            for rung, err in self._rung_error.items():
                if "unavailable" in err:
                    rep[rung + "_unavailable"] = err["unavailable"]
                else:
                    rep[rung + "_logits_err_rel"] = err.get("err_rel")
        self.selection_report = rep
        if self.verified is None:
            logger.info("layout_dm_amd: engine '%s' (as requested) — %s", sel, rep["expected_throughput"])
            return
        fmt = lambda e: "non-finite" if e is None or e != e or e == float("inf") else f"{e:.2e}"   # noqa: E731
        err = rep["fast_logits_err_rel"]
        err_s = fmt(err)
        if self.auto:
            why = ("inside" if sel == "fast_verified" else "OUTSIDE") + f" the {self.auto_tolerance:g} logits tolerance"
            for rung in self.RUNGS:
                if rung + "_logits_err_rel" in rep:
                    why += (f"; the {rung} engine's is {fmt(rep[rung + '_logits_err_rel'])}, " + ("inside" if sel == rung + "_verified" else "OUTSIDE"))
                elif rung + "_unavailable" in rep:
                    why += f"; no {rung} engine for this geometry"
            logger.info("layout_dm_amd: precision='auto' selected engine '%s': the fp16 engine's logits error on this checkpoint is %s "
                        "(relative, against the %s reference-precision engine), %s — expect %s", sel, err_s, self.verifier, why,
                        rep["expected_throughput"])
        elif self.precision in ("mixed_verified", "hybrid_verified"):
            logger.info("layout_dm_amd: engine '%s' (as requested); its logits error on this checkpoint %s (relative, against the "
                        "%s engine) — expect %s", self.precision, fmt(rep.get(self.precision[:-len("_verified")] + "_logits_err_rel")), self.verifier,
                        rep["expected_throughput"])
        else:
            logger.info("layout_dm_amd: engine '%s' (as requested); fp16 logits error on this checkpoint %s (relative, against the %s "
                        "engine) — expect %s", sel, err_s, self.verifier, rep["expected_throughput"])

    @property
    def calibration(self) -> Dict[str, float]:
        """fp16-engine-vs-verifier logits error measured at load time (precision fast_verified / auto; mixed_verified: the mixed
        engine's), else {}."""
        return dict(self._v_fast.calibration) if self._v_fast is not None else {}

    # -- the hot path ----------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, batch_size: Optional[int] = 1, cond: Optional[Dict] = None, sampling_cfg=None,
               get_intermediate_results: bool = False, seed: Optional[int] = None, first_layout: int = 0,
               return_device_tensor: bool = False, **kwargs) -> Union[torch.Tensor, List[torch.Tensor]]:
        """BaseMaskAndReplaceDiffusion.sample (base.py:293-371).  Extra keyword arguments
        (`cond_type=...` from test.py:195-200) are accepted and ignored like the reference does.

        seed: Philox key for the stochastic samplers.  None -> drawn from torch's global CPU
        generator, so `set_seed()` (helpers/util.py:10-13) keeps runs reproducible.
        first_layout: global index of row 0 (batch sharding across calls / GPUs)."""
        eng = self.engine
        T = self.num_timesteps
        t_eval = int(_cfg_get(sampling_cfg, "num_timesteps", T))
        td = float(_cfg_get(sampling_cfg, "time_difference", 0.0) or 0.0)
        t_model, t_post = timestep_schedule(T, t_eval, td)
        if cond and cond.get("type") == "relation":
            raise NotImplementedError(
                "cond=relation needs the relation graph (cond['batch_w_canvas']) turned into the kernel's CSR form: use "
                "layout_dm_amd.relation.sample_with_relation, which LayoutDM.sample does (the logit adjustment of "
                "logit_adjustment.py:88-126 then runs inside ldm_sample_loop)")
        if seed is None:
            # (deterministic decoding draws nothing — and must not advance torch's global generator either: the reference's argmax
            #  path consumes no random numbers (helpers/sampling.py:110-111), and test.py's get_cond of the NEXT batch reads that stream)
            seed = 0 if str(_cfg_get(sampling_cfg, "name")) == "deterministic" else int(torch.randint(0, 2 ** 62, (1,)).item())
        B = int(batch_size)
        if B == 0:  # an empty shard (distributed.shard_range with fewer layouts than ranks)
            empty = torch.empty((0, eng.S), dtype=torch.int32, device=eng.device)
            if get_intermediate_results:
                return [empty.long().cpu() for _ in t_model]
            return empty if return_device_tensor else empty.long().cpu()
        if cond:
            seq = torch.as_tensor(cond["seq"])
            if seq.size(0) == 1 and B > 1:  # duplicate_cond, helpers/task.py:235-248
                cond = dict(cond)
                for k, v in list(cond.items()):
                    # only what still has the single-layout batch dimension (a caller may hand over tensors that
                    # are already per-sample, e.g. a (B,C,S) refinement prior)
                    if isinstance(v, torch.Tensor) and v.dim() > 0 and v.size(0) == 1:
                        cond[k] = v.repeat([B] + [1] * (v.dim() - 1))
                seq = cond["seq"]
            assert seq.size(0) == B, "cond['seq'] batch does not match batch_size"
            tokens = seq.to(device=eng.device, dtype=torch.int32).contiguous().clone()
        else:
            tokens = torch.full((B, eng.S), eng.mask_id, dtype=torch.int32, device=eng.device)
        outs, inters = [], []
        off = 0
        for n in batch_cuts(B, eng.max_batch, eng.batch_round):  # the reference caps a call at 512 (Converter); we cut
            sub = None
            if cond:
                sub = {k: (v[off:off + n] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.size(0) == B else v)
                       for k, v in cond.items()}
            if (self.verified is not None and self.engine is self.verified.fast
                    and _cfg_get(sampling_cfg, "name") == "deterministic"):
                tk, inter = self.verified.sample_loop(tokens[off:off + n].contiguous(), t_model, t_post, cond=sub,
                                                      intermediates=get_intermediate_results)
            else:
                tk, inter = eng.sample_loop(tokens[off:off + n].contiguous(), t_model, t_post, sampling_cfg, cond=sub,
                                            seed=seed, first_layout=first_layout + off,
                                            intermediates=get_intermediate_results, use_graph=self.use_graph)
            outs.append(tk)
            inters.append(inter)
            off += n
        out = torch.cat(outs) if len(outs) > 1 else outs[0]
        if get_intermediate_results:
            inter = torch.cat(inters, dim=1) if len(inters) > 1 else inters[0]
            return [x.long().cpu() for x in inter]
        if return_device_tensor:
            return out
        return out.long().cpu()
