"""CPU: the per-token scalar tail of a reverse step (layout_dm_amd/csrc/ldm_post_token.h — log-softmax, constrained
posterior on the token's sub-vocabulary, cond overrides, draw with the kernel's Philox stream), compiled for the host and
run against the oracle on states of the REFERENCE's trajectories (tests/golden): greedy tokens, and stochastic draws on
identical uniforms for random / top-k / top-p; gumbel for support and determinism.  This is the form in which one lane of
the stack kernel can finish a token behind the fused vocabulary head."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

from oracle import restatement as R
from oracle import spec as SP
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = {"deterministic": 0, "random": 1, "top_p": 2, "top_k": 3, "gumbel": 4}
SCHED_KEYS = ("log_at", "log_bt", "log_ct", "log_cumprod_at", "log_cumprod_bt", "log_cumprod_ct")


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    cxx = shutil.which("g++") or shutil.which("c++")
    if cxx is None:
        pytest.skip("no host C++ compiler")
    exe = tmp_path_factory.mktemp("post_token") / "cpu_post_token_check"
    subprocess.run([cxx, "-O2", "-std=c++17", "-Wall", "-Wextra", "-Werror",
                    os.path.join(ROOT, "tests", "cpu_post_token_check.cpp"), "-o", str(exe)], check=True, cwd=ROOT)
    return str(exe)


def _run(harness, tmp_path, spec, W, tokens, t_post, logits, cfg, step, seed, first_layout, cond=None, f64=False,
         alias=True):
    """tokens (B,S) int64, logits (B,S,C) float32 -> tokens drawn by the scalar tail, (B,S) int64."""
    B, S = tokens.shape
    C, A, T = spec.n_class, spec.n_attr, spec.n_step
    N = B * S
    pos = np.tile(np.arange(S, dtype=np.int32), B)
    attr = pos % A
    start = np.array([int(spec.full_ids(a)[0]) for a in range(A)], np.int32)[attr]
    count = np.array([len(spec.full_ids(a)) - 2 for a in range(A)], np.int32)[attr]
    u = (t_post - 1 + (T + 1)) % (T + 1)  # constrained.py:114
    sched = np.zeros((A, 10), np.float32)
    for a, key in enumerate(SP.VAR_NAMES):
        g = lambda n, i: float(W[f"{key}_{n}"][i])
        sched[a] = [g("log_at", t_post), g("log_bt", t_post), g("log_ct", t_post),
                    g("log_cumprod_at", t_post), g("log_cumprod_bt", t_post), g("log_cumprod_ct", t_post),
                    g("log_cumprod_at", u), g("log_cumprod_bt", u), g("log_cumprod_ct", u),
                    g("log_1_min_cumprod_ct", u)]
    cond_tok = np.full(N, -1, np.int32)
    strong = np.zeros(N, np.int32)
    pad_dis = np.zeros(N, np.int32)
    weak = None
    if cond is not None:
        cs = np.asarray(cond["seq"]).reshape(-1).astype(np.int32)
        cond_tok = cs
        strong = np.asarray(cond["mask"]).reshape(-1).astype(np.int32)
        if cond.get("type") in ("c", "cwh", "refinement", "relation"):  # base.py:272-284
            pad_dis = ((attr != 0) & (cs != spec.pad_id)).astype(np.int32)
        if cond.get("type") == "refinement":
            weak = np.ascontiguousarray(np.asarray(cond["weak_logits"], np.float32).transpose(0, 2, 1)).reshape(N, C)
    layout = (np.repeat(np.arange(B, dtype=np.uint64), S) + np.uint64(first_layout))
    path_in, path_out = str(tmp_path / "case.bin"), str(tmp_path / "out.bin")
    with open(path_in, "wb") as f:
        f.write(np.array([0x4C444D31, N, C, spec.pad_id, spec.mask_id, KINDS[cfg["name"]], int(cfg.get("top_k", 1)),
                          1 if f64 else 0, 1 if weak is not None else 0, 1 if alias else 0], np.int32).tobytes())
        f.write(np.array([cfg.get("temperature", 1.0), cfg.get("top_p", 1.0)], np.float32).tobytes())
        f.write(struct.pack("<Q", seed))
        for arr in (tokens.numpy().reshape(-1).astype(np.int32), start, count, cond_tok, strong, pad_dis, pos,
                    np.full(N, step, np.int32)):
            f.write(np.ascontiguousarray(arr, np.int32).tobytes())
        f.write(layout.tobytes())
        f.write(np.ascontiguousarray(sched[attr], np.float32).tobytes())
        f.write(np.ascontiguousarray(logits.numpy().reshape(N, -1)[:, :C], np.float32).tobytes())
        if weak is not None:
            f.write(weak.tobytes())
    subprocess.run([harness, path_in, path_out], check=True, timeout=120)
    return torch.from_numpy(np.fromfile(path_out, np.int32).astype(np.int64)).view(B, S)


def _weights(ds):
    spec = SP.SPECS[ds]
    return spec, R.as_torch_weights(synth.synth_state_dict(spec, seed=1, perturb=True))


@pytest.mark.parametrize("f64", [True, False], ids=["f64_lse", "f32_lse"])
def test_scalar_tail_greedy_equals_reference_tokens(harness, tmp_path, golden_dir, f64):
    """Greedy: the reference's own argmax tokens on states of its stochastic trajectories (uncond, cond=c with the
    strong mask and the [PAD] disable, refinement with the additive prior)."""
    cases = []
    spec, W = _weights("rico25")
    g = np.load(os.path.join(golden_dir, "rico25_uncond_trajectory.npz"))
    cases.append((spec, W, g, None))
    g = np.load(os.path.join(golden_dir, "rico25_refinement_trajectory.npz"))
    table = torch.from_numpy(g["weak_table"])
    seq_orig = torch.from_numpy(g["seq_orig"].astype(np.int64))
    cases.append((spec, W, g, {"seq": g["cond_seq"].astype(np.int64), "mask": g["cond_mask"], "type": "refinement",
                               "weak_logits": table[seq_orig].permute(0, 2, 1).contiguous().numpy()}))
    spec_p, W_p = _weights("publaynet")
    g = np.load(os.path.join(golden_dir, "publaynet_cond_c_trajectory.npz"))
    cases.append((spec_p, W_p, g, {"seq": g["cond_seq"].astype(np.int64), "mask": g["cond_mask"], "type": "c"}))
    gv = np.load(os.path.join(golden_dir, "rico25_cond_variants.npz"))
    for ctype in ("cwh", "partial"):  # helpers/task.py:61-110
        sub = {k[len(ctype) + 1:]: gv[k] for k in gv.files if k.startswith(ctype + "_")}
        cases.append((spec, W, sub, {"seq": sub["cond_seq"].astype(np.int64), "mask": sub["cond_mask"], "type": ctype}))
    bad = total = 0
    for spec, W, g, cond in cases:
        for i in (0, 25, 50, 75, 99):
            t = int(g["steps"][i])
            toks = torch.from_numpy(g["states_before"][i].astype(np.int64))
            logits = R.denoiser_logits(W, spec, toks, t)
            # working storage: a separate buffer for the even states, the token's own logits row (as in the kernel) for the odd
            out = _run(harness, tmp_path, spec, W, toks, t, logits, {"name": "deterministic"}, i, 0, 0, cond, f64=f64,
                       alias=bool(i & 1))
            ref = torch.from_numpy(g["greedy_next"][i].astype(np.int64))
            mism = out != ref
            if mism.any():  # only where the reference's own top-2 margin is at the fp32 rounding level
                assert float(torch.from_numpy(g["greedy_margin"][i])[mism].max()) < 1e-4
            bad += int(mism.sum())
            total += ref.numel()
    assert bad <= (0 if f64 else 2), f"{bad}/{total}"


@pytest.mark.parametrize("cfg", [{"name": "random", "temperature": 1.0}, {"name": "random", "temperature": 0.7},
                                 {"name": "top_k", "top_k": 5, "temperature": 1.0},
                                 {"name": "top_p", "top_p": 0.9, "temperature": 1.0}],
                         ids=["random", "random_T0.7", "top_k5", "top_p0.9"])
def test_scalar_tail_draws_equal_oracle_on_identical_uniforms(harness, tmp_path, golden_dir, cfg):
    """Stochastic samplers: same Philox uniforms (seed, global layout index, step, position) as the oracle's inverse-CDF
    rule, i.e. as the wave-per-token kernel: token for token, except where fp32 rounding moves a CDF edge across u."""
    spec, W = _weights("rico25")
    g = np.load(os.path.join(golden_dir, "rico25_uncond_trajectory.npz"))
    bad = total = 0
    for i in (3, 40, 80, 99):
        t = int(g["steps"][i])
        toks = torch.from_numpy(g["states_before"][i].astype(np.int64))
        B, S = toks.shape
        logits = R.denoiser_logits(W, spec, toks, t)
        out = _run(harness, tmp_path, spec, W, toks, t, logits, cfg, i, 1234567890123, 500)
        u = R.token_uniforms(1234567890123, 500, B, S, i)[..., 0]
        ref = R.single_step(W, spec, toks, t, cfg, uniforms=u)
        bad += int((out != ref).sum())
        total += ref.numel()
    assert bad <= 2, f"{bad}/{total}"


def test_scalar_tail_gumbel_support_and_determinism(harness, tmp_path, golden_dir):
    spec, W = _weights("rico25")
    g = np.load(os.path.join(golden_dir, "rico25_uncond_trajectory.npz"))
    i = 60
    t = int(g["steps"][i])
    toks = torch.from_numpy(g["states_before"][i].astype(np.int64))
    logits = R.denoiser_logits(W, spec, toks, t)
    cfg = {"name": "gumbel", "temperature": 1.0}
    a = _run(harness, tmp_path, spec, W, toks, t, logits, cfg, i, 99, 0)
    b = _run(harness, tmp_path, spec, W, toks, t, logits, cfg, i, 99, 0)
    c = _run(harness, tmp_path, spec, W, toks, t, logits, cfg, i, 100, 0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    for at in range(spec.n_attr):
        assert torch.isin(a[:, at::spec.n_attr], torch.as_tensor(spec.full_ids(at))).all()
