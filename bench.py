#!/usr/bin/env python
"""bench.py — layouts/sec of the LayoutDM sampling hot path on MI355X.

One "step" = one full pass of the hot path over one batch: the T-step reverse loop (denoiser forward + posterior +
categorical draw per step) for `batch` layouts per GPU, starting from a state resident in HBM (all-[MASK] or the
conditioned sequence) and ending with the final int32 tokens on the device (gathered to every rank with ONE RCCL
all_gather when N>1, through layout_dm_amd.distributed.sample_sharded).

Workloads (BASELINE.json configs), selected with --config:
  2  Rico25    cond=unconditional  T=100  512 layouts/GPU   sampling=random    (default at --gpus 1; the config the
                                                                                 metric is quoted on)
  3  PubLayNet cond=c              T=100  1024 layouts/GPU  sampling=top_p 0.9 (cond built as helpers/task.py:94-110)
  4  Rico25    cond=unconditional  T=100  1024 layouts/GPU  sampling=random    (default at --gpus N>1: 8 GPUs = 8192)
Random-init synthetic weights with the reference's init distributions (no checkpoints offline).

The ONE JSON line carries the headline numerics mode (--precision, default `fast`) at top level and, at N=1:
  "modes"     every numerics mode on the headline workload, each with its own roofline against its own peak: exact (fp32
              MFMA, the mode of the bit-exact token tests), fast (fp16 operands / fp32 accumulate), and — config 2
              verbatim, i.e. GREEDY decoding — fast_verified (fast + exact re-decision of the near-tie layouts: the exact
              mode's tokens, checked here against the exact engine) next to the plain fast greedy figure;
  "configs"   the other BASELINE configurations and the "next" rows of SURVEY section 8f at a few steps each: 3 (PubLayNet
              cond=c top-p, 1024), 4 (the per-GPU shard of the scaling run, 1024), refinement, relation (512 each);
  "fid_features"  layouts/s of the FID feature extractor (section 8f row 3);
  "layout_metrics"  layouts/s of the alignment / overlap metrics kernel (section 8f row 1);
  "tokens_sha256" of the first 512 layouts of a fixed-seed Rico25 unconditional run: independent of N by construction
              (Philox keyed by global layout index), so a scaling run can prove it.

Contract: `python bench.py --gpus N --steps K --warmup W`.  For N > 1 either launch it under torch.distributed.run (one
rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment) or call it plainly — `python bench.py
--gpus N` then re-executes itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
127.0.0.1` on a free port.  Every N > 1 run uses the SAME per-GPU workload (config 4: 1024 layouts per GPU); the N = 1
line keeps config 2 as the headline and carries the config-4 figure under "scaling_point".  Rank 0 prints ONE JSON
line (with n_gpus, the RCCL world size actually seen, per-rank layouts/s min / max, tokens_sha256).
`--dry-run`: the same launch, shard, gather, barrier and reporting path on the gloo backend with a fake sampler (no
GPU): what tests/test_bench_spawn.py runs at world size 2.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_TOKEN_STEP = {  # 4 x [QKV + attn + out-proj + FFN] + head(2*464*C)
    "rico25": 21_740_256,
    "publaynet": 21_721_696,
}
# MI355X_MICROARCH.md dense MFMA peaks; split = 3 fp16 MFMA passes per product (divide by the passes)
# mixed: two fp16 passes per weight product, three per attention product (4.27 % of the flops): 2.043 passes per flop on average
# hybrid: one pass for the FFN and the head (64.1 % of the flops), two for in_proj / out_proj (31.7 %), three inside the attention: 1.403 on average
PEAK_TFLOPS = {"exact": 157.3, "fast": 2500.0, "split": 2500.0 / 3, "fast_verified": 2500.0, "mixed": 2500.0 / 2.043, "hybrid": 2500.0 / 1.403}
DTYPE = {"exact": "f32", "fast": "f16 (f32 accumulate)", "split": "f16x3 split (f32 accumulate)",
         "mixed": "f16 hi+lo activations x f16 weights (f32 accumulate)",
         "hybrid": "attention path f16 hi+lo activations x f16 weights, FFN / head plain f16 (f32 accumulate)",
         "fast_verified": "f16 (f32 accumulate) + f32 re-decision of near-tie layouts"}
CONFIGS = {
    2: dict(dataset="rico25", cond="unconditional", batch=512, sampling="random"),
    3: dict(dataset="publaynet", cond="c", batch=1024, sampling="top_p"),
    4: dict(dataset="rico25", cond="unconditional", batch=1024, sampling="random"),
}
GEMM_CLASSES = ("gemm", "ffn", "qkv", "layer")
FETCH_CALIBRATION = 2.0           # profiles/r03_fetch_size_calibration.txt
WEIGHT_IMAGE_BYTES = 23.4e6       # fast mode: LDS weight images of the 4 layers + head, fp16
SHA_LAYOUTS, SHA_SEED = 512, 20260926


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4], help="BASELINE config (0 = 2 at N=1, 4 at N>1)")
    ap.add_argument("--batch", type=int, default=0, help="layouts per GPU per step (0 = the config's)")
    ap.add_argument("--dataset", default=None, choices=["rico25", "publaynet"])
    ap.add_argument("--cond", default=None, choices=["unconditional", "c"])
    ap.add_argument("--sampling", default=None, choices=["random", "deterministic", "top_p", "top_k", "gumbel"])
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--precision", default=os.environ.get("LDM_BENCH_PRECISION", "fast"),
                    choices=["exact", "fast", "split", "mixed", "hybrid"])
    ap.add_argument("--modes", default=None,
                    help="comma list of numerics modes reported under 'modes' (default: exact,fast,fast_verified at N=1, "
                         "none at N>1; split = the fp16x3 cross-check mode, on request)")
    ap.add_argument("--no-extras", action="store_true", help="skip 'configs', 'fid_features' (N=1 extras)")
    ap.add_argument("--total", type=int, default=0,
                    help="STRONG scaling: total layouts per step over all GPUs (sharded by global layout index); "
                         "0 = weak scaling with --batch / the config's batch per GPU")
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=0, help="concurrent chunk pipelines (0 = library default)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the live rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes")
    ap.add_argument("--cpu-budget", type=float, default=22.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / shard / gather / report path only: gloo backend, fake sampler, no GPU")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------- self-spawn (N > 1)
def maybe_spawn(a):
    """`python bench.py --gpus N` outside a torch.distributed.run launch: re-execute under it (one rank per GPU)."""
    if a.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // a.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def _fake_tokens(first, count, seed, S=125):
    """Deterministic function of the GLOBAL layout index (what the Philox-keyed sampler guarantees on the GPU)."""
    import torch

    i = torch.arange(first, first + count, dtype=torch.int64)[:, None]
    s = torch.arange(S, dtype=torch.int64)[None, :]
    return ((((i * 2654435761 + s * 40503 + seed * 7919) & 0xFFFFFFFF) >> 7) % 150).to(torch.int32)


def dry_run_main(a, world, rank, local_rank):
    """The N-rank launch path without a GPU: gloo, fake sampler through distributed.sample_sharded, the same barrier +
    max-over-ranks timing, per-rank statistics and tokens_sha256 as the real run."""
    import hashlib

    import torch

    from layout_dm_amd.distributed import sample_sharded, shard_range

    dist = None
    if world > 1:
        import torch.distributed as dist

        sys.stdout.flush()
        saved = os.dup(1)       # gloo announces its connections on stdout: keep the ONE-line contract
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo")
            dist.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    config = a.config or (2 if world == 1 else 4)
    B = a.batch or CONFIGS[config]["batch"]
    n_global = a.total if a.total else world * B
    lo, hi = shard_range(n_global, rank, world) if a.total else (rank * B, (rank + 1) * B)
    seed_box, local_s = [0], [0.0]

    def sample_fn(first, count):
        assert world == 1 or (first == lo and count == hi - lo)
        t = time.perf_counter()
        out = _fake_tokens(first, count, seed_box[0])
        local_s[0] += time.perf_counter() - t
        return out

    def sync():
        if dist is not None:
            dist.barrier()

    final = None
    for i in range(a.warmup):
        seed_box[0] = i
        final = sample_sharded(sample_fn, n_global)
    sync()
    local_s[0] = 0.0
    t0 = time.perf_counter()
    for i in range(a.steps):
        seed_box[0] = a.warmup + i
        final = sample_sharded(sample_fn, n_global)
    sync()
    dt = time.perf_counter() - t0
    rates = [(hi - lo) * a.steps / max(local_s[0], 1e-9)]
    seen_world = 1
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        allr = [None] * world
        dist.all_gather_object(allr, rates[0])
        rates, seen_world = allr, dist.get_world_size()
    assert final.shape[0] == n_global and torch.equal(final, _fake_tokens(0, n_global, a.warmup + a.steps - 1))
    sha = hashlib.sha256(sample_sharded(lambda f, c: _fake_tokens(f, c, SHA_SEED), max(SHA_LAYOUTS, world))
                         [:SHA_LAYOUTS].contiguous().numpy().tobytes()).hexdigest()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({
            "metric": "layouts/sec (whole node), Rico25 uncond T=100", "value": round(n_global * a.steps / dt, 2),
            "unit": "layouts/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3), "higher_is_better": True,
            "scaling": "strong" if a.total else "weak", "vs_baseline": None, "dtype": "none (dry run)",
            "data": "synthetic", "dry_run": True,
            "config": {"workload": f"DRY RUN (fake sampler, gloo): BASELINE config {config} shape, batch={B}/GPU",
                       "parallelism": f"dp{world} (independent layout shards, one all_gather of the final tokens)"},
            "world_size_seen": seen_world,
            "per_rank_layouts_per_s": {"min": round(min(rates), 1), "max": round(max(rates), 1), "ranks": len(rates)},
            "tokens_sha256": {"sha256": sha, "of": "fake sampler; must not depend on n_gpus"}}), flush=True)


# ----------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline_reference(spec, sd, T, sampling, cond_np, budget_s=22.0):
    """The REAL reference's CPU path (kind = "reference"), timed on this host exactly as trainer/test.py:194-203 times it:
    time.time() around `model.sample(batch_size, cond, sampling_cfg)` of its own ConstrainedMaskAndReplaceDiffusion, imported
    from the reference tree or — on the GPU box — from the byte-compiled oracle/_ref/ that oracle/build_ref.py built from it
    (oracle/ref_harness.py supplies the hydra / omegaconf / torch_geometric stubs).  Returns None when neither is present.
    Sample: BASELINE config 1 as written (batch 4, all T steps), then batch 64 over a strided schedule of T/4 steps (the
    reference's own `sampling_cfg.num_timesteps`, base.py:310-315; every step costs one denoiser forward + posterior + draw)
    scaled to T steps."""
    import torch

    try:
        from oracle import ref_harness as rh

        if not rh.reference_importable():
            return None
        m, _tok = rh.build_reference_model(spec.name, seed=0, n_step=spec.n_step)
    except Exception as e:  # the reference needs more than the stubs provide: fall back to the port, and say why
        return {"unavailable": f"{type(e).__name__}: {e}"[:200]}
    try:   # (the whole reference leg is guarded — it executes byte-compiled reference code at the very end of main(); a failure
        #  here must fall back to the port, not abort bench.py before its one JSON line)
        return _cpu_baseline_reference_timed(rh, m, spec, sd, T, sampling, cond_np, budget_s)
    except Exception as e:
        return {"unavailable": f"{type(e).__name__}: {e}"[:200]}


def _cpu_baseline_reference_timed(rh, m, spec, sd, T, sampling, cond_np, budget_s):
    import torch

    m.load_state_dict({k.split("model.module.")[-1]: torch.as_tensor(v) for k, v in sd.items()})
    m.eval()
    ncpu = os.cpu_count() or 1
    threads = min(ncpu, 32)
    torch.set_num_threads(threads)

    def run(batch, t_eval):
        cfg = rh.sampling_cfg(sampling, num_timesteps=t_eval, top_p=0.9)
        cond = None
        if cond_np is not None:
            cond = {"seq": torch.from_numpy(cond_np["seq"][:batch]).long(), "mask": torch.from_numpy(cond_np["mask"][:batch]).bool(),
                    "type": "c"}
        torch.manual_seed(0)
        t0 = time.time()
        with torch.no_grad():
            ids = m.sample(batch_size=batch, cond=cond, sampling_cfg=cfg)   # test.py:195-200
        dt = time.time() - t0
        assert tuple(ids.shape) == (batch, spec.seq_len)
        return dt

    t_all = time.time()
    run(4, 4)                                    # warm-up (thread pool, allocator)
    dt4 = run(4, T)
    runs = [{"batch": 4, "steps": T, "seconds": round(dt4, 2), "layouts_per_s": round(4 / dt4, 3)}]
    left = budget_s - (time.time() - t_all)
    per_step64 = 16 * dt4 / T                     # (upper estimate: cost linear in the batch)
    n64 = int(max(0, min(T // 4, left / max(per_step64, 1e-9))))
    if n64 >= 5:
        dt64 = run(64, n64)
        runs.append({"batch": 64, "steps": n64, "seconds": round(dt64, 2), "layouts_per_s": round(64 / (dt64 * T / n64), 3),
                     "extrapolated": f"{n64} strided steps scaled to T = {T}"})
    best = max(runs, key=lambda r: r["layouts_per_s"])
    return {"value": best["layouts_per_s"], "unit": "layouts/s", "cores": threads, "kind": "reference",
            "value_from": f"batch {best['batch']}" + (" (EXTRAPOLATED from a strided run)" if "extrapolated" in best else " (all T steps timed)"),
            "value_full_T_run": runs[0]["layouts_per_s"],
            "host_logical_cpus": ncpu, "runs": runs,
            "source": "reference tree" if rh.reference_available() else "oracle/_ref (byte-compiled by oracle/build_ref.py)",
            "sample": f"the reference's own sample() (base.py:293-371) timed as test.py:194-203: batch 4 x all T={T} steps"
                      + (f", batch 64 x {n64} strided steps scaled to T" if len(runs) > 1 else "")
                      + f"; torch CPU fp32, {threads} threads; best = batch {best['batch']}"}


def cpu_baseline(spec, sd, T, sampling, cond_np, budget_s=22.0):
    """kind "reference" when the reference itself is importable here (cpu_baseline_reference), else the port below."""
    ref = cpu_baseline_reference(spec, sd, T, sampling, cond_np, budget_s)
    if ref is not None and "value" in ref:
        return ref
    out = cpu_baseline_port(spec, sd, T, sampling, cond_np, budget_s)
    if ref is not None:
        out["reference_unavailable"] = ref["unavailable"]
    return out


def cpu_baseline_port(spec, sd, T, sampling, cond_np, budget_s=22.0):
    """The oracle restatement of the reference's CPU path (kind = "port"), timed on this host on a BOUNDED sample of
    the same workload.  (batch, threads) is swept first — small-batch CPU inference degrades with too many threads —
    then the best configuration runs as many of the T reverse steps as fit in the remaining budget (every step costs
    the same: one denoiser forward + posterior + draw), fp32, torch CPU.  layouts/s = batch / (mean step time x T)."""
    import torch

    from oracle import restatement as R

    W = R.as_torch_weights(sd)
    ncpu = os.cpu_count() or 1
    cfg = {"name": sampling, "temperature": 1.0, "top_p": 0.9}
    steps = R.timestep_list(spec.n_step, T)

    def make(batch):
        if cond_np is None:
            return torch.full((batch, spec.seq_len), spec.mask_id, dtype=torch.long), None
        c = {"seq": torch.from_numpy(cond_np["seq"][:batch]), "mask": torch.from_numpy(cond_np["mask"][:batch]),
             "type": "c"}
        return c["seq"].clone(), c

    def run(batch, threads, n_steps, budget):
        torch.set_num_threads(threads)
        tokens, cond = make(batch)
        t0, done = time.time(), 0
        for i, t in enumerate(steps[:n_steps]):
            u = R.token_uniforms(0, 0, batch, spec.seq_len, i)[..., 0] if sampling != "deterministic" else None
            tokens = R.single_step(W, spec, tokens, t, cfg, uniforms=u, cond=cond)
            done += 1
            if time.time() - t0 > budget:
                break
        return (time.time() - t0) / done, done

    t_all = time.time()
    cands = [(16, min(ncpu, 32)), (64, min(ncpu, 32)), (64, min(ncpu, 64)), (64, min(ncpu, 128))]
    seen, best = set(), None
    for batch, threads in cands:
        if (batch, threads) in seen:
            continue
        seen.add((batch, threads))
        run(batch, threads, 1, 5.0)  # warm-up (thread pool, allocator)
        per_step, _ = run(batch, threads, 2, 4.0)
        rate = batch / per_step
        if best is None or rate > best[0]:
            best = (rate, batch, threads)
        if time.time() - t_all > 0.45 * budget_s:
            break
    _, batch, threads = best
    remaining = max(4.0, budget_s - (time.time() - t_all))
    per_step, done = run(batch, threads, T, remaining)
    return {"value": round(batch / (per_step * T), 3), "unit": "layouts/s", "cores": threads, "kind": "port",
            "host_logical_cpus": ncpu,
            "sample": f"{batch} layouts x {done} of T={T} reverse steps ({per_step * done:.1f} s; per-step cost is "
                      f"uniform), best of a (batch, threads) sweep {sorted(seen)}, oracle/restatement.py, torch CPU "
                      f"fp32, {threads} threads"}


# ----------------------------------------------------------------------------------------- HBM traffic (PMC)
def measure_traffic(kernel_substr: str, dataset: str, precision: str, timeout_s: int = 170):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC counters, collected live: two separate counter
    passes (FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md) over tools/pmc_probe.py (a
    4-step loop with the bench's launch shapes).  Returns (dict | None, note)."""
    exe = "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    vals = {}
    env = dict(os.environ, TMPDIR="/tmp", PMC_DATASET=dataset, PMC_PRECISION=precision)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix=f"ldm_pmc_{counter}_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "run", "--",
               sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py")]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                               timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out"
        if p.returncode != 0:
            return None, f"rocprofv3 --pmc {counter} failed rc={p.returncode}: {p.stdout[-300:]}"
        import csv

        acc = []
        for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if any(k in row.get("Kernel_Name", "") for k in kernel_substr.split("|")) and \
                        row.get("Counter_Name") == counter:
                    acc.append(float(row["Counter_Value"]))
        subprocess.run(["rm", "-rf", out_dir])
        if not acc:
            return None, f"no {counter} rows for kernel '{kernel_substr}'"
        vals[counter] = sum(acc) / len(acc)  # KB per launch
    return vals, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_probe.py), KB per launch"


def measure_traffic_per_kernel(dataset: str, precision: str, steps: int = 2, timeout_s: int = 170):
    """FETCH_SIZE / WRITE_SIZE per launch of EVERY kernel class of one sampling call in `precision` (two separate counter passes
    over tools/pmc_probe.py with PMC_STEPS = steps): {kernel symbol: {"fetch_bytes_raw", "fetch_bytes_calibrated", "write_bytes",
    "launches"}} or (None, note).  Used for the split mode, whose step is a chain of kernels (per-kernel bytes)."""
    exe = "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    import csv
    import re

    acc = {}
    env = dict(os.environ, TMPDIR="/tmp", PMC_DATASET=dataset, PMC_PRECISION=precision, PMC_STEPS=str(steps))
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out_dir = tempfile.mkdtemp(prefix=f"ldm_pmck_{counter}_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out_dir, "-o", "run", "--",
               sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py")]
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout_s)
        except subprocess.TimeoutExpired:
            return None, f"rocprofv3 --pmc {counter} timed out"
        if p.returncode != 0:
            return None, f"rocprofv3 --pmc {counter} failed rc={p.returncode}: {p.stdout[-300:]}"
        for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                name = row.get("Kernel_Name", "")
                if "ldm::" not in name or row.get("Counter_Name") != counter:
                    continue
                key = re.sub(r"\(.*$", "", name).replace("void ", "").replace("ldm::", "")
                acc.setdefault(key, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
        subprocess.run(["rm", "-rf", out_dir])
    out = {}
    for k, v in acc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v and len(v["FETCH_SIZE"]) >= 4:
            f_raw = 1024 * sum(v["FETCH_SIZE"]) / len(v["FETCH_SIZE"])
            w = 1024 * sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"])
            out[k] = {"fetch_bytes_raw": int(f_raw), "fetch_bytes_calibrated": int(f_raw * FETCH_CALIBRATION), "write_bytes": int(w),
                      "launches": len(v["FETCH_SIZE"])}
    return (out or None), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_probe.py), bytes per launch of 256 layouts"


# event-profile class -> substring of the kernel symbol in the PMC csv
KERNEL_SYMBOL = {"layers_fused": "stack_stream_k", "layers_fused_loop": "stack_stream_k",
                 "posterior_sample": "posterior_sample_k", "gemm_ffn2": "gemm_f", "gemm_ffn1": "gemm_f",
                 "attention": "attn_"}


# split / mixed / hybrid: event-profile class -> kernel symbol prefix (template arguments: <ADA, OUT, TM, ABL, PRE, NPM, NPP>: products per
# k16-step of the tile loop / of the GEMM prologue; attnout16x3_k<TM, W2>)
SPLIT_KERNEL_SYMBOL = {"gemm_ffn2_qkv_ln": "lngemm16x3_k<true, 2, false, 0, true, 3, 3>", "gemm_qkv_ln": "lngemm16x3_k<true, 2, false, 0, false, 3, 3>",
                       "gemm_ffn1_ln": "lngemm16x3_k<false, 1, false, 0, false, 3, 3>", "gemm_ffn2_head_ln": "lngemm16x3_k<false, 0, false, 0, true, 3, 3>",
                       "attn_out_fused": "attnout16x3_k<false, false, false>", "posterior_sample": "posterior_sample_k<16, true, false>"}
MIXED_KERNEL_SYMBOL = {"gemm_ffn2_qkv_ln": "lngemm16x3_k<true, 2, false, 0, true, 2, 2>", "gemm_qkv_ln": "lngemm16x3_k<true, 2, false, 0, false, 2, 2>",
                       "gemm_ffn1_ln": "lngemm16x3_k<false, 1, false, 0, false, 2, 2>", "gemm_ffn2_head_ln": "lngemm16x3_k<false, 0, false, 0, true, 2, 2>",
                       "attn_out_fused": "attnout16x3_k<false, true, false>", "posterior_sample": "posterior_sample_k<16, true, false>"}
HYBRID_KERNEL_SYMBOL = {"gemm_qkv_ln": "lngemm16x3_k<true, 2, false, 0, false, 2, 2>", "ffn_fused16": "ffn16_rows_k", "gemm_head_ln": "lngemm16x3_k<false, 0, false, 0, false, 1, 1>",
                        "attn_out_ffn_fused": "attnout16x3_k<false, true, true>",
                        "attn_out_fused": "attnout16x3_k<false, true, false>", "posterior_sample": "posterior_sample_k<16, true, false>"}


# ----------------------------------------------------------------------------------------- one workload, one mode
def run_mode(a, spec, sd, precision, B, steps, warmup, cond_np, rank, world, local_rank, dist, with_roofline,
             sampling=None, relation_graph=None, total=0, verified=False):
    """Times `steps` passes of one workload in one numerics mode.  B = this rank's layouts per step (weak scaling) or,
    with total > 0, the rank's shard of `total` (strong scaling).  cond_np: this rank's cond arrays (type c / refinement /
    relation) or None.  verified: greedy decoding through layout_dm_amd.verified (fast + exact re-decision).
    Returns (result dict, engine, last tokens)."""
    import torch

    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion, timestep_schedule
    from layout_dm_amd.distributed import sample_sharded, shard_range

    sampling = sampling or a.sampling
    model = HipMaskAndReplaceDiffusion(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem,
                                       d_model=spec.d_model, n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer,
                                       num_timesteps=spec.n_step, precision="fast_verified" if verified else precision,
                                       max_batch=B, chunk=a.chunk, device=local_rank, use_graph=not a.no_graph,
                                       lanes=a.lanes)
    model.load_state_dict(sd)
    eng = model.engine
    cfg = {"name": sampling, "temperature": 1.0, "top_p": 0.9, "top_k": 5, "num_timesteps": a.timesteps}
    t_model, t_post = timestep_schedule(spec.n_step, a.timesteps)
    dev = eng.device
    relation = None
    if cond_np is None:
        init = torch.full((B, eng.S), eng.mask_id, dtype=torch.int32, device=dev)
        lc_keep = None
    else:
        init = torch.from_numpy(cond_np["seq"]).to(device=dev, dtype=torch.int32).contiguous()
        cd = {"seq": init, "mask": torch.from_numpy(cond_np["mask"]).to(dev), "type": cond_np["type"]}
        if cond_np["type"] == "refinement":
            cd["weak_logits"] = torch.from_numpy(cond_np["weak_logits"]).to(dev)
        lc_keep = eng.make_cond(cd, B)
        if relation_graph is not None:
            from layout_dm_amd.synthetic import linear_bin_centres

            # hyper-parameters: the reference's defaults (hydra_configs.py:44-47); canvas = bins of the box (.5, .5, 1, 1)
            relation = eng.make_relation(relation_graph, linear_bin_centres(spec.n_bin),
                                         [spec.n_bin // 2, spec.n_bin // 2, spec.n_bin - 1, spec.n_bin - 1], 3e6, 3, B)
    tokens = torch.empty_like(init)
    seed_box = [0]
    my_lo = shard_range(total, rank, world)[0] if total else rank * B

    ev_pairs = []

    def sample_fn(first_layout, count):  # this rank's shard: `count` layouts starting at global index `first_layout`
        assert count == B and (world == 1 or first_layout == my_lo)
        tokens.copy_(init)  # inputs resident in HBM
        # HIP events around this rank's own sampling launches of EVERY timed step, on the stream they run on (the engine
        # launches on torch's current stream): the per-rank device time without the gather / barrier, and the launch
        # durations the roofline is computed from
        ev_pairs.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
        ev_pairs[-1][0].record()
        _sample(first_layout)
        ev_pairs[-1][1].record()
        return tokens

    def _sample(first_layout):
        if verified:
            model.verified.sample_loop(tokens, t_model, t_post, cond=None)
        else:
            eng.sample_loop(tokens, t_model, t_post, cfg, seed=seed_box[0], first_layout=first_layout,
                            use_graph=not a.no_graph, lc_keep=lc_keep, relation=relation)

    n_global = total if total else world * B

    def one_step(i):
        seed_box[0] = 1000 + i
        return sample_sharded(sample_fn, n_global)  # world == 1: no collective; else ONE all_gather of the tokens

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    final = None
    for i in range(warmup):
        final = one_step(i)
    sync()
    del ev_pairs[:]
    t0 = time.perf_counter()
    for i in range(steps):
        final = one_step(warmup + i)
    sync()
    dt = time.perf_counter() - t0
    ev_ms = [x.elapsed_time(y) for x, y in ev_pairs]  # (everything is complete: sync() above)
    per_rank = None
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        mine = torch.tensor([B * len(ev_ms) / max(sum(ev_ms) * 1e-3, 1e-9)],
                            dtype=torch.float64, device=dev)
        allr = torch.empty(dist.get_world_size(), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(allr, mine)
        per_rank = {"min": round(float(allr.min()), 1), "max": round(float(allr.max()), 1), "ranks": int(allr.numel()),
                    "what": "layouts/s of each rank's own sampling launches (HIP events, no gather / barrier)"}
    fin = final.cpu()
    assert fin.shape[0] == n_global
    assert (fin != eng.mask_id).all(), "sampling left [MASK] tokens"
    if cond_np is not None:  # conditioned tokens survive (strong mask)
        m = torch.from_numpy(cond_np["mask"])
        mine = fin[my_lo:my_lo + B]
        assert (mine[m] == torch.from_numpy(cond_np["seq"])[m].int()).all(), "strong-masked tokens changed"

    value = n_global * steps / dt
    flop_layout = FLOP_PER_TOKEN_STEP[spec.name] * spec.seq_len * a.timesteps
    res = {"value": round(value, 2), "unit": "layouts/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * dt / steps, 3), "dtype": DTYPE["fast_verified" if verified else precision],
           "sampling": sampling,
           "algorithmic_tflops": round(value * flop_layout / 1e12, 2),
           "frac_of_mfma_peak_whole_job": round(value * flop_layout / 1e12 / (world * PEAK_TFLOPS[precision]), 4)}
    if ev_ms:
        srt = sorted(ev_ms)
        res["timed_region_step_ms_hip_events"] = {"avg": round(sum(ev_ms) / len(ev_ms), 4), "min": round(srt[0], 4),
                                                  "median": round(srt[len(srt) // 2], 4), "max": round(srt[-1], 4),
                                                  "n": len(ev_ms)}
    if per_rank is not None:
        res["per_rank_layouts_per_s"] = per_rank
    if verified:
        res["verification"] = dict(model.verified.last_stats, verifier_engine=model.verifier)
        res["calibration"] = dict(model.verified.calibration)
    if with_roofline and rank == 0 and not verified:
        # per-kernel durations: HIP events around every launch, on the stream the kernels run on, over one more step of
        # the same workload (eager launches — events cannot bracket graph nodes; the one-launch loop of the fast mode is
        # the same launch either way)
        eng.set_profiling(True)
        tokens.copy_(init)
        eng.sample_loop(tokens, t_model, t_post, cfg, seed=999, first_layout=0, use_graph=False, lc_keep=lc_keep,
                        relation=relation)
        torch.cuda.synchronize()
        rows = eng.profile(reset=True)
        eng.set_profiling(False)
        tot = sum(r["ms"] for r in rows) or 1.0
        rows.sort(key=lambda r: -r["ms"])
        dom = rows[0]
        avg_ms = dom["ms"] / max(dom["launches"], 1)
        if dom["flops"] > 0:
            ach = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[precision]
            roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None}
        else:
            ach = dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(ach, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None}
        roof.update({"avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                     "share_of_step": round(dom["ms"] / tot, 3),
                     "algorithmic_per_launch": dom["flops"] / dom["launches"] if dom["flops"] > 0
                     else dom["bytes"] / dom["launches"],
                     "launch_configuration": "eager pass, one chunk pipeline" if dom["launches"] > 1 else
                     "the timed launch itself (one launch per sampling call)"})
        if dom["name"] == "layers_fused_loop":
            # one launch = every layout's whole loop; per (layout, reverse step) of a workgroup:
            n_wg_rounds = -(-B // 256)
            roof["ms_per_workgroup_step"] = round(avg_ms / (a.timesteps * n_wg_rounds), 4)
            if ev_ms and dom["launches"] == 1:
                # The roofline figures come from the TIMED region itself: a step there is this one launch (+ set_rng_k, 4 us),
                # bracketed by HIP events on its stream.  The profiling pass above is ONE launch after a host-side pause: the
                # chip has cooled and clocks higher for it (r04_final4: 131.0 ms against 133.3 in the timed steps, whose first
                # is the fastest for the same reason) — kept beside it, not used.
                t_avg = sum(ev_ms) / len(ev_ms)
                roof["profiling_pass"] = {"avg_launch_ms": roof["avg_launch_ms"], "achieved": roof["achieved"], "frac": roof["frac"],
                                          "what": "one more launch with per-launch events after the timed region"}
                ach_t = dom["flops"] / (t_avg * 1e-3) / 1e12
                roof.update({"avg_launch_ms": round(t_avg, 4), "achieved": round(ach_t, 2),
                             "frac": round(ach_t / PEAK_TFLOPS[precision], 4), "launches": len(ev_ms),
                             "ms_per_workgroup_step": round(t_avg / (a.timesteps * n_wg_rounds), 4),
                             "launch_configuration": "the timed launches themselves (one launch per sampling call; HIP events "
                                                     "around each timed step's launch)"})
        res["roofline"] = roof
        res["kernel_breakdown_ms"] = {r["name"]: round(r["ms"], 3) for r in rows}
        gemm_ms = sum(r["ms"] for r in rows if r["name"].startswith(GEMM_CLASSES))
        gemm_fl = sum(r["flops"] for r in rows if r["name"].startswith(GEMM_CLASSES))
        if gemm_ms > 0:
            res["gemm_mfma_utilisation"] = round(gemm_fl / (gemm_ms * 1e-3) / 1e12 / PEAK_TFLOPS[precision], 4)
    return res, eng, tokens


def tokens_sha256(a, sd, spec, rank, world, local_rank, dist):
    """sha256 of the first SHA_LAYOUTS layouts of a fixed-seed Rico25-shaped unconditional `random` run, sharded over the
    ranks like the timed workload: the same digest for every N (and every batch cut) or the sharding is wrong."""
    import hashlib

    import torch

    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion
    from layout_dm_amd.distributed import sample_sharded, shard_range

    total = max(SHA_LAYOUTS, world)
    lo, hi = shard_range(total, rank, world)
    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem,
                                   d_model=spec.d_model, n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer,
                                   num_timesteps=spec.n_step, precision=a.precision, max_batch=max(hi - lo, 1),
                                   device=local_rank)
    m.load_state_dict(sd)
    cfg = {"name": "random", "temperature": 1.0, "num_timesteps": a.timesteps}
    fn = lambda first, count: m.sample(batch_size=count, sampling_cfg=cfg, seed=SHA_SEED, first_layout=first,
                                       return_device_tensor=True)
    out = sample_sharded(fn, total)[:SHA_LAYOUTS].cpu().contiguous()
    m.engine.close()
    return hashlib.sha256(out.numpy().tobytes()).hexdigest()


def main():
    a = parse()
    maybe_spawn(a)          # `python bench.py --gpus N>1` re-executes itself under torch.distributed.run (does not return)
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.dry_run:
        return dry_run_main(a, world, rank, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    # one process per GPU: keep this rank's host threads on a contiguous slice of the cores (launch latency, and the
    # cpu_baseline leg must not spread over every rank's cores)
    if world > 1 and hasattr(os, "sched_setaffinity"):
        try:
            cores = sorted(os.sched_getaffinity(0))
            per = max(1, len(cores) // world)
            os.sched_setaffinity(0, set(cores[local_rank * per:(local_rank + 1) * per]))
        except OSError:
            pass
    dist = None
    # LDM_BENCH_FORCE_DIST=1 (dev): take the RCCL code path (init, all_gather, barrier, all_reduce) with ONE rank, so
    # the multi-GPU plumbing can be smoke-tested on a single-GPU box under torch.distributed.run --nproc-per-node 1
    force_dist = world == 1 and os.environ.get("LDM_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from layout_dm_amd import synthetic as SP
    from layout_dm_amd.distributed import shard_range

    config = a.config or (2 if world == 1 else 4)
    base = CONFIGS[config]
    a.dataset = a.dataset or base["dataset"]
    a.cond = a.cond or base["cond"]
    a.sampling = a.sampling or base["sampling"]
    spec = SP.SPECS[a.dataset]
    sd = SP.synth_state_dict(spec, seed=0, perturb=False)  # the reference's init distributions
    if a.total:  # strong scaling: `total` layouts per step, sharded by global layout index (ragged shards allowed)
        lo, hi = shard_range(a.total, rank, world)
        B, n_global = hi - lo, a.total
        if B < 1:
            raise SystemExit("--total smaller than the number of GPUs")
    else:
        B = a.batch or base["batch"]
        lo, hi, n_global = rank * B, (rank + 1) * B, world * B
    cond_global = cond_local = None
    if a.cond == "c":
        cond_global = SP.synth_cond_c(spec, n_global, seed=0)
        cond_local = {"seq": cond_global["seq"][lo:hi], "mask": cond_global["mask"][lo:hi], "type": "c"}

    res, eng, tokens = run_mode(a, spec, sd, a.precision, B, a.steps, a.warmup, cond_local, rank, world, local_rank,
                                dist, with_roofline=not a.no_roofline, total=a.total)
    desc = eng.describe()
    per_gpu = f"{B}/GPU" if not a.total else f"{a.total} total ({B} on rank 0)"
    workload = (f"BASELINE config {config}: {a.dataset} cond={a.cond} T={a.timesteps} batch={per_gpu} "
                f"sampling={a.sampling}" + (" top_p=0.9" if a.sampling == "top_p" else ""))
    dsname = {"rico25": "Rico25", "publaynet": "PubLayNet"}[a.dataset]
    cname = {"unconditional": "uncond", "c": "cond=c"}[a.cond]
    out = {
        "metric": f"layouts/sec (whole node), {dsname} {cname} T={a.timesteps}",
        "value": res["value"],
        "unit": "layouts/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": res["ms_per_step"],
        "higher_is_better": True,
        "scaling": "strong" if a.total else "weak",
        "vs_baseline": None,
        "dtype": DTYPE[a.precision],
        "data": "synthetic (random-init weights with the reference's init distributions; "
                + ("all-[MASK] start)" if a.cond == "unconditional" else
                   "cond=c sequences built as helpers/task.py:94-110, n~U{1..25} elements per layout)"),
        "config": {"workload": workload, "precision_mode": a.precision,
                   "launch": "one launch per sampling call (reverse loop resident in the layout's workgroup)"
                   if desc.get("loop") == "one_launch" else
                   ("per-step launches in hipGraphs" if not a.no_graph else "per-step launches, eager"),
                   "chunk_layouts": eng.chunk, "lanes": eng.lanes,
                   # what the LIBRARY says it runs (ldm_describe), incl. the development knobs it honoured (LDM_DEV=1
                   # only; ldm_create refuses a stray knob) — not what os.environ happens to hold
                   "library": desc,
                   "cpu_baseline_kind": None,
                   "parallelism": f"dp{world} (independent layout shards, one all_gather of the final tokens)"},
        "algorithmic_tflops": res["algorithmic_tflops"],
        "world_size_seen": dist.get_world_size() if dist is not None else 1,
        "dist_backend": dist.get_backend() if dist is not None else None,   # "nccl" = RCCL on ROCm
    }
    if "per_rank_layouts_per_s" in res:
        out["per_rank_layouts_per_s"] = res["per_rank_layouts_per_s"]
    if world > 1 and config == 4 and not a.total and not a.batch:
        out["scaling_point"] = {"workload": "BASELINE config 4 shard: rico25 uncond T=100 1024 layouts/GPU sampling=random",
                                "layouts_per_s_per_gpu": round(res["value"] / world, 2)}
    for k in ("roofline", "kernel_breakdown_ms", "gemm_mfma_utilisation", "timed_region_step_ms_hip_events"):
        if k in res:
            out[k] = res[k]

    if rank == 0:
        # the reference's own timer (test.py:194-203) also covers ids -> {bbox,label,mask} and the copy to the host:
        # reported beside the headline (which stops at tokens resident in HBM), never folded into `value`
        reps = 20
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            dec = eng.decode(tokens)
            host = {k: v.cpu() for k, v in dec.items()}
        torch.cuda.synchronize()
        dec_ms = 1e3 * (time.perf_counter() - t1) / reps
        assert host["bbox"].shape == (B, spec.max_elem, 4)
        out["decode"] = {"ms_per_batch_incl_d2h": round(dec_ms, 3), "valid_elements": int(host["mask"].sum()),
                         "layouts_per_s_incl_decode": round(n_global / ((res["ms_per_step"] + dec_ms) * 1e-3), 2)}
    eng.close()

    # every numerics mode in the one line (N=1): the bit-exact mode's throughput next to the headline's
    modes = a.modes if a.modes is not None else ("exact,split,mixed,hybrid,fast,fast_verified" if world == 1 else "")
    modes = [m for m in modes.split(",") if m and m != "none"]
    if modes:
        out["modes"] = {}
        for m in modes:
            if m == a.precision:
                r = dict(res)
            elif m == "fast_verified":
                # BASELINE config 2 verbatim: GREEDY decoding; fast + exact re-decision == the exact engine's tokens
                if a.cond != "unconditional":
                    continue
                kv = max(a.steps, 10)
                r, e2, tk = run_mode(a, spec, sd, "fast", B, kv, 1, None, rank, world, local_rank, dist,
                                     with_roofline=False, sampling="deterministic", verified=True)
                e2.close()
                g, e3, _ = run_mode(a, spec, sd, "fast", B, kv, 1, None, rank, world, local_rank, dist,
                                    with_roofline=False, sampling="deterministic")
                e3.close()
                x, e4, tx = run_mode(a, spec, sd, "exact", B, 1, 0, None, rank, world, local_rank, dist,
                                     with_roofline=False, sampling="deterministic")
                r["tokens_equal_exact_mode"] = bool(torch.equal(tk, tx))
                e4.close()
                r["plain_fast_greedy_layouts_per_s"] = g["value"]
                if config == 2 and B == 512 and a.timesteps == 100:
                    # BASELINE config 2 as written: greedy decode with the bit-exact check.  Greedy decoding of THIS
                    # diffusion keeps every token [MASK] until t <= 2 on ANY checkpoint (the posterior's mass on
                    # [MASK] is ~ (t-1)/t: SURVEY App. G), so the near-ties — and the exact re-checks — sit in the
                    # last steps; `nondegenerate` below starts the greedy loops from mid-trajectory states instead
                    out["config2_verbatim"] = {
                        "workload": "BASELINE config 2 verbatim: rico25 cond=unconditional T=100 batch=512 greedy decode",
                        "mode": "fast_verified", "value": r["value"], "unit": "layouts/s", "steps": kv,
                        "ms_per_step": r["ms_per_step"], "tokens_equal_exact_mode": r["tokens_equal_exact_mode"],
                        "exact_fraction": r["verification"]["exact_fraction"],
                        "relaunched_fraction": r["verification"]["relaunched_fraction"]}
                    if not a.no_extras:
                        r["nondegenerate"] = verified_nondegenerate(a, spec, sd, B, local_rank)
            else:
                k = a.steps if m == "fast" else 10  # exact steps take ~1.2 s each; >= 10 timed steps whatever --steps says
                r, e2, _ = run_mode(a, spec, sd, m, B, k, 1, cond_local, rank, world, local_rank, dist,
                                    with_roofline=not a.no_roofline)
                e2.close()
            r.pop("kernel_breakdown_ms", None)
            if m in ("split", "mixed", "hybrid") and rank == 0 and world == 1 and "roofline" in r and not a.no_roofline and not a.no_traffic:
                # HBM-side bytes of every kernel of the split / mixed step (r06): the mode's dominant kernel takes `traffic`
                per, note = measure_traffic_per_kernel(a.dataset, m)
                if per:
                    dom = SPLIT_KERNEL_SYMBOL.get(r["roofline"].get("kernel"))
                    if dom and m != "split":   # the two-product / plain-fp16 instantiations of the same classes
                        dom = {"mixed": MIXED_KERNEL_SYMBOL, "hybrid": HYBRID_KERNEL_SYMBOL}[m].get(r["roofline"].get("kernel"))

                    hit = [k for k in per if dom and k.startswith(dom)]
                    if hit:
                        r["roofline"]["traffic"] = per[hit[0]]["fetch_bytes_calibrated"] + per[hit[0]]["write_bytes"]
                    r["roofline"]["traffic_per_kernel"] = per
                    r["roofline"]["traffic_source"] = note + f"; FETCH_SIZE x {FETCH_CALIBRATION} (profiles/r03_fetch_size_calibration.txt)"
                else:
                    r["roofline"]["traffic_source"] = f"live PMC collection unavailable: {note}"
            out["modes"][m] = r

    # the reference-arithmetic throughput (what a checkpoint outside the fp16 engine's tolerance gets) as top-level scalars, and
    # inside `config`, where a reader that drops nested objects still sees them
    for mode_key, top_key in (("split", "reference_precision_layouts_per_s"), ("exact", "fp32_mfma_layouts_per_s")):
        v = out.get("modes", {}).get(mode_key, {}).get("value") if mode_key != a.precision else res["value"]
        if v is not None:
            out[top_key] = v
            out["config"][top_key] = v
    if world == 1 and not a.no_extras and not a.total and not a.batch:
        out["configs"] = extras(a, SP, config, rank, world, local_rank, dist)
        out["fid_features"] = fid_timing(SP, local_rank)
        out["layout_metrics"] = layout_metrics_timing(local_rank)
        if "4" in out["configs"]:   # the per-GPU workload of every N > 1 run, under the same key there
            out["scaling_point"] = {"workload": "BASELINE config 4 shard: rico25 uncond T=100 1024 layouts/GPU sampling=random",
                                    "layouts_per_s_per_gpu": out["configs"]["4"]["value"]}
        if config == 2 and a.cond == "unconditional":
            out["batch_shapes"] = batch_shapes(a, spec, sd, rank, world, local_rank, dist)
            # what a user gets on a checkpoint that is NOT init-like: LayoutDM's default precision="auto" on the trained-like
            # weight points (sigma 0.06 / 0.15, LayerNorm gains 1 +- 0.5, outlier channels, AdaLN x5): which engine the
            # load-time measurement selects and that engine's layouts/s on the headline workload — next to the plain fp16
            # engine on the same weights (a power-bound kernel's clock depends on operand statistics), which auto refuses
            # where its logits error is outside 1e-3
            out["weight_sensitivity"] = auto_on_trained_like(a, SP, spec, B, res["value"], rank, world, local_rank, dist)
    if True:   # (r06: also with --no-extras — the one-rank RCCL test of the GPU suite compares it with the plain run's; one short call)
        out["tokens_sha256"] = {"sha256": tokens_sha256(a, SP.synth_state_dict(SP.SPECS["rico25"], seed=0), SP.SPECS["rico25"],
                                                        rank, world, local_rank, dist),
                                "of": f"first {SHA_LAYOUTS} layouts, Rico25 uncond random T={a.timesteps}, seed {SHA_SEED}, "
                                      f"precision {a.precision}; must not depend on n_gpus"}

    if rank == 0 and world == 1 and not a.no_roofline and not a.no_traffic and "roofline" in out:
        sym = KERNEL_SYMBOL.get(out["roofline"]["kernel"])
        if sym:
            vals, note = measure_traffic(sym, a.dataset, a.precision)
            if vals:
                out["roofline"]["traffic"] = int((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)
                out["roofline"]["traffic_detail"] = traffic_detail(vals, note, out["roofline"]["kernel"], spec)
            else:
                out["roofline"]["traffic_detail"] = {"source": f"live PMC collection unavailable: {note}"}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, sd, a.timesteps, a.sampling, cond_global, a.cpu_budget)
        out["config"]["cpu_baseline_kind"] = out["cpu_baseline"]["kind"]
        out["cpu_baseline_kind"] = out["cpu_baseline"]["kind"]
    ws = out.get("weight_sensitivity", {})
    for point in ("mid", "wide", "fitted"):   # scalars a reader that drops nested objects still sees
        if ws.get(point):
            for key, val in ((f"auto_selected_{point}", ws[point]["auto_selected"]), (f"auto_layouts_per_s_{point}", ws[point]["value"])):
                out[key] = val              # top level ...
                out["config"][key] = val    # ... and inside `config`
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def batch_shapes(a, spec, sd, rank, world, local_rank, dist):
    """Batch quantisation of the one-launch loop: one workgroup per layout on 256 compute units, so a
    call costs whole rounds of 256 layouts — B = 300 or 488 cost what 512 costs, 640 what 768 costs.  The reference's
    default run (num_uncond_samples=1000, max_batch_size=512) is a 512 + a 488 batch (test.py / data/util.py:301-307)."""
    out = {"rule": "a call costs ceil(B / round) rounds (ldm_describe: round, batch_rule); HipMaskAndReplaceDiffusion.sample cuts "
                   "batches above max_batch into whole rounds (diffusion.batch_cuts)"}
    for B in (300, 488, 640):
        r, e, _ = run_mode(a, spec, sd, a.precision, B, 3, 1, None, rank, world, local_rank, dist, with_roofline=False)
        rnd = int(e.describe().get("round", "256"))
        e.close()
        rounds = -(-B // rnd)
        out[str(B)] = {"value": r["value"], "unit": "layouts/s", "ms_per_step": r["ms_per_step"], "steps": 3, "rounds": rounds,
                       "round": rnd, "ms_per_workgroup_step": round(r["ms_per_step"] / (a.timesteps * rounds), 4),
                       "fill_of_last_round": round((B - (rounds - 1) * rnd) / rnd, 3)}
    return out


def auto_on_trained_like(a, SP, spec, B, headline, rank, world, local_rank, dist):
    import torch

    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion

    out = {"weights": "trained_like points of layout_dm_amd/synthetic.py (mid: sigma 0.06, wide: sigma 0.15)",
           "workload": f"headline workload ({B} layouts, T={a.timesteps}, sampling={a.sampling}), LayoutDM's default precision='auto'"}
    cfg = {"name": a.sampling, "temperature": 1.0, "top_p": 0.9, "num_timesteps": a.timesteps}
    # r06: a checkpoint that was actually TRAINED — the reference model fitted by the reference's own loss /
    # optimizer for 1 500 steps on structured synthetic layouts (oracle/make_trained_fixture.py -> oracle/_fit/rico25_fitted.npz, a
    # build output that travels with the snapshot like oracle/_ref/; only the weight FILE is read here, no oracle code)
    fitted = os.path.join(ROOT, "oracle", "_fit", "rico25_fitted.npz")
    points = ("mid", "wide") + (("fitted",) if spec.name == "rico25" and os.path.exists(fitted) else ())
    if "fitted" in points:
        out["fitted"] = None
        out["weights"] += "; fitted: oracle/_fit/rico25_fitted.npz (the reference model trained with the reference's own step)"
    for point in points:
        if point == "fitted":
            import numpy as np

            with np.load(fitted) as wf:
                sdw = {k: wf[k] for k in wf.files}
        else:
            sdw = SP.trained_like_state_dict(spec, point, seed=2)
        m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem,
                                       d_model=spec.d_model, n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer,
                                       num_timesteps=spec.n_step, precision="auto", max_batch=B, device=local_rank)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        m.load_state_dict(sdw)
        torch.cuda.synchronize()
        load_s = time.perf_counter() - t0
        k = 3
        m.sample(batch_size=B, sampling_cfg=cfg, seed=1, return_device_tensor=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k):
            tk = m.sample(batch_size=B, sampling_cfg=cfg, seed=2 + i, return_device_tensor=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / k
        assert tk.shape == (B, spec.seq_len) and bool((tk != m.mask_id).all())
        cal = m.calibration
        entry = {"auto_selected": m.selected_precision, "value": round(B / dt, 2), "unit": "layouts/s", "steps": k,
                 "ms_per_step": round(1e3 * dt, 3), "fast_engine_err_rel_measured_at_load": cal.get("err_rel"),
                 "hybrid_engine_err_rel_measured_at_load": m.selection_report.get("hybrid_logits_err_rel"),
                 "mixed_engine_err_rel_measured_at_load": m.selection_report.get("mixed_logits_err_rel"),
                 "tolerance": m.auto_tolerance, "verifier_check": m.verifier_check, "load_and_calibrate_s": round(load_s, 2),
                 "ratio_to_headline": round(B / dt / headline, 4)}
        m.close()
        # the plain fp16 engine on the same weights (what precision="fast" would run, refused by auto where outside 1e-3)
        rw, ew, _ = run_mode(a, spec, sdw, "fast", B, 3, 1, None, rank, world, local_rank, dist, with_roofline=False)
        ew.close()
        entry["plain_fast_engine"] = {"value": rw["value"], "ratio_to_headline": round(rw["value"] / headline, 4),
                                      "inside_tolerance": bool(cal.get("err_rel", 1.0) <= m.auto_tolerance)}
        out[point] = entry
    return out


def verified_nondegenerate(a, spec, sd, B, local_rank):
    """fast_verified where the marks are NOT confined to a greedy run's last steps by construction: greedy free-running
    loops started from the states a stochastic (`random`) run visits at step 20 / 50 / 80, B layouts; each checked
    token for token against the exact engine's greedy loop from the same state."""
    import torch

    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion, timestep_schedule

    m = HipMaskAndReplaceDiffusion(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem,
                                   d_model=spec.d_model, n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer,
                                   num_timesteps=spec.n_step, precision="fast_verified", max_batch=B, device=local_rank)
    m.load_state_dict(sd)
    vg, fa = m.verified, m.verified.fast
    # the yardstick is the fp32-MFMA engine, whatever engine the verifier itself uses (default: split)
    mx = HipMaskAndReplaceDiffusion(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem,
                                    d_model=spec.d_model, n_head=spec.n_head, d_ff=spec.d_ff, n_layer=spec.n_layer,
                                    num_timesteps=spec.n_step, precision="exact", max_batch=B, device=local_rank)
    mx.load_state_dict(sd)
    ex = mx.engine
    t_model, t_post = timestep_schedule(spec.n_step, a.timesteps)
    n = len(t_model)
    greedy = {"name": "deterministic"}
    tok = torch.full((B, fa.S), fa.mask_id, dtype=torch.int32, device=fa.device)
    _, inter = fa.sample_loop(tok, t_model, t_post, {"name": "random", "temperature": 1.0}, seed=77, intermediates=True)
    inter = inter.clone()
    out = {"what": f"greedy loops from the states of a `random` run (seed 77) at step i0, {B} layouts; layouts/s-equivalent "
                   f"= {B} x (steps run) / {n} / time", "verifier_engine": m.verifier, "calibration": dict(vg.calibration)}
    reps = 3
    for i0 in (20, 50, 80):
        start = inter[i0 - 1].clone()
        want = ex.sample_loop(start.clone(), t_model[i0:], t_post[i0:], greedy)[0].clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ex.sample_loop(start.clone(), t_model[i0:], t_post[i0:], greedy)
        torch.cuda.synchronize()
        dt_exact = time.perf_counter() - t0
        vg.sample_loop(start.clone(), t_model[i0:], t_post[i0:])       # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            got = vg.sample_loop(start.clone(), t_model[i0:], t_post[i0:])[0]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        st = vg.last_stats
        out[f"from_step_{i0}"] = {
            "value": round(B * (n - i0) / n / dt, 1), "unit": "layouts/s-equivalent", "steps_run": n - i0, "reps": reps,
            "exact_engine_layouts_per_s_equivalent": round(B * (n - i0) / n / dt_exact, 1),
            "tokens_equal_exact_mode": bool(torch.equal(got, want)),
            "exact_fraction": round(st["exact_fraction"], 5), "relaunched_fraction": round(st["relaunched_fraction"], 6),
            "marked_layout_steps": st["marked_layout_steps"], "mismatch_layout_steps": st["mismatch_layout_steps"],
            "fast_passes": st["fast_passes"]}
    fa.close()
    m.verified.exact.close()
    ex.close()
    return out


def extras(a, SP, headline_config, rank, world, local_rank, dist):
    """The other BASELINE configurations and SURVEY section 8f's rows, a few steps each (N = 1)."""
    import copy
    import dataclasses

    out = {}
    for key in ("3", "4", "refinement", "relation", "5_refinement_T200", "5_relation_T200"):
        if key == str(headline_config):
            continue
        b = copy.copy(a)
        graph = None
        if key.startswith("5_"):
            # BASELINE config 5's shape: a T = 200 model (schedule buffers / AdaLN tables of 200 timesteps; base.py:310-311
            # needs num_timesteps <= the model's) sampled for 200 steps with cond=refinement / cond=relation, 512 layouts
            b.dataset, b.sampling, b.timesteps, B = "rico25", "random", 200, 512
            spec = dataclasses.replace(SP.SPECS["rico25"], n_step=200)
            if key == "5_refinement_T200":
                cond = SP.synth_cond_refinement(spec, B, seed=0)
            else:
                cond, graph = SP.synth_cond_relation(spec, B, seed=0)
            what = (f"BASELINE config 5 shape: rico25 cond={'refinement' if graph is None else 'relation'} T=200 (T=200 "
                    f"model, random-init weights) batch={B} sampling=random")
            sd = SP.synth_state_dict(spec, seed=0, perturb=False)
            r, e, _ = run_mode(b, spec, sd, a.precision, B, 3, 1, cond, rank, world, local_rank, dist,
                               with_roofline=not a.no_roofline, relation_graph=graph)
            e.close()
            entry = {"workload": what, "value": r["value"], "unit": "layouts/s", "ms_per_step": r["ms_per_step"], "steps": 3,
                     "algorithmic_tflops": r["algorithmic_tflops"]}
            if "roofline" in r:
                entry["dominant_kernel"] = {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "share_of_step")}
            out[key] = entry
            continue
        if key in ("3", "4"):
            base = CONFIGS[int(key)]
            b.dataset, b.cond, b.sampling, B = base["dataset"], base["cond"], base["sampling"], base["batch"]
            spec = SP.SPECS[b.dataset]
            cond = dict(SP.synth_cond_c(spec, B, seed=0), type="c") if b.cond == "c" else None
            what = f"BASELINE config {key}: {b.dataset} cond={b.cond} T={a.timesteps} batch={B} sampling={b.sampling}"
        else:
            b.dataset, b.sampling, B = "rico25", "random", 512
            spec = SP.SPECS["rico25"]
            if key == "refinement":
                cond = SP.synth_cond_refinement(spec, B, seed=0)
                what = (f"rico25 cond=refinement (prior table as helpers/task.py:154-224, lambda 3, offset 0.1) T={a.timesteps} "
                        f"batch={B} sampling=random")
            else:
                cond, graph = SP.synth_cond_relation(spec, B, seed=0)
                what = (f"rico25 cond=relation (graphs as data/util.py:111-177, edge_ratio 0.1; relation_lambda 3e6, "
                        f"3 updates, t >= 10) T={a.timesteps} batch={B} sampling=random")
        sd = SP.synth_state_dict(spec, seed=0, perturb=False)
        r, e, _ = run_mode(b, spec, sd, a.precision, B, 3, 1, cond, rank, world, local_rank, dist,
                           with_roofline=not a.no_roofline, relation_graph=graph)
        e.close()
        entry = {"workload": what, "value": r["value"], "unit": "layouts/s", "ms_per_step": r["ms_per_step"], "steps": 3}
        if "roofline" in r:
            entry["dominant_kernel"] = {k: r["roofline"][k] for k in ("kernel", "frac", "avg_launch_ms", "share_of_step")}
            entry["kernel_breakdown_ms"] = r["kernel_breakdown_ms"]
        out[key] = entry
    return out


def fid_timing(SP, local_rank):
    """FIDNetV3.extract_features (trainer/fid/model.py:123-164) on 512 x 25 random elements, random weights."""
    import numpy as np
    import torch

    from layout_dm_amd.fid import FIDNetV3

    B, N, L = 512, 25, 25
    m = FIDNetV3(num_label=L, max_bbox=N, device=local_rank)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in SP.synth_fid_state_dict(L, seed=0, max_bbox=N).items()})
    rng = np.random.default_rng(0)
    n = rng.integers(1, N + 1, size=B)
    dev = torch.device("cuda", local_rank)
    bbox = torch.from_numpy(rng.random((B, N, 4)).astype(np.float32)).to(dev)
    label = torch.from_numpy(rng.integers(0, L, size=(B, N))).to(dev)
    pm = torch.from_numpy(~(np.arange(N)[None] < n[:, None])).to(dev)
    for _ in range(3):
        f = m.extract_features(bbox, label, pm)
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        f = m.extract_features(bbox, label, pm)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    assert f.shape == (B, 256) and bool(torch.isfinite(f).all())
    return {"value": round(B / (ms * 1e-3), 1), "unit": "layouts/s", "ms_per_batch": round(ms, 4), "batch": B,
            "what": "fid_features_k: FIDNetV3 encoder (26 x 256, 4 layers) per layout, fp32, inputs resident in HBM"}


def layout_metrics_timing(local_rank):
    """compute_alignment + compute_overlap (trainer/helpers/metric.py:98-203) of 512 x 25 random elements, inputs resident."""
    import numpy as np
    import torch

    from layout_dm_amd.metrics import layout_metrics

    B, N = 512, 25
    rng = np.random.default_rng(0)
    n = rng.integers(1, N + 1, size=B)
    dev = torch.device("cuda", local_rank)
    bbox = torch.from_numpy(rng.random((B, N, 4)).astype(np.float32)).to(dev)
    mask = torch.from_numpy(np.arange(N)[None] < n[:, None]).to(dev).to(torch.uint8)
    for _ in range(3):
        o = layout_metrics(bbox, mask)
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        o = layout_metrics(bbox, mask)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / reps
    assert o.shape == (B, 6) and bool(torch.isfinite(o).all())
    return {"value": round(B / (ms * 1e-3), 1), "unit": "layouts/s", "ms_per_batch": round(ms, 4), "batch": B,
            "what": "layout_metrics_k: alignment (3 scores) + overlap (3 scores) per layout, fp32, one wavefront per layout, "
                    "inputs resident in HBM (call + stream sync included)"}


def traffic_detail(vals, note, kernel, spec):
    """Raw and calibrated HBM bytes per launch of the dominant kernel.  Calibration (profiles/r03_fetch_size_calibration.txt,
    tools/microbench/rowio under --pmc): FETCH_SIZE reports half the bytes of 16 B/lane coalesced streams on gfx950
    (global_load_dwordx4 and global_load_lds_dwordx4 alike), WRITE_SIZE is exact."""
    fetch, write = vals["FETCH_SIZE"] * 1024, vals["WRITE_SIZE"] * 1024
    d = {"fetch_bytes_raw": int(fetch), "write_bytes_raw": int(write),
         "fetch_bytes_calibrated": int(fetch * FETCH_CALIBRATION), "write_bytes_calibrated": int(write),
         "calibration": f"FETCH_SIZE x {FETCH_CALIBRATION} (profiles/r03_fetch_size_calibration.txt), WRITE_SIZE x 1",
         "source": note}
    if kernel == "layers_fused_loop":
        # tools/pmc_probe.py: 512 layouts x 100 reverse steps in ONE launch = the timed launch of config 2, i.e. 200
        # workgroup-round-steps of 256 layouts.  FETCH_SIZE counts the L2's fabric-side requests (Infinity-Cache hits
        # included, MI355X_MICROARCH.md), so it measures what misses the 8 per-XCD L2s, not HBM: the 23.4 MB of weight
        # images live in the 256-MiB Infinity Cache across steps; each XCD's 32 workgroups stream them through their 4-MiB
        # L2 once per step when they run in step with each other.
        steps, layouts = 100, 512
        rounds = -(-layouts // 256)
        per = (fetch * FETCH_CALIBRATION + write) / (steps * rounds)
        d["probe_launch"] = f"{layouts} layouts x {steps} reverse steps = {steps * rounds} rounds of 256 workgroup-steps"
        d["bytes_per_256_workgroup_steps_calibrated"] = int(per)
        d["weights_once_per_xcd_bytes"] = int(8 * WEIGHT_IMAGE_BYTES)
        d["ratio_to_weights_once_per_xcd"] = round(per / (8 * WEIGHT_IMAGE_BYTES), 2)
        d["algorithmic_hbm_bytes_per_launch"] = int(WEIGHT_IMAGE_BYTES + layouts * spec.seq_len * 8)
        d["weight_stream_bytes_requested_from_l2_per_launch"] = int(WEIGHT_IMAGE_BYTES * layouts * steps)
    return d


if __name__ == "__main__":
    main()
