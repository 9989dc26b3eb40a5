#!/usr/bin/env python
"""bench.py — layouts/sec of the LayoutDM sampling hot path on MI355X.

One "step" = one full pass of the hot path over one batch: the T=100-step reverse loop
(denoiser forward + posterior + categorical draw per step) for `--batch` Rico25-shaped layouts per
GPU, starting from the all-[MASK] state resident in HBM and ending with the final int32 tokens
(gathered to every rank with ONE RCCL all_gather when N>1).  Workload = BASELINE.json configs[1]
(Rico25, cond=unconditional, T=100, batch=512 per GPU, random-init synthetic weights).

Contract: `python bench.py --gpus N --steps K --warmup W` ; for N>1 launched under
torch.distributed.run (one rank per GPU).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_TOKEN_STEP = {  # 4 x [QKV + attn + out-proj + FFN] + head(2*464*C)
    "rico25": 21_740_256,
    "publaynet": 21_721_696,
}
PEAK_TFLOPS = {"exact": 157.3, "fast": 2500.0, "split": 2500.0}  # MI355X_MICROARCH.md (dense MFMA)
DTYPE = {"exact": "f32", "fast": "f16 (f32 accumulate)", "split": "f16x3 split (f32 accumulate)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=512, help="layouts per GPU per step")
    ap.add_argument("--dataset", default="rico25", choices=["rico25", "publaynet"])
    ap.add_argument("--timesteps", type=int, default=100)
    ap.add_argument("--precision", default=os.environ.get("LDM_BENCH_PRECISION", "fast"),
                    choices=["exact", "fast", "split"])
    ap.add_argument("--sampling", default="random", choices=["random", "deterministic", "top_p", "gumbel"])
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=16, help="layouts in the bounded CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(spec, sd, T, sampling, batch, budget_s=20.0, max_threads=32):
    """The oracle restatement of the reference's CPU path (kind = "port"), timed on this host on a
    BOUNDED sample of the same workload: `batch` layouts taken through as many of the T reverse
    steps as fit in ~budget_s seconds (every step costs the same: one denoiser forward + posterior
    + draw), fp32, torch CPU.  layouts/s = batch / (mean step time x T)."""
    import torch

    from oracle import restatement as R

    W = R.as_torch_weights(sd)
    cores = min(os.cpu_count() or 1, max_threads)  # small-batch CPU inference degrades beyond ~32 threads
    torch.set_num_threads(cores)
    cfg = {"name": sampling, "temperature": 1.0, "top_p": 0.9}
    steps = R.timestep_list(spec.n_step, T)
    tokens = torch.full((batch, spec.seq_len), spec.mask_id, dtype=torch.long)
    R.single_step(W, spec, tokens[:1], steps[0], cfg, uniforms=R.token_uniforms(0, 0, 1, spec.seq_len, 0)[..., 0])
    done, t0 = 0, time.time()
    for i, t in enumerate(steps):
        u = R.token_uniforms(0, 0, batch, spec.seq_len, i)[..., 0] if sampling != "deterministic" else None
        tokens = R.single_step(W, spec, tokens, t, cfg, uniforms=u)
        done += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    per_step = dt / done
    return {"value": round(batch / (per_step * T), 3), "unit": "layouts/s", "cores": cores, "kind": "port",
            "sample": f"{batch} layouts x {done} of T={T} reverse steps ({dt:.1f} s; per-step cost is uniform), "
                      f"oracle/restatement.py, torch CPU fp32, {cores} threads"}


def main():
    a = parse()
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if a.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    # LDM_BENCH_FORCE_DIST=1 (dev): take the RCCL code path (init, all_gather, barrier, all_reduce) with ONE rank, so
    # the multi-GPU plumbing can be smoke-tested on a single-GPU box under torch.distributed.run --nproc-per-node 1
    force_dist = world == 1 and os.environ.get("LDM_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
    if world > 1 or force_dist:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from layout_dm_amd.diffusion import HipMaskAndReplaceDiffusion, timestep_schedule
    from layout_dm_amd import synthetic as SP
    synth = SP

    spec = SP.SPECS[a.dataset]
    sd = synth.synth_state_dict(spec, seed=0, perturb=False)  # the reference's init distributions
    B = a.batch
    model = HipMaskAndReplaceDiffusion(n_category=spec.n_category, n_bin=spec.n_bin, max_elem=spec.max_elem,
                                       d_model=spec.d_model, n_head=spec.n_head, d_ff=spec.d_ff,
                                       n_layer=spec.n_layer, num_timesteps=spec.n_step, precision=a.precision,
                                       max_batch=B, chunk=a.chunk, device=local_rank, use_graph=not a.no_graph)
    model.load_state_dict(sd)
    eng = model.engine
    cfg = {"name": a.sampling, "temperature": 1.0, "top_p": 0.9, "num_timesteps": a.timesteps}
    t_model, t_post = timestep_schedule(spec.n_step, a.timesteps)
    dev = eng.device
    init = torch.full((B, eng.S), eng.mask_id, dtype=torch.int32, device=dev)
    tokens = torch.empty_like(init)
    gathered = torch.empty((world * B, eng.S), dtype=torch.int32, device=dev) if dist is not None else None

    def one_step(i):
        tokens.copy_(init)  # inputs resident in HBM
        eng.sample_loop(tokens, t_model, t_post, cfg, seed=1000 + i, first_layout=rank * B,
                        use_graph=not a.no_graph)
        if dist is not None:
            dist.all_gather_into_tensor(gathered, tokens)  # the single RCCL collective of the path

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        one_step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        one_step(a.warmup + i)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    final = (gathered if dist is not None else tokens).cpu()
    assert (final != eng.mask_id).all(), "sampling left [MASK] tokens"

    layouts = world * B * a.steps
    value = layouts / dt
    flop_layout = FLOP_PER_TOKEN_STEP[a.dataset] * spec.seq_len * a.timesteps
    out = {
        "metric": "layouts/sec (whole node), Rico25 uncond T=100",
        "value": round(value, 2),
        "unit": "layouts/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(1e3 * dt / a.steps, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE[a.precision],
        "data": "synthetic (random-init weights with the reference's init distributions, all-[MASK] start)",
        "config": {"workload": f"{a.dataset} cond=unconditional T={a.timesteps} batch={B}/GPU sampling={a.sampling}",
                   "precision_mode": a.precision, "hipgraph": not a.no_graph, "chunk_layouts": min(eng.cfg.chunk or 512, B),
                   "parallelism": f"dp{world} (independent layout shards, one all_gather of the final tokens)"},
        "algorithmic_tflops": round(value * flop_layout / 1e12, 2),
    }

    if rank == 0 and not a.no_roofline:
        # per-kernel durations: HIP events around every launch, on the stream the kernels run on,
        # over one more step of the same workload (eager launches — events cannot bracket graph nodes)
        eng.set_profiling(True)
        tokens.copy_(init)
        eng.sample_loop(tokens, t_model, t_post, cfg, seed=999, first_layout=0, use_graph=False)
        torch.cuda.synchronize()
        rows = eng.profile(reset=True)
        eng.set_profiling(False)
        tot = sum(r["ms"] for r in rows) or 1.0
        rows.sort(key=lambda r: -r["ms"])
        dom = rows[0]
        is_gemm = dom["flops"] > 0
        avg_ms = dom["ms"] / max(dom["launches"], 1)
        if is_gemm:
            ach = dom["flops"] / dom["launches"] / (avg_ms * 1e-3) / 1e12
            peak = PEAK_TFLOPS[a.precision]
            roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                    "share_of_step": round(dom["ms"] / tot, 3)}
        else:
            ach = dom["bytes"] / dom["launches"] / (avg_ms * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": dom["name"], "achieved": round(ach, 1), "peak": 8000.0,
                    "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": None,
                    "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                    "share_of_step": round(dom["ms"] / tot, 3)}
        # HBM traffic of the dominant kernel: PMC counters cannot be read from inside the process, so the
        # value comes from the committed rocprofv3 --pmc pass (profiles/r01_traffic.json), scaled to this
        # launch's row count; null when no measurement exists for the kernel.
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json"))).get(dom["name"])
            if tr and a.precision == "fast":
                rows_per_launch = min(eng.cfg.chunk or 512, B) * eng.S
                roof["traffic"] = int((tr["fetch_kb"] + tr["write_kb"]) * 1024 * rows_per_launch / tr["M"])
                roof["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 FETCH_SIZE+WRITE_SIZE, raw)"
        except Exception:
            pass
        out["roofline"] = roof
        out["kernel_breakdown_ms"] = {r["name"]: round(r["ms"], 3) for r in rows}
        gemm_ms = sum(r["ms"] for r in rows if r["name"].startswith(("gemm", "ffn", "qkv")))
        gemm_fl = sum(r["flops"] for r in rows if r["name"].startswith(("gemm", "ffn", "qkv")))
        if gemm_ms > 0:
            out["gemm_mfma_utilisation"] = round(gemm_fl / (gemm_ms * 1e-3) / 1e12 / PEAK_TFLOPS[a.precision], 4)
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
        step_ms = 1e3 * dt / a.steps
        out["decode"] = {"ms_per_batch_incl_d2h": round(dec_ms, 3), "valid_elements": int(host["mask"].sum()),
                         "layouts_per_s_incl_decode": round(world * B / ((step_ms + dec_ms) * 1e-3), 2)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spec, sd, a.timesteps, a.sampling, a.cpu_batch)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
