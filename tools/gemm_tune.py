"""Dev tool: time the fp16 tile configurations of the generic fast path (gemm16) on the denoiser's shapes (MI355X only)."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd.binding import load_library

lib = load_library()
lib.ldm_dev_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
shapes = {"qkv": (1536, 464), "attn_out": (464, 512), "ffn1": (1856, 464), "ffn2": (464, 1856), "head": (155, 464)}
cfgs = [5, 6]
out = {}
for name, (N, K) in shapes.items():
    for cfg in cfgs:
        ms = C.c_float()
        rc = lib.ldm_dev_bench_gemm(M, N, K, cfg, 20, C.byref(ms))
        tf = 2.0 * M * N * K / (ms.value * 1e-3) / 1e12 if rc == 0 and ms.value > 0 else 0.0
        out[f"{name}/cfg{cfg}"] = {"ms": round(ms.value, 4), "TF": round(tf, 1), "rc": rc}
        print(f"{name:9s} M={M} N={N} K={K} cfg{cfg}: {ms.value:.4f} ms  {tf:7.1f} TF  rc={rc}", flush=True)
json.dump(out, open(os.path.join("gpurun_out", f"gemm_tune_M{M}.json"), "w"), indent=1)
