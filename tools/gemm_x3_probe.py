"""Dev tool (GPU box): what bounds the split GEMM (gemm16x3_k, 256 x 256 tiles, 2-stage ring)?  Times the kernel on the denoiser's four shapes
(M = 32 000 rows = one 256-layout chunk) next to three ablations of itself: operand fills + barriers only, fragment reads +
MFMAs + barriers only, fills + MFMAs without the fragment reads.  A second argument list of row counts (e.g. 5376 10752:
126 / 252 tiles of the QKV shape = one round on half / all of the CUs) separates a per-CU from a chip-wide fill limit."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd.binding import load_library

lib = load_library()
lib.ldm_dev_bench_gemm_x3.argtypes = [C.c_int] * 6 + [C.POINTER(C.c_float)]
Ms = [int(x) for x in sys.argv[1:]] or [32000]
shapes = {"qkv": (1392, 464, 0), "attn_out": (464, 464, 0), "ffn1": (1856, 464, 1), "ffn2": (464, 1856, 0)}
names = {0: "kernel", 1: "fills only", 2: "reads+MFMA only", 3: "fills+MFMA (no frag reads)"}
for M in Ms:
    for name, (N, K, c16) in shapes.items():
        row = []
        for abl in (0, 1, 2, 3):
            ms = C.c_float()
            rc = lib.ldm_dev_bench_gemm_x3(M, N, K, abl, c16, 20, C.byref(ms))
            row.append(f"{names[abl]} {ms.value * 1e3:7.1f} us" if rc == 0 else f"{names[abl]} rc={rc}")
        flops = 2.0 * M * N * K
        tiles = -(-M // 256) * -(-N // 256)
        print(f"{name:9s} M={M} N={N} K={K} ({tiles} tiles): " + " | ".join(row) + f"   (MFMA-bound: {3 * flops / 2.5e15 * 1e6:.1f} us)", flush=True)
