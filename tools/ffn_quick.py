"""Dev tool (GPU box): time the fused FFN kernel alone (dev hook cfg 101) under the current LDM_FFN_* env."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd.binding import load_library
lib = load_library()
lib.ldm_dev_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
ms = C.c_float()
M = int(os.environ.get("PROBE_M", "64000"))
rc = lib.ldm_dev_bench_gemm(M, 464, 464, 101, 20, C.byref(ms))
fl = 2.0 * 2 * M * 464 * 1856
print(f"ffn M={M} dbg={os.environ.get('LDM_FFN_DBG','0')} var={os.environ.get('LDM_FFN_VAR','0')}: "
      f"{ms.value*1000:.1f} us  {fl/ms.value/1e9:.0f} TF rc={rc}", flush=True)
