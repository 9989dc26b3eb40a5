"""Dev tool: launch the fused FFN a few times (for rocprofv3 --pmc passes)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd.binding import load_library
lib = load_library()
lib.ldm_dev_bench_gemm.argtypes = [C.c_int] * 5 + [C.POINTER(C.c_float)]
ms = C.c_float()
lib.ldm_dev_bench_gemm(32000, 464, 464, 101, 3, C.byref(ms))
print(ms.value)
