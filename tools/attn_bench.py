import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd.binding import load_library
lib = load_library()
lib.ldm_dev_bench_attn.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float)]
ms = C.c_float()
for B in (256, 512):
    rc = lib.ldm_dev_bench_attn(B, 20, C.byref(ms))
    print(f"attn B={B} abl={os.environ.get('LDM_ATTN_ABL','0')}: {ms.value*1000:.1f} us rc={rc}", flush=True)
