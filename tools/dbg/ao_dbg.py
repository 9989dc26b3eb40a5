import ctypes, sys, torch
sys.path.insert(0, '/root/repo')
from layout_dm_amd import binding
lib = binding.load_library()
fn = lib.ldm_dev_attnout_check
fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
fn.restype = ctypes.c_int
torch.cuda.init()
for mask, seed in ((0, 8), (0, 9), (15, 8), (0, 10), (0, 11), (0, 12)):
    err = (ctypes.c_double * 25)()
    rc = fn(2, 125, 1.0, seed, err, mask)
    print(f"zero_lo={mask:2d} seed={seed} rc={rc} err={err[0]:.3e} per wave:", " ".join(f"{err[6+i]/err[2]:.1e}" for i in range(4)), flush=True)
