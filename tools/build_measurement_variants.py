"""Dev tool: MEASUREMENT builds of the library for same-box A/Bs of the loop kernel's weight stream (tools/gpu_calls/r04_call26.sh,
) — the shipped objects with kernels_stack.hip recompiled under other defines, loaded only through LDM_HIP_LIB:
  libldm_hip_abl_ffnwin1.so  -DLDM_ABL_FFN_WINDOW=1   FFN stream re-reads a 64-KiB window (L2-served; WRONG numbers)
  libldm_hip_abl_ffnwin2.so  -DLDM_ABL_FFN_WINDOW=2   ... a 16-KiB window (L1-served; WRONG numbers)
  libldm_hip_abl_lngemm.so   -DLDM_LNGEMM_ABL_BUILD   kernels_lngemm.hip with its compile-time timing variants (LDM_LNGEMM_ABL=mask:
                                                      2 no fragment reads, 4 no weight DMA, 8 no epilogue, 16 no epilogue stores, 32 stores to tile 0 columns; WRONG numbers)"""
# libldm_hip_abl_noslp.so: kernels_stack.hip under -fno-slp-vectorize (no SLP-packed v_pk_*_f32 beside the MFMAs; same numerics)
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd import build as B

VARIANTS = {"ffnwin1": ("kernels_stack.hip", ["-DLDM_ABL_FFN_WINDOW=1"]), "ffnwin2": ("kernels_stack.hip", ["-DLDM_ABL_FFN_WINDOW=2"]),
            "lngemm": ("kernels_lngemm.hip", ["-DLDM_LNGEMM_ABL_BUILD"]),
            "noslp": ("kernels_stack.hip", ["-fno-slp-vectorize"])}

B.build()
cc = B.hipcc()
bdir = os.path.join(B.HERE, "build")
for name in (sys.argv[1:] or list(VARIANTS)):
    src, defs = VARIANTS[name]
    obj = os.path.join(bdir, f"{src.rsplit('.', 1)[0]}_abl_{name}.o")
    subprocess.run([cc, "-x", "hip", *B.FLAGS, *defs, "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
    objs = [os.path.join(bdir, s.rsplit(".", 1)[0] + ".o") for s in B.SOURCES if s != src] + [obj]
    out = os.path.join(B.HERE, f"libldm_hip_abl_{name}.so")
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True)
    print(out)
