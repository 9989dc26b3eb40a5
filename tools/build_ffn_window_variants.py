"""Dev tool: MEASUREMENT builds of the library for same-box A/Bs of the loop kernel's weight stream (tools/gpu_calls/r04_call26.sh,
) — the shipped objects with kernels_stack.hip recompiled under other defines, loaded only through LDM_HIP_LIB:
  libldm_hip_abl_ffnwin1.so  -DLDM_ABL_FFN_WINDOW=1   FFN stream re-reads a 64-KiB window (L2-served; WRONG numbers)
  libldm_hip_abl_ffnwin2.so  -DLDM_ABL_FFN_WINDOW=2   ... a 16-KiB window (L1-served; WRONG numbers)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from layout_dm_amd import build as B

VARIANTS = {"ffnwin1": ["-DLDM_ABL_FFN_WINDOW=1"], "ffnwin2": ["-DLDM_ABL_FFN_WINDOW=2"]}

B.build()
cc = B.hipcc()
bdir = os.path.join(B.HERE, "build")
for name in (sys.argv[1:] or list(VARIANTS)):
    obj = os.path.join(bdir, f"kernels_stack_abl_{name}.o")
    subprocess.run([cc, "-x", "hip", *B.FLAGS, *VARIANTS[name], "-c", os.path.join(B.CSRC, "kernels_stack.hip"), "-o", obj], check=True)
    objs = [os.path.join(bdir, s.rsplit(".", 1)[0] + ".o") for s in B.SOURCES if s != "kernels_stack.hip"] + [obj]
    out = os.path.join(B.HERE, f"libldm_hip_abl_{name}.so")
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out], check=True)
    print(out)
