"""Dev tool (no GPU needed): compile one csrc/*.hip for gfx950 and print, per kernel, the register / scratch /
LDS budget hipcc reports (-Rpass-analysis=kernel-resource-usage), plus the assembly when asked.

    python tools/kernel_resources.py kernels_stack.hip [name-filter] [--asm out.s]

A non-zero "scratch" or "spill" column on a fused kernel is a bug: these kernels are written to live exactly inside
the 512-register file of one wave per SIMD."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    asm = "/dev/null"
    if "--asm" in argv:  # --asm <file>: keep the assembly
        i = argv.index("--asm")
        asm = argv[i + 1]
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    src = os.path.join(ROOT, "layout_dm_amd", "csrc", args[0])
    flt = args[1] if len(args) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
           "-o", asm, src, "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    if "error:" in out:
        print(out)
        sys.exit(1)
    cur = None
    rows = []
    for line in out.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE,
                                          text=True).stdout.strip().split("(")[0]}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vspill", r"VGPRs? Spill: (\d+)"),
                         ("sspill", r"SGPRs? Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    print(f"{'kernel':70s} vgpr agpr sgpr scratch vspill sspill occ")
    for r in rows:
        if flt and flt not in r["name"]:
            continue
        print(f"{r['name'][-70:]:70s} {r.get('vgpr', 0):4d} {r.get('agpr', 0):4d} {r.get('sgpr', 0):4d} "
              f"{r.get('scratch', 0):7d} {r.get('vspill', 0):6d} {r.get('sspill', 0):6d} {r.get('occ', 0):3d}")


if __name__ == "__main__":
    main()
