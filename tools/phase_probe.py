"""Dev tool (GPU box): per-phase s_memtime sums of the instrumented fused kernels.

Run with LDM_ATTN_TM=1 (the probe variant of the per-step stack kernel; the loop variant has none).  Prints, per
workgroup and launch, the shader cycles of every phase plus the shader clock derived from s_memtime / s_memrealtime
(100 MHz)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layout_dm_amd.binding import Engine, load_library
from layout_dm_amd import synthetic as SY

lib = load_library()
B = int(os.environ.get("PROBE_B", "512"))
spec = SY.SPECS["rico25"]
sd = SY.synth_state_dict(spec, seed=0)
e = Engine(n_category=spec.n_category, precision="fast", max_batch=B)
e.load_state_dict(sd)
tokens = torch.full((B, spec.seq_len), spec.mask_id, dtype=torch.int32)
for _ in range(2):
    e.denoise_logits(tokens, 50)
torch.cuda.synchronize()
lib.ldm_dev_stack_phases((C.c_ulonglong * 16)())  # reset after warm-up
N = 3
for _ in range(N):
    e.denoise_logits(tokens, 50)
torch.cuda.synchronize()
k16 = (C.c_ulonglong * 16)()
lib.ldm_dev_stack_phases(k16)
k = list(k16)
if k[0]:  # stack kernel, per-step form (HEAD 1): all layers + vocabulary head per launch, rows resident in the accumulators
    n = k[0]
    clk = k[1] / max(k[2], 1) * 100.0
    tot = k[1] / n
    print(f"stack: blocks={n} cycles/block={tot:.0f} clock={clk:.0f} MHz  us/block={tot/clk:.1f}  (4 layers + head per block)")
    names = ("prologue", "head streams (174 MFMA)", "attn core", "slab pairs (60 MFMA)", "LN2 (+ chunk-0 DMA)", "FFN chunk loop",
             "layer boundary", "vocabulary head")
    per = (1, 32, 32, 32, 4, 236, 4, 1)
    for name, v, kk in zip(names, k[3:11], per):
        print(f"   {name:24s} {v/n:9.0f} cyc/block  {100*v/n/tot:5.1f}%   per item {v/n/kk:8.1f}")
    print(f"   layer entry: barrier + first tiles + tables {k[11]/n/4:8.1f}, statistics {k[12]/n/4:8.1f}, transform {(k[9]-k[11]-k[12])/n/4:8.1f} cycles per layer")
    print(f"   sync waits: {k[13]/n/192:6.1f} per tile, {k[14]/n/64:6.1f} per slab")
    mf = 4 * (48 * 29 + 16 * 30 + 8 * 32 + 59 * 59) * 32 + 145 * 32
    print(f"   ideal MFMA cycles/block = {mf}  ({100*mf/tot:.1f}% of block)")
e.close()
