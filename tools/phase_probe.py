"""Dev tool (GPU box): per-phase s_memtime sums of the instrumented fused kernels.

Run with LDM_FFN_DBG=3 LDM_ATTN_TM=1.  Prints, per workgroup and launch, the shader cycles spent waiting
at the tile-top barrier (weight DMA + skew), in the MFMA runs, in epilogues and in the attention core,
plus the shader clock derived from s_memtime / s_memrealtime (100 MHz)."""
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
f8 = (C.c_ulonglong * 12)()
a16 = (C.c_ulonglong * 16)()
lib.ldm_dev_ffn_phases(f8)
lib.ldm_dev_attn_phases(a16)  # reset after warm-up
lib.ldm_dev_layer_phases((C.c_ulonglong * 16)())
lib.ldm_dev_stack_phases((C.c_ulonglong * 16)())
N = 3
for _ in range(N):
    e.denoise_logits(tokens, 50)
torch.cuda.synchronize()
lib.ldm_dev_ffn_phases(f8)
lib.ldm_dev_attn_phases(a16)
f = list(f8); a = list(a16)
if f[0]:
    n = f[0]
    clk = f[1] / max(f[2], 1) * 100.0
    print(f"ffn: blocks={n} cycles/block={f[1]/n:.0f} clock={clk:.0f} MHz  us/block={f[1]/n/clk:.1f}")
    tot = f[1] / n
    for name, v in zip(("wait(top)", "gemm1", "bubble", "gemm2"), f[3:7]):
        print(f"   {name:10s} {v/n:9.0f} cyc/block  {100*v/n/tot:5.1f}%   per chunk {v/n/58:7.1f}")
    print(f"   ideal MFMA cycles/block = {58*59*32}  ({100*58*59*32/tot:.1f}% of block)")
    # outside the chunk loop (the r01 probe did not see these): prologue = LN parameters + row loads + fragments,
    # epilogue = residual / bias / statistics / stores (incl. the drain of the stores)
    print(f"   prologue   {f[7]/n:9.0f} cyc/block  = {100*f[7]/n/tot:5.1f}% of the chunk loop    ({f[7]/n/clk:.1f} us)")
    print(f"   epilogue   {f[8]/n:9.0f} cyc/block  = {100*f[8]/n/tot:5.1f}% of the chunk loop    ({f[8]/n/clk:.1f} us)")
    print(f"   whole block = {(f[1]+f[7]+f[8])/n/clk:.1f} us")
if a[0]:
    n = a[0]
    clk = a[1] / max(a[2], 1) * 100.0
    tot = a[1] / n
    print(f"attn: blocks={n} cycles/block={tot:.0f} clock={clk:.0f} MHz  us/block={tot/clk:.1f}")
    names = ("prologue", "wait(top)", "run(29 MFMA)", "epilogue", "attn core", "outproj total", "  wait2", "  run2(32 MFMA)", "  epi2")
    per = (1, 48, 48, 48, 8, 1, 15, 15, 15)
    for name, v, k in zip(names, a[3:12], per):
        print(f"   {name:16s} {v/n:9.0f} cyc/block  {100*v/n/tot:5.1f}%   per item {v/n/k:7.1f}")
    mf = (48 * 29 + 15 * 32 + 8 * 32) * 32
    print(f"   ideal MFMA cycles/block = {mf}  ({100*mf/tot:.1f}% of block)")
l16 = (C.c_ulonglong * 16)()
lib.ldm_dev_layer_phases(l16)
l = list(l16)
if l[0]:  # stream version of the fused layer kernel (LDM_FUSED_ATTN=5)
    n = l[0]
    clk = l[1] / max(l[2], 1) * 100.0
    tot = l[1] / n
    print(f"layer(stream): blocks={n} cycles/block={tot:.0f} clock={clk:.0f} MHz  us/block={tot/clk:.1f}")
    names = ("prologue", "head streams (174 MFMA)", "attn core", "seed + slab stream", "LN2 (+ chunk-0 DMA)", "FFN chunk loop", "store epilogue")
    per = (1, 8, 8, 1, 1, 58, 1)
    for name, v, k in zip(names, l[3:10], per):
        print(f"   {name:24s} {v/n:9.0f} cyc/block  {100*v/n/tot:5.1f}%   per item {v/n/k:8.1f}")
    print(f"   of which: per-tile sync (vmcnt+barrier) {l[10]/n/48:7.1f} cyc per tile, per-slab sync {l[11]/n/15:7.1f} cyc per slab, per-FFN-chunk sync {l[12]/n/58:7.1f}")
    mf = (48 * 29 + 16 * 30 + 8 * 32 + 58 * 59) * 32
    print(f"   ideal MFMA cycles/block = {mf}  ({100*mf/tot:.1f}% of block)")
k16 = (C.c_ulonglong * 16)()
lib.ldm_dev_stack_phases(k16)
k = list(k16)
if k[0]:  # stack kernel (LDM_FUSED_ATTN=6): all layers per launch, rows resident in the accumulators
    n = k[0]
    clk = k[1] / max(k[2], 1) * 100.0
    tot = k[1] / n
    print(f"stack: blocks={n} cycles/block={tot:.0f} clock={clk:.0f} MHz  us/block={tot/clk:.1f}  (4 layers per block)")
    names = ("prologue", "head streams (174 MFMA)", "attn core", "slab pairs (60 MFMA)", "LN2 (+ chunk-0 DMA)", "FFN chunk loop",
             "layer boundary", "store epilogue")
    per = (1, 32, 32, 32, 4, 232, 4, 1)
    for name, v, kk in zip(names, k[3:11], per):
        print(f"   {name:24s} {v/n:9.0f} cyc/block  {100*v/n/tot:5.1f}%   per item {v/n/kk:8.1f}")
    print(f"   layer entry: barrier + first tiles + tables {k[11]/n/4:8.1f}, statistics {k[12]/n/4:8.1f}, transform {(k[9]-k[11]-k[12])/n/4:8.1f} cycles per layer")
    print(f"   sync waits: {k[13]/n/192:6.1f} per tile, {k[14]/n/64:6.1f} per slab")
    mf = 4 * (48 * 29 + 16 * 30 + 8 * 32 + 58 * 59) * 32
    print(f"   ideal MFMA cycles/block = {mf}  ({100*mf/tot:.1f}% of block)")
e.close()
