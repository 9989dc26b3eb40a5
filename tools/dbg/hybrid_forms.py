"""Dev probe (GPU): the hybrid engine's logits (this process's form: fused FFN, or LDM_DEV=1 LDM_HYB_FFN=0 the two-launch form) against the CPU emulation of the
hybrid operand format (tools/two_product_emulation.py: the same roundings at the same sites) and against the float64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import two_product_emulation as E
from oracle import restatement as R, spec as SP, synth
from layout_dm_amd.binding import Engine
spec = SP.RICO25
for point in ("mid", "init"):
    sd = synth.synth_state_dict(spec, seed=0) if point == "init" else synth.trained_like_state_dict(spec, point, seed=3)
    W, W64 = R.as_torch_weights(sd), R.as_torch_weights(sd, torch.float64)
    g = torch.Generator().manual_seed(9)
    tokens = torch.randint(0, spec.n_class, (4, spec.seq_len), generator=g)
    e = Engine(n_category=spec.n_category, precision=os.environ.get("PROBE_PREC", "hybrid"), max_batch=8)
    e.load_state_dict(sd)
    fmt = dict(E.FORMATS)
    sites = fmt["hybrid: weights + ln2, hid, hln fp16"] if os.environ.get("PROBE_PREC", "hybrid") == "hybrid" else fmt["mixed: weights fp16"]
    f = {s: (E.h if s in sites else E.h2) for s in E.SITES}
    for t in (90, 40, 3):
        out = e.denoise_logits(tokens.int(), t).cpu()[..., :spec.n_class].double()
        emu = E.fwd(W, tokens, t, f).double()
        ref = R.denoiser_logits(W64, spec, tokens, t, dtype=torch.float64)
        m = ref.abs().max()
        print(f"[{point} t={t}] {e.describe()['kernels'][:60]}: vs emulation {((out - emu).abs().max() / m).item():.2e}   vs float64 {((out - ref).abs().max() / m).item():.2e}   "
              f"emulation vs float64 {((emu - ref).abs().max() / m).item():.2e}", flush=True)
    e.close()
