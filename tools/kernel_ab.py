"""Dev tool (GPU box): A/B timing of kernel variants selected by environment variables.

    python tools/kernel_ab.py "LDM_FUSED_ATTN=6" "LDM_FUSED_ATTN=0" "LDM_HIP_LIB=tools/ab/libldm_hip_prev.so" ...

Every configuration runs in its own process (the variants are latched in static initialisers): one eager,
event-profiled denoiser pass at B=512 (the bench workload's launch shapes) repeated a few times; prints the
average launch time of every kernel class and the checksum of the logits (variants must agree with the baseline to
fp16-accumulation noise)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import torch
from layout_dm_amd.binding import Engine
from layout_dm_amd import synthetic as SY
B = int(os.environ.get("PROBE_B", "512"))
spec = SY.SPECS["rico25"]
e = Engine(n_category=spec.n_category, precision=os.environ.get("PROBE_PREC", "fast"), max_batch=B)
e.load_state_dict(SY.synth_state_dict(spec, seed=0))
g = torch.Generator().manual_seed(0)
tokens = torch.randint(0, spec.n_class, (B, spec.seq_len), generator=g).int()
for _ in range(2):
    out = e.denoise_logits(tokens, 50)
torch.cuda.synchronize()
e.set_profiling(True)
for _ in range(int(os.environ.get("PROBE_REPS", "10"))):
    out = e.denoise_logits(tokens, 50)
torch.cuda.synchronize()
rows = e.profile(reset=True)
e.set_profiling(False)
res = {r["name"]: round(1e3 * r["ms"] / r["launches"], 2) for r in rows}
res["_logits_abs_mean"] = float(out.abs().mean())
res["_logits_sum"] = float(out.double().sum())
print("AB_RESULT " + json.dumps(res))
''' % ROOT


def main():
    base = None
    for cfg in sys.argv[1:] or [""]:
        env = dict(os.environ)
        for kv in cfg.split():
            k, v = kv.split("=", 1)
            env[k] = v
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, timeout=600)
        line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
        if not line:
            print(f"[{cfg}] FAILED rc={p.returncode}\n{p.stdout[-2000:]}", flush=True)
            continue
        r = json.loads(line[0][len("AB_RESULT "):])
        if base is None:
            base = r
        d = abs(r["_logits_sum"] - base["_logits_sum"]) / (abs(base["_logits_sum"]) + 1e-9)
        ks = "  ".join(f"{k}={v}us" for k, v in r.items() if not k.startswith("_"))
        print(f"[{cfg or 'default'}] {ks}  | logits |mean|={r['_logits_abs_mean']:.5f} rel.sum.diff vs first={d:.2e}",
              flush=True)


if __name__ == "__main__":
    main()
