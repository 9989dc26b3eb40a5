"""Checker script (GPU box, not collected by pytest): numerics of each precision mode vs the oracle on teacher-forced
states -> gpurun_out/precision_report.json (committed copy: profiles/r01_precision_report.json).  Lives under tests/
because it imports the oracle."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from layout_dm_amd.binding import Engine
from oracle import restatement as R, spec as SP, synth

out = {}
for ds in ("rico25", "publaynet"):
    spec = SP.SPECS[ds]
    for wname, perturb, wseed in (("ref_init", False, 0), ("perturbed", True, 1)):
        sd = synth.synth_state_dict(spec, seed=wseed, perturb=perturb)
        W = R.as_torch_weights(sd)
        B = 32
        g = torch.Generator().manual_seed(5)
        cases = []
        for t in (95, 60, 30, 5):
            tokens = torch.empty(B, spec.seq_len, dtype=torch.long)
            for a in range(spec.n_attr):
                ids = torch.as_tensor(spec.full_ids(a))
                tokens[:, a::spec.n_attr] = ids[torch.randint(0, len(ids) - 1, (B, spec.max_elem), generator=g)]
            tokens[torch.rand(B, spec.seq_len, generator=g) < t / 99] = spec.mask_id
            nxt, logits, logp = R.single_step(W, spec, tokens, t, {"name": "deterministic"}, return_all=True)
            cases.append((t, tokens, nxt, logits))
        for prec in ("exact", "split", "fast"):
            e = Engine(n_category=spec.n_category, precision=prec, max_batch=B)
            e.load_state_dict(sd)
            rels, mism = [], 0
            for t, tokens, nxt, logits in cases:
                lg = e.denoise_logits(tokens.int(), t).cpu()
                rels.append(((lg - logits).abs().max() / logits.abs().max()).item())
                o = e.sample_step(tokens.int(), t, {"name": "deterministic"}).cpu().long()
                mism += int((o != nxt).sum())
            out[f"{ds}/{wname}/{prec}"] = {"max_rel_logit_err": max(rels), "greedy_token_mismatch": mism,
                                          "tokens": B * spec.seq_len * len(cases)}
            print(ds, wname, prec, out[f"{ds}/{wname}/{prec}"], flush=True)
            e.close()
json.dump(out, open("gpurun_out/precision_report.json", "w"), indent=1)
