"""CPU, no reference tree needed: the drop-in's pure host logic that restates reference code closely (VERDICT r4 weak #10:
layout_dm_amd/layoutdm.py refinement_prior_table / refinement_weak_logits ~ helpers/task.py:154-224) against what the reference
itself produced — tests/golden/rico25_getcond.npz carries the reference's own (C, C) refinement prior table (lambda applied) and the
`seq_orig` its get_cond built; tests/test_boundary_vs_reference.py does the same against the live reference where it is present.
Also: relation.graph_to_csr on the reference-collated graph of the same fixture (edges per layout, local node ids, canvas = node 0)."""
import os

import numpy as np
import torch

from _stub_tokenizer import StubTokenizer
from oracle import spec as SP


def _sub(g, prefix):
    return {k[len(prefix):]: g[k] for k in g.files if k.startswith(prefix)}


def test_refinement_prior_matches_the_reference_table(golden_dir):
    from layout_dm_amd.layoutdm import refinement_prior_table, refinement_weak_logits

    spec = SP.RICO25
    sub = _sub(np.load(os.path.join(golden_dir, "rico25_getcond.npz")), "refinement_")
    tok = StubTokenizer(spec)
    want = torch.from_numpy(sub["weak_table"])                       # the reference's table, refine_lambda = 3 applied
    table = refinement_prior_table(tok, "uniform", 0.1) * 3.0
    assert table.shape == want.shape == (spec.n_class, spec.n_class) and torch.equal(table, want)
    seq_orig = torch.from_numpy(sub["seq_orig"].astype(np.int64))
    cfg = {"refine_mode": "uniform", "refine_offset_ratio": 0.1, "refine_lambda": 3.0}
    cache = {}
    wl = refinement_weak_logits(tok, seq_orig, cfg, cache)
    assert wl.shape == (seq_orig.shape[0], spec.n_class, spec.seq_len)
    assert torch.equal(wl, want[seq_orig].permute(0, 2, 1).contiguous())
    assert torch.equal(refinement_weak_logits(tok, seq_orig, cfg, cache), wl) and cache["key"] == ("uniform", 0.1)   # cached table
    # the other two modes of task.py:166-201 keep their structure: 'negative' is the complement inside a sub-vocabulary with the
    # weight's sign flipped, 'gaussian' is -(distance)^2 between bin centres; identity outside the bbox sub-vocabularies
    neg = refinement_prior_table(tok, "negative", 0.1)
    uni = refinement_prior_table(tok, "uniform", 0.1)
    gau = refinement_prior_table(tok, "gaussian", 0.1)
    for a in range(4):
        sl = slice(spec.n_category + a * spec.n_bin, spec.n_category + (a + 1) * spec.n_bin)
        assert torch.equal(neg[sl, sl] + uni[sl, sl], torch.ones(spec.n_bin, spec.n_bin))
        assert float(gau[sl, sl].diagonal().abs().max()) == 0.0 and float(gau[sl, sl].max()) == 0.0
    outside = torch.ones(spec.n_class, dtype=torch.bool)
    outside[spec.n_category:spec.n_category + 4 * spec.n_bin] = False
    for t in (neg, uni, gau):
        assert torch.equal(t[outside][:, outside], torch.eye(int(outside.sum())))


def test_graph_to_csr_on_the_reference_collated_graph(golden_dir):
    from layout_dm_amd.relation import graph_to_csr

    sub = _sub(np.load(os.path.join(golden_dir, "rico25_getcond.npz")), "relation_")
    B = sub["cond_seq"].shape[0]

    class G:
        edge_index, edge_attr, batch = (torch.from_numpy(sub[k]) for k in ("edge_index", "edge_attr", "batch"))
        y = torch.from_numpy(sub["y"])

    off, src, dst, ea = graph_to_csr(G, B)
    ei, bt = sub["edge_index"], sub["batch"]
    assert len(off) == B + 1 and int(off[-1]) == ei.shape[1] == len(src) == len(dst) == len(ea)
    first = np.array([int(np.argmax(bt == b)) for b in range(B)])    # first node of every graph in the collated batch
    for b in range(B):
        sel = np.nonzero(bt[ei[0]] == b)[0]
        assert int(off[b + 1] - off[b]) == len(sel)
        got = sorted(zip(src[off[b]:off[b + 1]].tolist(), dst[off[b]:off[b + 1]].tolist()))
        want = sorted(zip((ei[0, sel] - first[b]).tolist(), (ei[1, sel] - first[b]).tolist()))
        assert got == want                                             # local node ids, the canvas element is node 0
