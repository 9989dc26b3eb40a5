"""CPU, build container only (skipped where the reference tree is absent): the host-side logic of the drop-in
(layout_dm_amd/layoutdm.py, relation.py — pure functions, no GPU) against the REAL reference objects built through
oracle/ref_harness.py: the reference's own LayoutSequenceTokenizer / BboxTokenizer, its refinement prior, its
aggregate_sampling_settings, its relation constants.  (the boundary was only ever exercised
with mock tokenizers.)"""
import copy

import numpy as np
import pytest
import torch

from oracle import ref_harness as rh

pytestmark = pytest.mark.skipif(not rh.reference_available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref_tok():
    out = {}
    for ds in ("rico25", "publaynet"):
        _m, tok = rh.build_reference_model(ds, seed=0)
        out[ds] = tok
    return out


def _random_layout_tokens(tok, B, g):
    """Valid LayoutDM sequences through the reference's own tokenizer.encode."""
    n = torch.randint(1, tok.max_seq_length + 1, (B,), generator=g)
    mask = torch.arange(tok.max_seq_length)[None] < n[:, None]
    label = torch.randint(0, tok.N_category, (B, tok.max_seq_length), generator=g)
    bbox = torch.rand(B, tok.max_seq_length, 4, generator=g) * 0.8 + 0.1
    return tok.encode({"label": label, "mask": mask, "bbox": bbox})


@pytest.mark.parametrize("mode", ["uniform", "negative", "gaussian"])
def test_refinement_weak_logits_equal_reference(ref_tok, mode):
    """layoutdm.refinement_weak_logits == set_additional_conditions_for_refinement (helpers/task.py:204-224)."""
    from trainer.helpers.task import set_additional_conditions_for_refinement

    from layout_dm_amd.layoutdm import refinement_weak_logits

    tok = ref_tok["rico25"]
    g = torch.Generator().manual_seed(1)
    enc = _random_layout_tokens(tok, 5, g)
    cfg = rh.sampling_cfg("random", refine_mode=mode, refine_offset_ratio=0.07, refine_lambda=2.5)
    cond = {"seq": enc["seq"].clone(), "mask": enc["mask"].clone(), "seq_orig": enc["seq"].clone(), "type": "refinement"}
    ref = set_additional_conditions_for_refinement(copy.deepcopy(cond), tok, cfg)
    ours = refinement_weak_logits(tok, cond["seq_orig"], cfg, cache={})
    assert ours.shape == ref["weak_logits"].shape == (5, tok.N_total, tok.max_token_length)
    assert torch.equal(ours, ref["weak_logits"].float())
    # the engine applies the prior where !mask for every class: exactly the reference's weak_mask
    assert torch.equal(ref["weak_mask"], (~cond["mask"])[:, None, :].expand(-1, tok.N_total, -1))
    # a single conditioning layout stays (1,C,S): duplicate_cond repeats it afterwards
    one = refinement_weak_logits(tok, cond["seq_orig"][:1], cfg, cache={})
    assert one.shape == (1, tok.N_total, tok.max_token_length) and torch.equal(one[0], ours[0])


def test_aggregate_sampling_settings_equal_reference(ref_tok):
    """layoutdm.aggregate_sampling_settings == BaseModel.aggregate_sampling_settings + LayoutDM's time_difference
    (models/base_model.py:124-150, models/layoutdm.py:90-97) for every cond type / override combination."""
    import trainer.models.layoutdm as ref_layoutdm

    from layout_dm_amd.layoutdm import aggregate_sampling_settings

    tok = ref_tok["rico25"]
    ref_model = object.__new__(ref_layoutdm.LayoutDM)
    torch.nn.Module.__init__(ref_model)
    ref_model.tokenizer = tok
    for cond in ("unconditional", "c", "cwh", "partial", "refinement", "relation"):
        for refine_lambda in (0.0, 3.0):
            for relation_lambda in (0.0, 3e6):
                for td in (0.0, 0.2):
                    for preset_T in (None, 50):
                        args = rh.to_cfg(dict(cond=cond, refine_lambda=refine_lambda, refine_mode="uniform",
                                              refine_offset_ratio=0.1, relation_lambda=relation_lambda,
                                              relation_mode="average", relation_tau=1.0, relation_num_update=3,
                                              num_timesteps=100, time_difference=td))
                        base = rh.sampling_cfg("top_p", top_p=0.9)
                        del base["num_timesteps"]
                        if preset_T is not None:
                            base["num_timesteps"] = preset_T
                        want = ref_model.aggregate_sampling_settings(copy.deepcopy(base), args)
                        got = aggregate_sampling_settings(tok, copy.deepcopy(base), args)
                        assert dict(got) == dict(want), (cond, refine_lambda, relation_lambda, td, preset_T)


def test_device_decode_plan_and_relation_geometry_with_real_tokenizer(ref_tok):
    """The LayoutDM tokenizer qualifies for the device-side decode (linear bins: no centre table), and the relation
    constants equal what _stochastic_convert derives (logit_adjustment.py:30-41,78-82)."""
    from layout_dm_amd.layoutdm import device_decode_plan
    from layout_dm_amd.relation import relation_geometry

    for ds, tok in ref_tok.items():
        ok, centres = device_decode_plan(tok)
        assert ok and centres is None
        cs, bins = relation_geometry(tok)
        bt, N = tok.bbox_tokenizer, tok.N_bbox_per_var
        ref_centres = torch.cat([torch.from_numpy(bt.clustering_models[f"{k}-{N}"].cluster_centers_)
                                 for k in bt.var_names], dim=1)                      # (N, 4) as in l.78-81
        assert np.array_equal(cs, ref_centres.numpy().T)
        canvas_ids = bt.encode(torch.FloatTensor([[[0.5, 0.5, 1.0, 1.0]]])).long().view(-1)   # l.38
        assert [int(canvas_ids[i]) - i * N for i in range(4)] == bins
        # the canvas token of coordinate x selects centre bins[x] of that coordinate: the box (0.5,0.5,1,1) itself
        assert np.allclose([cs[i][bins[i]] for i in range(4)], [0.5, 0.5, 1.0, 1.0], atol=1.0 / N)


def test_relation_plan_csr_on_reference_built_graph(ref_tok):
    """graph_to_csr on a DataBatch-shaped graph produced by the reference's own AddCanvasElement /
    AddRelationConstraints transforms (data/util.py:111-177): per-layout edge lists with local node ids."""
    from trainer.data.util import AddCanvasElement, AddRelationConstraints

    from layout_dm_amd.relation import graph_to_csr

    g = torch.Generator().manual_seed(3)
    rel = AddRelationConstraints(seed=3, edge_ratio=0.6)
    ys, eis, eas, bts, per_layout, off = [], [], [], [], [], 0
    for b in range(4):
        n = int(torch.randint(2, 8, (1,), generator=g))
        data = type("Data", (), {})()
        data.x = torch.rand(n, 4, generator=g) * 0.6 + 0.2
        data.y = torch.randint(0, 25, (n,), generator=g)
        data.attr = {"has_canvas_element": torch.tensor([False])}
        data = rel(AddCanvasElement()(data))
        ei = data.edge_index.view(2, -1)
        per_layout.append((ei.clone(), data.edge_attr.clone()))
        ys.append(data.y); eis.append(ei + off); eas.append(data.edge_attr)
        bts.append(torch.full((n + 1,), b, dtype=torch.long))
        off += n + 1
    graph = rh.GraphBatch(torch.cat(ys), torch.cat(eis, dim=1), torch.cat(eas), torch.cat(bts))
    o, src, dst, attr = graph_to_csr(graph, 4)
    assert o[0] == 0 and o[-1] == graph.edge_attr.numel()
    for b, (ei, ea) in enumerate(per_layout):
        sl = slice(int(o[b]), int(o[b + 1]))
        assert torch.equal(src[sl].long(), ei[0]) and torch.equal(dst[sl].long(), ei[1]) and torch.equal(attr[sl].long(), ea)


def test_reference_tokenizer_drives_dropin_constructor_checks(ref_tok):
    """Everything LayoutDM.__init__ reads from the tokenizer exists on the real object with the values the C-ABI is
    configured from (the GPU tests use a duck-typed mock)."""
    from layout_dm_amd import synthetic as SY

    for ds, tok in ref_tok.items():
        spec = SY.SPECS[ds]
        assert tok.id_to_name(tok.N_total - 1) == "mask" and list(tok.var_names) == ["c", "x", "y", "w", "h"]
        assert (tok.N_category, tok.N_bbox_per_var, tok.max_seq_length, tok.N_var_per_element) == \
               (spec.n_category, spec.n_bin, spec.max_elem, spec.n_attr)
        assert (tok.N_total, tok.max_token_length) == (spec.n_class, spec.seq_len)
        assert tok.name_to_id("pad") == spec.pad_id and tok.name_to_id("mask") == spec.mask_id
        assert list(tok.special_tokens) == ["pad", "mask"]
