"""CPU: product-side geometry / synthetic checkpoints == the oracle's independent copy."""
import numpy as np
import pytest

from layout_dm_amd import synthetic as P
from oracle import spec as OS
from oracle import synth as O


@pytest.mark.parametrize("ds", ["rico25", "publaynet"])
def test_same_checkpoint(ds):
    a = P.synth_state_dict(P.SPECS[ds], seed=3, perturb=True)
    b = O.synth_state_dict(OS.SPECS[ds], seed=3, perturb=True)
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert P.SPECS[ds].n_class == OS.SPECS[ds].n_class
    for at in range(5):
        assert np.array_equal(P.SPECS[ds].full_ids(at), OS.SPECS[ds].full_ids(at))


@pytest.mark.parametrize("point", ["init", "mid", "wide"])
def test_same_trained_like_checkpoint(point):
    a = P.trained_like_state_dict(P.SPECS["rico25"], point, seed=2)
    b = O.trained_like_state_dict(OS.SPECS["rico25"], point, seed=2)
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_same_synthetic_cond_c():
    for ds in ("rico25", "publaynet"):
        a = P.synth_cond_c(P.SPECS[ds], 37, seed=4)
        b = O.synth_cond_c(OS.SPECS[ds], 37, seed=4)
        assert np.array_equal(a["seq"], b["seq"]) and np.array_equal(a["mask"], b["mask"])


def test_timestep_schedule_matches_oracle():
    from layout_dm_amd.diffusion import timestep_schedule
    from oracle import restatement as R

    for te in (100, 50, 25, 10, 7, 1):
        tm, tp = timestep_schedule(100, te)
        assert tm == R.timestep_list(100, te)
        prev = 100
        for t, p in zip(tm, tp):
            skip = prev - t - 1
            assert p == (t - skip if (skip > 0 and t > skip) else t)
            prev = t
    # with time_difference (base.py:218-226, then the skip-step shift of l.227-235) — the rule oracle.single_step
    # applies, which tests/test_oracle_golden.py pins to a reference run with time_difference = 0.15
    for te, td in ((100, 0.15), (100, 0.5), (25, 0.1), (10, 0.99)):
        tm, tp = timestep_schedule(100, te, td)
        assert tm == R.timestep_list(100, te)
        prev = 100
        for t, p in zip(tm, tp):
            skip = prev - t - 1
            noise_t = min(max(t - int(100 * td), 0), 99)
            assert p == (noise_t - skip if (skip > 0 and noise_t > skip) else noise_t)
            assert 0 <= p < 100
            prev = t


def test_relation_graph_to_csr_roundtrip():
    """Host logic of cond=relation: DataBatch-style global edge lists -> per-layout CSR with local node ids."""
    import pytest
    import torch

    from layout_dm_amd.relation import graph_to_csr

    g = torch.Generator().manual_seed(0)
    B = 6
    n_nodes = [1, 4, 1, 7, 3, 2]                      # canvas only / several elements
    batch = torch.cat([torch.full((n,), b, dtype=torch.long) for b, n in enumerate(n_nodes)])
    first = [0]
    for n in n_nodes[:-1]:
        first.append(first[-1] + n)
    edges = []
    for b, n in enumerate(n_nodes):
        for i in range(n):
            for j in range(i + 1, n):
                if torch.rand(1, generator=g).item() < 0.7:
                    edges.append((first[b] + i, first[b] + j, int(torch.randint(1, 1024, (1,), generator=g))))
    perm = torch.randperm(len(edges), generator=g).tolist()   # DataBatch order is not required to be grouped
    edges = [edges[k] for k in perm]
    ei = torch.tensor([[e[0] for e in edges], [e[1] for e in edges]])
    ea = torch.tensor([e[2] for e in edges])
    off, src, dst, attr = graph_to_csr({"batch": batch, "edge_index": ei, "edge_attr": ea}, B)
    assert off.dtype == torch.int32 and off.tolist()[0] == 0 and off.tolist()[-1] == len(edges)
    rebuilt = []
    for b in range(B):
        for k in range(int(off[b]), int(off[b + 1])):
            assert 0 <= int(src[k]) < n_nodes[b] and 0 <= int(dst[k]) < n_nodes[b]
            rebuilt.append((first[b] + int(src[k]), first[b] + int(dst[k]), int(attr[k])))
    assert sorted(rebuilt) == sorted(edges)
    per_layout_order = [e for e in edges if batch[e[0]] == 3]          # original order inside a layout is kept
    assert rebuilt[int(off[3]):int(off[4])] == per_layout_order
    e_off, e_src, _, _ = graph_to_csr({"batch": batch, "edge_index": torch.zeros((2, 0), dtype=torch.long),
                                      "edge_attr": torch.zeros(0, dtype=torch.long)}, B)
    assert e_off.tolist() == [0] * (B + 1) and e_src.numel() == 0
    with pytest.raises(ValueError):
        graph_to_csr({"batch": batch, "edge_index": torch.tensor([[0], [1]]), "edge_attr": torch.tensor([3])}, B)


def test_device_decode_plan_selection():
    """Host logic of LayoutDM.sample: which tokenizer configurations are decoded on the GPU (everything else falls back
    to the caller's tokenizer.decode, like the reference)."""
    import numpy as np
    import torch

    from layout_dm_amd.layoutdm import LayoutDM

    def bbt(quant="linear", shared="x-y-w-h", order=("x", "y", "w", "h"), n=32, two_d=False):
        models = {f"{k}-{n}": type("M", (), {"cluster_centers_": (np.random.rand(n, 2) if two_d else
                                                                  np.sort(np.random.rand(n, 1), axis=0))})()
                  for k in "xywh"}
        return type("B", (), {"shared_bbox_vocab": shared, "bbox_quantization": quant, "var_names": ["x", "y", "w", "h"],
                              "_var_order": list(order), "clustering_models": models})()

    def plan(tok):
        fake = type("F", (), {"tokenizer": tok, "_decode_plan": None})()
        return LayoutDM._device_decode_centres(fake)

    def tok(special=("pad", "mask"), names=("c", "x", "y", "w", "h"), **kw):
        return type("T", (), {"special_tokens": list(special), "var_names": list(names), "N_bbox_per_var": 32,
                              "bbox_tokenizer": bbt(**kw)})()

    ok, c = plan(tok())
    assert ok and c is None
    ok, c = plan(tok(quant="kmeans"))
    assert ok and isinstance(c, torch.Tensor) and c.shape == (4, 32) and c.dtype == torch.float64
    assert plan(tok(quant="percentile"))[0]
    assert not plan(tok(special=("pad", "bos", "eos", "mask")))[0]          # autoregressive tokenizers: host decode
    assert not plan(tok(shared="xywh"))[0]                                   # one shared bbox vocabulary
    assert not plan(tok(order=("w", "h", "x", "y")))[0]                      # permuted variable order
    assert not plan(tok(quant="kmeans", two_d=True))[0]                      # product-space clusters
    assert not plan(type("T", (), {"special_tokens": ["pad", "mask"], "var_names": ["c", "x", "y", "w", "h"]})())[0]
