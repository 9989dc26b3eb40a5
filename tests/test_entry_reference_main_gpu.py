"""GPU: the reference's own `trainer.test.main()` (/root/reference/src/trainer/trainer/test.py:57-283) end to end on the MI355X with
the REAL engine.

`tests/test_entry_reference_main.py` runs the same entry point in the build container with the engine faked below the C-ABI; here
nothing is faked on our side: `python -m layout_dm_amd.test_entry` imports the reference's `trainer` package (on the GPU box: the
byte-compiled `oracle/_ref/`, built by `oracle/build_ref.py`; hydra / omegaconf / torch_geometric stand-ins from
`oracle/ref_harness.py`, as in the CPU test), swaps `trainer.models.layoutdm.LayoutDM` for the drop-in class and calls the
reference's `main()`: its job_dir / config.yaml / best_model.pt handling, its tokenizer, its test-time transforms and `get_cond` per
batch, `model.sample` -> libldm_hip.so, its result pickles.

Answer key: the SAME `main()` with the reference's OWN `LayoutDM` (torch, on the host cores) on the same job_dir, for the six cond
types, greedy decoding (`sampling=deterministic`: no random draw on either side, so both runs see identical `get_cond` randomness
under main()'s `set_seed`).  The result pickles must carry the same layouts — the wire format `eval.py:105-108` reads."""
import os
import pickle
import sys

import numpy as np
import pytest
import torch
import yaml

from oracle import ref_harness as rh

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not rh.reference_importable(), reason="neither /root/reference nor oracle/_ref/ present")]

TRAIN_CFG = {
    "model": {"_target_": "trainer.models.layoutdm.LayoutDM", "_partial_": True, "q_type": "constrained"},
    "backbone": {"_target_": "trainer.models.transformer_utils.TransformerEncoder",
                 "encoder_layer": {"_target_": "trainer.models.transformer_utils.Block", "d_model": 512, "nhead": 8,
                                   "dim_feedforward": 2048, "dropout": 0.0, "batch_first": True, "norm_first": True,
                                   "timestep_type": "adalayernorm", "diffusion_step": 100},
                 "num_layers": 4},
    "data": {"num_bin_bboxes": 32, "pad_until_max": True, "shared_bbox_vocab": "x-y-w-h", "bbox_quantization": "linear",
             "special_tokens": ["pad", "mask"], "var_order": "c-x-y-w-h", "transforms": ["RandomOrder"]},
    "dataset": {"_target_": "trainer.datasets.rico.Rico25Dataset", "_partial_": True, "dir": "???", "max_seq_length": 25},
}


@pytest.fixture()
def job(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from layout_dm_amd import synthetic as SY

    rh.install_entry_stubs()
    import trainer.datasets.rico as rico
    import trainer.test as ref_test

    job_dir = tmp_path / "job"
    job_dir.mkdir()
    (job_dir / "config.yaml").write_text(yaml.safe_dump(TRAIN_CFG))
    sd = {k: torch.from_numpy(v) for k, v in SY.synth_state_dict(SY.RICO25, seed=1, perturb=True).items()}
    torch.save(sd, job_dir / "best_model.pt")
    synth_ds = type("Rico25Dataset", (rh.SynthLayoutDataset,), {"labels": rico.Rico25Dataset.labels})
    monkeypatch.setattr(rico, "Rico25Dataset", synth_ds)
    monkeypatch.setattr(ref_test, "save_image", lambda *a, **k: None)
    monkeypatch.setattr(sys, "argv", list(sys.argv))
    monkeypatch.chdir(tmp_path)
    return str(job_dir), tmp_path


def _run_main(argv, use_dropin, monkeypatch):
    import trainer.models.layoutdm as ref_layoutdm
    import trainer.test as ref_test

    from layout_dm_amd import test_entry as TE

    if use_dropin:
        return TE.main(argv)                      # swaps the class, then the reference's main()
    # the answer key: the reference's own class, on the host (its main() would otherwise move the model to "cuda")
    with pytest.MonkeyPatch.context() as mp:
        mp.setattr(ref_layoutdm, "LayoutDM", _REF_CLASS[0])
        mp.setattr(torch.cuda, "is_available", lambda: False)
        mp.setattr(sys, "argv", [sys.argv[0]] + list(argv))
        ref_test.filter_args_for_ai_platform()
        return ref_test.main()


_REF_CLASS = []


@pytest.mark.parametrize("cond", ["unconditional", "c", "cwh", "partial", "refinement", "relation"])
def test_reference_main_on_the_gpu_equals_the_reference_model(job, cond, capsys, monkeypatch):
    job_dir, tmp = job
    import trainer.models.layoutdm as ref_layoutdm

    from layout_dm_amd.layoutdm import LayoutDM

    if not _REF_CLASS:
        _REF_CLASS.append(ref_layoutdm.LayoutDM if ref_layoutdm.LayoutDM is not LayoutDM else None)
    assert _REF_CLASS[0] is not None
    # (num_timesteps=25: the reference's strided schedule, base.py:310-315 — a quarter of the cost of the answer key, which runs on the host cores)
    common = [f"cond={cond}", f"job_dir={job_dir}", "max_batch_size=4", "num_uncond_samples=6", "sampling=deterministic", "num_timesteps=25"]
    # ---- the drop-in class + the HIP engine
    _run_main(common + [f"result_dir={tmp / 'ours'}"], True, monkeypatch)
    printed = capsys.readouterr().out
    assert ref_layoutdm.LayoutDM is LayoutDM and "ms per sample" in printed          # test.py:257-258
    # ---- the reference's own class
    _run_main(common + [f"result_dir={tmp / 'ref'}"], False, monkeypatch)

    def load(which):
        d = os.listdir(tmp / which)
        assert len(d) == 1 and d[0].startswith(cond + "_")
        return pickle.load(open(tmp / which / d[0] / "seed_0.pkl", "rb"))

    ours, ref = load("ours"), load("ref")
    assert set(ours) == set(ref) and len(ours["results"]) == len(ref["results"]) == 6        # eval.py:105-108 reads "results"
    assert ours["test_cfg"]["cond"] == cond
    n_tok = n_bad = 0
    for (b1, l1), (b2, l2) in zip(ours["results"], ref["results"]):
        assert b1.dtype == b2.dtype and b1.shape[1] == 4
        if b1.shape != b2.shape:      # an element count differs: at least one category token differs
            n_bad += max(len(l1), len(l2))
            n_tok += max(len(l1), len(l2)) * 5
            continue
        n_tok += l1.size * 5
        n_bad += int((l1 != l2).sum()) + int((b1 != b2).sum())
    print(f"cond={cond}: {n_bad} of {n_tok} decoded attributes differ from the reference model's")
    assert n_bad == 0
    for k in ("inputs", "relations"):
        if k in ref:
            assert k in ours and len(ours[k]) == len(ref[k])
    if cond in ("partial", "refinement"):     # test.py:215-228: the decoded conditioning inputs travel in the pickle too
        for (b1, l1), (b2, l2) in zip(ours["inputs"], ref["inputs"]):
            assert np.array_equal(b1, b2) and np.array_equal(l1, l2)
