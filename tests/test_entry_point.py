"""The hydra-less entry point (layout_dm_amd/test_entry.py): CLI parsing / config plumbing on the CPU, and one full
`cond=unconditional` run on the GPU against a job_dir in the reference's layout (config.yaml + best_model.pt)."""
import os
import pickle

import numpy as np
import pytest
import torch
import yaml

from layout_dm_amd import test_entry as TE

TRAIN_CFG = {
    "backbone": {"_target_": "trainer.models.transformer_utils.TransformerEncoder",
                 "encoder_layer": {"_target_": "trainer.models.transformer_utils.Block", "d_model": 512, "nhead": 8,
                                   "dim_feedforward": 2048, "dropout": 0.0, "batch_first": True, "norm_first": True,
                                   "timestep_type": "adalayernorm", "diffusion_step": 100},
                 "num_layers": 4},
    "data": {"batch_size": 64, "bbox_quantization": "linear", "num_bin_bboxes": 32, "pad_until_max": True,
             "shared_bbox_vocab": "x-y-w-h", "special_tokens": ["pad", "mask"], "transforms": ["RandomOrder"],
             "var_order": "c-x-y-w-h"},
    "dataset": {"_target_": "trainer.datasets.rico.Rico25Dataset", "_partial_": True, "dir": "???", "max_seq_length": 25},
    "model": {"_target_": "trainer.models.layoutdm.LayoutDM", "_partial_": True, "q_type": "constrained"},
    "seed": 0,
}


def test_cli_parser_matches_testconfig_defaults_and_types():
    cfg = TE.parse_cli(["job_dir=/j", "result_dir=/r"])
    assert cfg.cond == "unconditional" and cfg.max_batch_size == 512 and cfg.num_timesteps == 100
    assert cfg.sampling == "random" and cfg.top_p == 0.9 and cfg.relation_lambda == 3e6 and cfg.debug is False
    cfg = TE.parse_cli(["job_dir=/j", "result_dir=/r", "cond=refinement", "max_batch_size=64", "top_p=0.8",
                        "debug=true", "num_uncond_samples=17", "time_difference=0.1", "sampling=top_p"])
    assert (cfg.cond, cfg.max_batch_size, cfg.top_p, cfg.debug, cfg.num_uncond_samples, cfg.time_difference) == \
           ("refinement", 64, 0.8, True, 17, 0.1)
    with pytest.raises(SystemExit):
        TE.parse_cli(["job_dir=/j"])                       # result_dir is mandatory (hydra: "???")
    with pytest.raises(SystemExit):
        TE.parse_cli(["job_dir=/j", "result_dir=/r", "no_such_key=1"])
    with pytest.raises(ValueError):
        TE.parse_cli(["job_dir=/j", "result_dir=/r", "max_batch_size=many"])


def test_geometry_tokenizer_and_job_dir_discovery(tmp_path):
    tok = TE.GeometryTokenizer(TE.to_attr(TRAIN_CFG["data"]), TE.to_attr(TRAIN_CFG["dataset"]))
    assert (tok.N_category, tok.N_total, tok.max_token_length) == (25, 155, 125)
    assert tok.id_to_name(tok.N_total - 1) == "mask" and tok.name_to_id("pad") == 153
    from layout_dm_amd.layoutdm import device_decode_plan

    assert device_decode_plan(tok) == (True, None)
    with pytest.raises(NotImplementedError):
        TE.GeometryTokenizer(TE.to_attr(dict(TRAIN_CFG["data"], shared_bbox_vocab="xywh")), TE.to_attr(TRAIN_CFG["dataset"]))
    # kmeans: centres come from the reference's clustering pickle layout (bbox_tokenizer.py:51-68)
    cl = tmp_path / "clustering_weights"
    cl.mkdir()
    rng = np.random.default_rng(0)
    models = {f"{k}-32": TE.AttrDict(cluster_centers_=rng.random((32, 1))) for k in "xywh"}
    models["x-64"] = TE.AttrDict(cluster_centers_=rng.random((64, 1)))
    with open(cl / "rico25_max25_kmeans_train_clusters.pkl", "wb") as f:
        pickle.dump(models, f)
    tokk = TE.GeometryTokenizer(TE.to_attr(dict(TRAIN_CFG["data"], bbox_quantization="kmeans")),
                                TE.to_attr(TRAIN_CFG["dataset"]), str(cl))
    ok, centres = device_decode_plan(tokk)
    assert ok and centres.shape == (4, 32) and bool((centres[:, 1:] >= centres[:, :-1]).all())
    # multi-seed job_dir (test.py:70-89)
    for s in (0, 1):
        d = tmp_path / "job" / str(s)
        d.mkdir(parents=True)
        (d / "config.yaml").write_text(yaml.safe_dump(TRAIN_CFG))
    cfg, dirs = TE._find_ckpt_dirs(str(tmp_path / "job"))
    assert [os.path.basename(d) for d in dirs] == ["0", "1"] and cfg.model.q_type == "constrained"
    with pytest.raises(SystemExit):
        TE.run_builtin(TE.parse_cli([f"job_dir={tmp_path / 'job'}", f"result_dir={tmp_path}", "cond=c"]))


@pytest.mark.gpu
def test_entry_point_unconditional_end_to_end(tmp_path):
    """`python -m layout_dm_amd.test_entry cond=unconditional job_dir=... result_dir=...` (built-in runner): result
    pickle in the reference's wire format, reproducible under the reference's set_seed convention, equal to calling
    LayoutDM.sample directly."""
    from layout_dm_amd import synthetic as SY
    from layout_dm_amd.layoutdm import LayoutDM

    job = tmp_path / "job"
    job.mkdir()
    (job / "config.yaml").write_text(yaml.safe_dump(TRAIN_CFG))
    sd = {k: torch.from_numpy(v) for k, v in SY.synth_state_dict(SY.RICO25, seed=1, perturb=True).items()}
    torch.save(sd, job / "best_model.pt")
    # (run_builtin, not main: main() drives the reference's own entry point whenever hydra + trainer are importable — which they are
    #  in a session that ran tests/test_entry_reference_main*.py before this file; that path has its own tests)
    out = TE.run_builtin(TE.parse_cli([f"job_dir={job}", f"result_dir={tmp_path / 'res'}", "num_uncond_samples=10", "max_batch_size=4",
                                       "num_timesteps=20", "sampling=random"]))
    assert os.path.basename(out["result_dir"]) == "unconditional_temperature_1.0_name_random_num_timesteps_20"
    data = pickle.load(open(out["pickles"][0], "rb"))
    assert set(data) == {"results", "train_cfg", "test_cfg"} and len(data["results"]) == 10
    for bbox, label in data["results"]:
        assert bbox.ndim == 2 and bbox.shape[1] == 4 and bbox.dtype == np.float32 and label.shape == (bbox.shape[0],)
        assert ((bbox >= 0) & (bbox <= 1)).all() and ((label >= 0) & (label < 25)).all()
    assert data["test_cfg"]["cond"] == "unconditional" and data["train_cfg"]["sampling"]["name"] == "random"
    # the same three batches through the model class directly, seeded like test.py (set_seed(0) before the loop)
    tok = TE.GeometryTokenizer(TE.to_attr(TRAIN_CFG["data"]), TE.to_attr(TRAIN_CFG["dataset"]))
    m = LayoutDM(backbone_cfg=TE.to_attr(TRAIN_CFG["backbone"]), tokenizer=tok, q_type="constrained", max_batch=4)
    m.load_state_dict(sd)
    torch.manual_seed(0)
    direct = []
    for bs in (4, 4, 2):
        lay = m.sample(batch_size=bs, cond=None, sampling_cfg=TE.AttrDict(name="random", temperature=1.0, num_timesteps=20))
        direct.extend(TE._filter_invalid(lay))
    for (b1, l1), (b2, l2) in zip(data["results"], direct):
        assert np.array_equal(b1, b2) and np.array_equal(l1, l2)


@pytest.mark.gpu
def test_check_checkpoint_reports_what_auto_selects(tmp_path, caplog):
    """`python -m layout_dm_amd.check_checkpoint job_dir=...`: one JSON-able record per checkpoint with the
    engine `precision="auto"` selects, the MEASURED fp16 logits error, the tolerance and the throughput class — on the init-like
    synthetic checkpoint the fp16 engine stays (error ~4e-4), on the sigma = 0.15 one the reference-precision engine takes over —
    and the same as an INFO record on the `layout_dm_amd` logger."""
    import json
    import logging

    from layout_dm_amd import check_checkpoint as CC
    from layout_dm_amd import synthetic as SY

    for point, want in (("init", "fast_verified"), ("wide", "split")):
        job = tmp_path / f"job_{point}"
        job.mkdir()
        (job / "config.yaml").write_text(yaml.safe_dump(TRAIN_CFG))
        sd = {k: torch.from_numpy(v) for k, v in SY.trained_like_state_dict(SY.RICO25, point, seed=2).items()}
        torch.save(sd, job / "best_model.pt")
        with caplog.at_level(logging.INFO, logger="layout_dm_amd"):
            caplog.clear()
            reps = CC.check(str(job), max_batch=8)
        assert len(reps) == 1
        rep = reps[0]
        json.dumps(rep, default=str)
        assert rep["engine_selected"] == want and rep["tolerance"] == 1e-3 and rep["checkpoint"].endswith("best_model.pt")
        assert (rep["fast_logits_err_rel"] <= 1e-3) == (want == "fast_verified")
        assert rep["verifier_check"]["finite"] and "layouts/s" in rep["expected_throughput"]
        msgs = [r.getMessage() for r in caplog.records if r.name == "layout_dm_amd"]
        assert len(msgs) == 1 and f"'{want}'" in msgs[0] and f"{rep['fast_logits_err_rel']:.2e}" in msgs[0]
