"""`python -m layout_dm_amd.test_entry cond=... job_dir=... result_dir=...`

Runs the REFERENCE's own hydra entry point (trainer/test.py:57-288) with its model class swapped for
the MI355X drop-in — CLI keys, cond= plumbing, checkpoint format and result pickles are the
reference's, only `model.sample` runs in libldm_hip.so.  Needs the reference package (`trainer`) and
its dependencies (hydra, omegaconf, torch_geometric, …) to be importable, exactly as the reference's
own `python -m src.trainer.trainer.test` does."""
from __future__ import annotations

import sys


def main() -> None:
    try:
        import trainer.models.layoutdm as ref_layoutdm
    except Exception as e:
        raise SystemExit(
            "layout_dm_amd.test_entry drives the reference's `trainer.test` entry point: install the "
            f"layout-dm package (poetry install) first — import failed with: {e!r}")
    from .layoutdm import LayoutDM

    ref_layoutdm.LayoutDM = LayoutDM  # hydra resolves `_target_: trainer.models.layoutdm.LayoutDM` to this
    import trainer.test as ref_test

    ref_test.filter_args_for_ai_platform()
    ref_test.main()


if __name__ == "__main__":
    sys.exit(main())
