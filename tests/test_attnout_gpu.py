"""GPU: the split mode's fused attention + out_proj launch (layout_dm_amd/csrc/kernels_attnout.hip) on synthetic operands against a
float64 host computation of the same block (csrc/ldm_dev.cpp ldm_dev_attnout_check) — the kernel alone, without the denoiser
around it: a dropped lo term, a stale MFMA operand or a wrong fragment address shows here at 1e-4 .. 1, where the logits of a
whole pass would still look "close".  Reference: torch.nn.MultiheadAttention inside Block.forward
(/root/reference/src/trainer/trainer/models/transformer_utils.py:175-178,197-204)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,S,amp", [(1, 125, 1.0), (3, 125, 2.5), (2, 105, 1.0), (2, 128, 0.3), (6, 125, 1.5)])
def test_fused_attention_out_proj_vs_float64(B, S, amp):
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    from layout_dm_amd import binding

    lib = binding.load_library()
    fn = lib.ldm_dev_attnout_check
    fn.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_uint32, ctypes.POINTER(ctypes.c_double), ctypes.c_int]
    fn.restype = ctypes.c_int
    err = (ctypes.c_double * 25)()
    torch.cuda.init()
    rc = fn(B, S, amp, 7 + B, err, 0)
    assert rc == 0, rc
    full, hi_only, mag = err[0], err[1], err[2]
    print(f"B={B} S={S} amp={amp}: max rel err {full:.2e} (fp16-hi-only yardstick {hi_only:.2e}, block magnitude {mag:.2f})")
    # three-MFMA products carry 2^-22 relative per operand pair; a missing lo term would sit at the yardstick (~1e-3)
    assert full < 3e-6 and full < hi_only * 1e-2
