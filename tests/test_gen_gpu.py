"""The HIP generator used by bench.py reproduces tests/datagen.py bit for bit (so the CPU baseline
and parity checks can regenerate any row of a corpus that only ever existed in HBM)."""
import numpy as np
import pytest

from tests import datagen as dg
from vers_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("d", [8, 128, 300, 768])
def test_generator_matches_numpy(d):
    import torch
    n, ld = 333, (d + 3) // 4 * 4
    out = torch.empty(n, ld, device="cuda:0")
    capi.gen_rows_dev(out.data_ptr(), n, d, ld, 0, 0x5EED0001, start_row=17)
    torch.cuda.synchronize()
    want = dg.dist_u(0x5EED0001, n, d, start=17)
    got = out.cpu().numpy()
    assert np.array_equal(got[:, :d].view(np.uint32), want.view(np.uint32))
    assert not got[:, d:].any()
    sig = dg.default_sigma(d)
    capi.gen_rows_dev(out.data_ptr(), n, d, ld, 1, 0x5EED0002, seed_centres=0xC0FFEE, n_modes=7, sigma=float(sig), start_row=5)
    torch.cuda.synchronize()
    want = dg.dist_c(0x5EED0002, n, d, 7, sig, seed_c=0xC0FFEE, start=5)
    assert np.array_equal(out.cpu().numpy()[:, :d].view(np.uint32), want.view(np.uint32))
