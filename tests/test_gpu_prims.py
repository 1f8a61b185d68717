"""Scan and stable radix sort of the binning stage, through the C ABI test hooks: bit-exact vs torch."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    import diff_gaussian_rasterization as D
    assert torch.cuda.is_available()
    return D._load(), D


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.mark.parametrize("n", [1, 63, 64, 4095, 4096, 4097, 100_000, 1_234_567, 20_000_001])
def test_exclusive_scan(n):
    lib, D = _lib()
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 1000, (n,), generator=g, dtype=torch.int32).cuda()
    out = torch.empty(n + 1, dtype=torch.int32, device="cuda")
    rc = lib.gsr_test_scan(x.data_ptr(), out.data_ptr(), n, _stream())
    assert rc == 0, D._err(lib)
    ref = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(x.cpu().long(), 0)])
    assert torch.equal(out.cpu().long() & 0xFFFFFFFF, ref & 0xFFFFFFFF)     # 32-bit prefix sums (the 20 M case wraps)


@pytest.mark.parametrize("n,lo,hi", [(1, 0, 32), (1000, 0, 32), (4096, 0, 13), (4097, 0, 8), (300_001, 0, 32),
                                     (2_000_003, 0, 13), (50_000, 0, 15), (70_000, 3, 9)])
def test_stable_radix_sort_pairs(n, lo, hi):
    lib, D = _lib()
    g = torch.Generator().manual_seed(n + hi)
    # few distinct keys => many ties => stability is exercised
    keys = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int64)
    if hi - lo < 32:
        keys = keys % (1 << min(hi, 20))
    keys32 = keys.to(torch.int32).cuda()
    vals = torch.randint(0, 2 ** 31 - 1, (n,), generator=g, dtype=torch.int32).cuda()
    k, v = keys32.clone(), vals.clone()
    rc = lib.gsr_test_sort_pairs(k.data_ptr(), v.data_ptr(), n, lo, hi, 0, _stream())
    assert rc == 0, D._err(lib)
    digit = (keys >> lo) & ((1 << (hi - lo)) - 1)
    order = torch.argsort(digit, stable=True)
    assert torch.equal(k.cpu().long(), keys[order])
    assert torch.equal(v.cpu(), vals.cpu()[order])
    # argsort form (iota): vals are the permutation
    k2, v2 = keys32.clone(), torch.empty_like(vals)
    rc = lib.gsr_test_sort_pairs(k2.data_ptr(), v2.data_ptr(), n, lo, hi, 1, _stream())
    assert rc == 0, D._err(lib)
    assert torch.equal(v2.cpu().long(), order)


def test_float_depth_keys_sort_like_floats():
    lib, D = _lib()
    n = 200_000
    d = (torch.rand(n, generator=torch.Generator().manual_seed(5)) * 100 + 0.2).float()
    d[::7] = d[3]            # ties
    k = d.view(torch.int32).cuda()
    v = torch.empty(n, dtype=torch.int32, device="cuda")
    assert lib.gsr_test_sort_pairs(k.data_ptr(), v.data_ptr(), n, 0, 32, 1, _stream()) == 0
    assert torch.equal(v.cpu().long(), torch.argsort(d, stable=True))
