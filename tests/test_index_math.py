"""Host replicas of the division shortcuts the kernels use for index decompositions, checked exhaustively over the
ranges the launch code admits (conv.hip / convs.hip: idiv_small, P < 2^22; litechain.hip: 16-bit magic multiply)."""
import numpy as np
import pytest


def idiv_small(i, d):
    """fastmot_amd/csrc/net.h::idiv_small in float32 arithmetic: (int)(((float)i + 0.5f) * (1.f / (float)d))"""
    inv = np.float32(1.0) / np.float32(d)
    return ((i.astype(np.float32) + np.float32(0.5)) * inv).astype(np.int32)       # (float32 product, truncation)


@pytest.mark.parametrize('d', [1, 2, 3, 5, 7, 8, 9, 10, 19, 38, 46, 64, 76, 91, 152, 184, 255, 304, 361, 368, 400, 608,
                               640, 1024, 1280, 1444, 2304, 4096, 5776, 23104, 92416, 369664, 1638400, 4194303])
def test_idiv_small_exact_below_2_22(d):
    i = np.arange(0, 1 << 22, dtype=np.int64)
    np.testing.assert_array_equal(idiv_small(i, d), (i // d).astype(np.int32))


def test_idiv_small_random_divisors():
    rng = np.random.default_rng(0)
    i = rng.integers(0, 1 << 22, 200000)
    for d in rng.integers(1, 1 << 22, 300):
        np.testing.assert_array_equal(idiv_small(i, int(d)), (i // int(d)).astype(np.int32))


def test_litechain_magic_division():
    """floor(i / d) == (i * ceil(2^16 / d)) >> 16 for every i < 2048 and d <= 32 (regions are at most 24 x 24)"""
    i = np.arange(2048, dtype=np.uint32)
    for d in range(1, 33):
        rcp = np.uint32((65536 + d - 1) // d)
        np.testing.assert_array_equal((i * rcp) >> 16, i // d)
