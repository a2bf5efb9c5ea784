"""GPU: the audio front-end's GEMM kernel alone (amuse_debug_gemm / amuse_debug_tile through the C ABI) against a torch fp32
product of the same bf16 operands: the reference's Linear layers of a DeiT block (models/audio/audio_main_new.py:174-204) are
y = x W^T + b.  Covers the tile-major layout round trip (bit exact), the ragged last row tile, the smallest and the four
production shapes' N / K, both debug epilogues, and run-to-run determinism."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from amuse_amd import _lib
    return _lib


def _p(t):
    return C.c_void_p(t.data_ptr())


def _pack_w(w):
    """[N, K] -> the kernel's fragment order (amuse_audio_api.hip pack_w): [span][x = (p, q)][k-step][lane = (g, i = (a, b))][e],
    feature = 64 span + 32 p + 8 a + 4 q + b, k = 32 ks + 8 g + e."""
    N, K = w.shape
    v = w.view(N // 64, 2, 4, 2, 4, K // 32, 4, 8)          # span, p, a, q, b, ks, g, e
    return v.permute(0, 1, 3, 5, 6, 2, 4, 7).contiguous().view(-1)


def _tile_index(M, F):
    """element index of (row, f) in the bf16 tile-major layout (amuse_audio.hpp tm_bf16)"""
    r = torch.arange(M).view(-1, 1)
    f = torch.arange(F).view(1, -1)
    return ((r // 16) * (F // 32) + f // 32) * 512 + (((f % 32) // 8) * 16 + r % 16) * 8 + f % 8


def test_tile_round_trip_is_bit_exact_and_matches_the_documented_index(lib):
    L = lib.load()
    for (M, F) in ((1, 32), (37, 64), (300, 768), (1216, 256)):
        Mp = (M + 127) // 128 * 128
        a = torch.randn(M, F, device="cuda").to(torch.bfloat16)
        at = torch.full((Mp * F,), 7.0, device="cuda", dtype=torch.bfloat16)
        lib.check(L.amuse_debug_tile(_p(a), _p(at), M, F, 0, None))
        back = torch.empty(M, F, device="cuda", dtype=torch.bfloat16)
        lib.check(L.amuse_debug_tile(_p(at), _p(back), M, F, 1, None))
        assert torch.equal(back, a)
        idx = _tile_index(M, F).cuda()
        assert torch.equal(at[idx.view(-1)].view(M, F), a)
        pad = torch.ones(Mp * F, dtype=torch.bool, device="cuda")
        pad[_tile_index(Mp, F).cuda()[:M].reshape(-1)] = False
        assert bool((at[pad] == 0).all())                      # pad rows are zeroed


@pytest.mark.parametrize("M,N,K", [(128, 256, 64), (100, 256, 256), (1216, 768, 768), (1216 * 2 + 77, 2304, 768),
                                   (640, 3072, 768), (513, 768, 3072)])
def test_gemm_matches_fp32_product_of_the_bf16_operands(lib, M, N, K):
    L = lib.load()
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    Mp = (M + 127) // 128 * 128
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    at = torch.empty(Mp * K, device="cuda", dtype=torch.bfloat16)
    lib.check(L.amuse_debug_tile(_p(a), _p(at), M, K, 0, None))
    wpk = _pack_w(w)
    ref = a.float() @ w.float().T + bias
    scale = float(ref.abs().max())
    # epilogue 3: fp32, tile-major
    ot = torch.empty(Mp * N, device="cuda", dtype=torch.float32)
    lib.check(L.amuse_debug_gemm(_p(at), _p(wpk), _p(bias), M, N, K, 3, _p(ot), None))
    o32 = torch.empty(M, N, device="cuda", dtype=torch.float32)
    lib.check(L.amuse_debug_tile(_p(ot), _p(o32), M, N, 2, None))
    assert float((o32 - ref).abs().max()) < 2e-5 * scale * (K / 64) ** 0.5 + 1e-5      # fp32 accumulation order only
    # epilogue 0: bf16, tile-major - the fp32 result rounded once
    ob = torch.empty(Mp * N, device="cuda", dtype=torch.bfloat16)
    lib.check(L.amuse_debug_gemm(_p(at), _p(wpk), _p(bias), M, N, K, 0, _p(ob), None))
    o16 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    lib.check(L.amuse_debug_tile(_p(ob), _p(o16), M, N, 1, None))
    assert torch.equal(o16, o32.to(torch.bfloat16))
    # the same launch again: bit-identical
    ob2 = torch.empty_like(ob)
    lib.check(L.amuse_debug_gemm(_p(at), _p(wpk), _p(bias), M, N, K, 0, _p(ob2), None))
    lib.check(L.amuse_debug_tile(_p(ob2), _p(o16), M, N, 1, None))
    assert torch.equal(o16, o32.to(torch.bfloat16))


def test_small_launch_shape_is_bitwise_the_production_shape(lib):
    """Launches of few tiles run 128-feature tiles with a nine-stage ring, large ones 256-feature tiles with three stages: a row's
    result must not depend on which (the audio front-end's features do not depend on the batch size)."""
    L = lib.load()
    g = torch.Generator(device="cuda").manual_seed(3)
    for (N, K) in ((768, 3072), (2304, 768)):
        Mbig, Msmall = 128 * 70, 1216                                   # 210+ tiles against <= 90 tiles of 256 features
        a = torch.randn(Mbig, K, device="cuda", generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device="cuda", generator=g)
        wpk = _pack_w(w)
        outs = []
        for M in (Mbig, Msmall):
            Mp = (M + 127) // 128 * 128
            at = torch.empty(Mp * K, device="cuda", dtype=torch.bfloat16)
            lib.check(L.amuse_debug_tile(_p(a[:M].contiguous()), _p(at), M, K, 0, None))
            ot = torch.empty(Mp * N, device="cuda", dtype=torch.float32)
            lib.check(L.amuse_debug_gemm(_p(at), _p(wpk), _p(bias), M, N, K, 3, _p(ot), None))
            o = torch.empty(M, N, device="cuda", dtype=torch.float32)
            lib.check(L.amuse_debug_tile(_p(ot), _p(o), M, N, 2, None))
            outs.append(o)
        assert torch.equal(outs[0][:Msmall], outs[1])


def test_gemm_rejects_bad_shapes(lib):
    L = lib.load()
    x = torch.zeros(128 * 64, device="cuda", dtype=torch.bfloat16)
    b = torch.zeros(256, device="cuda")
    for (M, N, K, epi) in ((128, 128, 64, 0), (128, 256, 32, 0), (0, 256, 64, 0), (128, 256, 64, 1)):
        assert L.amuse_debug_gemm(_p(x), _p(x), _p(b), M, N, K, epi, _p(x), None) != 0
    assert L.amuse_debug_tile(_p(x), _p(x), 16, 48, 0, None) != 0
