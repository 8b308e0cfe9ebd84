"""GPU: every C-ABI op of libconformer_gfx950 against an independent computation on the same inputs.

The checker here is plain torch math in f32/f64 (on the 16-bit-rounded operands where the op rounds them), plus the
numpy oracle for the bit-exact mask path.  All calls go through the ctypes binding -> C ABI -> HIP kernels.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfm():
    import cfm as c
    assert torch.cuda.is_available(), "these tests need the MI355X"
    assert c.lib().cfm_device_ok() == 1, c.lib().cfm_last_error()
    return c


def rnd(shape, seed, scale=1.0, device="cuda"):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(device)


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


W_DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


@pytest.mark.parametrize("M,N,K", [(7968, 256, 256), (200, 144, 144), (98, 576, 144), (98, 144, 576), (333, 432, 144),
                                   (64, 64, 64), (1, 256, 256), (130, 132, 72), (513, 2048, 256)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
@pytest.mark.parametrize("tile", [0, 1, 2, 3])
def test_gemm_plain(cfm, M, N, K, wdt, tile):
    a = rnd((M, K), 1)
    w = rnd((N, K), 2, K ** -0.5)
    bias = rnd((N,), 3, 0.1)
    w16 = w.to(W_DT[wdt])
    a16 = a.to(W_DT[wdt])
    ref = a16.float() @ w16.float().t() + bias
    # 16-bit A, f32 out
    out = cfm.gemm(a16, w16, bias=bias, tile=tile)
    assert out.dtype == torch.float32 and relerr(out, ref) < 2e-5
    # f32 A (rounded while staging), 16-bit out
    out16 = cfm.gemm(a, w16, bias=bias, out_dtype=W_DT[wdt], tile=tile)
    assert out16.dtype == W_DT[wdt]
    assert relerr(out16.float(), ref.to(W_DT[wdt]).float()) < (1e-2 if wdt == "bf16" else 2e-3)


@pytest.mark.parametrize("M,N,K", [(7968, 256, 2048), (250, 144, 576), (77, 72, 40)])
def test_gemm_epilogues(cfm, M, N, K):
    a = rnd((M, K), 4).bfloat16()
    w = rnd((N, K), 5, K ** -0.5).bfloat16()
    bias = rnd((N,), 6, 0.1)
    res = rnd((M, N), 7)
    mask = (torch.rand(M, device="cuda") > 0.3).to(torch.uint8)
    lin = a.float() @ w.float().t()
    # silu / relu
    assert relerr(cfm.gemm(a, w, bias=bias, act=cfm.ACT_SILU), torch.nn.functional.silu(lin + bias)) < 2e-5
    assert relerr(cfm.gemm(a, w, bias=bias, act=cfm.ACT_RELU), torch.relu(lin + bias)) < 2e-5
    # residual + alpha, in place and out of place; inputs must not change when out-of-place
    res0 = res.clone()
    out = cfm.gemm(a, w, bias=bias, residual=res, alpha=0.5)
    assert torch.equal(res, res0)
    assert relerr(out, res0 + 0.5 * (lin + bias)) < 2e-5
    cfm.gemm(a, w, bias=bias, residual=res, alpha=0.5, out=res)
    assert relerr(res, res0 + 0.5 * (lin + bias)) < 2e-5
    # output-row mask then residual (pointwise_conv2 + masked_fill + residual)
    out = cfm.gemm(a, w, bias=bias, residual=res0, alpha=1.0, row_mask=mask)
    assert relerr(out, res0 + (lin + bias) * mask[:, None].float()) < 2e-5
    # input-row mask (masked_fill before pointwise_conv1): dead rows still get the bias
    out = cfm.gemm(a, w, bias=bias, row_mask=mask, mask_mode=1)
    assert relerr(out, lin * mask[:, None].float() + bias) < 2e-5


@pytest.mark.parametrize("M,D", [(7968, 256), (98, 144), (31, 512)])
def test_gemm_glu(cfm, M, D):
    a = rnd((M, D), 8).bfloat16()
    w = rnd((2 * D, D), 9, D ** -0.5)
    bias = rnd((2 * D,), 10, 0.2)
    # interleave as the packer does: GEMM col blk*32 + half*16 + i  <-  weight row half*D + blk*16 + i
    idx = torch.arange(2 * D, device="cuda")
    src = ((idx % 32) // 16) * D + (idx // 32) * 16 + (idx % 16)
    wp = w[src].bfloat16().contiguous()
    bp = bias[src].contiguous()
    lin = a.float() @ w.bfloat16().float().t() + bias
    ref = lin[:, :D] * torch.sigmoid(lin[:, D:])
    out = cfm.gemm(a, wp, bias=bp, act=cfm.ACT_GLU)
    assert out.shape == (M, D) and relerr(out, ref) < 2e-5


@pytest.mark.parametrize("M,N,K", [(7968, 256, 256), (98, 144, 576), (333, 576, 144)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3])
def test_gemm_split_is_f32_accurate(cfm, M, N, K, tile):
    a = rnd((M, K), 11)
    w = rnd((N, K), 12, K ** -0.5)
    bias = rnd((N,), 13, 0.1)
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    ref = a.double() @ w.double().t() + bias.double()
    out = cfm.gemm(a, hi, bias=bias, w_lo=lo, tile=tile)
    assert relerr(out, ref) < 4e-5
    single = cfm.gemm(a, hi, bias=bias, tile=tile)
    assert relerr(single, ref) > 10 * relerr(out, ref)      # the split really buys precision


@pytest.mark.parametrize("B,T1,F1,C,N", [(2, 21, 39, 144, 144), (3, 99, 39, 256, 256), (1, 7, 9, 16, 32)])
def test_gemm_conv3x3s2(cfm, B, T1, F1, C, N):
    img = torch.relu(rnd((B, T1, F1, C), 14)).bfloat16()
    w = rnd((N, C, 3, 3), 15, (9 * C) ** -0.5)
    bias = rnd((N,), 16, 0.1)
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    wk = w.permute(0, 2, 3, 1).reshape(N, 9 * C).bfloat16().contiguous()       # [co][kt][kf][ci]
    ref = torch.nn.functional.conv2d(img.float().permute(0, 3, 1, 2), w.bfloat16().float(), bias, stride=2)
    ref = torch.relu(ref).permute(0, 2, 3, 1).reshape(B * T2 * F2, N)
    for tile in (0, 1, 2, 3):
        out = cfm.gemm(img, wk, bias=bias, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, B * T2 * F2), tile=tile)
        assert relerr(out, ref) < 2e-5, tile
    hi = w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous()
    whi = hi.bfloat16()
    wlo = (hi - whi.float()).bfloat16()
    img32 = torch.relu(rnd((B, T1, F1, C), 14))
    ref = torch.relu(torch.nn.functional.conv2d(img32.double().permute(0, 3, 1, 2), w.double(), bias.double(), stride=2))
    ref = ref.permute(0, 2, 3, 1).reshape(B * T2 * F2, N)
    out = cfm.gemm(img32, whi, bias=bias, w_lo=wlo, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, B * T2 * F2))
    assert relerr(out, ref) < 4e-5


@pytest.mark.parametrize("M,N,K,ldc", [(300, 5002, 512, 5002), (129, 70, 64, 70), (77, 130, 72, 134), (64, 128, 64, 130), (5, 2, 8, 2)])
@pytest.mark.parametrize("tile", [0, 1, 3])
def test_gemm_pair_stores(cfm, M, N, K, ldc, tile):
    """N or the output row stride only a multiple of 2 (a vocabulary of 5002 columns written in the reference's own layout):
    column-pair stores, the last pair of a row predicated; nothing outside [M, N] is written."""
    a = rnd((M, K), 61).bfloat16()
    w = rnd((N, K), 62, K ** -0.5).bfloat16()
    bias = rnd((N,), 63, 0.1)
    ref = a.float() @ w.float().t() + bias
    for odt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1e-2), (torch.float16, 2e-3)):
        buf = torch.full((M, ldc), 7.0, dtype=odt, device="cuda")
        out = cfm.gemm(a, w, bias=bias, out=buf[:, :N], tile=tile)
        assert relerr(out.float(), ref) < tol, (odt, relerr(out.float(), ref))
        assert bool((buf[:, N:] == 7.0).all())
    if N % 4 or ldc % 4:
        with pytest.raises(RuntimeError, match="multiples of 4"):
            cfm.gemm(a, w, bias=bias, out=torch.empty((M, ldc), device="cuda")[:, :N], residual=torch.zeros((M, ldc), device="cuda")[:, :N])
    with pytest.raises(RuntimeError, match="multiple of 2"):
        cfm.gemm(a, rnd((N + 1, K), 64).bfloat16())


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(2380, 256, 2048), (77, 132, 1096), (333, 64, 1024), (40, 256, 72)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
@pytest.mark.parametrize("tile", [9, 10, 11])
def test_gemm_two_k_groups_matches_one(cfm, M, N, K, wdt, tile):
    """Tile ids 9 / 10 / 11 (32 x 64 k2, 64 x 64 k4, 64 x 64 k2: K groups per tile that meet through LDS): against the one-group tile with every epilogue family -- K tails,
    an odd tile count per group, ragged M / N, residual + alpha, SiLU, GLU, row masks; the sums are regrouped (half + half), so f32 close."""
    torch.manual_seed(M + K)
    dt = W_DT[wdt]
    a = torch.randn((M, K), device="cuda").to(dt)
    w = (torch.randn((N, K), device="cuda") / K ** 0.5).to(dt)
    bias = torch.randn((N,), device="cuda")
    res = torch.randn((M, N), device="cuda")
    mask = (torch.rand((M,), device="cuda") > 0.2).to(torch.uint8)
    for kw in (dict(), dict(act=cfm.ACT_SILU, out_dtype=dt), dict(residual=res, alpha=0.5), dict(row_mask=mask)) + ((dict(act=cfm.ACT_GLU),) if N % 64 == 0 else ()):
        ref = cfm.gemm(a, w, bias=bias, tile=5, **kw)
        out = cfm.gemm(a, w, bias=bias, tile=tile, **kw)
        assert out.shape == ref.shape and relerr(out, ref) < (2e-6 if out.dtype == torch.float32 else 1e-2), kw.keys()
    again = cfm.gemm(a, w, bias=bias, tile=tile)
    assert torch.equal(again, cfm.gemm(a, w, bias=bias, tile=tile))          # fixed merge order: reproducible


@pytest.mark.parametrize("M,N,K", [(4096, 2304, 192), (20000, 1000, 576), (300, 5002, 512), (70000, 1282, 64), (100, 100, 64)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_gemm_persistent_matches_tiled(cfm, M, N, K, wdt):
    """Persistent-workgroup GEMM (tile id 7: 512 resident workgroups walking the tile list, prefetch ring across tile boundaries) is
    bit-identical to the one-tile-per-workgroup kernel: several tiles per workgroup, K tails, ragged M / N edges, N % 4 == 2."""
    a = rnd((M, K), 81).to(W_DT[wdt])
    w = rnd((N, K), 82, K ** -0.5).to(W_DT[wdt])
    bias = rnd((N,), 83, 0.1)
    for odt in (torch.float32, W_DT[wdt]):
        for act in (cfm.ACT_NONE, cfm.ACT_SILU):
            ref = cfm.gemm(a, w, bias=bias, out_dtype=odt, act=act, tile=1)
            out = cfm.gemm(a, w, bias=bias, out_dtype=odt, act=act, tile=7)
            assert torch.equal(out, ref), (odt, act)
    assert torch.equal(cfm.gemm(a, w, out_dtype=torch.float32, tile=7), cfm.gemm(a, w, out_dtype=torch.float32, tile=1))
    with pytest.raises(RuntimeError, match="K % 64"):
        cfm.gemm(a[:, :K - 8], w[:, :K - 8].contiguous(), out_dtype=torch.float32, tile=7)
    with pytest.raises(RuntimeError, match="persistent"):
        cfm.gemm(a, w, out_dtype=torch.float32, tile=7, residual=torch.zeros((M, N), device="cuda")) if N % 4 == 0 else cfm.gemm(
            a, w, out_dtype=torch.float32, tile=7, row_mask=torch.ones(M, dtype=torch.uint8, device="cuda"))


@pytest.mark.parametrize("M,N,K", [(4096, 2304, 192), (20000, 1000, 576), (300, 5002, 512), (7000, 1282, 64), (100, 100, 64), (257, 514, 128)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_gemm_256_tile_matches_tiled(cfm, M, N, K, wdt):
    """256 x 256 tile with LDS-DMA staging (tile id 8, csrc/gemm256.hip) is bit-identical to the 128 x 128 register-staged kernel:
    ragged M / N edges, N % 4 == 2, one K tile, every output dtype and epilogue it takes."""
    a = rnd((M, K), 91).to(W_DT[wdt])
    w = rnd((N, K), 92, K ** -0.5).to(W_DT[wdt])
    bias = rnd((N,), 93, 0.1)
    for odt in (torch.float32, W_DT[wdt]):
        for act in (cfm.ACT_NONE, cfm.ACT_SILU, cfm.ACT_RELU):
            ref = cfm.gemm(a, w, bias=bias, out_dtype=odt, act=act, tile=1)
            out = cfm.gemm(a, w, bias=bias, out_dtype=odt, act=act, tile=8)
            assert torch.equal(out, ref), (odt, act)
    assert torch.equal(cfm.gemm(a, w, out_dtype=torch.float32, tile=8), cfm.gemm(a, w, out_dtype=torch.float32, tile=1))
    with pytest.raises(RuntimeError, match="K % 64"):
        cfm.gemm(a[:, :K - 8], w[:, :K - 8].contiguous(), out_dtype=torch.float32, tile=8)
    with pytest.raises(RuntimeError, match="256x256"):
        cfm.gemm(a, w, out_dtype=torch.float32, tile=8, row_mask=torch.ones(M, dtype=torch.uint8, device="cuda"))


def test_gemm_auto_split_matches_single_kernel(cfm):
    """tile=0 on a shape whose 256 x 256 tiles make 2.3 rounds on 256 CUs (the front-end convolution at config 2; a plain product of the
    same M, N): whole rounds go to the 256 x 256 kernel, the remaining rows to 128 x 128 tiles -- two launches, one result, bit-identical
    to either kernel alone."""
    B, T1, F1, C, N = 32, 499, 39, 256, 256
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    img = rnd((B, T1, F1, C), 101).bfloat16()
    w = rnd((N, 9 * C), 102, (9 * C) ** -0.5).bfloat16()
    bias = rnd((N,), 103, 0.1)
    conv = (C, T1, F1, T2, F2, B * T2 * F2)
    ref = cfm.gemm(img, w, bias=bias, act=cfm.ACT_RELU, conv=conv, out_dtype=torch.bfloat16, tile=1)
    cfm.prof_reset(); cfm.prof_enable(True)
    out = cfm.gemm(img, w, bias=bias, act=cfm.ACT_RELU, conv=conv, out_dtype=torch.bfloat16)
    torch.cuda.synchronize(); cfm.prof_enable(False)
    assert torch.equal(out, ref)
    names = set(cfm.prof_table())
    assert any(n.endswith("256x256") for n in names) and any(n.endswith("128x128") for n in names), names
    a = rnd((B * T2 * F2, 512), 104).bfloat16()
    w2 = rnd((N, 512), 105, 512 ** -0.5).bfloat16()
    assert torch.equal(cfm.gemm(a, w2, bias=bias, out_dtype=torch.float32), cfm.gemm(a, w2, bias=bias, out_dtype=torch.float32, tile=1))
    # two N tiles of 256, fp16 operands, SiLU, a ragged last M tile: 548 tiles of 256 x 256 = 2.14 rounds -> split at row 65 536
    a = rnd((70000, 512), 106).half()
    w3 = rnd((512, 512), 107, 512 ** -0.5).half()
    b3 = rnd((512,), 108, 0.1)
    for odt in (torch.float32, torch.float16):
        cfm.prof_reset(); cfm.prof_enable(True)
        out = cfm.gemm(a, w3, bias=b3, act=cfm.ACT_SILU, out_dtype=odt)
        torch.cuda.synchronize(); cfm.prof_enable(False)
        names = set(cfm.prof_table())
        assert len(names) == 2 and any(n.endswith("256x256") for n in names), names      # the tail picks its own (smaller) tile
        assert torch.equal(out, cfm.gemm(a, w3, bias=b3, act=cfm.ACT_SILU, out_dtype=odt, tile=1))


@pytest.mark.parametrize("B,T1,F1,C,N", [(2, 21, 17, 64, 64), (3, 45, 39, 256, 256), (1, 9, 9, 128, 320)])
def test_gemm_256_tile_conv(cfm, B, T1, F1, C, N):
    """The implicit 3x3 / stride-2 convolution through the 256 x 256 tile equals the 128 x 128 kernel bit for bit."""
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    img = rnd((B, T1, F1, C), 95).bfloat16()
    w = rnd((N, 9 * C), 96, (9 * C) ** -0.5).bfloat16()
    bias = rnd((N,), 97, 0.1)
    conv = (C, T1, F1, T2, F2, B * T2 * F2)
    for odt in (torch.float32, torch.bfloat16):
        ref = cfm.gemm(img, w, bias=bias, act=cfm.ACT_RELU, conv=conv, out_dtype=odt, tile=1)
        out = cfm.gemm(img, w, bias=bias, act=cfm.ACT_RELU, conv=conv, out_dtype=odt, tile=8)
        assert torch.equal(out, ref)


@pytest.mark.parametrize("B,T,U,J", [(2, 7, 5, 64), (1, 1, 1, 512), (3, 33, 9, 512), (2, 5, 3, 72)])
def test_joint_act(cfm, B, T, U, J):
    """cfm_joint_act: tanh(enc[b,t] + pred[b,u]) as a [B*T*U, J] operand, against torch (joint.py:31-37)."""
    enc, pred = rnd((B * T, J), 71, 1.5), rnd((B * U, J), 72, 1.5)
    enc[0, :4] = torch.tensor([40.0, -40.0, 0.0, 1e-4])     # saturation and the small-argument end
    ref = torch.tanh(enc.view(B, T, 1, J).double() + pred.view(B, 1, U, J).double()).view(B * T * U, J)
    out = cfm.joint_act(enc, pred, B, T, U, torch.float32)
    assert float((out.double() - ref).abs().max()) < 5e-7
    for odt, tol in ((torch.bfloat16, 4e-3), (torch.float16, 5e-4)):
        out = cfm.joint_act(enc, pred, B, T, U, odt)
        assert out.dtype == odt and float((out.double() - ref).abs().max()) < tol
    wide = torch.zeros((B * T, J + 8), device="cuda")
    wide[:, :J] = enc
    assert torch.equal(cfm.joint_act(wide[:, :J], pred, B, T, U, torch.float32), cfm.joint_act(enc, pred, B, T, U, torch.float32))
    with pytest.raises(ValueError):
        cfm.joint_act(enc, pred, B, T + 1, U, torch.float32)


def test_gemm_rejects_bad_arguments(cfm):
    a = rnd((8, 12), 1).bfloat16()
    w = rnd((8, 12), 2).bfloat16()
    with pytest.raises(RuntimeError, match="multiple of 8"):
        cfm.gemm(a, w)
    with pytest.raises(RuntimeError, match="no CPU path"):
        cfm.gemm(a.cpu(), w.cpu())


@pytest.mark.parametrize("M,D", [(7968, 256), (98, 144), (5, 512), (3, 1024), (2, 2048)])
def test_layernorm(cfm, M, D):
    x = rnd((M, D), 20, 2.0) + 0.5
    g1, b1 = 1 + 0.1 * rnd((D,), 21), 0.1 * rnd((D,), 22)
    g2, b2 = 1 + 0.1 * rnd((D,), 23), 0.1 * rnd((D,), 24)
    ref1 = torch.nn.functional.layer_norm(x, (D,), g1, b1, 1e-5)
    ref2 = torch.nn.functional.layer_norm(ref1, (D,), g2, b2, 1e-5)
    y1, _ = cfm.layernorm(x, g1, b1)
    assert relerr(y1, ref1) < 2e-6
    mask = (torch.rand(M, device="cuda") > 0.3).to(torch.uint8)
    y1, y2 = cfm.layernorm(x, g1, b1, g2=g2, b2=b2, out2_dtype=torch.bfloat16, row_mask=mask)
    assert relerr(y1, ref1) < 2e-6
    assert relerr(y2.float(), (ref2 * mask[:, None]).bfloat16().float()) < 1e-2
    _, y3 = cfm.layernorm(x, g1, b1, want1=False, out2_dtype=torch.float16)
    assert relerr(y3.float(), ref1) < 2e-3
    xc = x.clone()
    cfm.layernorm(xc, g1, b1, out1=xc)                       # in place
    assert relerr(xc, ref1) < 2e-6


def attn_reference(q, k, v, p, u, vb, mask, scale):
    """q (B,Tq,H,dk) k,v (B,Tk,H,dk) p (B,P,H,dk)|None, all float64."""
    qh, kh, vh = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    if p is not None:
        s = torch.einsum("bhid,bhjd->bhij", qh + u[None, :, None, :], kh)
        s = s + torch.einsum("bhid,bhjd->bhij", qh + vb[None, :, None, :], p.permute(0, 2, 1, 3))
    else:
        s = torch.einsum("bhid,bhjd->bhij", qh, kh)
    s = s * scale
    if mask is not None:
        dead = mask.unsqueeze(1) == 0
        s = s.masked_fill(dead, float("-inf"))
        a = torch.softmax(s, -1).masked_fill(dead, 0.0)
        a = torch.nan_to_num(a, nan=0.0)
    else:
        a = torch.softmax(s, -1)
    o = torch.einsum("bhij,bhjd->bhid", a, vh)
    return o.permute(0, 2, 1, 3).reshape(q.size(0), q.size(1), -1)


@pytest.mark.parametrize("B,H,Tq,Tk,dk", [(32, 4, 249, 249, 64), (2, 4, 49, 49, 36), (1, 4, 16, 80, 36), (3, 8, 70, 130, 64), (2, 2, 5, 5, 8),
                                          (2, 4, 411, 411, 64), (1, 4, 16, 700, 64), (2, 4, 300, 257, 64), (1, 2, 33, 513, 36)])   # > 256 keys: several super-tiles
@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("pos", ["none", "broadcast", "perkey"])
@pytest.mark.parametrize("masking", ["none", "pad", "full"])
def test_attention(cfm, B, H, Tq, Tk, dk, mode, pos, masking):
    D = H * dk
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[mode]
    qkv = rnd((B, Tq, 3 * D), 30).to(dt)
    kv_src = qkv if Tk == Tq else rnd((B, Tk, 3 * D), 31).to(dt)
    q = qkv[..., :D].reshape(B, Tq, H, dk)
    k = kv_src[..., D:2 * D].reshape(B, Tk, H, dk)
    v = kv_src[..., 2 * D:].reshape(B, Tk, H, dk)
    u, vb = rnd((H, dk), 32, 0.3), rnd((H, dk), 33, 0.3)
    P = {"none": 0, "broadcast": 1, "perkey": Tk}[pos]
    p = rnd((B, P, D), 34).to(dt) if P else None
    mask = None
    if masking == "pad":
        lens = torch.randint(1, Tk + 1, (B,), generator=torch.Generator().manual_seed(35))
        lens[0] = Tk
        mask = (torch.arange(Tk)[None, :] < lens[:, None]).unsqueeze(1).cuda()              # (B,1,Tk)
    elif masking == "full":
        mask = (torch.rand(B, Tq, Tk, generator=torch.Generator().manual_seed(36)) > 0.4).cuda()
        mask[:, min(3, Tq - 1), :] = False                                                     # a fully masked row (Q2)
    out = torch.empty((B, Tq, D), dtype=dt, device="cuda")
    m8 = cfm.as_u8_mask(mask) if mask is not None else None
    mstr = (0, 0) if mask is None else (m8.stride(0), m8.stride(1) if m8.size(1) > 1 else 0)
    cfm.attention(qkv, kv_src[..., D:], kv_src[..., 2 * D:], B, H, Tq, Tk, dk,
                  (Tq * 3 * D, 3 * D), (Tk * 3 * D, 3 * D, dk), (Tk * 3 * D, 3 * D, dk), out,
                  p=p, p_str=(P * D, D if P > 1 else 0), bias_u=u if P else None, bias_v=vb if P else None,
                  mask=m8, mask_str=mstr, mma_code=cfm.F16 if mode == "fp16" else cfm.BF16, split=mode == "fp32")
    ref = attn_reference(q.double(), k.double(), v.double(), p.double().reshape(B, P, H, dk) if P else None, u.double(), vb.double(),
                         mask, 1 / math.sqrt(dk))
    tol = {"bf16": 2e-2, "fp16": 3e-3, "fp32": 2e-4}[mode]
    assert torch.isfinite(out.float()).all()
    assert relerr(out.float(), ref) < tol
    if masking == "full":
        assert float(out[:, min(3, Tq - 1)].float().abs().max()) == 0.0


def test_attention_kv_cache_layout(cfm):
    B, H, Tc, Tn, dk = 1, 4, 20, 16, 36
    D = H * dk
    qkv = rnd((B, Tn, 3 * D), 40).bfloat16()
    cache = rnd((B, H, Tc, 2 * dk), 41)
    new = cfm.kv_cache_pack(cache, qkv[..., D:], qkv[..., 2 * D:], (Tn * 3 * D, 3 * D), (Tn * 3 * D, 3 * D), B, H, Tn, dk)
    k_new = qkv[..., D:2 * D].float().reshape(B, Tn, H, dk).permute(0, 2, 1, 3)
    v_new = qkv[..., 2 * D:].float().reshape(B, Tn, H, dk).permute(0, 2, 1, 3)
    ref = torch.cat([torch.cat([cache[..., :dk], k_new], 2), torch.cat([cache[..., dk:], v_new], 2)], -1)
    assert torch.equal(new, ref)
    # attention reading K/V straight from the (B,H,Tk,2dk) f32 cache layout
    Tk = Tc + Tn
    out = torch.empty((B, Tn, D), dtype=torch.bfloat16, device="cuda")
    cfm.attention(qkv, new, new[..., dk:], B, H, Tn, Tk, dk, (Tn * 3 * D, 3 * D), (H * Tk * 2 * dk, 2 * dk, Tk * 2 * dk),
                  (H * Tk * 2 * dk, 2 * dk, Tk * 2 * dk), out)
    q = qkv[..., :D].double().reshape(B, Tn, H, dk)
    ref_o = attn_reference(q, ref[..., :dk].double().permute(0, 2, 1, 3), ref[..., dk:].double().permute(0, 2, 1, 3), None, None, None,
                           None, 1 / math.sqrt(dk))
    assert relerr(out.float(), ref_o) < 2e-2


@pytest.mark.parametrize("B,T,D,K", [(32, 249, 256, 15), (3, 37, 144, 15), (2, 5, 16, 15), (2, 40, 32, 7), (3, 61, 512, 15), (2, 33, 384, 15), (2, 20, 64, 15)])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
def test_dwconv_bn_silu(cfm, B, T, D, K, dt):
    x = rnd((B, T, D), 50).to(dt)
    w = rnd((D, K), 51, 0.3)
    db, sc, sh = rnd((D,), 52, 0.1), 1 + 0.2 * rnd((D,), 53), rnd((D,), 54, 0.1)
    y = cfm.dwconv_bn_silu(x, w, db, sc, sh)
    ref = torch.nn.functional.conv1d(x.float().transpose(1, 2), w[:, None, :], db, padding=(K - 1) // 2, groups=D)
    ref = torch.nn.functional.silu(ref * sc[None, :, None] + sh[None, :, None]).transpose(1, 2)
    tol = {torch.bfloat16: 1e-2, torch.float16: 2e-3, torch.float32: 1e-5}[dt]
    assert y.dtype == dt and relerr(y.float(), ref) < tol


@pytest.mark.parametrize("F", [80, 83, 40, 24, 7])           # F % 4 == 0: row-window kernel; otherwise the scattered-position kernel
@pytest.mark.parametrize("B,T,C", [(2, 200, 144), (3, 83, 256), (1, 7, 8)])
def test_conv1_relu(cfm, B, T, C, F):
    x = rnd((B, T, F), 60)
    w = rnd((C, 1, 3, 3), 61, 1 / 3)
    b = rnd((C,), 62, 0.1)
    w9c = w.reshape(C, 9).t().contiguous()
    ref = torch.relu(torch.nn.functional.conv2d(x[:, None], w, b, stride=2)).permute(0, 2, 3, 1)
    y = cfm.conv1_relu(x, w9c, b, torch.float32)
    assert relerr(y, ref) < 2e-6
    y = cfm.conv1_relu(x, w9c, b, torch.bfloat16)
    assert relerr(y.float(), ref.bfloat16().float()) < 1e-2


@pytest.mark.parametrize("F", [80, 83, 7])
@pytest.mark.parametrize("B,T,C", [(2, 200, 144), (3, 83, 256), (1, 7, 16), (32, 1000, 256)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_conv1_relu_mma(cfm, B, T, C, F, wdt):
    """First convolution on the matrix pipe: equals the f32 convolution of the 16-bit-ROUNDED inputs and weights (what the MFMA
    multiplies), output rounded once; with CMVN folded in (applied in f32 before the rounding); close to the f32 FMA form."""
    if B == 32 and F != 80:
        pytest.skip("full config-2 size once")
    dt = W_DT[wdt]
    x = rnd((B, T, F), 60)
    w = rnd((C, 1, 3, 3), 61, 1 / 3)
    b = rnd((C,), 62, 0.1)
    w9c = w.reshape(C, 9).t().contiguous()
    assert cfm.conv1_relu_mma_supported(C, dt)
    y = cfm.conv1_relu(x, w9c, b, dt, mma=True)
    ref = torch.relu(torch.nn.functional.conv2d(x.to(dt).double()[:, None], w.to(dt).double(), b.double(), stride=2)).permute(0, 2, 3, 1)
    tol = 6e-3 if wdt == "bf16" else 8e-4                     # one output rounding
    assert y.dtype == dt and relerr(y.double(), ref) < tol
    fma = cfm.conv1_relu(x, w9c, b, dt)
    assert relerr(y.float(), fma.float()) < (2e-2 if wdt == "bf16" else 3e-3)
    mean, istd = rnd((F,), 63, 0.5), (1.0 + 0.2 * rnd((F,), 64)).abs() + 0.5
    yn = cfm.conv1_relu(x, w9c, b, dt, cmvn=(mean, istd), mma=True)
    xn = ((x - mean) * istd).to(dt).double()
    refn = torch.relu(torch.nn.functional.conv2d(xn[:, None], w.to(dt).double(), b.double(), stride=2)).permute(0, 2, 3, 1)
    assert relerr(yn.double(), refn) < tol
    with pytest.raises(RuntimeError, match="bf16 or fp16"):
        cfm.conv1_relu(x, w9c, b, torch.float32, mma=True)


def test_masks_bit_exact(cfm):
    from oracle import conformer_oracle as O
    lens = torch.tensor([200, 163, 7, 0, 1000], dtype=torch.int32, device="cuda")
    for T in (9, 200, 1000):
        got = cfm.valid_mask(lens, T).cpu().numpy()
        assert np.array_equal(got, ~O.pad_mask(lens.cpu().numpy(), T))
        tp = O.subsampled_len(T)
        if tp > 0:
            sub = cfm.valid_mask(lens.long(), tp, first=6, stride=4).cpu().numpy()
            assert np.array_equal(sub, O.subsample_mask((~O.pad_mask(lens.cpu().numpy(), T))[:, None, :])[:, 0, :])
    for size, c, left in [(17, 4, -1), (17, 4, 2), (16, 16, 0), (9, 1, 0), (9, 3, 1), (5, 8, -1), (49, 4, 2), (1, 1, 0), (249, 16, 4)]:
        assert np.array_equal(cfm.chunk_mask(size, c, left, "cuda").cpu().numpy(), O.chunk_mask(size, c, left)), (size, c, left)
    valid = cfm.valid_mask(torch.tensor([13, 8], dtype=torch.int32, device="cuda"), 13).unsqueeze(1)
    ch = cfm.chunk_mask(13, 3, 1, "cuda")
    assert np.array_equal(cfm.attn_mask_combine(valid, ch).cpu().numpy(), valid.cpu().numpy() & O.chunk_mask(13, 3, 1)[None])


def test_cast_and_add_rows(cfm):
    x = rnd((5, 7, 16), 70)
    assert torch.equal(cfm.cast(x, torch.bfloat16), x.bfloat16())
    assert torch.equal(cfm.cast(x.half(), torch.float32), x.half().float())
    y = rnd((6, 16), 71)
    add = rnd((2, 16), 72)
    ref = y + add.repeat_interleave(3, 0)
    assert torch.equal(cfm.add_rows(y, add, 3), ref)


def test_profiling_table(cfm):
    a = rnd((512, 256), 80).bfloat16()
    w = rnd((256, 256), 81).bfloat16()
    cfm.prof_reset()
    cfm.prof_enable(True)
    for _ in range(5):
        cfm.gemm(a, w)
    cfm.prof_enable(False)
    tab = cfm.prof_table()
    (name, e), = tab.items()
    assert name.startswith("gemm_bf16") and e["calls"] == 5 and e["ms"] > 0 and e["flops"] == 5 * 2.0 * 512 * 256 * 256
    cfm.prof_reset()


@pytest.mark.parametrize("M,D,FF", [(7968, 256, 2048), (98, 144, 576), (33, 256, 2048), (5, 144, 64), (64, 256, 160)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
@pytest.mark.parametrize("act", ["silu", "relu"])
def test_ffn_fused(cfm, M, D, FF, wdt, act):
    """One-launch LN -> W1 -> act -> W2 -> residual -> LN -> LN against the same chain in torch f32."""
    from cfm import packing
    dt = W_DT[wdt]
    x = rnd((M, D), 90, 1.5) + 0.3
    w1 = rnd((FF, D), 91, D ** -0.5)
    w2 = rnd((D, FF), 92, FF ** -0.5)
    b1, b2 = rnd((FF,), 93, 0.1), rnd((D,), 94, 0.1)
    lns = [(1 + 0.1 * rnd((D,), 95 + i), 0.1 * rnd((D,), 98 + i)) for i in range(3)]
    w1f, w2f = packing.pack_ffn_fragments(w1, w2, dt)
    actf = torch.nn.functional.silu if act == "silu" else torch.relu
    code = cfm.ACT_SILU if act == "silu" else cfm.ACT_RELU

    def ref(x, ln, alpha, add_x, ln1, ln2):
        a = torch.nn.functional.layer_norm(x, (D,), ln[0], ln[1], 1e-5) if ln else x
        h = actf(a.to(dt).float() @ w1.to(dt).float().t() + b1)
        y = alpha * (h.to(dt).float() @ w2.to(dt).float().t() + b2)
        if add_x:
            y = y + x
        y1 = torch.nn.functional.layer_norm(y, (D,), ln1[0], ln1[1], 1e-5) if ln1 else y
        y2 = torch.nn.functional.layer_norm(y1, (D,), ln2[0], ln2[1], 1e-5) if ln2 else y1
        return y1, y2

    tol = 6e-3 if wdt == "bf16" else 8e-4          # the hidden activation is rounded to 16 bit before the second product
    # macaron form: LN in, residual, second norm to a 16-bit operand
    x0 = x.clone()
    o32, o16 = cfm.ffn_fused(x, w1f, w2f, b1, b2, FF, act=code, ln=lns[0], alpha=0.5, add_x=True, ln2=lns[2], out16_dtype=dt)
    r1, r2 = ref(x, lns[0], 0.5, True, None, lns[2])
    assert torch.equal(x, x0)
    assert relerr(o32, r1) < tol and relerr(o16.float(), r2) < tol + (1e-2 if wdt == "bf16" else 2e-3)
    # final form: LN in, residual, norm_final, IN PLACE
    xi = x.clone()
    cfm.ffn_fused(xi, w1f, w2f, b1, b2, FF, act=code, ln=lns[0], alpha=0.5, add_x=True, ln1=lns[1], out_f32=xi)
    r1, _ = ref(x, lns[0], 0.5, True, lns[1], None)
    assert relerr(xi, r1) < tol
    # bare module form: no norms, no residual
    o32, _ = cfm.ffn_fused(x, w1f, w2f, b1, b2, FF, act=code)
    r1, _ = ref(x, None, 1.0, False, None, None)
    assert relerr(o32, r1) < tol
    # bitwise reproducible
    o32b, _ = cfm.ffn_fused(x, w1f, w2f, b1, b2, FF, act=code)
    assert torch.equal(o32, o32b)


@pytest.mark.parametrize("M,D,FF", [(7968, 256, 2048), (98, 144, 576), (33, 256, 2048), (1, 256, 2048), (31, 144, 576), (32, 256, 2048)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_rowchain_three_roles(cfm, M, D, FF, wdt):
    """The macaron / conv-in / final chains of a conformer block, each one launch, against the same chain in torch."""
    from cfm import packing
    dt = W_DT[wdt]
    code = cfm.BF16 if wdt == "bf16" else cfm.F16
    F = torch.nn.functional
    x = rnd((M, D), 100, 1.5) + 0.3
    a16 = rnd((M, D), 101).to(dt)
    w1, w2 = rnd((FF, D), 102, D ** -0.5), rnd((D, FF), 103, FF ** -0.5)
    b1, b2 = rnd((FF,), 104, 0.1), rnd((D,), 105, 0.1)
    wh, bh = rnd((D, D), 106, D ** -0.5), rnd((D,), 107, 0.1)
    wq, bq = rnd((3 * D, D), 108, D ** -0.5), rnd((3 * D,), 109, 0.1)
    wg, bg = rnd((2 * D, D), 110, D ** -0.5), rnd((2 * D,), 111, 0.2)
    lns = [(1 + 0.1 * rnd((D,), 112 + i), 0.1 * rnd((D,), 116 + i)) for i in range(3)]
    mask = (torch.rand(M, device="cuda") > 0.3).to(torch.uint8)
    w1f, w2n = packing.pack_frag_major(w1, dt), packing.pack_frag_major(w2, dt)
    r16 = lambda t: t.to(dt).float()
    lin = lambda a, w, b: r16(a) @ r16(w).t() + b
    ffn = lambda xn: lin(F.silu(lin(xn, w1, b1)), w2, b2)
    ln = lambda t, p: F.layer_norm(t, (D,), p[0], p[1], 1e-5)
    tol = 8e-3 if wdt == "bf16" else 1e-3

    # macaron: x1 = x + 1/2 FFN(LN(x)); qkv = LN_mha(x1) . Wqkv^T + b
    out = torch.empty_like(x)
    qkv = torch.empty((M, 3 * D), dtype=dt, device="cuda")
    cfm.rowchain(M, D, code, x=x, ln=lns[0], ffn=(w1f, w2n, b1, b2, FF), alpha=0.5, ln2=lns[1], out_f32=out,
                 tail=(packing.pack_frag_major(wq, dt), bq, 3 * D, False, qkv))
    x1 = x + 0.5 * ffn(ln(x, lns[0]))
    assert relerr(out, x1) < tol
    assert relerr(qkv.float(), lin(ln(x1, lns[1]), wq, bq)) < tol + (1e-2 if wdt == "bf16" else 2e-3)

    # conv-in: x2 = x + a16 . Wo^T + b (in place); glu = GLU(mask(LN_conv(x2)) . Wpw1^T + b)
    idx = packing.glu_interleave_index(D, "cuda")
    xi = x.clone()
    glu = torch.empty((M, D), dtype=dt, device="cuda")
    cfm.rowchain(M, D, code, head=(a16, packing.pack_frag_major(wh, dt), bh, xi, None), ln=lns[2], ln_mask=mask, out_f32=xi,
                 tail=(packing.pack_frag_major(wg[idx], dt), bg[idx].contiguous(), 2 * D, True, glu))
    x2 = x + lin(a16.float(), wh, bh)
    assert relerr(xi, x2) < tol
    pre = lin(ln(x2, lns[2]) * mask[:, None].float(), wg, bg)
    assert relerr(glu.float(), pre[:, :D] * torch.sigmoid(pre[:, D:])) < tol + (1e-2 if wdt == "bf16" else 2e-3)

    # final: x3 = x + mask(a16 . Wpw2^T + b); out = LN_final(x3 + 1/2 FFN(LN_ff(x3))), in place
    xi = x.clone()
    cfm.rowchain(M, D, code, head=(a16, packing.pack_frag_major(wh, dt), bh, xi, mask), ln=lns[0], ffn=(w1f, w2n, b1, b2, FF), alpha=0.5,
                 ln1=lns[1], out_f32=xi)
    x3 = x + lin(a16.float(), wh, bh) * mask[:, None].float()
    ref = ln(x3 + 0.5 * ffn(ln(x3, lns[0])), lns[1])
    assert relerr(xi, ref) < tol
    xj = x.clone()
    cfm.rowchain(M, D, code, head=(a16, packing.pack_frag_major(wh, dt), bh, xj, mask), ln=lns[0], ffn=(w1f, w2n, b1, b2, FF), alpha=0.5,
                 ln1=lns[1], out_f32=xj)
    assert torch.equal(xi, xj)          # bitwise reproducible


@pytest.mark.parametrize("B,T,D,FF", [(32, 249, 256, 2048), (3, 41, 144, 576), (2, 5, 256, 2048), (5, 32, 256, 2048), (1, 1, 256, 2048), (4, 3, 144, 576)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_rowchain_depthwise_input_stage(cfm, B, T, D, FF, wdt):
    """final chain with the depthwise conv + BatchNorm + SiLU (convolution.py:43-45) in its input stage == cfm_dwconv_bn_silu, then the
    plain final chain: same fp32 operation order (bit-identical in bf16), including frames next to utterance edges and a ragged last tile."""
    from cfm import packing
    dt = torch.bfloat16 if wdt == "bf16" else torch.float16
    code = cfm.BF16 if wdt == "bf16" else cfm.F16
    M = B * T
    x = rnd((M, D), 200, 1.2)
    glu = rnd((B, T, D), 201).to(dt)
    w1, w2 = rnd((FF, D), 202, D ** -0.5), rnd((D, FF), 203, FF ** -0.5)
    b1, b2 = rnd((FF,), 204, 0.1), rnd((D,), 205, 0.1)
    wh, bh = rnd((D, D), 206, D ** -0.5), rnd((D,), 207, 0.1)
    taps, tb = rnd((D, 15), 208, 0.3), rnd((D,), 209, 0.1)
    sc, sh = 1 + 0.2 * rnd((D,), 210), 0.1 * rnd((D,), 211)
    lns = [(1 + 0.1 * rnd((D,), 212 + i), 0.1 * rnd((D,), 216 + i)) for i in range(2)]
    mask = (torch.rand(M, device="cuda") > 0.2).to(torch.uint8)
    w1f, w2n, whf = packing.pack_frag_major(w1, dt), packing.pack_frag_major(w2, dt), packing.pack_frag_major(wh, dt)
    dwo = cfm.dwconv_bn_silu(glu, taps, tb, sc, sh, out_dtype=dt)
    ref = x.clone()
    cfm.rowchain(M, D, code, head=(dwo.view(M, D), whf, bh, ref, mask), ln=lns[0], ffn=(w1f, w2n, b1, b2, FF), alpha=0.5, ln1=lns[1], out_f32=ref)
    out = x.clone()
    cfm.rowchain(M, D, code, head=(glu.view(M, D), whf, bh, out, mask), ln=lns[0], ffn=(w1f, w2n, b1, b2, FF), alpha=0.5, ln1=lns[1], out_f32=out,
                 dw=(taps, tb, sc, sh, T))
    assert torch.isfinite(out).all()
    if wdt == "bf16":
        assert torch.equal(out, ref)
    else:
        # fp16: the fused kernel's compiler folds SiLU's last multiply into the f32->f16 conversion (v_fma_mixlo_f16, ONE rounding)
        # where the stand-alone kernel rounds the product to f32 first: a handful of conv outputs differ by one f16 ulp.
        assert relerr(out, ref) < 1e-3 and (out - ref).abs().max().item() < 2e-3 and (out != ref).float().mean().item() < 0.05
    with pytest.raises(RuntimeError):     # the stage belongs to the head + feed-forward chain only
        cfm.rowchain(M, D, code, x=x, ln=lns[0], out_f32=out, dw=(taps, tb, sc, sh, T))




# ------------------------------------------------------------------------------------------------------------ feed-forward split over FF
@pytest.mark.gpu
@pytest.mark.parametrize("M", [1, 16, 45, 1024])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_ffn_split_modes(cfm, M, wdt):
    """cfm_ffn_split (csrc/ffnsplit.hip) against plain torch math on the same 16-bit-rounded operands: the feed-forward as partial slabs, the reduce
    with residual / bias / alpha / LayerNorms, and the projection mode; ragged last tile, a single row."""
    from cfm import packing
    torch.manual_seed(M)
    D, FF, dt, code = 256, 2048, W_DT[wdt], (cfm.BF16 if wdt == "bf16" else cfm.F16)
    x = torch.randn((M, D), device="cuda")
    w1 = torch.randn((FF, D), device="cuda") / D ** 0.5
    w2 = torch.randn((D, FF), device="cuda") / FF ** 0.5
    b1, b2 = 0.1 * torch.randn((FF,), device="cuda"), 0.1 * torch.randn((D,), device="cuda")
    lng = [(1 + 0.1 * torch.randn((D,), device="cuda"), 0.1 * torch.randn((D,), device="cuda")) for _ in range(4)]
    w1f, w2n = packing.pack_frag_major(w1, dt), packing.pack_frag_major(w2, dt)
    ln = lambda t, p: torch.nn.functional.layer_norm(t, (D,), p[0], p[1], 1e-5)
    r16 = lambda t: t.to(dt).float()
    # mode 2: partial slabs; their sum is the second product (without b2)
    slabs = torch.full((FF // 256, M, D), float("nan"), device="cuda")
    cfm.ffn_split(x, code, 2, ln=lng[0], w1=w1f, b1=b1, n1=FF, act=cfm.ACT_SILU, w2=w2n, psum_out=slabs)
    h = r16(torch.nn.functional.silu(r16(ln(x, lng[0])) @ r16(w1).t() + b1))
    ref = h @ r16(w2).t()
    assert relerr(slabs.sum(0), ref) < 2e-3
    per = torch.stack([h[:, g * 256:(g + 1) * 256] @ r16(w2)[:, g * 256:(g + 1) * 256].t() for g in range(FF // 256)])
    assert relerr(slabs, per) < 2e-3
    # mode 0: x + 0.5 (sum + b2) -> LN1 -> rows_out -> LN2 -> rows2_out; in place over x is allowed
    y, y2, xin = torch.empty_like(x), torch.empty_like(x), x.clone()
    cfm.ffn_split(xin, code, 0, psum=slabs, psum_b2=b2, psum_alpha=0.5, ln1=lng[1], ln2=lng[2], rows_out=y, rows2_out=y2)
    want = ln(x + 0.5 * (slabs.sum(0) + b2), lng[1])
    assert relerr(y, want) < 1e-5 and relerr(y2, ln(want, lng[2])) < 1e-5
    cfm.ffn_split(xin, code, 0, psum=slabs, psum_b2=b2, psum_alpha=0.5, ln1=lng[1], rows_out=xin)
    assert torch.equal(xin, y)
    # mode 1: rows (reduced, no LN1) written once, LN -> projection with bias, three slices
    wq, bq = torch.randn((3 * D, D), device="cuda") / D ** 0.5, 0.1 * torch.randn((3 * D,), device="cuda")
    rows, qkv = torch.empty_like(x), torch.full((M, 3 * D), float("nan"), device="cuda").to(dt)
    cfm.ffn_split(x, code, 1, psum=slabs, psum_b2=b2, psum_alpha=0.5, rows_out=rows, ln=lng[3], w1=packing.pack_frag_major(wq, dt), b1=bq, n1=3 * D, out16=qkv)
    x1 = x + 0.5 * (slabs.sum(0) + b2)
    assert relerr(rows, x1) < 1e-6
    assert relerr(qkv.float(), r16(ln(x1, lng[3])) @ r16(wq).t() + bq) < (1.2e-2 if wdt == "bf16" else 2e-3)
    # ... and with a K/V ring: the key / value columns land in each stream's ring slots exactly as cfm_kv_ring_write puts them
    if M % 16 == 0:
        Bs, Tq, H, dk, ring_T = M // 16, 16, 4, 64, 40
        offs = torch.randint(0, 1000, (Bs,), dtype=torch.int32, device="cuda")
        kv, kv_ref = torch.zeros((Bs, H, ring_T, 2 * dk), device="cuda"), torch.zeros((Bs, H, ring_T, 2 * dk), device="cuda")
        qkv2 = torch.empty_like(qkv)
        cfm.ffn_split(x, code, 1, psum=slabs, psum_b2=b2, psum_alpha=0.5, ln=lng[3], w1=packing.pack_frag_major(wq, dt), b1=bq, n1=3 * D, out16=qkv2,
                      ring=(kv, offs, Tq))
        q3 = qkv2.view(Bs, Tq, 3 * D)
        kq, vq = q3[:, :, D:2 * D], q3[:, :, 2 * D:]
        cfm.check(cfm.lib().cfm_kv_ring_write(kq.data_ptr(), vq.data_ptr(), code, Tq * 3 * D, 3 * D, Tq * 3 * D, 3 * D, kv_ref.data_ptr(), offs.data_ptr(),
                                              Bs, H, Tq, dk, ring_T, cfm.stream()), "cfm_kv_ring_write")
        assert torch.equal(qkv2, qkv) and torch.equal(kv, kv_ref)
    # reproducible
    again = torch.empty_like(slabs)
    cfm.ffn_split(x, code, 2, ln=lng[0], w1=w1f, b1=b1, n1=FF, act=cfm.ACT_SILU, w2=w2n, psum_out=again)
    assert torch.equal(again, slabs)
