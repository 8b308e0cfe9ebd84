"""GPU: every TRAINING entry point of libconformer_gfx950 (include/cfm.h "Training") against torch autograd on the same inputs.

Checker: plain torch math in f32 / f64 on the CPU or GPU, differentiated by torch.autograd -- an independent computation of the same
derivative (the module-level tests compare against the reference's own gradients; these isolate each kernel).  All calls go through
the ctypes binding -> C ABI -> HIP kernels.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W_DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


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
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------------------ weight gradients
@pytest.mark.parametrize("M,N,K", [(1992, 2048, 256), (1992, 256, 2048), (333, 144, 144), (98, 576, 144), (64, 64, 64), (1, 8, 8), (130, 136, 72),
                                   (2000, 5008, 256), (7968, 768, 256)])
@pytest.mark.parametrize("adt,bdt", [("16", "16"), ("f32", "16"), ("f32", "f32"), ("16", "f32")])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_gemm_tn(cfm, M, N, K, adt, bdt, wdt):
    dt = W_DT[wdt]
    a = rnd((M, N), 1)
    b = rnd((M, K), 2)
    mask = (torch.rand(M, generator=torch.Generator().manual_seed(3)) > 0.2).to("cuda")
    a_in = a if adt == "f32" else a.to(dt)
    b_in = b if bdt == "f32" else b.to(dt)
    ar, br = a.to(dt).double(), b.to(dt).double()         # the kernel rounds f32 operands to the MFMA type while staging
    ref = 0.5 * (ar * mask[:, None]).t() @ br
    ref_cs = 0.5 * (ar * mask[:, None]).sum(0)
    out, cs = cfm.gemm_tn(a_in, b_in, want_colsum=True, row_mask=mask.view(torch.uint8), alpha=0.5, mma_code=cfm.dt_code(dt))
    assert relerr(out, ref) < 1e-4 and relerr(cs, ref_cs) < 1e-4
    # one split per tile: no atomics, bitwise reproducible
    o1, _ = cfm.gemm_tn(a_in, b_in, alpha=0.5, splits=1, mma_code=cfm.dt_code(dt), row_mask=mask.view(torch.uint8))
    o2, _ = cfm.gemm_tn(a_in, b_in, alpha=0.5, splits=1, mma_code=cfm.dt_code(dt), row_mask=mask.view(torch.uint8))
    assert torch.equal(o1, o2) and relerr(o1, ref) < 1e-4
    # accumulate
    o3 = o1.clone()
    cfm.gemm_tn(a_in, b_in, out=o3, alpha=0.5, accumulate=True, mma_code=cfm.dt_code(dt), row_mask=mask.view(torch.uint8))
    assert relerr(o3, 2 * ref) < 1e-4


@pytest.mark.parametrize("M,N,K", [(1992, 256, 256), (300, 144, 576)])
def test_gemm_tn_split_is_f32_accurate(cfm, M, N, K):
    a, b = rnd((M, N), 4), rnd((M, K), 5)
    ref = a.double().t() @ b.double()
    out, cs = cfm.gemm_tn(a, b, want_colsum=True, split=True)
    assert relerr(out, ref) < 3e-5 and relerr(cs, a.double().sum(0)) < 1e-5
    # the single-pass bf16 product of the same operands is ~100x worse: the split is doing its job
    out16, _ = cfm.gemm_tn(a, b)
    assert relerr(out16, ref) > 10 * relerr(out, ref)


@pytest.mark.parametrize("B,T1,F1,C", [(2, 21, 39, 144), (3, 9, 7, 16), (2, 99, 39, 256)])
@pytest.mark.parametrize("split", [False, True])
def test_gemm_tn_conv(cfm, B, T1, F1, C, split):
    """weight gradient of Conv2d(C,C,3,2) over a channels-last image: dW[co, (kt,kf,ci)] = sum_m dY[m,co] * im2col(x)[m,(kt,kf,ci)]."""
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    dt = torch.float32 if split else torch.bfloat16
    img = rnd((B, T1, F1, C), 6).to(dt)
    dy = rnd((B * T2 * F2, C), 7).to(dt)
    x = img.double().permute(0, 3, 1, 2).requires_grad_(False)
    w = torch.zeros((C, C, 3, 3), dtype=torch.float64, device="cuda", requires_grad=True)
    y = torch.nn.functional.conv2d(x, w, stride=2)                       # (B,C,T2,F2)
    y.backward(dy.double().view(B, T2, F2, C).permute(0, 3, 1, 2))
    ref = w.grad.permute(0, 2, 3, 1).reshape(C, 9 * C)                   # [co][kt][kf][ci]
    out, _ = cfm.gemm_tn(dy, img, conv=(C, T1, F1, T2, F2), split=split)
    assert relerr(out, ref) < (3e-5 if split else 1e-4)


@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
@pytest.mark.parametrize("M", [2380, 4130, 257])
def test_gemm_tn_group_equals_single_products(cfm, M, wdt):
    """cfm_gemm_tn_group: the eight weight gradients of a d = 256 / ff = 2048 block in ONE launch, every product against the f64 product of
    the rounded operands and against cfm_gemm_tn on the same operands; the scatter tables (row_off / colsum_off / colsum_off2) as the fused
    q|k|v product uses them; accumulation into a running sum; one split = bitwise reproducible."""
    dt = W_DT[wdt]
    D, FF = 256, 2048
    shapes = [(D, FF), (FF, D), (D, D), (2 * D, D), (D, D), (3 * D, D), (D, FF), (FF, D)]      # (N, K) in a block's backward order
    prods, refs = [], []
    for i, (N, K) in enumerate(shapes):
        a, b = rnd((M, N), 100 + i).to(dt), rnd((M, K), 200 + i).to(dt)
        alpha = 0.5 if i in (0, 6) else 1.0
        prods.append(dict(a=a, b=b, out=torch.zeros((N, K), device="cuda"), colsum=torch.zeros((N,), device="cuda"), alpha=alpha))
        refs.append((alpha * a.double().t() @ b.double(), alpha * a.double().sum(0)))
    cfm.gemm_tn_group(prods, mma_code=cfm.dt_code(dt))
    for pr, (rw, rb) in zip(prods, refs):
        assert relerr(pr["out"], rw) < 1e-4 and relerr(pr["colsum"], rb) < 1e-4
        single, cs = cfm.gemm_tn(pr["a"], pr["b"], want_colsum=True, alpha=pr["alpha"], mma_code=cfm.dt_code(dt))
        assert relerr(pr["out"], single) < 2e-6 and relerr(pr["colsum"], cs) < 2e-6
    # a second call ACCUMULATES
    cfm.gemm_tn_group(prods, mma_code=cfm.dt_code(dt))
    for pr, (rw, rb) in zip(prods, refs):
        assert relerr(pr["out"], 2 * rw) < 1e-4 and relerr(pr["colsum"], 2 * rb) < 1e-4
    # splits = 1: no atomics on the products, two runs agree bit for bit
    runs = []
    for _ in range(2):
        ps = [dict(pr, out=torch.zeros_like(pr["out"]), colsum=None) for pr in prods]
        cfm.gemm_tn_group(ps, mma_code=cfm.dt_code(dt), splits=1)
        runs.append(ps)
    assert all(torch.equal(x["out"], y["out"]) for x, y in zip(*runs))
    # scatter tables: rows of a fused 3D x D product land where three separate parameters live in a flat slab, the column sums too, and the
    # first D column sums a second time (pos_bias_u = linear_q.bias' gradient)
    a, b = prods[5]["a"], prods[5]["b"]
    slab = torch.zeros((3 * D * D + 3 * D + D + 64,), device="cuda")
    ar = torch.arange(D, device="cuda")
    w_off, b_off, u_off = (2 * D * D + 16, 16, D * D + 16), (3 * D * D + 32 + D, 3 * D * D + 16, 3 * D * D + 48 + 2 * D), 3 * D * D + 48 + 3 * D
    row_off = torch.cat([w_off[j] + ar * D for j in range(3)])
    cs_off = torch.cat([b_off[j] + ar for j in range(3)])
    cs_off2 = torch.cat([u_off + ar, torch.full((2 * D,), -1, device="cuda", dtype=torch.int64)])
    extra = dict(a=prods[2]["a"], b=prods[2]["b"], out=torch.zeros((D, D), device="cuda"), colsum=torch.zeros((D,), device="cuda"))
    cfm.gemm_tn_group([dict(a=a, b=b, out=slab, colsum=slab, row_off=row_off, colsum_off=cs_off, colsum_off2=cs_off2), extra], mma_code=cfm.dt_code(dt))
    rw, rb = refs[5]
    for j in range(3):
        assert relerr(slab[w_off[j]:w_off[j] + D * D].view(D, D), rw[j * D:(j + 1) * D]) < 1e-4
        assert relerr(slab[b_off[j]:b_off[j] + D], rb[j * D:(j + 1) * D]) < 1e-4
    assert relerr(slab[u_off:u_off + D], rb[:D]) < 1e-4
    assert relerr(extra["out"], refs[2][0]) < 1e-4


def test_gemm_tn_group_falls_back_for_f32_operands(cfm):
    """operands the grouped kernel does not take (f32 rows: the f32-accurate mode) run as single launches behind the same entry point."""
    M = 700
    prods, refs = [], []
    for i, (N, K) in enumerate([(144, 576), (576, 144)]):
        a, b = rnd((M, N), 300 + i), rnd((M, K), 310 + i)
        prods.append(dict(a=a, b=b, out=torch.zeros((N, K), device="cuda"), colsum=torch.zeros((N,), device="cuda")))
        refs.append(a.bfloat16().double().t() @ b.bfloat16().double())
    cfm.gemm_tn_group(prods)
    for pr, r in zip(prods, refs):
        assert relerr(pr["out"], r) < 1e-4


# ------------------------------------------------------------------------------------------------------------ GEMM training epilogues
@pytest.mark.parametrize("M,N,K", [(500, 576, 144), (1992, 2048, 256)])
@pytest.mark.parametrize("wdt", ["bf16", "fp16"])
def test_gemm_preact_and_dsilu(cfm, M, N, K, wdt):
    dt = W_DT[wdt]
    a, w, bias = rnd((M, K), 8).to(dt), rnd((N, K), 9, K ** -0.5).to(dt), rnd((N,), 10, 0.1)
    z_ref = a.float() @ w.float().t() + bias
    pre = torch.empty((M, N), dtype=dt, device="cuda")
    h = cfm.gemm(a, w, bias=bias, act=cfm.ACT_SILU, out_dtype=dt, pre_out=pre)
    assert relerr(pre.float(), z_ref.to(dt).float()) < 1e-2 and relerr(h.float(), torch.nn.functional.silu(z_ref)) < 2e-2
    # backward epilogue: dz = 0.5 * (dy . W2) * silu'(z)
    dy, w2 = rnd((M, K), 11).to(dt), rnd((N, K), 12, K ** -0.5).to(dt)        # dy [M,K] . w2^T [K,N]  (w2 here is already the [N,K] pack)
    zz = pre.float().requires_grad_(True)
    torch.nn.functional.silu(zz).backward(0.5 * (dy.float() @ w2.float().t()))
    dz = cfm.gemm(dy, w2, act=cfm.ACT_DSILU, aux=pre, alpha=0.5, out_dtype=torch.float32)
    assert relerr(dz, zz.grad) < 1e-4
    dr = cfm.gemm(dy, w2, act=cfm.ACT_DRELU, aux=pre, alpha=1.0, out_dtype=dt)
    assert relerr(dr.float(), ((dy.float() @ w2.float().t()) * (pre.float() > 0)).to(dt).float()) < 1e-2


def test_gemm_glu_preact(cfm):
    M, D, K = 300, 144, 144
    a, w, bias = rnd((M, K), 13).bfloat16(), rnd((2 * D, K), 14, K ** -0.5).bfloat16(), rnd((2 * D,), 15, 0.1)
    pre = torch.empty((M, 2 * D), dtype=torch.float32, device="cuda")
    g = cfm.gemm(a, w, bias=bias, act=cfm.ACT_GLU, out_dtype=torch.float32, pre_out=pre)
    z = a.float() @ w.float().t() + bias
    assert relerr(pre, z) < 1e-5
    zz = z.view(M, D // 16, 2, 16)
    assert relerr(g, (zz[:, :, 0] * torch.sigmoid(zz[:, :, 1])).reshape(M, D)) < 1e-5
    # GLU backward on that layout
    dg = rnd((M, D), 16)
    zr = z.clone().requires_grad_(True)
    zv = zr.view(M, D // 16, 2, 16)
    (zv[:, :, 0] * torch.sigmoid(zv[:, :, 1])).reshape(M, D).backward(dg)
    du = cfm.glu_bwd(pre, dg, torch.float32)
    assert relerr(du, zr.grad) < 1e-5
    du16 = cfm.glu_bwd(pre.bfloat16(), dg.bfloat16(), torch.bfloat16)
    assert relerr(du16.float(), zr.grad) < 3e-2


# ------------------------------------------------------------------------------------------------------------ LayerNorm backward
@pytest.mark.parametrize("M,D", [(1992, 256), (111, 144), (37, 512), (5, 1024), (7968, 256)])
@pytest.mark.parametrize("dyt", [torch.float32, torch.bfloat16])
def test_layernorm_bwd(cfm, M, D, dyt):
    x, dy = rnd((M, D), 17), rnd((M, D), 18).to(dyt)
    g, b = 1 + rnd((D,), 19, 0.1), rnd((D,), 20, 0.1)
    dres = rnd((M, D), 21)
    mask = (torch.rand(M, generator=torch.Generator().manual_seed(22)) > 0.3).to("cuda")
    xr, gr, br = x.double().requires_grad_(True), g.double().requires_grad_(True), b.double().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5) * mask[:, None]
    y.backward(dy.double())
    dx, dg, db = cfm.layernorm_bwd(x, dy, g, row_mask=mask.view(torch.uint8), dres=dres)
    assert relerr(dx, xr.grad + dres.double()) < 2e-5 and relerr(dg, gr.grad) < 2e-5 and relerr(db, br.grad) < 2e-5
    # in place over the residual gradient, no mask
    xr.grad = gr.grad = br.grad = None
    torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-5).backward(dy.double())
    buf = dres.clone()
    dx, dg, db = cfm.layernorm_bwd(x, dy, g, dres=buf, dx=buf)
    assert dx.data_ptr() == buf.data_ptr() and relerr(dx, xr.grad + dres.double()) < 2e-5 and relerr(dg, gr.grad) < 2e-5


# ------------------------------------------------------------------------------------------------------------ depthwise + BatchNorm(train) + SiLU
@pytest.mark.parametrize("B,T,D", [(3, 37, 144), (2, 249, 256), (5, 16, 512), (1, 3, 16), (4, 411, 256)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_dwconv_bn_train_fwd_bwd(cfm, B, T, D, dt):
    K = 15
    g = rnd((B, T, D), 23).to(dt)
    w, bias = rnd((D, K), 24, K ** -0.5), rnd((D,), 25, 0.1)
    gamma, beta = 1 + rnd((D,), 26, 0.1), rnd((D,), 27, 0.1)
    rm, rv = rnd((D,), 28, 0.1), 1 + rnd((D,), 29, 0.1).abs()
    rm0, rv0 = rm.clone(), rv.clone()
    c, stats, s = cfm.dwconv_bn_train(g, w, bias, gamma, beta, rm, rv, 0.1, 1e-5, dt)

    leaves = [t.double().requires_grad_(True) for t in (g.float(), w, bias, gamma, beta)]
    gr, wr, br, gar, ber = leaves
    bn = torch.nn.BatchNorm1d(D, eps=1e-5, momentum=0.1).double().cuda().train()
    bn.running_mean.copy_(rm0.double())
    bn.running_var.copy_(rv0.double())
    cr = torch.nn.functional.conv1d(gr.transpose(1, 2), wr.unsqueeze(1), br, padding=7, groups=D)           # (B,D,T)
    yr = torch.nn.functional.batch_norm(cr, bn.running_mean, bn.running_var, gar, ber, True, 0.1, 1e-5)
    sr = torch.nn.functional.silu(yr).transpose(1, 2)
    assert relerr(c, cr.transpose(1, 2)) < 1e-5
    assert relerr(s.float(), sr) < (1e-5 if dt == torch.float32 else 1e-2)
    if B * T > 1:
        assert relerr(rm, bn.running_mean) < 1e-5 and relerr(rv, bn.running_var) < 1e-5
    ds = rnd((B, T, D), 30).to(dt)
    sr.backward(ds.double())
    dg, dw_w, dw_b, dgamma, dbeta = cfm.dwconv_bn_train_bwd(ds, c, stats, g, w, dt)
    tol = 1e-4 if dt == torch.float32 else 1e-2
    if B * T > 1:
        assert relerr(dg.float(), gr.grad) < tol
        assert relerr(dw_w, wr.grad) < 1e-4 and relerr(dgamma, gar.grad) < 1e-4 and relerr(dbeta, ber.grad) < 1e-4
        # the conv bias gradient is zero under batch statistics (torch: rounding noise); ours must be tiny relative to the others
        assert float(dw_b.abs().max()) < 1e-3 * float(dbeta.abs().max() + 1e-6)


# ------------------------------------------------------------------------------------------------------------ front-end backward
@pytest.mark.parametrize("B,T,C", [(2, 83, 144), (3, 200, 256), (1, 7, 16)])
def test_frontend_backward_pieces(cfm, B, T, C):
    F = 80
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    x = rnd((B, T, F), 31)
    h1 = torch.relu(rnd((B, T1, F1, C), 32))
    dcol = rnd((B * T2 * F2, 9 * C), 33)
    # reference col2im through conv_transpose: dcol rows are the im2col rows of a 3x3 stride-2 conv over [B,C,T1,F1], K order (kt,kf,c)
    cols = dcol.double().view(B, T2 * F2, 3, 3, C).permute(0, 4, 2, 3, 1).reshape(B, C * 9, T2 * F2)
    ref = torch.nn.functional.fold(cols, (T1, F1), kernel_size=3, stride=2).permute(0, 2, 3, 1) * (h1 > 0)
    dh1 = cfm.col2im_relu_bwd(dcol, h1, torch.float32)
    assert relerr(dh1, ref) < 1e-5
    dh16 = cfm.col2im_relu_bwd(dcol.bfloat16(), h1.bfloat16(), torch.bfloat16)
    assert relerr(dh16.float(), ref) < 2e-2
    # conv1 weight gradient
    w = torch.zeros((C, 1, 3, 3), dtype=torch.float64, device="cuda", requires_grad=True)
    b = torch.zeros((C,), dtype=torch.float64, device="cuda", requires_grad=True)
    mean, istd = rnd((F,), 34, 0.1), 1 + rnd((F,), 35, 0.1).abs()
    y = torch.nn.functional.conv2d(((x - mean) * istd).double().unsqueeze(1), w, b, stride=2)       # (B,C,T1,F1)
    y.backward(dh1.double().permute(0, 3, 1, 2))
    dw, db = cfm.conv1_wgrad(dh1, x, cmvn=(mean, istd))
    assert relerr(dw, w.grad.reshape(C, 9).t()) < 2e-5 and relerr(db, b.grad) < 2e-5


# ------------------------------------------------------------------------------------------------------------ attention backward
def _attn_ref(q, k, v, mask, scale):
    """q,k,v (B,T,H,dk) double, mask bool (B,1|Tq,Tk) or None -> (B,Tq,H*dk); masked softmax with fully masked rows giving zero."""
    s = torch.einsum("bihd,bjhd->bhij", q, k) * scale
    if mask is not None:
        dead = ~mask.unsqueeze(1)
        s = s.masked_fill(dead, float("-inf"))
        p = torch.softmax(s, -1)
        p = torch.where(dead.expand_as(p), torch.zeros((), dtype=p.dtype, device=p.device), p)
    else:
        p = torch.softmax(s, -1)
    o = torch.einsum("bhij,bjhd->bihd", p, v)
    return o.reshape(o.shape[0], o.shape[1], -1)


@pytest.mark.parametrize("B,T,H,dk", [(3, 37, 4, 36), (2, 249, 4, 64), (2, 100, 8, 64), (1, 411, 4, 64), (2, 65, 2, 16)])
@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("mkind", ["none", "pad", "chunk"])
def test_attention_backward(cfm, B, T, H, dk, mode, mkind):
    D = H * dk
    split = mode == "fp32"
    dt = torch.float32 if split else W_DT[mode]
    mma = cfm.BF16 if mode != "fp16" else cfm.F16
    qkv = rnd((B * T, 3 * D), 36, 0.7).to(dt)
    dout = rnd((B * T, D), 37).to(dt)
    lens = [T, max(1, T - T // 4), max(1, T // 2)][:B] + [T] * max(0, B - 3)
    valid = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).cuda()
    mask = None
    if mkind == "pad":
        mask = valid[:, None, :]
    elif mkind == "chunk":                                                        # has fully masked rows (padded queries of a chunk with no valid key)
        blk = torch.arange(T) // 8
        chunk = ((blk[None, :] <= blk[:, None]) & (blk[None, :] >= blk[:, None] - 1)).cuda()
        mask = chunk[None] & valid[:, None, :]
    scale = dk ** -0.5
    x = qkv.double().view(B, T, 3, H, dk).requires_grad_(True)
    o_ref = _attn_ref(x[:, :, 0], x[:, :, 1], x[:, :, 2], mask, scale)
    o_ref.backward(dout.double().view(B, T, D))
    g_ref = x.grad.reshape(B * T, 3 * D)

    ctx = torch.empty((B * T, D), dtype=dt, device="cuda")
    lse = torch.empty((B, H, T), dtype=torch.float32, device="cuda")
    m8 = None if mask is None else mask.contiguous().view(torch.uint8)
    mstr = (0, 0) if mask is None else (mask.shape[1] * T, T if mask.shape[1] > 1 else 0)
    st = (T * 3 * D, 3 * D)
    cfm.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], B, H, T, T, dk, st, st + (dk,), st + (dk,), ctx, mask=m8, mask_str=mstr, mma_code=mma, split=split, lse=lse)
    ftol = dict(bf16=2e-2, fp16=3e-3, fp32=3e-5)[mode]
    assert relerr(ctx.float(), o_ref.reshape(B * T, D)) < ftol
    dqkv = torch.zeros_like(qkv)
    cfm.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], ctx, dout, lse, B, H, T, T, dk, st, st, st, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], mask=m8, mask_str=mstr,
                      mma_code=mma, split=split)
    gtol = dict(bf16=3e-2, fp16=5e-3, fp32=1e-4)[mode]
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert relerr(dqkv[:, sl].float(), g_ref[:, sl]) < gtol, (name, relerr(dqkv[:, sl].float(), g_ref[:, sl]))


# ------------------------------------------------------------------------------------------------------------ CTC backward
@pytest.mark.parametrize("B,T,V,Umax", [(3, 49, 73, 9), (2, 60, 5002, 12), (4, 30, 11, 5), (2, 249, 5002, 40)])
def test_ctc_gradient(cfm, B, T, V, Umax):
    rs = np.random.RandomState(B * 100 + T)
    Vp = (V + 7) // 8 * 8
    logits = torch.zeros((B, T, Vp), device="cuda")
    logits[:, :, :V] = rnd((B, T, V), 38, 2.0)
    enc_lens = np.sort(rs.randint(max(2 * Umax + 1, T // 2), T + 1, size=B))[::-1].copy()
    enc_lens[0] = T
    label_lens = rs.randint(1, Umax + 1, size=B)
    label_lens[0] = Umax
    labels = rs.randint(1, V, size=(B, Umax))
    labels[1, 1:3] = labels[1, 0]
    for b in range(B):
        labels[b, label_lens[b]:] = 0
    i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device="cuda", dtype=torch.int32)
    el, lb, ll = i32(enc_lens), i32(labels), i32(label_lens)
    nll, state = cfm.ctc_nll_train(logits, V, el, lb, ll)
    lr = logits[:, :, :V].detach().cpu().double().requires_grad_(True)
    per = torch.nn.functional.ctc_loss(lr.transpose(0, 1).log_softmax(2), torch.from_numpy(labels), torch.from_numpy(enc_lens), torch.from_numpy(label_lens), reduction="none")
    (per.sum() / Umax).backward()
    assert relerr(nll.cpu(), per.detach()) < 2e-5
    gdev = torch.full((1,), 2.0, device="cuda")
    grad = cfm.ctc_grad(logits, V, el, lb, ll, state, gscale=0.5 / Umax, gscale_dev=gdev)
    assert float(grad[:, :, V:].abs().max()) == 0.0 if Vp > V else True
    assert relerr(grad[:, :, :V].cpu(), lr.grad) < 5e-4          # fp32 recursions vs torch in fp64: grows ~linearly with T (2.4e-4 at T = 249)
    for b in range(B):
        assert float(grad[b, int(enc_lens[b]):].abs().max()) == 0.0 if enc_lens[b] < T else True
    # the backward recursion inside ctc_grad (in place over alpha) instead of beside the forward one: the same bits
    nll2, state2 = cfm.ctc_nll_train(logits, V, el, lb, ll, beta_now=False)
    assert state[4] is not None and state2[4] is None and torch.equal(nll, nll2)
    assert torch.equal(cfm.ctc_grad(logits, V, el, lb, ll, state2, gscale=0.5 / Umax, gscale_dev=gdev), grad)


def test_ctc_gradient_infeasible_is_zero(cfm):
    logits = rnd((2, 6, 16), 39)
    i32 = lambda a: torch.tensor(a, device="cuda", dtype=torch.int32)
    el, lb, ll = i32([6, 3]), i32([[3, 3, 5, 0], [1, 2, 3, 4]]), i32([3, 4])            # utterance 1: 4 labels in 3 frames
    nll, state = cfm.ctc_nll_train(logits, 16, el, lb, ll)
    assert math.isinf(float(nll[1])) and math.isfinite(float(nll[0]))
    grad = cfm.ctc_grad(logits, 16, el, lb, ll, state)
    assert bool(torch.isfinite(grad).all()) and float(grad[1].abs().max()) == 0.0 and float(grad[0].abs().max()) > 0


# ------------------------------------------------------------------------------------------------------------ optimizer
@pytest.mark.parametrize("n", [1, 1023, 4096 * 3 + 1, 1 << 20])
def test_adam_and_sumsq(cfm, n):
    p, g = rnd((n,), 40), rnd((n,), 41)
    ref_p = torch.nn.Parameter(p.clone())
    opt = torch.optim.Adam([ref_p], lr=1e-3, betas=(0.9, 0.98), eps=1e-9, weight_decay=0.0)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    scale = torch.full((1,), 0.37, device="cuda")
    for step in range(1, 4):
        ref_p.grad = g * 0.37
        opt.step()
        cfm.adam_step(p, g, m, v, 1e-3, (0.9, 0.98), 1e-9, 0.0, step, grad_scale=scale)
    assert float((p - ref_p.detach()).abs().max()) < 1e-6
    assert relerr(cfm.sumsq(g), (g.double() ** 2).sum().view(1)) < 1e-5


# ------------------------------------------------------------------------------------------------------------ dropout
def test_dropout_generator_statistics_and_rows(cfm):
    n = 1 << 20
    for p in (0.1, 0.5):
        m = cfm.dropout_mask(n, p, 12345, "cuda")
        rate = float(m.float().mean())
        assert abs(rate - (1 - p)) < 4 * math.sqrt(p * (1 - p) / n) + 1e-4, (p, rate)
        m2 = cfm.dropout_mask(n, p, 12346, "cuda")
        assert abs(float((m & m2).float().mean()) - (1 - p) ** 2) < 3e-3          # different seeds: independent masks
        assert torch.equal(m, cfm.dropout_mask(n, p, 12345, "cuda"))             # a pure function of (seed, index)
    M, N = 777, 256
    x = rnd((M, N), 50)
    rm = (torch.rand(M, generator=torch.Generator().manual_seed(51)) > 0.3).to("cuda")
    y = cfm.dropout_rows(x, torch.float32, alpha=0.5, drop=(0.1, 99), drop2=(0.2, 100), row_mask=rm.view(torch.uint8))
    k = cfm.dropout_mask(M * N, 0.1, 99, "cuda").view(M, N) & cfm.dropout_mask(M * N, 0.2, 100, "cuda").view(M, N)
    ref = 0.5 * x * k / (0.9 * 0.8) * rm[:, None]
    assert relerr(y, ref) < 1e-6


@pytest.mark.parametrize("M,N,K", [(500, 576, 144), (1992, 256, 2048)])
def test_gemm_epilogue_dropout(cfm, M, N, K):
    dt = torch.bfloat16
    a, w, bias, res = rnd((M, K), 52).to(dt), rnd((N, K), 53, K ** -0.5).to(dt), rnd((N,), 54, 0.1), rnd((M, N), 55)
    keep = cfm.dropout_mask(M * N, 0.1, 777, "cuda").view(M, N)
    z = a.float() @ w.float().t() + bias
    # SiLU + dropout (the FFN's hidden activation), pre-activation untouched
    pre = torch.empty((M, N), dtype=torch.float32, device="cuda")
    h = cfm.gemm(a, w, bias=bias, act=cfm.ACT_SILU, out_dtype=torch.float32, pre_out=pre, drop=(0.1, 777))
    assert relerr(pre, z) < 1e-5 and relerr(h, torch.nn.functional.silu(z) * keep / 0.9) < 1e-5
    # residual + alpha * dropout(v)
    y = cfm.gemm(a, w, bias=bias, residual=res, alpha=0.5, drop=(0.1, 777))
    assert relerr(y, res + 0.5 * z * keep / 0.9) < 1e-5
    # backward epilogue: dz = (dy . W2) * mask/(1-p) * silu'(z)
    dy, w2 = rnd((M, K), 56).to(dt), rnd((N, K), 57, K ** -0.5).to(dt)
    zz = pre.clone().requires_grad_(True)
    (torch.nn.functional.silu(zz) * keep / 0.9).backward(dy.float() @ w2.float().t())
    dz = cfm.gemm(dy, w2, act=cfm.ACT_DSILU, aux=pre, alpha=1.0, out_dtype=torch.float32, drop=(0.1, 777))
    assert relerr(dz, zz.grad) < 1e-4


@pytest.mark.parametrize("B,T,H,dk", [(2, 100, 4, 64), (3, 37, 4, 36)])
@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_attention_dropout_forward_backward(cfm, B, T, H, dk, mode):
    D = H * dk
    split = mode == "fp32"
    dt = torch.float32 if split else torch.bfloat16
    p, seed = 0.1, 4242
    qkv = rnd((B * T, 3 * D), 58, 0.7).to(dt)
    dout = rnd((B * T, D), 59).to(dt)
    lens = [T, max(1, T - T // 4), max(1, T // 2)][:B]
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).cuda()[:, None, :]
    keep = cfm.dropout_mask(B * H * T * T, p, seed, "cuda").view(B, H, T, T)
    x = qkv.double().view(B, T, 3, H, dk).requires_grad_(True)
    s = torch.einsum("bihd,bjhd->bhij", x[:, :, 0], x[:, :, 1]) * dk ** -0.5
    s = s.masked_fill(~mask.unsqueeze(1), float("-inf"))
    pr = torch.softmax(s, -1) * keep / (1 - p)
    o_ref = torch.einsum("bhij,bjhd->bihd", pr, x[:, :, 2]).reshape(B, T, D)
    o_ref.backward(dout.double().view(B, T, D))
    g_ref = x.grad.reshape(B * T, 3 * D)
    ctx = torch.empty((B * T, D), dtype=dt, device="cuda")
    lse = torch.empty((B, H, T), dtype=torch.float32, device="cuda")
    m8, mstr, st = mask.contiguous().view(torch.uint8), (T, 0), (T * 3 * D, 3 * D)
    cfm.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], B, H, T, T, dk, st, st + (dk,), st + (dk,), ctx, mask=m8, mask_str=mstr, split=split, lse=lse, drop=(p, seed))
    assert relerr(ctx.float(), o_ref.reshape(B * T, D)) < (3e-5 if split else 2e-2)
    dqkv = torch.zeros_like(qkv)
    cfm.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], ctx, dout, lse, B, H, T, T, dk, st, st, st, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], mask=m8, mask_str=mstr,
                      split=split, drop=(p, seed))
    for sl in (slice(0, D), slice(D, 2 * D), slice(2 * D, 3 * D)):
        assert relerr(dqkv[:, sl].float(), g_ref[:, sl]) < (1e-4 if split else 3e-2)


@pytest.mark.parametrize("mode", ["bf16", "fp16", "fp32"])
def test_pack_kernel_equals_torch_packs(mode):
    """cfm_pack_matrices (one launch for the 8 matrices of a block: casts, transposes, the fused q|k|v concatenation, the GLU row
    interleave, hi/lo planes in the f32-accurate mode) against the torch-op packs it replaces: bit-identical, also after an in-place
    weight update through a raw pointer + packing.bump_epoch() (what the trainer's Adam kernel does)."""
    import cfm
    import encoder_layer
    from cfm import packing
    torch.manual_seed(3)
    prec = cfm.Precision(mode)
    layer = encoder_layer.ConformerEncoderLayer(256, 15, 0.1, 0.1, 2048, 4, True).to("cuda")
    for rep in range(2):
        got = packing.pack_layer_train(layer, prec, True)
        ref = (packing.pack_ffn_train(layer.feed_forward_macaron, prec), packing.pack_mhsa_train(layer.self_attn, prec, True),
               packing.pack_conv_module_train(layer.conv_module, prec), packing.pack_ffn_train(layer.feed_forward, prec))
        n = 0
        for g, r in zip(got, ref):
            for k, v in r.__dict__.items():
                w = getattr(g, k)
                if v is None:
                    assert w is None, k
                    continue
                assert isinstance(w, torch.Tensor) and w.shape == v.shape and w.dtype == v.dtype, k
                assert torch.equal(w, v), (mode, rep, k)
                n += 1
        assert n >= 29
        with torch.no_grad():
            for p in layer.parameters():
                p.data.add_(0.01 * torch.randn_like(p))            # (bumps versions too; the epoch covers raw-pointer writers)
        packing.bump_epoch()


@pytest.mark.parametrize("accumulate", [0, 1])
def test_layernorm_bwd_fused_equals_separate_launches(accumulate):
    """cfm_layernorm_bwd_fused: dx identical to cfm_layernorm_bwd; the second output identical to cfm_dropout_rows on dx (two dropouts, bf16);
    parameter gradients identical through the workspace (accumulate = 0) and added on top of what is there by atomics (accumulate = 1: order of
    the f32 sums differs, 1e-5)."""
    import ctypes
    import cfm
    torch.manual_seed(11)
    M, D = 1237, 256
    x = torch.randn((M, D), device="cuda")
    dy = torch.randn((M, D), device="cuda")
    dres = torch.randn((M, D), device="cuda")
    gamma = torch.randn((D,), device="cuda")
    mask = (torch.rand((M,), device="cuda") > 0.2).to(torch.uint8)
    dx_ref, dg_ref, db_ref = cfm.layernorm_bwd(x, dy, gamma, row_mask=mask, dres=dres)
    y_ref = cfm.dropout_rows(dx_ref, torch.bfloat16, alpha=0.5, drop=(0.1, 77), drop2=(0.2, 78))
    dx = torch.empty_like(x)
    dx2 = torch.empty((M, D), dtype=torch.bfloat16, device="cuda")
    base_g, base_b = torch.randn((D,), device="cuda"), torch.randn((D,), device="cuda")
    dg, db = base_g.clone(), base_b.clone()
    ws = torch.empty((int(cfm.lib().cfm_layernorm_bwd_ws(M, D)),), device="cuda")
    d = cfm.LnBwdDesc()
    d.x, d.dy, d.gamma, d.row_mask, d.dres, d.dx, d.dgamma, d.dbeta, d.ws, d.dx2 = (t.data_ptr() for t in (x, dy, gamma, mask, dres, dx, dg, db, ws, dx2))
    d.M, d.D, d.dy_dtype, d.dx2_dtype, d.accumulate = M, D, cfm.F32, cfm.BF16, accumulate
    d.eps, d.alpha2, d.p1, d.p2, d.seed1, d.seed2 = 1e-5, 0.5, 0.1, 0.2, 77, 78
    cfm.check(cfm.lib().cfm_layernorm_bwd_fused(ctypes.byref(d), cfm.stream()), "cfm_layernorm_bwd_fused")
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    assert torch.equal(dx2.view(torch.int16), y_ref.view(torch.int16))
    if accumulate:
        assert float((dg - base_g - dg_ref).abs().max()) < 1e-5 * float(dg_ref.abs().max()) + 1e-5
        assert float((db - base_b - db_ref).abs().max()) < 1e-5 * float(db_ref.abs().max()) + 1e-5
    else:
        assert torch.equal(dg, dg_ref) and torch.equal(db, db_ref)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("mkind,drop", [("none", None), ("pad", None), ("pad", (0.1, 991)), ("chunk", None), ("chunk", (0.1, 991))])
def test_attention_backward_fast_kernels_are_bit_identical_to_general(cfm, mode, mkind, drop):
    """d_k = 64 with 16-bit rows takes the fast backward kernels (tiles staged as they lie in memory, transposed operands by
    ds_read_b64_tr_b16, prefetched tiles); cfm_attention_bwd_force_general switches them off: same products in the same order -- identical
    bits, on a ragged batch whose length is not a multiple of the 64-row tile."""
    B, T, H, dk = 3, 217, 4, 64
    D = H * dk
    dt = W_DT[mode]
    mma = cfm.BF16 if mode == "bf16" else cfm.F16
    qkv = rnd((B * T, 3 * D), 61, 0.7).to(dt)
    dout = rnd((B * T, D), 62).to(dt)
    mask = None
    valid = (torch.arange(T)[None, :] < torch.tensor([T, 150, 64])[:, None]).cuda()
    if mkind == "pad":
        mask = valid[:, None, :]
    elif mkind == "chunk":                              # (B, T, T): chunk window & padding -- the dynamic-chunk training masks (round 3: fast kernels too)
        blk = torch.arange(T) // 16
        mask = ((blk[None, :] <= blk[:, None]) & (blk[None, :] >= blk[:, None] - 2)).cuda()[None] & valid[:, None, :]
    m8 = None if mask is None else mask.contiguous().view(torch.uint8)
    mstr = (0, 0) if mask is None else (mask.shape[1] * T, T if mask.shape[1] > 1 else 0)
    st = (T * 3 * D, 3 * D)
    ctx = torch.empty((B * T, D), dtype=dt, device="cuda")
    lse = torch.empty((B, H, T), dtype=torch.float32, device="cuda")
    cfm.attention(qkv, qkv[:, D:], qkv[:, 2 * D:], B, H, T, T, dk, st, st + (dk,), st + (dk,), ctx, mask=m8, mask_str=mstr, mma_code=mma, lse=lse, drop=drop)
    res = []
    try:
        for force in (0, 1):
            cfm.lib().cfm_attention_bwd_force_general(force)
            dqkv = torch.zeros_like(qkv)
            cfm.attention_bwd(qkv, qkv[:, D:], qkv[:, 2 * D:], ctx, dout, lse, B, H, T, T, dk, st, st, st, dqkv, dqkv[:, D:], dqkv[:, 2 * D:], mask=m8, mask_str=mstr,
                              mma_code=mma, drop=drop)
            torch.cuda.synchronize()
            res.append(dqkv)
    finally:
        cfm.lib().cfm_attention_bwd_force_general(0)
    assert float(res[0].float().abs().max()) > 1e-3
    assert torch.equal(res[0].view(torch.int16), res[1].view(torch.int16)), float((res[0].float() - res[1].float()).abs().max())
