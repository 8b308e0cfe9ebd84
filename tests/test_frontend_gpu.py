"""The fused front-end (csrc/frontend.hip: conv1 recomputed inside conv2's operand producer) against the two-kernel path
(cfm_conv1_relu_mma + cfm_gemm with the implicit-im2col A operand) on the same packed weights: BIT-IDENTICAL by construction --
same MFMA operands for every conv1 value, same K order in the main loop -- so the parity of the two-kernel path against the
oracle / the reference goldens (test_modules_gpu.py) carries over unchanged.  Shapes cover every tile variant (256-row tiles,
the 64 / 96 / 128-row tail tiles, ragged last tiles), every supported channel count, CMVN and both 16-bit types."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _two_kernel(cfm, x, w1, b1, w2, b2, cmvn):
    B, T, F = x.shape
    C = w1.shape[1]
    T1, F1 = (T - 3) // 2 + 1, (F - 3) // 2 + 1
    T2, F2 = (T1 - 3) // 2 + 1, (F1 - 3) // 2 + 1
    if C <= 256:
        h1 = cfm.conv1_relu(x, w1, b1, w2.dtype, cmvn=cmvn, mma=True)
    else:       # the stand-alone MFMA conv1 stops at 256 channels: two channel halves side by side are the same values
        h1 = torch.cat([cfm.conv1_relu(x, w1[:, c0:c0 + 256].contiguous(), b1[c0:c0 + 256].contiguous(), w2.dtype, cmvn=cmvn, mma=True).view(B, T1, F1, 256)
                        for c0 in range(0, C, 256)], dim=-1).contiguous()
    return cfm.gemm(h1, w2, bias=b2, act=cfm.ACT_RELU, conv=(C, T1, F1, T2, F2, B * T2 * F2), out_dtype=w2.dtype)


# (B, T, F, C): M = B*T2*F2 rows
SHAPES = [
    (1, 7, 7, 64),          # one output position
    (2, 67, 80, 256),       # 570 rows: 64-row tail tiles only
    (3, 131, 80, 128),      # 1 824 rows
    (2, 200, 83, 192),      # odd F, three channel slabs (odd K-tile parity per tap)
    (24, 1000, 80, 256),    # 113 544 rows: one whole round of 256-row tiles + a 128-row tail... (depends on the CU count)
    (32, 1000, 80, 256),    # BASELINE config 2: two whole rounds + 20 320 rows on 96-row tiles
    (40, 403, 80, 256),     # 75 240 rows: more than half a round left after one whole round
    (2, 131, 80, 512),      # C = 512: two workgroups (halves of N) per row tile, eight channel slabs; tail tiles only
    (16, 1000, 80, 512),    # BASELINE config 4: 75 696 rows = two whole rounds of (128 tiles x 2 halves) + a 96-row-tile round
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("use_cmvn", [False, True])
def test_fused_equals_two_kernels(shape, dtype, use_cmvn):
    import cfm
    B, T, F, C = shape
    if use_cmvn and B > 3:
        pytest.skip("CMVN covered on the small shapes")
    g = torch.Generator().manual_seed(B * 1000 + T + C)
    x = (torch.randn((B, T, F), generator=g) * 3.0 + 1.0).to(DEV)
    w1 = (torch.randn((9, C), generator=g) * 0.3).to(DEV)
    b1 = (torch.randn((C,), generator=g) * 0.2).to(DEV)
    w2 = (torch.randn((C, 9 * C), generator=g) * (1.0 / (9 * C) ** 0.5)).to(DEV).to(dtype)
    b2 = (torch.randn((C,), generator=g) * 0.1).to(DEV)
    cmvn = None
    if use_cmvn:
        cmvn = ((torch.randn((F,), generator=g)).to(DEV), (torch.rand((F,), generator=g) + 0.5).to(DEV))
    assert cfm.conv12_supported(C, dtype)
    ref = _two_kernel(cfm, x, w1, b1, w2, b2, cmvn)
    got = cfm.conv12_relu(x, w1, b1, w2, b2, cmvn=cmvn)
    torch.cuda.synchronize()
    assert got.shape == ref.shape and got.dtype == ref.dtype
    assert float(ref.float().abs().max()) > 0.1 and float((ref != 0).float().mean()) > 0.2     # a live comparison, not zeros against zeros
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), (shape, dtype, float((got.float() - ref.float()).abs().max()))


def test_module_uses_fused_path_and_matches():
    """ConvolutionSubSampling.embed_frames with and without the fused kernel (bit-identical), mean-only CMVN."""
    import cfm
    import attention
    import convolution
    torch.manual_seed(5)
    pos = attention.RelativePositionalEncoding(256, 0.0)
    mod = convolution.ConvolutionSubSampling(80, 256, pos).to(DEV).eval()
    x = torch.randn((4, 333, 80), device=DEV)
    mean = torch.randn((80,), device=DEV)
    for mode in ("bf16", "fp16"):
        mod.precision = mode
        try:
            with torch.no_grad():
                convolution.FUSE_CONVS = True
                a = mod.embed_frames(x, cmvn=(mean, None))
                convolution.FUSE_CONVS = False
                b = mod.embed_frames(x, cmvn=(mean, None))
        finally:
            convolution.FUSE_CONVS = True
        assert torch.equal(a, b), mode


def test_unsupported_sizes_are_refused():
    import cfm
    assert not cfm.conv12_supported(80, torch.bfloat16)
    assert cfm.conv12_supported(512, torch.bfloat16) and not cfm.conv12_supported(384, torch.bfloat16) and not cfm.conv12_supported(1024, torch.bfloat16)
    assert not cfm.conv12_supported(256, torch.float32)
    x = torch.zeros((1, 16, 16), device=DEV)
    with pytest.raises(RuntimeError):
        cfm.conv12_relu(x, torch.zeros((9, 80), device=DEV), torch.zeros((80,), device=DEV), torch.zeros((80, 720), device=DEV, dtype=torch.bfloat16),
                        torch.zeros((80,), device=DEV))
