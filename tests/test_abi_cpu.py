"""CPU (no GPU): the C-ABI shared library loads, exports every symbol include/cfm.h declares, the ctypes mirrors of its
structs have the C sizes, and the host-side logic (weight packing index maps, module construction, loud failures) is right.
No compute entry point is called."""
import ctypes
import json
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden

HDR = os.path.join(ROOT, "include", "cfm.h")


@pytest.fixture(scope="module")
def cfm():
    import cfm as c
    if not os.path.exists(c.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return c


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HDR).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(cfm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(cfm):
    lib = cfm.lib()
    names = declared_functions()
    assert len(names) >= 20 and "cfm_gemm" in names and "cfm_ffn_fused" in names and "cfm_encoder_layer_forward" in names
    for n in names:
        assert hasattr(lib, n), "libconformer_gfx950.so does not export %s" % n
    assert lib.cfm_version() == 302
    assert isinstance(lib.cfm_last_error(), bytes)


def test_ctypes_structs_match_c_sizes(cfm, tmp_path):
    structs = {"cfm_gemm_desc": cfm.GemmDesc, "cfm_attn_desc": cfm.AttnDesc, "cfm_ffn_desc": cfm.FfnDesc,
               "cfm_layer_weights": cfm.LayerWeights, "cfm_layer_scratch": cfm.LayerScratch, "cfm_layer_io": cfm.LayerIO,
               "cfm_gemm_tn_desc": cfm.GemmTnDesc, "cfm_attn_bwd_desc": cfm.AttnBwdDesc, "cfm_rowchain_desc": cfm.RowChainDesc,
               "cfm_layer_train_weights": cfm.LayerTrainWeights, "cfm_layer_train_io": cfm.LayerTrainIO, "cfm_layer_train_saved": cfm.LayerTrainSaved,
               "cfm_layer_train_scratch": cfm.LayerTrainScratch, "cfm_layer_train_grads": cfm.LayerTrainGrads,
               "cfm_ln_bwd_desc": cfm.LnBwdDesc, "cfm_train_group": cfm.TrainGroup, "cfm_ctc_group": cfm.CtcGroup, "cfm_ffn_train_desc": cfm.FfnTrainDesc, "cfm_greedy_desc": cfm.GreedyDesc, "cfm_ffn_split_desc": cfm.FfnSplitDesc}
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "cfm.h"\nint main(){' +
                   "".join('printf("%s %%zu\\n", sizeof(%s));' % (n, n) for n in structs) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for n, cls in structs.items():
        assert int(out[n]) == ctypes.sizeof(cls), (n, out[n], ctypes.sizeof(cls))


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "c.c"
    src.write_text('#include "cfm.h"\nint main(void){return CFM_VERSION == 302 ? 0 : 1;}\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "c")], check=True)


def test_ffn_fragment_packing_index_maps(cfm):
    from cfm import packing
    for D, FF in ((256, 2048), (144, 576), (16, 32)):
        w1 = torch.arange(FF * D, dtype=torch.float32).reshape(FF, D) % 251          # exact in bf16 up to 256
        w2 = (torch.arange(D * FF, dtype=torch.float32).reshape(D, FF) * 7) % 241
        w1f, w2f = packing.pack_ffn_fragments(w1, w2, torch.bfloat16)
        ks1, nf2 = (D + 31) // 32, D // 16
        w1f = w1f.float().reshape(FF // 16, ks1, 64, 8)
        w2f = w2f.float().reshape(FF // 32, nf2, 64, 8)
        rs = np.random.RandomState(0)
        for _ in range(200):
            ffb, kk, lane, j = rs.randint(FF // 16), rs.randint(ks1), rs.randint(64), rs.randint(8)
            k = kk * 32 + 8 * (lane >> 4) + j
            want = float(w1[ffb * 16 + (lane & 15), k]) if k < D else 0.0
            assert float(w1f[ffb, kk, lane, j]) == want
            fs, nf = rs.randint(FF // 32), rs.randint(nf2)
            ff = fs * 32 + (0 if j < 4 else 16) + 4 * (lane >> 4) + (j & 3)
            assert float(w2f[fs, nf, lane, j]) == float(w2[nf * 16 + (lane & 15), ff])


def test_glu_interleave_and_bn_fold(cfm):
    from cfm import packing
    D = 32
    idx = packing.glu_interleave_index(D, "cpu")
    assert sorted(idx.tolist()) == list(range(2 * D))
    for col in range(2 * D):                      # GEMM column blk*32 + half*16 + i  <-  weight row half*D + blk*16 + i
        blk, half, i = col // 32, (col % 32) // 16, col % 16
        assert int(idx[col]) == half * D + blk * 16 + i
    import convolution
    m = convolution.ConvolutionModule(D, 15, 64).eval()
    with torch.no_grad():
        m.norm.running_mean.normal_()
        m.norm.running_var.uniform_(0.5, 1.5)
        m.norm.weight.normal_()
        m.norm.bias.normal_()
    pk = packing.pack_conv_module(m, cfm.Precision("bf16"))
    x = torch.randn(7, D)
    ref = torch.nn.functional.batch_norm(x, m.norm.running_mean, m.norm.running_var, m.norm.weight, m.norm.bias, False, 0.0, m.norm.eps)
    assert torch.allclose(x * pk.bn_scale + pk.bn_shift, ref, atol=1e-5)
    assert pk.dw_w.shape == (D, 15) and pk.pw1_w.shape == (2 * D, D) and pk.pw1_w.dtype == torch.bfloat16


def test_modules_have_the_reference_parameter_names():
    import encoder
    g, meta = load_golden("enc_cfg1")
    enc = encoder.ConformerEncoder(cmvn=None, **meta["cfg"])
    assert {k: list(v.shape) for k, v in enc.state_dict().items()} == meta["state"]
    g, meta = load_golden("enc_cfg1_norel")
    enc = encoder.ConformerEncoder(cmvn=None, **meta["cfg"])
    assert {k: list(v.shape) for k, v in enc.state_dict().items()} == meta["state"]
    g, _ = load_golden("mods_d144")
    import encoder_layer
    layer = encoder_layer.ConformerEncoderLayer(144, 15, 0.1, 0.1, 576, 4, True)
    assert {k: list(v.shape) for k, v in layer.state_dict().items()} == json.loads(bytes(g["layer_manifest"]).decode())


def test_no_cpu_fallback_anywhere(cfm):
    import attention
    import convolution
    import encoder
    import feedforward
    import utils
    x = torch.zeros(2, 9, 16)
    empty = torch.ones((0, 0, 0), dtype=torch.bool)
    with pytest.raises(RuntimeError, match="no CPU path"):
        feedforward.PositionwiseFeedForwardModule(16, 0.0, 32).eval()(x)
    with pytest.raises(RuntimeError, match="no CPU path"):
        attention.MultiHeadSelfAttentionModule(16, 2, 0.0).eval()(x, x, x, empty)
    with pytest.raises(RuntimeError, match="no CPU path"):
        convolution.ConvolutionModule(16, 15).eval()(x, empty)
    with pytest.raises(RuntimeError, match="no CPU path"):
        utils.make_pad_mask(torch.tensor([3, 2], dtype=torch.int32), 5)
    enc = encoder.ConformerEncoder(80, 15, 16, 0.0, 0.0, 0.0, 32, 2, 1, use_relative=True).eval()
    with pytest.raises(RuntimeError, match="no CPU path"):
        enc(torch.zeros(1, 50, 80), torch.tensor([50], dtype=torch.int32))
    with pytest.raises(ValueError):
        cfm.Precision("int8")


def test_product_code_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import it (or fall back to it)."""
    pkg = os.path.join(ROOT, "conformer-pytorch-lightning_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.*oracle)", re.M)
    checked = 0
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                checked += 1
                assert not pat.search(open(os.path.join(base, f)).read()), os.path.join(base, f)
    assert checked >= 10
