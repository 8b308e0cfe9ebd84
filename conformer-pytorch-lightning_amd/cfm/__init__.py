"""cfm -- ctypes binding of libconformer_gfx950.so (the C ABI declared in include/cfm.h).

This is plumbing only: it turns torch tensors that already live on an MI355X into raw device
pointers + sizes, passes the current HIP stream, and converts negative status codes into
``RuntimeError`` carrying ``cfm_last_error()``.  There is NO CPU fallback and no other backend: if the
shared library is missing, or a tensor is not on a HIP device, every entry point raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "libconformer_gfx950.so")

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_SILU, ACT_RELU, ACT_GLU, ACT_DSILU, ACT_DRELU = 0, 1, 2, 3, 4, 5
TILE_AUTO_TRAIN = -1            # cfm_gemm_desc.tile: automatic choice with the K-group tiles allowed (training paths; cfm.h)

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
_TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}

c_p = ctypes.c_void_p
c_i64 = ctypes.c_int64
c_i32 = ctypes.c_int32
c_f = ctypes.c_float


class GemmDesc(ctypes.Structure):
    _fields_ = [("A", c_p), ("W", c_p), ("W_lo", c_p), ("bias", c_p), ("residual", c_p), ("row_mask", c_p), ("C", c_p),
                ("lda", c_i64), ("ldc", c_i64), ("ldr", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("a_dtype", c_i32), ("w_dtype", c_i32), ("c_dtype", c_i32),
                ("act", c_i32), ("alpha", ctypes.c_float),
                ("conv_C", c_i32), ("conv_T1", c_i32), ("conv_F1", c_i32), ("conv_T2", c_i32), ("conv_F2", c_i32),
                ("tile", c_i32), ("mask_mode", c_i32),
                ("C_pre", c_p), ("ld_pre", c_i64), ("pre_dtype", c_i32), ("aux_dtype", c_i32), ("aux", c_p), ("ld_aux", c_i64),
                ("drop_p", ctypes.c_float), ("drop2_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32), ("drop2_seed", ctypes.c_uint32)]


class GemmTnDesc(ctypes.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("C", c_p), ("colsum", c_p), ("row_mask", c_p),
                ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64), ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("a_dtype", c_i32), ("b_dtype", c_i32), ("mma_dtype", c_i32), ("split", c_i32), ("accumulate", c_i32), ("splits", c_i32),
                ("alpha", ctypes.c_float),
                ("conv_C", c_i32), ("conv_T1", c_i32), ("conv_F1", c_i32), ("conv_T2", c_i32), ("conv_F2", c_i32),
                ("row_off", c_p), ("colsum_off", c_p), ("tile", c_i32), ("colsum_off2", c_p)]


class AttnBwdDesc(ctypes.Structure):
    _fields_ = [("q", c_p), ("k", c_p), ("v", c_p), ("mask", c_p), ("out", c_p), ("dout", c_p), ("lse", c_p),
                ("grad_q", c_p), ("grad_k", c_p), ("grad_v", c_p), ("delta", c_p),
                ("q_sb", c_i64), ("q_st", c_i64), ("k_sb", c_i64), ("k_st", c_i64), ("v_sb", c_i64), ("v_st", c_i64), ("m_sb", c_i64), ("m_sq", c_i64),
                ("B", c_i32), ("H", c_i32), ("Tq", c_i32), ("Tk", c_i32), ("dk", c_i32),
                ("io_dtype", c_i32), ("dout_dtype", c_i32), ("mma_dtype", c_i32), ("split", c_i32), ("scale", ctypes.c_float),
                ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32)]


class AttnDesc(ctypes.Structure):
    _fields_ = [("q", c_p), ("k", c_p), ("v", c_p), ("p", c_p), ("bias_u", c_p), ("bias_v", c_p), ("mask", c_p), ("out", c_p),
                ("q_sb", c_i64), ("q_st", c_i64), ("k_sb", c_i64), ("k_st", c_i64), ("k_sh", c_i64),
                ("v_sb", c_i64), ("v_st", c_i64), ("v_sh", c_i64), ("p_sb", c_i64), ("p_st", c_i64),
                ("m_sb", c_i64), ("m_sq", c_i64),
                ("B", c_i32), ("H", c_i32), ("Tq", c_i32), ("Tk", c_i32), ("dk", c_i32),
                ("q_dtype", c_i32), ("kv_dtype", c_i32), ("p_dtype", c_i32), ("out_dtype", c_i32), ("mma_dtype", c_i32),
                ("split", c_i32), ("scale", ctypes.c_float), ("lse", c_p), ("drop_p", ctypes.c_float), ("drop_seed", ctypes.c_uint32)]


class FfnDesc(ctypes.Structure):
    _fields_ = [("x", c_p), ("ln_g", c_p), ("ln_b", c_p), ("w1f", c_p), ("w2f", c_p), ("b1", c_p), ("b2", c_p),
                ("ln1_g", c_p), ("ln1_b", c_p), ("ln2_g", c_p), ("ln2_b", c_p), ("out_f32", c_p), ("out16", c_p),
                ("M", c_i64), ("D", c_i32), ("FF", c_i32), ("w_dtype", c_i32), ("out16_dtype", c_i32), ("act", c_i32),
                ("add_x", c_i32), ("alpha", ctypes.c_float), ("eps", ctypes.c_float)]


class RowChainDesc(ctypes.Structure):
    _fields_ = [("x", c_p), ("head_a", c_p), ("head_w", c_p), ("head_b", c_p), ("head_res", c_p), ("head_mask", c_p),
                ("ln_g", c_p), ("ln_b", c_p), ("ln_mask", c_p), ("w1f", c_p), ("w2n", c_p), ("b1", c_p), ("b2", c_p),
                ("ln1_g", c_p), ("ln1_b", c_p), ("ln2_g", c_p), ("ln2_b", c_p), ("out_f32", c_p), ("out16", c_p),
                ("tail_w", c_p), ("tail_b", c_p), ("tail_out", c_p), ("M", c_i64), ("D", c_i32), ("FF", c_i32),
                ("tail_N", c_i32), ("tail_glu", c_i32), ("w_dtype", c_i32), ("alpha", ctypes.c_float), ("eps", ctypes.c_float),
                ("out2_f32", c_p), ("dw_w", c_p), ("dw_b", c_p), ("dw_scale", c_p), ("dw_shift", c_p), ("dw_T", c_i32), ("dw_K", c_i32),
                ("tail_vt", c_p), ("vt_T", c_i32), ("vt_ld", c_i32),
                ("att_qkv", c_p), ("att_vt", c_p), ("att_p", c_p), ("att_bias_u", c_p), ("att_bias_v", c_p), ("att_mask", c_p),
                ("att_p_sb", c_i64), ("att_m_sb", c_i64), ("att_T", c_i32), ("att_H", c_i32), ("att_vt_ld", c_i32), ("att_scale", ctypes.c_float),
                ("s2_ln_g", c_p), ("s2_ln_b", c_p), ("s2_w1f", c_p), ("s2_w2n", c_p), ("s2_b1", c_p), ("s2_b2", c_p), ("s2_out_f32", c_p),
                ("s2_alpha", ctypes.c_float), ("psum_out", c_p), ("psum_in", c_p), ("psum_b2", c_p), ("psum_alpha", ctypes.c_float),
                ("cin_a", c_p), ("cin_w", c_p), ("cin_b", c_p), ("cin_res", c_p), ("cin_out", c_p), ("cin_ln_g", c_p), ("cin_ln_b", c_p), ("cin_mask", c_p),
                ("cin_tail_w", c_p), ("cin_tail_b", c_p), ("tail_pair", c_i32)]


_LAYER_W_FIELDS = [
    "ln_ffm_g", "ln_ffm_b", "ln_mha_g", "ln_mha_b", "ln_conv_g", "ln_conv_b", "ln_ff_g", "ln_ff_b", "ln_final_g", "ln_final_b",
    "ffm_w1", "ffm_w1_lo", "ffm_w2", "ffm_w2_lo", "ffm_b1", "ffm_b2",
    "ff_w1", "ff_w1_lo", "ff_w2", "ff_w2_lo", "ff_b1", "ff_b2", "ffm_w1f", "ffm_w2f", "ff_w1f", "ff_w2f", "ffm_w2n", "ff_w2n", "qkv_wf", "out_wf", "pw1_wf", "pw2_wf",
    "qkv_w", "qkv_w_lo", "pos_w", "pos_w_lo", "out_w", "out_w_lo", "qkv_b", "out_b", "bias_u", "bias_v",
    "pw1_w", "pw1_w_lo", "pw2_w", "pw2_w_lo", "pw1_b", "pw2_b", "dw_w", "dw_b", "bn_scale", "bn_shift"]


_TRAIN_W_PTRS = ["ln_ffm_g", "ln_ffm_b", "ln_mha_g", "ln_mha_b", "ln_conv_g", "ln_conv_b", "ln_ff_g", "ln_ff_b", "ln_final_g", "ln_final_b",
                 "ffm_w1", "ffm_w1_lo", "ffm_w2", "ffm_w2_lo", "ffm_w1t", "ffm_w1t_lo", "ffm_w2t", "ffm_w2t_lo", "ffm_b1", "ffm_b2",
                 "ff_w1", "ff_w1_lo", "ff_w2", "ff_w2_lo", "ff_w1t", "ff_w1t_lo", "ff_w2t", "ff_w2t_lo", "ff_b1", "ff_b2",
                 "qkv_w", "qkv_w_lo", "qkv_t", "qkv_t_lo", "out_w", "out_w_lo", "out_t", "out_t_lo", "qkv_b", "out_b",
                 "pw1_w", "pw1_w_lo", "pw1_t", "pw1_t_lo", "pw2_w", "pw2_w_lo", "pw2_t", "pw2_t_lo", "pw1_b", "pw2_b", "dw_w", "dw_b", "bn_gamma", "bn_beta",
                 "bn_running_mean", "bn_running_var"]


class LayerTrainWeights(ctypes.Structure):
    _fields_ = [(n, c_p) for n in _TRAIN_W_PTRS] + [("bn_momentum", ctypes.c_float), ("bn_eps", ctypes.c_float)] + \
               [(n, c_p) for n in ("ffm_w1f", "ffm_w2f", "ff_w1f", "ff_w2f")]


class FfnTrainDesc(ctypes.Structure):
    _fields_ = [("x", c_p), ("ln_g", c_p), ("ln_b", c_p), ("w1f", c_p), ("w2f", c_p), ("b1", c_p), ("b2", c_p), ("y", c_p), ("xn_out", c_p), ("z_out", c_p),
                ("h_out", c_p), ("M", c_i64), ("D", c_i32), ("FF", c_i32), ("w_dtype", c_i32), ("alpha", c_f), ("eps", c_f), ("p_hidden", c_f), ("p_out", c_f),
                ("seed_hidden", ctypes.c_uint32), ("seed_out", ctypes.c_uint32)]


class TrainGroup(ctypes.Structure):
    _fields_ = [("B", c_i32), ("T", c_i32), ("row0", c_i64), ("attn_mask", c_p), ("am_sb", c_i64), ("am_sq", c_i64)]


LAYER_DONE_FN = ctypes.CFUNCTYPE(None, c_i32, c_p)


class LayerTrainIO(ctypes.Structure):
    _fields_ = [("B", c_i32), ("T", c_i32), ("D", c_i32), ("H", c_i32), ("FF", c_i32), ("ktaps", c_i32), ("act_dtype", c_i32), ("w_dtype", c_i32),
                ("attn_mask", c_p), ("am_sb", c_i64), ("am_sq", c_i64), ("pad_valid", c_p),
                ("p_hidden_m", ctypes.c_float), ("p_hidden", ctypes.c_float), ("p_branch", ctypes.c_float), ("p_attn", ctypes.c_float),
                ("p_attn_out", ctypes.c_float), ("seed", ctypes.c_uint32), ("deterministic", c_i32), ("grads_accumulate", c_i32), ("side_stream", c_p),
                ("n_groups", c_i32), ("groups", ctypes.POINTER(TrainGroup)), ("defer_wgrad", c_i32)]


_TRAIN_SAVED = ["xn1", "z1", "h1", "xn2", "qkv", "ctx", "xn3", "u", "glu", "s", "xn4", "z2", "h2", "x1", "x2", "x3", "x4", "c", "lse", "stats"]
_TRAIN_SCRATCH = ["dxn", "dz", "dyb", "ds", "dglu", "du", "dctx", "dqkv", "delta", "ln_ws", "dwbn_ws", "dy_ws", "dz2", "dyb2", "dyb3", "dyb4"]
_TRAIN_GRADS = ["slab", "ln_ffm_g", "ln_ffm_b", "ln_mha_g", "ln_mha_b", "ln_conv_g", "ln_conv_b", "ln_ff_g", "ln_ff_b", "ln_final_g", "ln_final_b",
                "ffm_w1", "ffm_b1", "ffm_w2", "ffm_b2", "ff_w1", "ff_b1", "ff_w2", "ff_b2", "out_w", "out_b", "pw2_w", "pw2_b", "dw_w", "dw_b", "bn_g", "bn_b",
                "pos_bias_u", "q_bias", "qkv_row_off", "qkv_bias_off", "pw1_row_off", "pw1_bias_off", "qkv_bias_off2"]


class LayerTrainSaved(ctypes.Structure):
    _fields_ = [(n, c_p) for n in _TRAIN_SAVED]


class LayerTrainScratch(ctypes.Structure):
    _fields_ = [(n, c_p) for n in _TRAIN_SCRATCH]


class LayerTrainGrads(ctypes.Structure):
    _fields_ = [(n, c_p) for n in _TRAIN_GRADS]


class LayerWeights(ctypes.Structure):
    _fields_ = [(n, c_p) for n in _LAYER_W_FIELDS]


class CtcGroup(ctypes.Structure):
    _fields_ = [("logits", c_p), ("ld", c_i64), ("B", c_i32), ("T", c_i32), ("Umax", c_i32), ("enc_lens", c_p), ("labels", c_p), ("label_lens", c_p),
                ("work", c_p), ("alpha", c_p), ("lse", c_p), ("nll", c_p), ("nll_shifted", c_p), ("beta", c_p)]


class LnBwdDesc(ctypes.Structure):
    _fields_ = [("x", c_p), ("dy", c_p), ("gamma", c_p), ("row_mask", c_p), ("dres", c_p), ("dx", c_p), ("dgamma", c_p), ("dbeta", c_p), ("ws", c_p),
                ("dx2", c_p), ("M", c_i64), ("D", c_i32), ("dy_dtype", c_i32), ("dx2_dtype", c_i32), ("accumulate", c_i32),
                ("eps", ctypes.c_float), ("alpha2", ctypes.c_float), ("p1", ctypes.c_float), ("p2", ctypes.c_float),
                ("seed1", ctypes.c_uint32), ("seed2", ctypes.c_uint32), ("dx2_row_mask", c_p),
                ("chain_x", c_p), ("chain_gamma", c_p), ("chain_dgamma", c_p), ("chain_dbeta", c_p)]


class GreedyDesc(ctypes.Structure):
    _fields_ = [("embed", c_p), ("lstm_w", c_p * 4), ("lstm_b", c_p * 4), ("proj_w", c_p), ("proj_b", c_p), ("pf_w", c_p), ("pf_b", c_p), ("out_w", c_p),
                ("out_b", c_p), ("enc_proj", c_p), ("token", c_p), ("t", c_p), ("count", c_p), ("frame_count", c_p), ("hyps", c_p), ("lens", c_p),
                ("h", c_p), ("c", c_p), ("h_new", c_p), ("c_new", c_p), ("pred", c_p), ("act", c_p), ("pmax", c_p), ("pidx", c_p), ("done", c_p),
                ("n_done", c_p), ("hyp_cap", c_i64), ("hyp_ld", c_i64), ("B", c_i32), ("T", c_i32), ("L", c_i32), ("E", c_i32), ("H", c_i32),
                ("P", c_i32), ("J", c_i32), ("Vp", c_i32), ("blank", c_i32), ("n_steps", c_i32)]


class FfnSplitDesc(ctypes.Structure):
    _fields_ = [("x", c_p), ("psum", c_p), ("psum_b2", c_p), ("psum_splits", c_i32), ("psum_alpha", c_f), ("ln1_g", c_p), ("ln1_b", c_p), ("ln2_g", c_p),
                ("ln2_b", c_p), ("rows_out", c_p), ("rows2_out", c_p), ("ln_g", c_p), ("ln_b", c_p), ("w1", c_p), ("b1", c_p), ("N1", c_i32), ("act", c_i32),
                ("w2", c_p), ("psum_out", c_p), ("out16", c_p), ("ldo", c_i64), ("M", c_i32), ("D", c_i32), ("mode", c_i32), ("w_dtype", c_i32), ("eps", c_f),
                ("kv_ring", c_p), ("ring_offsets", c_p), ("ring_T", c_i32), ("ring_H", c_i32), ("ring_Tq", c_i32)]


class LayerScratch(ctypes.Structure):
    _fields_ = [(n, c_p) for n in ("xn", "hid", "qkv", "pos", "ctx", "glu", "dw", "vt")] + [("vt_ld", c_i32), ("psum", c_p), ("psum_splits", c_i32)]


class LayerIO(ctypes.Structure):
    _fields_ = [("B", c_i32), ("T", c_i32), ("D", c_i32), ("H", c_i32), ("FF", c_i32), ("ktaps", c_i32),
                ("act_dtype", c_i32), ("w_dtype", c_i32),
                ("attn_mask", c_p), ("am_sb", c_i64), ("am_sq", c_i64),
                ("pad_valid", c_p), ("pos_embed", c_p), ("pos_rows", c_i32),
                ("pos_proj", c_p), ("pos_proj_ld", c_i64),
                ("attn_cache", c_p), ("cache_T", c_i32), ("new_cache", c_p), ("after_g", c_p), ("after_b", c_p), ("after_out", c_p),
                ("kv_ring", c_p), ("stream_offset", c_p), ("ring_T", c_i32), ("causal_conv", c_i32), ("conv_cache", c_p), ("pos_shared", c_i32),
                ("next_w", c_p), ("next_x_out", c_p), ("macaron_done", c_i32)]


_lib = None
_lock = threading.Lock()


def lib():
    """The loaded shared library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libconformer_gfx950.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C conformer-pytorch-lightning_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.cfm_version.restype = ctypes.c_int
        L.cfm_last_error.restype = ctypes.c_char_p
        L.cfm_device_ok.restype = ctypes.c_int
        L.cfm_gemm.argtypes = [ctypes.POINTER(GemmDesc), c_p]
        L.cfm_attention.argtypes = [ctypes.POINTER(AttnDesc), c_p]
        L.cfm_ffn_fused.argtypes = [ctypes.POINTER(FfnDesc), c_p]
        L.cfm_ffn_train_forward.argtypes = [ctypes.POINTER(FfnTrainDesc), c_p]
        L.cfm_ffn_train_supported.argtypes = [c_i32, c_i32]
        L.cfm_pack_ffn_fragments.argtypes = [c_p, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_rowchain.argtypes = [ctypes.POINTER(RowChainDesc), c_p]
        L.cfm_rowchain_supported.argtypes = [c_i32, c_i32]
        L.cfm_rowchain_pair_supported.argtypes = [c_i32, c_i32]
        L.cfm_layernorm.argtypes = [c_p, c_p, c_p, c_p, c_i32, c_p, c_p, c_p, c_i32, c_p, ctypes.c_float, c_i64, c_i32, c_p]
        L.cfm_kv_cache_pack.argtypes = [c_p, c_i32, c_p, c_p, c_i32, c_i64, c_i64, c_i64, c_i64, c_p, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_dwconv_bn_silu.argtypes = [c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_conv1_relu.argtypes = [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]
        L.cfm_conv1_relu_mma.argtypes = [c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]
        L.cfm_conv12_supported.argtypes = [c_i32, c_i32]
        L.cfm_pack_matrices.argtypes = [c_p, c_i32, c_i64, c_i32, c_i32, c_p]
        L.cfm_pack_vectors.argtypes = [c_p, c_p, c_p, c_i64, c_p]
        L.cfm_greedy_step.argtypes = [ctypes.POINTER(GreedyDesc), c_p]
        L.cfm_ffn_split.argtypes = [ctypes.POINTER(FfnSplitDesc), c_p]
        L.cfm_ffn_split_supported.argtypes = [c_i32, c_i32]
        L.cfm_attention_bwd_force_general.argtypes = [c_i32]
        L.cfm_attention_bwd_force_general.restype = None
        L.cfm_set_cin_merge.argtypes = [c_i32]
        L.cfm_set_cin_merge.restype = c_i32
        L.cfm_dwconv_bn_train_bwd_acc.argtypes = [c_p, c_i32, c_p, c_p, c_p, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_layernorm_bwd_fused.argtypes = [ctypes.POINTER(LnBwdDesc), c_p]
        L.cfm_conv12_relu.argtypes = [c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p, c_p, c_p]
        L.cfm_valid_mask.argtypes = [c_p, c_i32, c_p, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_chunk_mask.argtypes = [c_p, c_i32, c_i32, c_i32, c_p]
        L.cfm_attn_mask.argtypes = [c_p, c_p, c_p, c_i32, c_i32, c_p]
        L.cfm_cast.argtypes = [c_p, c_i32, c_p, c_i32, c_i64, c_p]
        L.cfm_add_rows.argtypes = [c_p, c_p, c_i64, c_i32, c_i32, c_p]
        L.cfm_encoder_layer_forward.argtypes = [ctypes.POINTER(LayerWeights), ctypes.POINTER(LayerScratch),
                                                ctypes.POINTER(LayerIO), c_p, c_p, c_i32, c_p, c_p, c_p]
        L.cfm_ctc_nll.argtypes = [c_p, c_i64, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p]
        L.cfm_joint_act.argtypes = [c_p, c_i64, c_p, c_i64, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]
        c_f = ctypes.c_float
        L.cfm_gemm_tn.argtypes = [ctypes.POINTER(GemmTnDesc), c_p]
        L.cfm_gemm_tn_group.argtypes = [ctypes.POINTER(GemmTnDesc), c_i32, c_p]
        L.cfm_attention_bwd.argtypes = [ctypes.POINTER(AttnBwdDesc), c_p]
        L.cfm_layernorm_bwd_ws.argtypes = [c_i64, c_i32]
        L.cfm_layernorm_bwd.argtypes = [c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_i64, c_i32, c_p]
        L.cfm_glu_bwd.argtypes = [c_p, c_i32, c_p, c_i32, c_p, c_i32, c_i64, c_i32, c_p]
        L.cfm_dwconv_bn_ws.argtypes = [c_i32, c_i32, c_i32]
        L.cfm_dwconv_bn_train.argtypes = [c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_i32, c_p, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_dwconv_bn_train_bwd.argtypes = [c_p, c_i32, c_p, c_p, c_p, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_col2im_relu_bwd.argtypes = [c_p, c_i32, c_p, c_i32, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_conv1_wgrad_ws.argtypes = [c_i32, c_i32, c_i32]
        L.cfm_conv1_wgrad.argtypes = [c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_ctc_nll_train.argtypes = [c_p, c_i64, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]
        L.cfm_ctc_nll_train_groups.argtypes = [ctypes.POINTER(CtcGroup), c_i32, c_i32, c_p]
        L.cfm_ctc_grad.argtypes = [c_p, c_i64, c_i32, c_i32, c_i32, c_p, c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p]
        L.cfm_adam_step.argtypes = [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_p, c_p]
        L.cfm_sumsq.argtypes = [c_p, c_i64, c_p, c_i32, c_p, c_p]
        L.cfm_adam_clip_step.argtypes = [c_p, c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_f, c_i64, c_p, c_f, c_f, c_i32, c_p, c_p]
        L.cfm_dropout_rows.argtypes = [c_p, c_i32, c_p, c_i32, c_p, c_f, c_f, ctypes.c_uint32, c_f, ctypes.c_uint32, c_i64, c_i32, c_p]
        L.cfm_dropout_mask.argtypes = [c_p, c_i64, c_f, ctypes.c_uint32, c_p]
        L.cfm_encoder_layer_train_forward.argtypes = [ctypes.POINTER(LayerTrainWeights), ctypes.POINTER(LayerTrainIO), ctypes.POINTER(LayerTrainSaved),
                                                      ctypes.POINTER(LayerTrainScratch), c_p, c_p, c_p]
        L.cfm_encoder_layer_train_backward.argtypes = [ctypes.POINTER(LayerTrainWeights), ctypes.POINTER(LayerTrainIO), ctypes.POINTER(LayerTrainSaved),
                                                       ctypes.POINTER(LayerTrainScratch), ctypes.POINTER(LayerTrainGrads), c_p, c_p, c_p, c_p]
        L.cfm_encoder_train_forward.argtypes = [c_i32, ctypes.POINTER(LayerTrainWeights), ctypes.POINTER(LayerTrainIO), ctypes.POINTER(LayerTrainSaved),
                                                ctypes.POINTER(LayerTrainScratch), ctypes.POINTER(c_p), c_p]
        L.cfm_encoder_train_backward.argtypes = [c_i32, ctypes.POINTER(LayerTrainWeights), ctypes.POINTER(LayerTrainIO), ctypes.POINTER(LayerTrainSaved),
                                                 ctypes.POINTER(LayerTrainScratch), c_i32, ctypes.POINTER(LayerTrainGrads), ctypes.POINTER(c_p), c_p, c_p, c_p,
                                                 LAYER_DONE_FN, c_p, ctypes.POINTER(c_p), c_p]
        L.cfm_stream_prep.argtypes = [c_p, c_i32, c_i32, c_i32, c_i32, c_p, c_i32, c_i32, c_p, c_p, c_p, c_p]
        L.cfm_kv_ring_write.argtypes = [c_p, c_p, c_i32, c_i64, c_i64, c_i64, c_i64, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_stream_advance.argtypes = [c_p, c_p, c_i32, c_i32, c_p]
        L.cfm_dwconv_causal_bn_silu.argtypes = [c_p, c_i32, c_p, c_p, c_p, c_p, c_p, c_p, c_i32, c_i32, c_i32, c_i32, c_i32, c_p]
        L.cfm_conv_cache_update.argtypes = [c_p, c_i32, c_p, c_i32, c_i32, c_i32, c_i32, c_p]
        for name in ("cfm_layernorm_bwd_ws", "cfm_dwconv_bn_ws", "cfm_conv1_wgrad_ws"):
            getattr(L, name).restype = c_i64
        L.cfm_prof_enable.argtypes = [c_i32]
        L.cfm_prof_enable.restype = None
        L.cfm_prof_reset.restype = None
        L.cfm_prof_collect.restype = ctypes.c_int
        L.cfm_prof_entry.argtypes = [c_i32, ctypes.c_char_p, c_i32, ctypes.POINTER(c_i64), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
        for name in ("cfm_gemm", "cfm_ffn_fused", "cfm_ffn_train_forward", "cfm_ffn_train_supported", "cfm_pack_ffn_fragments", "cfm_rowchain", "cfm_rowchain_supported", "cfm_attention", "cfm_layernorm", "cfm_kv_cache_pack", "cfm_dwconv_bn_silu", "cfm_conv1_relu", "cfm_conv1_relu_mma", "cfm_conv12_relu", "cfm_conv12_supported",
                     "cfm_valid_mask", "cfm_chunk_mask", "cfm_attn_mask", "cfm_cast", "cfm_add_rows",
                     "cfm_encoder_layer_forward", "cfm_ctc_nll", "cfm_joint_act", "cfm_prof_entry", "cfm_gemm_tn", "cfm_gemm_tn_group", "cfm_attention_bwd",
                     "cfm_layernorm_bwd", "cfm_glu_bwd", "cfm_dwconv_bn_train", "cfm_dwconv_bn_train_bwd", "cfm_col2im_relu_bwd", "cfm_conv1_wgrad",
                     "cfm_ctc_nll_train", "cfm_ctc_nll_train_groups", "cfm_ctc_grad", "cfm_adam_step", "cfm_adam_clip_step", "cfm_sumsq", "cfm_dropout_rows", "cfm_dropout_mask", "cfm_pack_matrices", "cfm_pack_vectors", "cfm_greedy_step", "cfm_ffn_split", "cfm_ffn_split_supported", "cfm_layernorm_bwd_fused", "cfm_dwconv_bn_train_bwd_acc",
                     "cfm_encoder_layer_train_forward", "cfm_encoder_layer_train_backward", "cfm_encoder_train_forward", "cfm_encoder_train_backward", "cfm_stream_prep", "cfm_kv_ring_write", "cfm_stream_advance", "cfm_dwconv_causal_bn_silu", "cfm_conv_cache_update"):
            getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, lib().cfm_last_error().decode(errors="replace")))


def dt_code(t):
    try:
        return _DT[t.dtype if isinstance(t, torch.Tensor) else t]
    except KeyError:
        raise TypeError("cfm: unsupported dtype %s (float32, bfloat16, float16 only)" % (t.dtype if isinstance(t, torch.Tensor) else t))


def torch_dtype(code):
    return _TORCH_DT[code]


def require_hip(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("cfm: tensor on %s -- this framework runs on MI355X (HIP) only; there is no CPU path" % t.device)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# ----------------------------------------------------------------------------------------------------------------------
# precision modes
# ----------------------------------------------------------------------------------------------------------------------
class Precision:
    """How the dense contractions are evaluated.

    bf16 : bf16 MFMA operands, f32 accumulate, bf16 intermediates, f32 residual stream (BASELINE.json config 2)
    fp16 : same speed, fp16 operands (11-bit mantissa)
    fp32 : "f32-accurate": f32 intermediates, every product as 3 bf16 MFMAs on hi/lo splits (~16 mantissa bits)
    """

    def __init__(self, name):
        if name not in ("bf16", "fp16", "fp32"):
            raise ValueError("precision must be bf16, fp16 or fp32, got %r" % (name,))
        self.name = name
        self.split = name == "fp32"
        self.w_code = F16 if name == "fp16" else BF16
        self.act_code = F32 if self.split else self.w_code
        self.w_dtype = torch_dtype(self.w_code)
        self.act_dtype = torch_dtype(self.act_code)

    def __repr__(self):
        return "Precision(%s)" % self.name


_precision = Precision(os.environ.get("CFM_PRECISION", "bf16"))


def set_precision(name):
    global _precision
    _precision = name if isinstance(name, Precision) else Precision(name)
    return _precision


def get_precision():
    return _precision


def resolve_precision(module=None):
    """The precision a module computes in: its own `precision` attribute when set (a name or a Precision; see
    ConformerEncoder.set_precision), else the process default.  Resolved per call -- two encoders at different precisions can run
    side by side."""
    p = getattr(module, "precision", None) if module is not None else None
    if p is None:
        return _precision
    return p if isinstance(p, Precision) else Precision(p)


_warned = set()


def check_mode(module, what):
    """Shared train / eval gate of the drop-in modules.  Returns True when the module must take its TRAIN path (module.training):
    BatchNorm batch statistics, dropout from the kernels' counter-based generator, autograd Functions.
    In eval mode the forward is inference-only: gradients do not flow (warned once when someone might expect them to)."""
    if module.training:
        return True
    if torch.is_grad_enabled() and what not in _warned and any(q.requires_grad for q in module.parameters()):
        _warned.add(what)
        import warnings
        warnings.warn("%s: eval-mode forward runs inference kernels; its output is detached from the parameters. "
                      "Call .train() for a differentiable forward, or wrap inference in torch.no_grad()." % what)
    return False


from .ops import *  # noqa: E402,F401,F403
