"""GPU, one rank: the data-parallel step harness (trainer.DataParallelTrainer) on the real train path -- encoder (train mode) + CTC loss +
backward through the HIP kernels, flat-buffer clip + Adam kernel -- equals a plain autograd step with torch's clip_grad_norm_ and
torch.optim.Adam on a copy of the model; and the RCCL all-reduce path rehearsed with a one-rank process group."""
import copy
import os

import numpy as np
import pytest
import torch

import synth
from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda"


STRUCT_ZERO = ("linear_k.bias", "pos_bias_v", "linear_pos.weight", "depthwise_conv.bias")      # tests/test_train_modules_gpu.py


def build(meta, seed_shift=0):
    import decoder
    import encoder
    enc = synth.load_synth_(encoder.ConformerEncoder(cmvn=None, **meta["cfg"]), meta["wseed"] + seed_shift)
    dec = synth.load_synth_(decoder.CTCDecoder(meta["V"], meta["cfg"]["encoder_dim"], 0.0), meta["cseed"] + seed_shift)
    return enc.to(DEV).train(), dec.to(DEV).train()


def make_loss(enc, dec):
    def loss_fn(mb):
        x, lens, labels, label_lens = mb
        y, m = enc(x, lens)
        return dec(y, m.squeeze(1).sum(1), labels, label_lens)
    return loss_fn


def make_window_loss(enc, dec):
    def window_loss_fn(window):
        outs = enc.forward_window([(mb[0], mb[1]) for mb in window])
        return [dec(y, m.squeeze(1).sum(1), mb[2], mb[3]) for (y, m), mb in zip(outs, window)]
    return window_loss_fn


def micro_batches(n, seed):
    import trainer as T
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        feats, lens, labels, label_lens = T.librispeech_shaped_batch(rs, max_frames_in_batch=1600, min_len=120, max_len=400, vocab=73)
        out.append(tuple(torch.from_numpy(a).to(DEV) for a in (feats, lens, labels, label_lens)))
    return out


@pytest.mark.parametrize("use_pg,window", [(False, False), (True, False), (False, True), (True, True)])
def test_trainer_step_equals_plain_autograd_step(use_pg, window):
    import cfm
    import trainer as T
    import torch.distributed as dist
    cfm.set_precision("fp32")
    cfm.set_deterministic(True)        # Adam turns a gradient's SIGN into the update: rounding-noise gradients (structurally zero ones) must
    if use_pg:                         # be bit-identical between the two runs, so the weight-gradient GEMMs run without split-M atomics
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", HSA_ENABLE_IPC_MODE_LEGACY="0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        g, meta = load_golden("train_cfg1")
        enc, dec = build(meta)
        enc_r, dec_r = build(meta)
        steps, accum, clip, lr, warmup = 3, 2, 4.0, 1e-3, 2
        data = [micro_batches(accum, 500 + s) for s in range(steps)]
        tr = T.DataParallelTrainer([enc, dec], make_loss(enc, dec), lr=lr, warmup_steps=warmup, accum_grad=accum, grad_clip=clip, bucket_mb=1.0,
                                   always_reduce=use_pg, window_loss_fn=make_window_loss(enc, dec) if window else None)
        assert len(tr.buckets) >= 4 and tr.params[0] is dec.ctc_lo.bias
        params_r = list(enc_r.parameters()) + list(dec_r.parameters())
        opt = torch.optim.Adam(params_r, lr=lr)
        loss_r = make_loss(enc_r, dec_r)
        named = list(enc.named_parameters()) + list(dec.named_parameters())
        bufs, bufs_r = dict(enc.named_buffers()), dict(enc_r.named_buffers())
        worst = (0.0, "")
        for s in range(steps):
            # both models start every step from the SAME weights / BatchNorm statistics / Adam moments-by-construction: the trajectories of
            # two separately rounded Adam runs drift apart at the noise level (structurally-zero gradients, ReLU mask flips), which says
            # nothing about the harness.  What is compared is ONE step: accumulate 2 micro-batches, clip, Adam, WarmupLR.
            with torch.no_grad():
                for (k, a), b in zip(named, params_r):
                    b.copy_(a)
                for k in bufs:
                    bufs_r[k].copy_(bufs[k])
            before = [a.detach().clone() for _, a in named]
            loss = tr.step(data[s])
            assert tr.reduce_log == list(range(len(tr.buckets))), "every bucket launched once, in gradient-ready order"
            opt.zero_grad()
            tot = 0.0
            for mb in data[s]:
                l = loss_r(mb)
                (l / accum).backward()
                tot += float(l)
            norm = torch.nn.utils.clip_grad_norm_(params_r, clip)
            for gp in opt.param_groups:
                gp["lr"] = T.warmup_lr(lr, warmup, s + 1)
            opt.step()
            assert abs(float(loss) - tot / accum) < 1e-5 * abs(tot / accum)
            assert abs(float(tr.last_grad_norm) - float(norm)) < 1e-4 * float(norm)
            for ((k, a), b), a0 in zip(zip(named, params_r), before):
                if window and k.endswith(STRUCT_ZERO):
                    # gradients that are zero in exact arithmetic hold rounding noise, and Adam turns the noise's SIGN into a full +-lr step.  The
                    # window sums its rows in another order than two passes do, so the noise differs -- in any implementation
                    continue
                upd = float((b - a0).abs().max())                       # how far torch's step moved this parameter
                worst = max(worst, (float((a - b).abs().max()) / max(upd, 1e-12), k))
            for k in bufs:
                if "running" in k:
                    assert float((bufs[k] - bufs_r[k]).abs().max()) < 1e-6, k
        print("  trainer step vs plain autograd + clip_grad_norm_ + torch Adam: worst |difference| / |update| = %.3e (%s)" % worst)
        # torch keeps the Adam moments of ITS trajectory, the trainer of its own: identical by construction here (same gradients every step)
        # micro-batch loop: measured 1.5e-4 (round 2 and round 3 boxes).  Window: its gradients equal the loop's to 6e-7 of the largest entry
        # (tests/test_train_modules_gpu.py::test_accumulation_window_equals_sequential_micro_batches) -- the order of f32 sums -- and Adam's
        # first steps move an element by lr * g / (|g| + eps): entries with |g| ~ 1e-6 max|g| turn that noise into a per-cent of lr (measured 6.7e-3)
        assert worst[0] < (2e-2 if window else 3e-4), worst
    finally:
        if use_pg:
            dist.destroy_process_group()
        cfm.set_precision("bf16")
        cfm.set_deterministic(False)


@pytest.mark.parametrize("window", [False, True])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_trainer_bf16_loss_goes_down(p_drop, window):
    """20 optimizer steps on a fixed pair of micro-batches in the headline dtype, without and WITH the reference's dropout 0.1 (train.sh):
    the loss must fall (the step is wired end to end)."""
    import cfm
    import trainer as T
    cfm.set_precision("bf16")
    g, meta = load_golden("train_cfg1")
    meta = dict(meta, cfg=dict(meta["cfg"], dropout=p_drop, attention_dropout=p_drop, pos_enc_dropout=p_drop))
    torch.manual_seed(7)
    enc, dec = build(meta)
    tr = T.DataParallelTrainer([enc, dec], make_loss(enc, dec), lr=2e-3, warmup_steps=5, accum_grad=2, grad_clip=4.0,
                               window_loss_fn=make_window_loss(enc, dec) if window else None)
    data = micro_batches(2, 777)
    losses = [float(tr.step(data)) for _ in range(20)]
    print("  bf16 training loss (dropout %.1f, %s): %.3f -> %.3f" % (p_drop, "one pass per accumulation window" if window else "micro-batch loop", losses[0], losses[-1]))
    assert all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0]


def test_direct_gradient_sink_equals_accumulate_grad():
    """Blocks registered with a gradient sink can add their gradients straight into the trainer's flat gradient buffer (cfm/autograd.py
    DIRECT_GRADS, opt-in -- it measured slower: no zero-filled slab, no AccumulateGrad add) -- the flat gradient after 2 accumulated micro-batches equals the
    autograd-accumulated one up to the order of f32 atomic sums, the buckets are still launched once each and in order, and the buffer is
    clean again after the step."""
    import cfm
    import trainer as T
    from cfm import autograd as ag
    cfm.set_precision("fp32")
    try:
        g, meta = load_golden("train_cfg1")
        data = micro_batches(2, 900)
        grads = {}
        for direct in (True, False):
            ag.DIRECT_GRADS = direct
            enc, dec = build(meta)
            tr = T.DataParallelTrainer([enc, dec], make_loss(enc, dec), lr=1e-3, warmup_steps=2, accum_grad=2, grad_clip=4.0, bucket_mb=1.0, always_reduce=False)
            seen = {}
            orig_finish = tr.finish
            tr.finish = lambda: seen.setdefault("g", tr.flat_g.clone())        # look at the accumulated gradient instead of stepping
            torch.manual_seed(5)
            tr.step(data)
            assert tr.reduce_log == list(range(len(tr.buckets)))
            grads[direct] = seen["g"]
            tr.finish = orig_finish
        a, b = grads[True], grads[False]
        scale = float(b.abs().max())
        assert scale > 0 and float((a - b).abs().max()) < 2e-5 * scale, float((a - b).abs().max()) / scale
    finally:
        ag.DIRECT_GRADS = False
        cfm.set_precision("bf16")
