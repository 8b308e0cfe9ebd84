"""Data-parallel training step for the hot path (BASELINE config 3): encoder + CTC loss + backward, gradient all-reduce over RCCL/xGMI.

What the reference does, implicitly, through Lightning (src/executor.py:137-154 devices=N / accumulate_grad_batches / gradient_clip_val,
train.sh:35-36 clip 4 / accum 2, src/module.py:140-143 Adam + WarmupLR stepped per optimizer step, src/dataset.py:54-58 data[rank::world]):
one process per GPU, DDP's bucketed gradient all-reduce overlapped with backward, `no_sync` on all but the last accumulated micro-batch,
clip-by-global-norm after the reduce, Adam.  The MI355X-first restatement:

  * ONE flat f32 buffer each for parameters, gradients and the two Adam moments.  Every nn.Parameter is a view into the first, every
    `.grad` a view into the second, laid out in the order gradients become READY (CTC head, after_norm, last block ... first block,
    front-end), so a bucket is a contiguous slice and "bucket ready" is a counter.
  * buckets (25 MB like DDP's default: 6 all-reduces for the 139 MB of encoder + CTC gradients; xGMI is 7 point-to-point links per GPU,
    and RCCL's ring is per-link bound, so fewer, larger messages than NVSwitch-tuned 1-5 MB buckets) are all-reduced with
    torch.distributed (backend "nccl" = RCCL) as soon as their last gradient has been accumulated -- post-accumulate-grad hooks fire from
    the autograd thread because the Functions in cfm/autograd.py RETURN their gradients -- asynchronously, on RCCL's stream, overlapped
    with the rest of the backward; nothing is reduced for the first accum_grad-1 micro-batches (no_sync).
  * after the last bucket: sum of squares of the flat gradient (one kernel), clip coefficient computed ON THE DEVICE, fused
    clip-scale + average + Adam over the flat buffers (one kernel, cfm_adam_step), gradient buffer zeroed.  No host synchronisation
    anywhere in the step; the learning rate is host scalar math (WarmupLR, src/scheduler.py:36-52).
  * replicas start EQUAL: the constructor broadcasts rank 0's flat parameter buffer and every module buffer (BatchNorm running
    statistics, num_batches_tracked) -- DDP's construction-time `_sync_module_states`, which the reference gets from Lightning's
    DDPStrategy (src/executor.py:137-139,153: one process per GPU, each building its own model at :60-100) -- and then asserts, with
    a MIN/MAX all-reduce of a checksum, that every rank holds the same bits.  Ranks seeded differently, or a checkpoint loaded on
    rank 0 only, therefore cannot diverge silently.
  * BatchNorm batch statistics are per rank (the reference uses no SyncBatchNorm: SURVEY quirk Q6); the RUNNING statistics follow
    DDP's default `broadcast_buffers=True`: float buffers live in one flat buffer that rank 0 broadcasts before the first forward of
    every optimizer step (DDP syncs buffers in the forward that follows a gradient sync: once per optimizer step under accumulation),
    so every rank evaluates / checkpoints rank 0's statistics, as under Lightning.  `broadcast_buffers=False` keeps them per rank.
  * optional 16-bit gradient buckets (`grad_comm_dtype=torch.bfloat16`): a bucket is rounded to bf16 into a staging buffer, all-reduced
    (half the xGMI bytes: 69.5 MB instead of 139 MB per step at config 3) and widened back into the f32 gradient buffer; the sum over
    ranks is then rounded per hop by the collective, so it is opt-in and tested against the f32 path with a stated tolerance.

The elementwise kernels come from a small `kernels` object (default: the HIP ones, no fallback); the gloo/CPU tests of the bucket /
accumulate / clip logic inject a torch implementation from tests/.
"""
import math

import functools

import torch
import torch.distributed as dist


class HipStepKernels:
    """sum of squares and the fused Adam step on the MI355X (csrc/train.hip)."""

    def sumsq(self, flat):
        import cfm
        return cfm.sumsq(flat)

    def adam_step(self, p, g, m, v, lr, betas, eps, weight_decay, step, grad_scale):
        import cfm
        cfm.adam_step(p, g, m, v, lr, betas, eps, weight_decay, step, grad_scale=grad_scale)

    def adam_clip_step(self, p, g, m, v, lr, betas, eps, weight_decay, step, clip, inv_world):
        """sum of squares, then ONE launch: clip coefficient from it, 1/world averaging, Adam, gradient buffer zeroed.  Returns the norm."""
        import cfm
        return cfm.adam_clip_step(p, g, m, v, lr, betas, eps, weight_decay, step, cfm.sumsq(g) if clip else None, clip, inv_world, zero_grad=True)

    def weights_changed(self):
        from cfm import packing
        packing.bump_epoch()               # the flat update bypasses torch's version counters: invalidate the packed weights


def warmup_lr(base_lr, warmup_steps, step_num):
    """WarmupLR of the reference (src/scheduler.py:36-52): peak = base_lr at step_num == warmup_steps."""
    if warmup_steps == 0:
        return base_lr * step_num ** -0.5
    return base_lr * warmup_steps ** 0.5 * min(step_num ** -0.5, step_num * warmup_steps ** -1.5)


class DataParallelTrainer:
    def __init__(self, modules, loss_fn, lr=1e-3, warmup_steps=25000, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, accum_grad=2,
                 grad_clip=4.0, bucket_mb=25.0, process_group=None, kernels=None, always_reduce=False, broadcast_buffers=True,
                 grad_comm_dtype=None, window_loss_fn=None):
        """modules: nn.Modules in FORWARD order (their parameters are flattened in reverse); loss_fn(batch) -> scalar loss tensor.
        always_reduce: issue the collectives even with one rank (lets a 1-GPU box rehearse the RCCL path).
        broadcast_buffers: rank 0's float buffers overwrite every rank's before each optimizer step's first forward (DDP's default).
        grad_comm_dtype: None (f32 all-reduce) or torch.bfloat16 / torch.float16 (16-bit bucket payloads).
        window_loss_fn(micro_batches) -> [loss per micro-batch]: the whole accumulation window in ONE forward / backward (the weights do not
        change between the micro-batches of a window, so their rows can share every row-local launch: ConformerEncoder.forward_window); the
        gradients are the same sums the micro-batch loop accumulates, BatchNorm statistics stay per micro-batch.  Without it: loss_fn per
        micro-batch, `no_sync` on all but the last, as the reference's Lightning loop does."""
        self.modules = list(modules)
        self.loss_fn = loss_fn
        self.window_loss_fn = window_loss_fn
        self.base_lr, self.warmup_steps, self.betas, self.eps, self.weight_decay = lr, warmup_steps, betas, eps, weight_decay
        self.accum_grad, self.grad_clip = int(accum_grad), grad_clip
        self.kernels = kernels if kernels is not None else HipStepKernels()
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.always_reduce = bool(always_reduce) and dist.is_available() and dist.is_initialized()
        self.step_count = 0
        self.broadcast_buffers = bool(broadcast_buffers)
        if grad_comm_dtype not in (None, torch.float32, torch.bfloat16, torch.float16):
            raise ValueError("grad_comm_dtype must be None, float32, bfloat16 or float16")
        self.grad_comm_dtype = None if grad_comm_dtype in (None, torch.float32) else grad_comm_dtype

        forward_order = []
        seen = set()
        for m in self.modules:
            for p in m.parameters():
                if p.requires_grad and id(p) not in seen:
                    seen.add(id(p))
                    forward_order.append(p)
        self.params = forward_order[::-1]                                  # gradient-ready order
        if not self.params:
            raise ValueError("DataParallelTrainer: no trainable parameters")
        dev = self.params[0].device
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("DataParallelTrainer: parameters must be float32 on one device")
        # 16-byte aligned offsets so the vectorised kernels and the collectives see aligned slices
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                self.flat_p[o:o + p.numel()].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_g[o:o + p.numel()].view(p.shape)
        self._flatten_buffers(dev)
        self.sync_module_states()
        self.kernels.weights_changed()

        # Autograd UNITS.  A conformer block whose backward writes all its parameter gradients into one contiguous slab (cfm/autograd.py
        # EncoderLayerFn on the composite path) takes part as ONE leaf -- a view of the flat parameter buffer over the block's range, its
        # .grad the same range of the flat gradient buffer: one AccumulateGrad (one add over the slab) and one hook per block instead of
        # ~33.  Every other parameter is its own unit.
        off_of = {id(p): o for p, o in zip(self.params, offs)}
        in_block = {}
        self.leaves = []
        for m in self.modules:
            for blk in m.modules():
                if not (hasattr(blk, "train_forward") and hasattr(blk, "_weights")):
                    continue
                ps = [p for p in blk.parameters() if p.requires_grad]
                if not ps or any(id(p) not in off_of for p in ps) or len(ps) != len(list(blk.parameters())):
                    continue
                lo = min(off_of[id(p)] for p in ps)
                hi = max(off_of[id(p)] + (p.numel() + 3) // 4 * 4 for p in ps)
                if sum((p.numel() + 3) // 4 * 4 for p in ps) != hi - lo:
                    continue                                                # not contiguous in the flat order (shared parameters): stay per-parameter
                leaf = self.flat_p[lo:hi].detach().requires_grad_(True)
                leaf.grad = self.flat_g[lo:hi]
                blk.__dict__["_flat_leaf"] = leaf
                blk.__dict__["_flat_grad_offsets"] = {n: off_of[id(p)] - lo for n, p in blk.named_parameters()}
                # gradient sink: the block's backward adds into this range directly and then calls the ready hook itself (cfm/autograd.py)
                blk.__dict__["_flat_grad_sink"] = (self.flat_g[lo:hi], functools.partial(self._on_grad, leaf))
                self.leaves.append(leaf)
                for p in ps:
                    in_block[id(p)] = leaf
        units, seen_leaf = [], set()                                        # (tensor to hook, start, end) in ready order
        for p, o in zip(self.params, offs):
            leaf = in_block.get(id(p))
            if leaf is None:
                units.append((p, o, o + (p.numel() + 3) // 4 * 4))
            elif id(leaf) not in seen_leaf:
                seen_leaf.add(id(leaf))
                lo = leaf.data_ptr() - self.flat_p.data_ptr()
                units.append((leaf, lo // 4, lo // 4 + leaf.numel()))
        # buckets: contiguous runs of units (ready order) of at least bucket_mb
        limit = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []                                                   # (start, end, n_units)
        start, count = 0, 0
        self._bucket_of = {}
        for i, (t, lo, hi) in enumerate(units):
            self._bucket_of[id(t)] = len(self.buckets)
            count += 1
            if hi - start >= limit or i == len(units) - 1:
                self.buckets.append((start, hi, count))
                start, count = hi, 0
        self._ready = [0] * len(self.buckets)
        self._next = 0
        self._works = []
        self._stage, self._tmpbuf = None, None
        self._sync = False
        self.reduce_log = []                                                # bucket indices in launch order of the last step (tests)
        for t, _, _ in units:
            t.register_post_accumulate_grad_hook(self._on_grad)

    # ------------------------------------------------------------------------------------------------------------
    def _flatten_buffers(self, dev):
        """Float buffers (BatchNorm running_mean / running_var) become views into ONE flat buffer -- one broadcast per optimizer step instead
        of one per tensor; integer buffers (num_batches_tracked) advance identically on every rank and are broadcast at construction only."""
        fl, other, seen = [], [], set()
        for m in self.modules:
            for b in m.buffers():
                if id(b) in seen or b.numel() == 0:
                    continue
                seen.add(id(b))
                (fl if (b.dtype == torch.float32 and b.device == dev) else other).append(b)
        self.float_buffers, self.other_buffers = fl, other
        n = sum((b.numel() + 3) // 4 * 4 for b in fl)
        self.flat_b = torch.zeros(n, dtype=torch.float32, device=dev)
        o = 0
        with torch.no_grad():
            for b in fl:
                self.flat_b[o:o + b.numel()].copy_(b.reshape(-1))
                b.data = self.flat_b[o:o + b.numel()].view(b.shape)       # in-place updates (the BatchNorm kernels) land in the flat buffer
                o += (b.numel() + 3) // 4 * 4

    def _collective(self):
        return self.world > 1 or self.always_reduce

    def sync_module_states(self):
        """Rank 0's parameters and buffers overwrite every rank's (DDP's `_sync_module_states` at construction; executor.py:137-139), then
        every rank checks that the job holds ONE model: a bit-level checksum of the flat parameter and buffer images, MIN == MAX over ranks."""
        if not self._collective():
            return
        dist.broadcast(self.flat_p, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
        self._broadcast_buffers(initial=True)
        self.assert_replicas_equal()

    def _broadcast_buffers(self, initial=False):
        src = dist.get_global_rank(self.pg, 0) if self.pg is not None else 0
        if self.flat_b.numel():
            dist.broadcast(self.flat_b, src=src, group=self.pg)
        if initial:
            for b in self.other_buffers:
                dist.broadcast(b, src=src, group=self.pg)

    def replica_checksum(self):
        """4 int64 words over the BITS of the flat parameter and float-buffer images (sum and position-weighted sum of the int32 view)."""
        words = []
        for t in (self.flat_p, self.flat_b):
            if t.numel() == 0:
                words += [torch.zeros((), dtype=torch.int64, device=t.device)] * 2
                continue
            bits = t.view(torch.int32).to(torch.int64)
            idx = torch.arange(1, bits.numel() + 1, dtype=torch.int64, device=t.device) % 65521
            words += [bits.sum(), (bits * idx).sum()]
        return torch.stack(words)

    def assert_replicas_equal(self):
        """Raises on EVERY rank (the MIN / MAX all-reduces are collectives all ranks see) when any rank's parameters or float buffers differ."""
        if not self._collective():
            return
        c = self.replica_checksum()
        lo, hi = c.clone(), c.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        if not torch.equal(lo, hi):
            raise RuntimeError("DataParallelTrainer: replicas hold different parameters / buffers (checksum min %s max %s)"
                               % (lo.tolist(), hi.tolist()))

    # ------------------------------------------------------------------------------------------------------------
    def _on_grad(self, p):
        if not self._sync:
            return
        b = self._bucket_of[id(p)]
        self._ready[b] += 1
        # buckets are launched strictly in index order (a collective must be issued in the same order on every rank, whatever order the
        # autograd engine happens to run the accumulation hooks of one Function's many parameters in)
        while self._next < len(self.buckets) and self._ready[self._next] == self.buckets[self._next][2]:
            self._launch(self._next)
            self._next += 1

    def _launch(self, b):
        self.reduce_log.append(b)
        if self._collective():
            s, e, _ = self.buckets[b]
            if self.grad_comm_dtype is None:
                self._works.append((dist.all_reduce(self.flat_g[s:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), None))
            else:
                # 16-bit payload: round the bucket into its staging slice, reduce that, widen back after the wait.  Pre-divided by the world
                # size so the running sum stays in range of fp16 as well (finish() then skips its own division)
                if self._stage is None:
                    self._stage = torch.empty(self.numel, dtype=self.grad_comm_dtype, device=self.flat_g.device)
                st = self._stage[s:e]
                torch.mul(self.flat_g[s:e], 1.0 / self.world, out=self._tmp(e - s))
                st.copy_(self._tmp(e - s))
                self._works.append((dist.all_reduce(st, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), (s, e, st)))

    def _tmp(self, n):
        if self._tmpbuf is None or self._tmpbuf.numel() < n:
            self._tmpbuf = torch.empty(max(e - s for s, e, _ in self.buckets), dtype=torch.float32, device=self.flat_g.device)
        return self._tmpbuf[:n]

    # ------------------------------------------------------------------------------------------------------------
    def lr(self):
        return warmup_lr(self.base_lr, self.warmup_steps, self.step_count + 1)

    def step(self, micro_batches):
        """One optimizer step over accum_grad micro-batches.  Returns the mean loss (a device tensor; no sync)."""
        if len(micro_batches) != self.accum_grad:
            raise ValueError("step() wants %d micro-batches (accum_grad), got %d" % (self.accum_grad, len(micro_batches)))
        total = None
        self.reduce_log = []
        if self.broadcast_buffers and self._collective():
            self._broadcast_buffers()                                       # DDP broadcast_buffers=True: rank 0's running statistics
        if self.window_loss_fn is not None:
            self._sync = True                                              # one backward for the window: every bucket is reduced as it completes
            self._ready = [0] * len(self.buckets)
            self._next = 0
            losses = self.window_loss_fn(micro_batches)
            if len(losses) != self.accum_grad:
                raise ValueError("window_loss_fn returned %d losses for %d micro-batches" % (len(losses), self.accum_grad))
            total = losses[0]
            for l in losses[1:]:
                total = total + l
            (total / self.accum_grad).backward()
            total = total.detach()
        else:
            for i, mb in enumerate(micro_batches):
                self._sync = i == self.accum_grad - 1                      # no_sync on all but the last micro-batch
                self._ready = [0] * len(self.buckets)
                self._next = 0
                loss = self.loss_fn(mb)
                (loss / self.accum_grad).backward()
                total = loss.detach() if total is None else total + loss.detach()
        self._sync = False
        while self._next < len(self.buckets):                               # parameters that took no part in this graph: still reduce
            self._launch(self._next)
            self._next += 1
        for w, widen in self._works:
            w.wait()                                                        # the current stream waits for RCCL's
            if widen is not None:
                s, e, st = widen
                self.flat_g[s:e].copy_(st)
        self._works = []
        self.finish()
        return total / self.accum_grad

    def finish(self):
        """clip by the global norm of the AVERAGED gradient (executor.py:150), Adam, zero the gradient buffer."""
        inv_world = 1.0 / self.world if self.grad_comm_dtype is None else 1.0    # 16-bit buckets arrive averaged
        if hasattr(self.kernels, "adam_clip_step"):                        # the HIP kernels: scalar glue and the zero-fill inside the Adam launch
            self.step_count += 1
            clip = self.grad_clip if (self.grad_clip is not None and self.grad_clip > 0) else 0.0
            norm = self.kernels.adam_clip_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, warmup_lr(self.base_lr, self.warmup_steps, self.step_count),
                                               self.betas, self.eps, self.weight_decay, self.step_count, clip, inv_world)
            if clip:
                self.last_grad_norm = norm
            self.kernels.weights_changed()
            return
        scale = None
        if self.grad_clip is not None and self.grad_clip > 0:
            norm = self.kernels.sumsq(self.flat_g).sqrt() * inv_world       # 1-element device tensors: no sync
            scale = (self.grad_clip / (norm + 1e-6)).clamp(max=1.0) * inv_world
            self.last_grad_norm = norm
        elif inv_world != 1.0:
            scale = torch.full((1,), inv_world, dtype=torch.float32, device=self.flat_g.device)
        self.step_count += 1
        self.kernels.adam_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, warmup_lr(self.base_lr, self.warmup_steps, self.step_count),
                               self.betas, self.eps, self.weight_decay, self.step_count, scale)
        self.flat_g.zero_()
        self.kernels.weights_changed()


# ----------------------------------------------------------------------------------------------------------------------
# synthetic "LibriSpeech-shaped" batches: what the reference's dynamic batcher hands the model (processor.py:267-316)
# ----------------------------------------------------------------------------------------------------------------------
def librispeech_shaped_batch(rs, max_frames_in_batch=8000, min_len=200, max_len=1650, vocab=5002, feat_dim=80):
    """One dynamic batch: utterance lengths ~ U{min_len..max_len} added while B * T_max <= max_frames_in_batch (data_config.json:39,15),
    sorted by length descending (processor.py:295), labels in [2, vocab-2] of length ~T/30 clipped to [1, 200] padded with 0
    (data_config.json:17, processor.py:305-308).  rs: numpy RandomState (seed 1234 + rank).  Returns numpy arrays."""
    import numpy as np
    lens = []
    while True:
        L = int(rs.randint(min_len, max_len + 1))
        if (len(lens) + 1) * max(lens + [L]) > max_frames_in_batch:
            break
        lens.append(L)
    lens = sorted(lens, reverse=True)
    B, T = len(lens), lens[0]
    feats = rs.standard_normal((B, T, feat_dim)).astype(np.float32)
    for b, L in enumerate(lens):
        feats[b, L:] = 0.0
    label_lens = np.array([min(200, max(1, L // 30)) for L in lens], dtype=np.int64)
    labels = np.zeros((B, int(label_lens.max())), dtype=np.int64)
    for b in range(B):
        labels[b, :label_lens[b]] = rs.randint(2, vocab - 1, size=label_lens[b])
    return feats, np.array(lens, dtype=np.int32), labels, label_lens
