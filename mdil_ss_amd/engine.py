"""Optimizer + data-parallel step engines of the incremental-training hot loops (step 1, step 2
and the two-old-domain step 3).

* ``FlatAdam``: the reference's ``Adam(grouped_parameters, 5e-4, (0.9, 0.999), eps=1e-8,
  weight_decay=1e-4)`` (train_new_task_step2.py:229-239) with every group's parameters, gradients
  and both moments re-homed into ONE flat fp32 buffer each, so an optimizer step is one fused HIP
  launch per learning-rate group (instead of 278 per-tensor updates) and the gradient exchange is
  a collective on a contiguous buffer.
* ``poly_factor``: the LambdaLR rule of :244-245 (``scheduler.step(epoch)`` closed form).
* ``Step2Engine``: one hot-loop iteration (:285-306) -- 2 student forwards (train mode) + frozen
  teacher forward (eval) + CE + lambda*KLD + backward + all-reduce + Adam -- with one process per
  GPU and RCCL all-reduce over xGMI replacing nn.DataParallel (:474-475).  The backward is issued
  as two passes (CE graph, then KD graph): the parameters only the CE graph touches (new decoder,
  new-domain adapters / BN) are final after the first pass and their bucket is all-reduced on a
  side stream while the KD graph's backward runs.  Sums are commutative, so the gradients are
  bit-identical to a single ``total.backward()``.
"""
import torch
import torch.distributed as dist

from . import ops


def poly_factor(epoch: int, num_epochs: int) -> float:
    return pow(1 - ((epoch - 1) / num_epochs), 0.9)


class FlatAdam:
    def __init__(self, groups, lr=5e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4):
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        self.param_groups = []
        params = []
        for g in groups:
            ps = [p for p in g["params"] if p.requires_grad]
            glr = g.get("lr", lr)
            self.param_groups.append({"params": ps, "lr": glr, "initial_lr": glr, "step": 0})
            params += ps
        assert params, "FlatAdam: no trainable parameters"
        dev = params[0].device
        total = sum(p.numel() for p in params)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_count = 0
        off = 0
        for g in self.param_groups:
            g["offset"] = off
            for p in g["params"]:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.data.reshape(-1))
                p.data = self.flat_param[off:off + n].view(p.shape)
                p.grad = self.flat_grad[off:off + n].view(p.shape)
                p._mdil_grad_sink = p.grad      # kernels accumulate here directly (ops._sink)
                off += n
            g["numel"] = off - g["offset"]
        ops.sinks_changed()
        ops.invalidate_packs()
        # every engine builds exactly one FlatAdam at construction: drain the device and zero the ticket
        # words of the fused BatchNorm finalizes here, so no engine starts on a stale arrival count
        ops.reset_tickets()

    def zero_grad(self):
        self.flat_grad.zero_()

    def set_epoch(self, epoch: int, num_epochs: int):
        f = poly_factor(epoch, num_epochs)
        for g in self.param_groups:
            g["lr"] = g["initial_lr"] * f

    def step(self, grad_scale: float = 1.0, groups=None):
        """``groups``: indices of the parameter groups to update (default all).  torch.optim.Adam
        keeps a step count per parameter and skips parameters whose .grad is None; a group whose
        parameters received no gradient is left out by the caller, so the count is per group."""
        self.step_count += 1
        for gi, g in enumerate(self.param_groups):
            if groups is not None and gi not in groups:
                continue
            g["step"] += 1
            a, b = g["offset"], g["offset"] + g["numel"]
            ops.adam_step(self.flat_param[a:b], self.flat_grad[a:b], self.exp_avg[a:b],
                          self.exp_avg_sq[a:b], g["step"], g["lr"], self.betas[0],
                          self.betas[1], self.eps, self.weight_decay, grad_scale)
        # one launch re-packs the weight images of what was just updated (selected by storage, not by requires_grad)
        ops.refresh_packs(trainable_only=True, updated=(self.flat_param.data_ptr(), self.flat_param.numel() * 4))

    def state_dict(self):
        """torch.optim.Adam-shaped dict (per-parameter state by running index)."""
        state, idx, groups = {}, 0, []
        for g in self.param_groups:
            off = g["offset"]
            ids = []
            for p in g["params"]:
                n = p.numel()
                state[idx] = {"step": torch.tensor(float(g["step"])),
                              "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                              "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
                ids.append(idx)
                idx += 1
                off += n
            groups.append({"lr": g["lr"], "initial_lr": g["initial_lr"], "betas": self.betas,
                           "eps": self.eps, "weight_decay": self.weight_decay, "amsgrad": False,
                           "params": ids})
        return {"state": state, "param_groups": groups}


def broadcast_replicas(modules, process_group=None, src=0):
    """Make every rank's replica identical to rank ``src``'s: parameters AND buffers.
    ``nn.DataParallel`` (train_new_task_step2.py:474-475) re-broadcasts the rank-0 module before
    every forward; with one process per GPU the replicas are synchronised once, here, and stay
    identical because every rank then applies the same averaged gradient.  Whatever a checkpoint
    does not cover (the new ``decoder.t.output_conv``, everything in step 1) would otherwise start
    from each process's own random init.  No-op without a process group / on one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    src_global = dist.get_global_rank(process_group, src) if process_group is not None else src
    by_type = {}
    for m in modules:
        if m is None:
            continue
        for t in list(m.parameters()) + list(m.buffers()):
            by_type.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for (dtype, device), ts in by_type.items():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src_global, group=process_group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n
    ops.refresh_packs()


def broadcast_buffers(module, process_group=None, src=0):
    """Rank ``src``'s buffers (BN running statistics, ``num_batches_tracked``) -> every rank.
    Running statistics are rank-local during training (each rank normalises with its own batch
    statistics, like a DataParallel replica) and rank 0's are what a checkpoint holds --
    ``nn.DataParallel`` keeps replica 0's buffers and re-broadcasts them before every forward
    (train_new_task_step2.py:474-475).  Every validation pass calls this first, so the model that
    is scored -- on whatever shard of the validation set a rank holds -- is the model rank 0
    saves.  No-op without a process group / on one rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return
    src_global = dist.get_global_rank(process_group, src) if process_group is not None else src
    by_type = {}
    for t in module.buffers():
        by_type.setdefault((t.dtype, t.device), []).append(t)
    with torch.no_grad():
        for ts in by_type.values():
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            dist.broadcast(flat, src=src_global, group=process_group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n


def replica_mask_generator(device, process_group=None, model_index=0):
    """Per-rank generator for the Dropout2d masks: replicas must NOT draw identical masks (each
    DataParallel replica draws its own), whatever the processes' global seeds are.
    ``model_index`` separates the streams of several models that draw masks in one engine (step 3:
    the student and the train-mode previous model) -- identically seeded generators would make the
    KD target's masks replay the student's first draws."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    rank = dist.get_rank(process_group)
    g = torch.Generator(device=device)
    g.manual_seed((torch.initial_seed() + 1000003 * (rank + 1) + 7919 * model_index) % (2 ** 63))
    return g


def global_weighted_ce(ce_local, targets, weight, process_group=None):
    """DataParallel's loss semantics under one-process-per-GPU data parallelism.
    ``nn.DataParallel`` gathers the replicas' logits and takes ONE weighted mean over the whole
    batch (train_new_task_step2.py:285,293): CE = sum_r sum_p w*nll / sum_r sum_p w.  A rank only
    sees its shard's weighted mean CE_r = S_r / W_r, and averaging gradients over ranks gives
    mean_r(CE_r).  Scaling the local loss by world * W_r / sum_r W_r makes the rank-averaged
    gradient equal the gradient of the global weighted mean: (1/world) sum_r world*(W_r/W)*CE_r =
    sum_r S_r / W.  W_r = sum of the class weights of the shard's target pixels (device scalar,
    all-reduced: 4 bytes, no host sync).  Returns the scaled loss (a device scalar)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(process_group) == 1:
        return ce_local
    w_local = weight[targets.reshape(-1)].sum()
    w_all = w_local.clone()
    dist.all_reduce(w_all, op=dist.ReduceOp.SUM, group=process_group)
    return ce_local * (w_local * float(dist.get_world_size(process_group)) / w_all)


class GradExchange:
    """Bucketed SUM all-reduce of slices of the flat gradient buffer.  On GPUs the collective
    (RCCL over xGMI) runs on a side stream, ordered after the work already enqueued on the compute
    stream, so it overlaps whatever backward work is enqueued next; ``join`` makes the compute
    stream wait for it.  On CPU tensors (gloo, used by the tests) it degrades to a synchronous
    all-reduce with identical results."""

    def __init__(self, process_group=None):
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.comm_stream = None

    def start(self, bucket, after=(), pre=None):
        """``after``: events (recorded on other streams) the collective must also wait for;
        ``pre``: work to run on the collective's stream right before it (e.g. adding the second
        graph's share of a bucket), ordered after ``after`` and after the current stream."""
        if self.world == 1:
            if pre is not None:
                for ev in after:
                    torch.cuda.current_stream().wait_event(ev)
                pre()
            return
        if bucket.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = _shared_stream("communication", 0)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for ev in after:
                self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if pre is not None:
                    pre()
                dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg)
        else:
            if pre is not None:
                pre()
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.pg)

    def join(self):
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)


class Step1Engine:
    """First-domain training (train_RAPFT_step1.py:260-330): one train-mode forward, weighted CE,
    backward, Adam over every trainable parameter at one learning rate."""

    def __init__(self, model, weight, current_task=0, lr=5e-4, weight_decay=1e-4,
                 process_group=None):
        self.model, self.weight, self.t = model, weight, current_task
        broadcast_replicas([model], process_group)
        model.mask_generator = replica_mask_generator(weight.device, process_group)
        self.optimizer = FlatAdam([{"params": list(model.parameters())}], lr, (0.9, 0.999), 1e-8,
                                  weight_decay)
        self.exchange = GradExchange(process_group)
        self.world = self.exchange.world

    def iteration(self, images, targets):
        if not self.model.training:
            self.model.train()
        ce = _ce_of(self, self.model, images, targets, self.t, self.weight)
        self.optimizer.zero_grad()
        _backward(ce)
        self.exchange.start(self.optimizer.flat_grad)
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world)
        return ce.detach()


def _ce_of(eng, model, images, targets, task, weight):
    """Train-mode forward of ``model`` for ``task`` + weighted cross entropy.  With the fused head
    (default; ops.HEAD_FUSE) the logits are never materialised (ops.head_ce) unless the engine's
    ``want_logits`` is set (--iouTrain reads ``last_outputs``); otherwise forward() +
    ops.cross_entropy2d.  -> loss (device scalar); ``eng.last_outputs`` = logits or None."""
    if ops.HEAD_FUSE:
        feat = model.features(images, task)
        want = getattr(eng, "want_logits", False)
        out = ops.head_ce(feat, *model.head_params(task), targets[:, 0], weight, want)
        ce, eng.last_outputs = (out if want else (out, None))
        return ce
    outputs = model(images, task)
    eng.last_outputs = outputs.detach()
    return ops.cross_entropy2d(outputs, targets[:, 0], weight)


# ROCm multiplexes HIP streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default).  The first
# engine of a process gets a queue per stream; the streams a SECOND engine creates (step 1 -> 2 -> 3 chained in
# one process, test suites) land on queues that are already taken and two of its "concurrent" streams end up
# behind one another: measured 22.8 -> 30.9 ms per step-2 iteration for the second engine of a process.
# Engines therefore share one set of streams per device and role (they run one after the other).
_SHARED_STREAMS = {}


def _shared_stream(role, priority=0):
    key = (torch.cuda.current_device(), role, priority)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(priority=priority)
    return st


def _set_stream(st):
    """torch.cuda.set_stream without its Python layers."""
    torch._C._cuda_setStream(stream_id=st.stream_id, device_index=st.device_index,
                             device_type=st.device_type)


def _backward(loss, streams=()):
    """``loss.backward()`` with the weight-gradient reductions of the factorised blocks batched
    (ops.DEFER_WGRAD): ~140 latency-bound 8 us launches between chip-filling MFMA kernels become
    one launch per 16.  ``streams``: the side streams the graph's nodes ran on besides the current
    one; every stream's queue is flushed before this returns, so the flat gradient buffer is
    complete once the caller has joined those streams."""
    ops.DEFER_WGRAD = True
    try:
        loss.backward()
    finally:
        ops.DEFER_WGRAD = False
    if ops.ASYNC_WGRAD:
        # weight-gradient launches handed to companion side streams: whatever the caller does next with
        # the gradients (a bucket exchange, the optimizer) must be ordered behind them.  The
        # single-stream iteration had no such join: with warm caches its Adam step overtook the last
        # launches (an order-dependent test failure found in round 4).
        ops.join_side_streams(torch.cuda.current_stream())
    ops.flush_wgrad()
    if streams:
        cur = torch.cuda.current_stream()
        try:
            for st in streams:
                _set_stream(st)
                ops.flush_wgrad()
        finally:
            _set_stream(cur)
    assert not ops.pending_wgrad(), "weight-gradient reductions left on a stream nobody flushed"


class Step2Engine:
    """Owns student / teacher, the criterion and the optimizer for the CS->BDD style step."""

    def __init__(self, student, teacher, weight, current_task=1, lambdac=0.1, lr=5e-4,
                 shared_lr=5e-6, weight_decay=1e-4, is_shared=None, is_ds_curr=None,
                 process_group=None, async_wgrad=False, streams=True, global_ce=False):
        self.async_wgrad = async_wgrad
        ops.ASYNC_WGRAD = bool(async_wgrad)     # from the FIRST iteration on (it runs before enable_streams)
        self.global_ce = global_ce      # DataParallel's global weighted mean (see global_weighted_ce)
        self.want_streams = streams
        # three-stream schedule: plan step after which the old-domain graph starts (None: lock step)
        _sg = __import__("os").environ.get("MDIL_STAGGER", "8")
        self.stagger = None if _sg in ("", "off", "lockstep") else int(_sg)
        self.iterations = 0
        self.student, self.teacher = student, teacher
        self.t = current_task
        self.lambdac = lambdac
        self.weight = weight
        broadcast_replicas([student, teacher], process_group)
        student.mask_generator = replica_mask_generator(weight.device, process_group)
        named = [("module." + n, p) for n, p in student.named_parameters()]
        self.optimizer = FlatAdam(
            [{"params": [p for n, p in named if is_shared(n)], "lr": shared_lr},
             {"params": [p for n, p in named if is_ds_curr(n)]}], lr, (0.9, 0.999), 1e-8,
            weight_decay)
        self.exchange = GradExchange(process_group)
        self.world = self.exchange.world
        g1 = self.optimizer.param_groups[1]
        g0 = self.optimizer.param_groups[0]
        fg = self.optimizer.flat_grad
        self.bucket_ds = fg[g1["offset"]:g1["offset"] + g1["numel"]]
        self.bucket_shared = fg[g0["offset"]:g0["offset"] + g0["numel"]]
        # the new decoder's parameters are the tail of the DS group (named_parameters order:
        # encoder.* before decoder.*); their gradients are final as soon as the backward has
        # crossed the decoder, long before the encoder's -> own bucket, reduced under the rest
        ds = [(n, p) for n, p in named if is_ds_curr(n) and p.requires_grad]
        n_dec = sum(p.numel() for n, p in ds if "decoder" in n)
        first_dec = next((i for i, (n, _) in enumerate(ds) if "decoder" in n), len(ds))
        assert all("decoder" in n for n, _ in ds[first_dec:]), "decoder parameters must be trailing"
        end = g1["offset"] + g1["numel"]
        self.bucket_dec = fg[end - n_dec:end]
        self.bucket_ds_enc = fg[g1["offset"]:end - n_dec]
        # The shared-encoder bucket (1,868,252 floats = 7.5 MB, 80 % of the exchanged bytes) is cut by
        # encoder depth.  named_parameters order is stem -> deep and the backward runs deep -> stem,
        # so the TAIL of the bucket is final first: stage 0 = encoder.layers.11-14, stage 1 =
        # layers.7-10 (the eight C=128 blocks: 2 x 3.15 MB), the rest (stem, C=64 blocks, both
        # downsamplers: 1.17 MB) goes out when the backward has drained.  A stage's all-reduce
        # starts, on the communication stream, as soon as BOTH student graphs have passed it.
        def layer_of(n):
            return int(n.split("encoder.layers.")[1].split(".")[0]) if "encoder.layers." in n else -1
        sh = [(n, p) for n, p in named if is_shared(n) and p.requires_grad]
        assert all(layer_of(a[0]) <= layer_of(b[0]) for a, b in zip(sh, sh[1:])), "shared group not in depth order"
        cut = lambda first: sum(p.numel() for n, p in sh if layer_of(n) < first)
        c7, c11, c_end = cut(7), cut(11), g0["numel"]
        # (first plan step of the stage -- its INPUT activation carries the hook --, slice of the bucket)
        self.shared_stages = [(12, (c11, c_end)), (8, (c7, c11))] if c_end > c11 > c7 > 0 else []
        if self.async_wgrad and __import__("os").environ.get("MDIL_ASYNC_STAGES") is None:
            # weight gradients on companion side streams: the whole shared bucket goes out after the
            # backward has drained and every side stream has been joined
            self.shared_stages = []
        self.shared_rest = (0, c7 if self.shared_stages else c_end)

    # ------------------------------------------------------------------------------------------
    # three-stream schedule: the new-task graph, the old-task (KD) graph and the frozen teacher
    # are independent until the losses, so their kernels are enqueued on separate HIP streams.
    # Each conv kernel has an HBM-bound head (first touch of its input) and tail (output store)
    # around an MFMA-bound body, and every BN / elementwise kernel is purely HBM-bound: with two
    # graphs in flight the hardware fills one graph's HBM phases with the other's MFMA phases.
    # Gradients of the shared encoder are accumulated into two separate flat buffers (one per
    # graph) and summed once, so the result does not depend on the interleaving.
    # ------------------------------------------------------------------------------------------
    def enable_streams(self):
        opt = self.optimizer
        g0 = opt.param_groups[0]
        self.flat_grad2 = torch.zeros(g0["numel"], dtype=torch.float32, device=opt.flat_grad.device)
        off = 0
        for p in g0["params"]:
            n = p.numel()
            p._mdil_grad_sink2 = self.flat_grad2[off:off + n].view(p.shape)
            off += n
        ops.sinks_changed()
        # the two student graphs (forward AND backward: the step's critical path) on high-priority
        # HIP streams, the frozen model's forward-only graph on a normal one: +0.5 % measured
        import os
        pr = [int(v) for v in os.environ.get("MDIL_STREAM_PRIO", "-1,-1,0").split(",")]
        self.s_new, self.s_old = _shared_stream("graph a", pr[0]), _shared_stream("graph b", pr[1])
        if os.environ.get("MDIL_STUDENT_ONE_STREAM") is not None:
            # measurement switch (profiles/r06_experiments.txt #P2): both student graphs on ONE stream beside the frozen
            # model's -- the stream structure a launch-paired student would have
            self.s_old = self.s_new
        self.s_t = _shared_stream("frozen", pr[2])
        self.multi_stream = True
        self.graph = None
        ops.ASYNC_WGRAD = self.async_wgrad

    def _teacher_forward(self, images):
        """Frozen previous-step model on ``images`` on its own stream -> NHWC logits."""
        main = torch.cuda.current_stream()
        self.s_t.wait_stream(main)
        with torch.cuda.stream(self.s_t), torch.no_grad():
            y = ops.to_nhwc(images)
            images.record_stream(self.s_t)
            ops.SINK_SLOT = 0
            for f in self.teacher.plan(self.t - 1, head=not ops.HEAD_FUSE):
                y = f(y)
        return y

    def _fwd_bwd_streams(self, images, targets, next_images=None):
        """Forward x3 + losses + backward of both graphs, forked over three streams and joined back
        on the current stream with the summed gradients in the flat buffer.  -> (ce, kld).

        Two schedules, bit-identical in their results (every stream runs the same launches in the
        same order; the shared-encoder gradients of the two graphs land in separate flat buffers
        and are summed once):

        * ``self.stagger is None`` -- LOCK STEP (rounds 1-3): the three forwards advance block by
          block together and ONE ``total.backward()`` replays both student graphs, each node on
          its forward stream.  All three graphs then sit in the same layer at the same time: the
          HBM-bound ends of the network (stem / down-samplers, the 16-channel decoder blocks at
          256 x 512, heads) of all graphs coincide -- 5.2 ms of a 24 ms step with no matrix-pipe
          kernel resident (tools/timeline.py on the round-4 trace) -- and so do their MFMA-bound
          middles, with HBM idle.
        * ``self.stagger = k`` -- STAGGERED (round 4): the old-domain graph starts its forward when
          the new-domain graph has finished plan step k, and EACH graph's backward follows its own
          loss on its own stream (``ce.backward()`` does not wait for the KD graph): one graph's
          HBM-bound phases now run beside the other's MFMA-bound phases.  The frozen model's
          forward for the NEXT batch (``next_images``) is released on the third stream when the
          new-domain graph's backward has drained, so it fills the end of this step and the head
          of the next, where a student graph would otherwise run alone."""
        s, t = self.student, self.t
        if not s.training:
            s.train()
        if self.teacher.training:
            self.teacher.eval()
        main = torch.cuda.current_stream()
        self.optimizer.zero_grad()
        self.flat_grad2.zero_()
        x = ops.to_nhwc(images)           # NHWC, shared by all three
        n = x.shape[0]
        masks_new = s.draw_masks(n, x.device)
        masks_old = s.draw_masks(n, x.device)
        pre, self._teacher_pre = getattr(self, "_teacher_pre", None), None
        fuse = ops.HEAD_FUSE           # plans stop at the decoder features; output_conv rides in the loss
        plans = [(self.s_new, s.plan(t, masks_new, head=not fuse), 0, True),
                 (self.s_old, s.plan(t - 1, masks_old, head=not fuse), 1, True)]
        ys = [x, x]
        if pre is not None and pre[0] is images:
            y_teacher = pre[1]                                        # computed during the last backward
        else:
            plans.append((self.s_t, self.teacher.plan(t - 1, head=not fuse), 0, False))
            ys.append(x)
            y_teacher = None
        for st, _, _, _ in plans:
            st.wait_stream(main)
            x.record_stream(st)
        n_enc = 1 + len(s.encoder.layers)            # plan steps that belong to the encoder
        self._dec_reduced = False
        self._stage_events = [[None, None] for _ in self.shared_stages]
        self._stages_sent = []
        grad_was = torch.is_grad_enabled()
        capturing = torch.cuda.is_current_stream_capturing()
        nsteps = len(plans[0][1])
        stagger = self.stagger if self.stagger is None else max(0, min(int(self.stagger), nsteps))

        def advance(k, i):
            """plan step i of graph k on its stream (+ the gradient-exchange hooks of that step)"""
            st, plan, slot, grad = plans[k]
            # ~70 stream switches per iteration: the raw setter (1 us) instead of the
            # torch.cuda.stream context manager (~20 us enter + exit)
            _set_stream(st)
            torch._C._set_grad_enabled(grad)
            try:
                ops.SINK_SLOT = slot
                ys[k] = plan[i](ys[k])
            finally:
                _set_stream(main)
                torch._C._set_grad_enabled(grad_was)
            if k > 1 or self.world <= 1 or capturing:
                return
            for si, (first, (a, b)) in enumerate(self.shared_stages):
                if i != first - 1:
                    continue
                # ys[k] is the input of the stage's first block in this student graph: its gradient
                # exists once the graph's backward has passed the stage
                def _stage_done(grad, self=self, si=si, k=k, a=a, b=b):
                    if self.async_wgrad:         # the stage's weight gradients may sit on side streams:
                        ops.join_side_streams(torch.cuda.current_stream())   # order them before the event
                    ops.flush_wgrad()            # this graph's queued weight-gradient reductions
                    ev = torch.cuda.Event()
                    ev.record()
                    st = self._stage_events[si]
                    st[k] = ev
                    if st[0] is not None and st[1] is not None:
                        dst, src = self.bucket_shared[a:b], self.flat_grad2[a:b]
                        self.exchange.start(dst, after=tuple(st), pre=lambda: dst.add_(src))
                        self._stages_sent.append(si)
                ys[k].register_hook(_stage_done)
            if k == 0 and i == n_enc - 1 and self.bucket_dec.numel():
                # fires (on the new-task graph's stream) once the backward has crossed the new
                # decoder: its gradient bucket is all-reduced over xGMI under the encoder backward
                def _dec_done(grad, self=self):
                    if self.async_wgrad:       # the decoder's weight gradients may sit on side streams
                        ops.join_side_streams(torch.cuda.current_stream())
                    ops.flush_wgrad()          # the decoder's queued weight-gradient reductions
                    self.exchange.start(self.bucket_dec)
                    self._dec_reduced = True
                ys[0].register_hook(_dec_done)

        if stagger is None:
            for i in range(nsteps):
                for k in range(len(plans)):
                    advance(k, i)
        else:
            # host order = the order the GPU needs the work in: new-domain steps 0 .. k-1 (with the
            # frozen model beside them), then new-domain step i with old-domain step i - k
            for i in range(nsteps + stagger):
                if i < nsteps:
                    advance(0, i)
                    if len(plans) > 2:
                        advance(2, i)
                if i == stagger - 1:             # the old-domain graph may start now
                    ev = torch.cuda.Event()
                    ev.record(self.s_new)
                    self.s_old.wait_event(ev)
                if i >= stagger:
                    advance(1, i - stagger)
        ops.SINK_SLOT = 0
        if y_teacher is None:
            y_teacher = ys[2]
        with torch.cuda.stream(self.s_new):
            if fuse:
                want = getattr(self, "want_logits", False)
                out = ops.head_ce(ys[0], *s.head_params(t), targets[:, 0], self.weight, want)
                ce, self.last_outputs = (out if want else (out, None))
            else:
                out_new = ys[0].permute(0, 3, 1, 2)
                self.last_outputs = out_new.detach()      # new-task logits (trainer: --iouTrain)
                ce = ops.cross_entropy2d(out_new, targets[:, 0], self.weight)
            if self.global_ce:
                ce = global_weighted_ce(ce, targets[:, 0], self.weight, self.exchange.pg)
            ev_new = None
            if stagger is not None:
                # the root gradient is produced on THIS stream: nothing of the KD graph is waited for
                _backward(ce)
                ev_new = torch.cuda.Event()
                ev_new.record(self.s_new)
        with torch.cuda.stream(self.s_old):
            self.s_old.wait_stream(self.s_t)
            y_teacher.record_stream(self.s_old)
            if fuse:
                kld = ops.head_kld(ys[1], *s.head_params(t - 1), y_teacher, *self.teacher.head_params(t - 1))
            else:
                kld = ops.kld_prob(ys[1].permute(0, 3, 1, 2), y_teacher.permute(0, 3, 1, 2))
            if stagger is not None:
                _backward(self.lambdac * kld)                          # train_new_task_step2.py:301-304
        if stagger is None:
            main.wait_stream(self.s_new)
            main.wait_stream(self.s_old)
            total = ce + self.lambdac * kld                               # train_new_task_step2.py:301
            if next_images is not None and not capturing:
                self._teacher_pre = (next_images, self._teacher_forward(next_images))
            _backward(total, (self.s_new, self.s_old))                    # :304
            main.wait_stream(self.s_t)
        elif next_images is not None and not capturing:
            # released when the new-domain graph's backward has drained (earlier release points -- beside
            # the old-domain forward's last blocks -- measured 3 % slower: profiles/r04_experiments.txt)
            self.s_t.wait_event(ev_new)
            self._teacher_pre = (next_images, self._teacher_forward(next_images))
        else:
            main.wait_stream(self.s_t)
        main.wait_stream(self.s_new)
        main.wait_stream(self.s_old)
        ops.join_side_streams(main)                        # asynchronous weight-gradient launches
        # CE-graph + KD-graph shared gradients: the stages that already went out were summed on the
        # communication stream; what is left is summed here
        sent = set(self._stages_sent)
        self._shared_pending = [self.shared_rest] + [r for si, (_, r) in enumerate(self.shared_stages)
                                                     if si not in sent]
        if not sent:
            self.bucket_shared.add_(self.flat_grad2)
        else:
            for a, b in self._shared_pending:
                self.bucket_shared[a:b].add_(self.flat_grad2[a:b])
        ce, kld = ce.detach(), kld.detach()
        for v in (ce, kld):
            v.record_stream(main)
        return ce, kld

    def _iteration_streams(self, images, targets, next_images=None):
        if self.graph is not None:
            self.static_images.copy_(images, non_blocking=True)
            self.static_targets.copy_(targets, non_blocking=True)
            self.graph.replay()
            ce, kld = self.static_ce, self.static_kld
        else:
            ce, kld = self._fwd_bwd_streams(images, targets, next_images)
        if getattr(self, "_dec_reduced", False) and self.graph is None:
            self.exchange.start(self.bucket_ds_enc)        # decoder bucket went out during backward
        else:
            self.exchange.start(self.bucket_ds)
        if self.graph is None and getattr(self, "_stages_sent", None):
            for a, b in self._shared_pending:              # the deep stages went out during backward
                self.exchange.start(self.bucket_shared[a:b])
        else:
            self.exchange.start(self.bucket_shared)
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world)
        return ce + self.lambdac * kld, ce, kld

    def enable_graph(self, images, targets):
        """Capture forward + losses + backward (about 1,750 launches over three streams) into one
        hipGraph and replay it every iteration: the host then issues ~6 calls per step instead of
        ~1,750, so the GPU never waits for Python.  Requires enable_streams() and at least one
        eager iteration first (weight images and scratch buffers must already exist; captured
        kernels keep their pointers).  The all-reduce and Adam stay eager: their host-side scalars
        (step count, learning rate) change every step."""
        assert getattr(self, "multi_stream", False), "call enable_streams() first"
        self.static_images = images.clone()
        self.static_targets = targets.clone()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        gen = getattr(self.student, "mask_generator", None)
        if gen is not None:
            # the per-rank Dropout2d generator (data parallel) is not torch's default one: a graph
            # only advances the RNG offsets of generators registered with it
            g.register_generator_state(gen)
        with torch.cuda.graph(g):
            ce, kld = self._fwd_bwd_streams(self.static_images, self.static_targets)
        self.static_ce, self.static_kld = ce, kld
        self.graph = g

    def iteration(self, images, targets, next_images=None):
        """-> (total, ce, kld) device scalars (no host sync here).  The very first iteration runs
        on one stream (it creates the packed weight images every stream will read afterwards);
        from the second one on the three-stream schedule is used when ``streams=True``.
        ``next_images``: the images of the following call, if known (same tensor object then):
        lets the frozen model's forward for that batch overlap this batch's backward."""
        self.iterations += 1
        if self.want_streams and self.iterations > 1 and not getattr(self, "multi_stream", False):
            torch.cuda.current_stream().synchronize()
            self.enable_streams()
        if getattr(self, "multi_stream", False):
            return self._iteration_streams(images, targets, next_images)
        s, t = self.student, self.t
        if not s.training:
            s.train()
        if self.teacher.training:
            self.teacher.eval()
        if ops.HEAD_FUSE:
            # output_conv + loss fused: the three logit tensors are never written (csrc/head.hip)
            ce = _ce_of(self, s, images, targets, t, self.weight)
            f_prev = s.features(images, t - 1)
            with torch.no_grad():
                f_teacher = self.teacher.features(images, t - 1)
            kld = ops.head_kld(f_prev, *s.head_params(t - 1), f_teacher, *self.teacher.head_params(t - 1))
        else:
            outputs = s(images, t)
            outputs_prev_task = s(images, t - 1)
            with torch.no_grad():
                outputs_prev_model = self.teacher(images, t - 1)
            self.last_outputs = outputs.detach()
            ce = ops.cross_entropy2d(outputs, targets[:, 0], self.weight)
            kld = ops.kld_prob(outputs_prev_task, outputs_prev_model)
        if self.global_ce:
            ce = global_weighted_ce(ce, targets[:, 0], self.weight, self.exchange.pg)
        self.optimizer.zero_grad()
        _backward(ce)
        self.exchange.start(self.bucket_ds)                # overlaps the KD graph's backward
        _backward(self.lambdac * kld)
        self.exchange.start(self.bucket_shared)
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world)
        return ce.detach() + self.lambdac * kld.detach(), ce.detach(), kld.detach()


class Step3Engine:
    """Third-domain step (train_new_task_step3.py:303-356): every iteration makes TWO optimizer
    steps -- (A) weighted CE on the new domain, (B) lambdac * (KLD(t-1) + KLD(t-2)) against the
    previous model on both old domains.

    Reference behaviour kept, quirks included:
      * the previous model is never switched to eval mode in that file (only ``model.train()``,
        :301): it normalises with batch statistics, keeps updating its own running statistics
        and has dropout active.  ``teacher_train=False`` gives the step-2 style frozen-eval
        teacher instead.
      * ``optimizer.zero_grad()`` (torch>=2: grads -> None) precedes both backwards, so step (B)
        only updates what the KD graphs reach: the shared encoder group.  The domain-specific
        group (new decoder, new-domain BN / adapters) takes one Adam step per iteration, the
        shared group two (per-group step counts in FlatAdam).  ``legacy_zero_grad=True`` restores
        the torch<=1.x behaviour (zeroed grads: the DS group also steps in (B), moved only by
        weight decay and its moments).

    Schedule: the previous model does not depend on the student, so its two forwards are enqueued
    on side streams next to phase (A); in phase (B) the two old-domain student graphs advance in
    lock step on two streams with separate gradient sinks (summed once, fixed order), like
    Step2Engine."""

    def __init__(self, student, teacher, weight, current_task=2, lambdac=0.1, lr=5e-4,
                 shared_lr=5e-6, weight_decay=1e-4, is_shared=None, is_ds_curr=None,
                 process_group=None, streams=True, teacher_train=True, legacy_zero_grad=False):
        self.student, self.teacher, self.weight = student, teacher, weight
        self.t, self.lambdac = current_task, lambdac
        self.want_streams, self.teacher_train = streams, teacher_train
        self.legacy_zero_grad = legacy_zero_grad
        _sg = __import__("os").environ.get("MDIL_STAGGER3", "off")
        self.stagger = None if _sg in ("", "off", "lockstep") else int(_sg)   # phase B, like Step2Engine
        self.iterations = 0
        broadcast_replicas([student, teacher], process_group)
        student.mask_generator = replica_mask_generator(weight.device, process_group)
        teacher.mask_generator = replica_mask_generator(weight.device, process_group, model_index=1)
        named = [("module." + n, p) for n, p in student.named_parameters()]
        self.optimizer = FlatAdam(
            [{"params": [p for n, p in named if is_shared(n)], "lr": shared_lr},
             {"params": [p for n, p in named if is_ds_curr(n)]}], lr, (0.9, 0.999), 1e-8,
            weight_decay)
        self.exchange = GradExchange(process_group)
        self.world = self.exchange.world
        g0, g1 = self.optimizer.param_groups
        fg = self.optimizer.flat_grad
        self.bucket_shared = fg[g0["offset"]:g0["offset"] + g0["numel"]]
        self.bucket_ds = fg[g1["offset"]:g1["offset"] + g1["numel"]]
        self.multi_stream = False

    def _modes(self):
        if not self.student.training:
            self.student.train()
        if self.teacher.training != self.teacher_train:
            self.teacher.train(self.teacher_train)

    def _sync_teacher_stats(self):
        """Running statistics the previous model accumulates in train mode are averaged over the
        ranks (nn.DataParallel computed them on the whole batch on one device; the rank mean of
        per-shard statistics is the data-parallel equivalent).  No-op on one GPU."""
        if self.world == 1 or not self.teacher_train:
            return
        bufs = [b for n, b in self.teacher.named_buffers() if b.dtype.is_floating_point]
        flat = torch.cat([b.reshape(-1) for b in bufs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.exchange.pg)
        flat.div_(self.world)
        off = 0
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view(b.shape))
            off += b.numel()

    def enable_streams(self):
        g0 = self.optimizer.param_groups[0]
        self.flat_grad2 = torch.zeros(g0["numel"], dtype=torch.float32,
                                      device=self.optimizer.flat_grad.device)
        off = 0
        for p in g0["params"]:
            n = p.numel()
            p._mdil_grad_sink2 = self.flat_grad2[off:off + n].view(p.shape)
            off += n
        ops.sinks_changed()
        self.s_a, self.s_b = _shared_stream("graph a", -1), _shared_stream("graph b", -1)
        self.s_t1, self.s_t0 = _shared_stream("frozen", 0), _shared_stream("frozen 2", 0)
        self.multi_stream = True

    # -------------------------------------------------------------------------------- one stream
    def _iteration_single(self, images, targets):
        s, te, t = self.student, self.teacher, self.t
        ce = _ce_of(self, s, images, targets, t, self.weight)
        self.optimizer.zero_grad()
        _backward(ce)
        self.exchange.start(self.optimizer.flat_grad)
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world)
        if ops.HEAD_FUSE:                 # output_conv rides in the loss: no logits in memory
            f1, f0 = s.features(images, t - 1), s.features(images, t - 2)
            with torch.no_grad():
                g1, g0 = te.features(images, t - 1), te.features(images, t - 2)
            k1 = ops.head_kld(f1, *s.head_params(t - 1), g1, *te.head_params(t - 1))
            k0 = ops.head_kld(f0, *s.head_params(t - 2), g0, *te.head_params(t - 2))
        else:
            p1 = s(images, t - 1)
            p0 = s(images, t - 2)
            with torch.no_grad():
                t1 = te(images, t - 1)
                t0 = te(images, t - 2)
            k1, k0 = ops.kld_prob(p1, t1), ops.kld_prob(p0, t0)
        kd = self.lambdac * (k1 + k0)
        self.optimizer.zero_grad()
        _backward(kd)
        self.exchange.start(self.bucket_shared)
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world,
                            groups=None if self.legacy_zero_grad else (0,))
        return ce.detach(), k1.detach(), k0.detach()

    # ------------------------------------------------------------------------------ four streams
    @staticmethod
    def _lockstep(plans, ys):
        main = torch.cuda.current_stream()
        grad_was = torch.is_grad_enabled()
        try:
            for i in range(len(plans[0][1])):
                for k, (st, plan, slot, grad) in enumerate(plans):
                    _set_stream(st)
                    torch._C._set_grad_enabled(grad)
                    ops.SINK_SLOT = slot
                    ys[k] = plan[i](ys[k])
        finally:
            _set_stream(main)
            torch._C._set_grad_enabled(grad_was)
            ops.SINK_SLOT = 0
        return ys

    def _iteration_streams(self, images, targets):
        s, te, t = self.student, self.teacher, self.t
        main = torch.cuda.current_stream()
        x = ops.to_nhwc(images)
        n = x.shape[0]
        streams = (self.s_a, self.s_b, self.s_t1, self.s_t0)
        for st in streams:
            st.wait_stream(main)
            x.record_stream(st)
        # phase A: new-domain graph || both previous-model forwards
        self.optimizer.zero_grad()
        tm1 = te.draw_masks(n, x.device) if self.teacher_train else None
        tm0 = te.draw_masks(n, x.device) if self.teacher_train else None
        fuse = ops.HEAD_FUSE              # plans stop at the decoder features; output_conv rides in the loss
        plans = ((self.s_a, s.plan(t, s.draw_masks(n, x.device), head=not fuse), 0, True),
                 (self.s_t1, te.plan(t - 1, tm1, head=not fuse), 0, False),
                 (self.s_t0, te.plan(t - 2, tm0, head=not fuse), 0, False))
        y_new, y_t1, y_t0 = self._lockstep(plans, [x, x, x])
        with torch.cuda.stream(self.s_a):
            if fuse:
                want = getattr(self, "want_logits", False)
                out = ops.head_ce(y_new, *s.head_params(t), targets[:, 0], self.weight, want)
                ce, self.last_outputs = (out if want else (out, None))
            else:
                self.last_outputs = y_new.detach().permute(0, 3, 1, 2)
                ce = ops.cross_entropy2d(y_new.permute(0, 3, 1, 2), targets[:, 0], self.weight)
        main.wait_stream(self.s_a)
        _backward(ce, (self.s_a,))
        main.wait_stream(self.s_a)
        self.exchange.start(self.optimizer.flat_grad)
        self.exchange.join()
        main.wait_stream(self.s_t1)      # the step re-packs every cached weight image, the previous
        main.wait_stream(self.s_t0)      # model's included: its forwards must have drained
        self.optimizer.step(grad_scale=1.0 / self.world)
        # phase B: the two old-domain student graphs in lock step
        self.optimizer.zero_grad()
        self.flat_grad2.zero_()
        for st in (self.s_a, self.s_b):
            st.wait_stream(main)
        plans = ((self.s_a, s.plan(t - 1, s.draw_masks(n, x.device), head=not fuse), 0, True),
                 (self.s_b, s.plan(t - 2, s.draw_masks(n, x.device), head=not fuse), 1, True))
        stagger = self.stagger
        if stagger is None:
            y_p1, y_p0 = self._lockstep(plans, [x, x])
        else:
            # staggered like Step2Engine: the second old-domain graph starts `stagger` plan
            # steps behind the first and each graph's backward follows its own KD term on its own stream
            nsteps = len(plans[0][1])
            stagger = max(0, min(int(stagger), nsteps))
            ys = [x, x]
            grad_was = torch.is_grad_enabled()
            try:
                for i in range(nsteps + stagger):
                    for k, j in ((0, i), (1, i - stagger)):
                        if not 0 <= j < nsteps:
                            continue
                        st, plan, slot, grad = plans[k]
                        _set_stream(st)
                        torch._C._set_grad_enabled(grad)
                        ops.SINK_SLOT = slot
                        ys[k] = plan[j](ys[k])
                        if k == 0 and j == stagger - 1:
                            ev = torch.cuda.Event()
                            ev.record(self.s_a)
                            self.s_b.wait_event(ev)
            finally:
                _set_stream(main)
                torch._C._set_grad_enabled(grad_was)
                ops.SINK_SLOT = 0
            y_p1, y_p0 = ys

        def kd(y_p, y_t, task):
            if fuse:
                return ops.head_kld(y_p, *s.head_params(task), y_t, *te.head_params(task))
            return ops.kld_prob(y_p.permute(0, 3, 1, 2), y_t.permute(0, 3, 1, 2))
        with torch.cuda.stream(self.s_a):
            self.s_a.wait_stream(self.s_t1)
            y_t1.record_stream(self.s_a)
            k1 = kd(y_p1, y_t1, t - 1)
            if stagger is not None:
                _backward(self.lambdac * k1)         # root gradient on this graph's stream: no join with the other
        with torch.cuda.stream(self.s_b):
            self.s_b.wait_stream(self.s_t0)
            y_t0.record_stream(self.s_b)
            k0 = kd(y_p0, y_t0, t - 2)
            if stagger is not None:
                _backward(self.lambdac * k0)
        main.wait_stream(self.s_a)
        main.wait_stream(self.s_b)
        if stagger is None:
            kd = self.lambdac * (k1 + k0)
            _backward(kd, (self.s_a, self.s_b))
        for st in streams:
            main.wait_stream(st)
        self.bucket_shared.add_(self.flat_grad2)
        self.exchange.start(self.bucket_shared)
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world,
                            groups=None if self.legacy_zero_grad else (0,))
        out = (ce.detach(), k1.detach(), k0.detach())
        for v in out:
            v.record_stream(main)
        return out

    def iteration(self, images, targets):
        """-> (ce, kld_{t-1}, kld_{t-2}) device scalars; two optimizer steps were made."""
        self.iterations += 1
        self._modes()
        if self.want_streams and self.iterations > 1 and not self.multi_stream:
            torch.cuda.current_stream().synchronize()
            self.enable_streams()
        if self.multi_stream:
            out = self._iteration_streams(images, targets)
        else:
            out = self._iteration_single(images, targets)
        self._sync_teacher_stats()
        return out


class MultiTaskEngine:
    """Joint multi-task training (train_multi_task.py:212-265): one shared encoder, one decoder
    head per dataset; the inner loop visits the datasets round-robin and makes one optimizer step
    per dataset.  ``zero_grad()`` (grads -> None) before every backward means a sub-step only
    updates the encoder and the visited head, and torch's Adam keeps a step count per parameter:
    the flat optimizer therefore carries one group per head (the reference's second group split
    by head; same learning rate), stepped individually."""

    def __init__(self, model, weights, lr=5e-4, weight_decay=1e-4, process_group=None):
        self.model, self.weights = model, weights
        broadcast_replicas([model], process_group)
        model.mask_generator = replica_mask_generator(weights[0].device, process_group)
        nb = len(model.decoder)
        named = list(model.named_parameters())
        groups = [{"params": [p for n, p in named if "encoder" in n], "lr": lr / nb}]   # :214
        for i in range(nb):
            groups.append({"params": [p for n, p in named if n.startswith(f"decoder.{i}.")]})
        self.optimizer = FlatAdam(groups, lr, (0.9, 0.999), 1e-8, weight_decay)
        self.exchange = GradExchange(process_group)
        self.world = self.exchange.world
        fg = self.optimizer.flat_grad
        self.buckets = [fg[g["offset"]:g["offset"] + g["numel"]] for g in self.optimizer.param_groups]

    def sub_step(self, ind, images, targets):
        """forward(head ``ind``) + CE + backward + all-reduce + Adam on (encoder, head ``ind``)."""
        if not self.model.training:
            self.model.train()
        ce = _ce_of(self, self.model, images, targets, ind, self.weights[ind])
        self.optimizer.zero_grad()
        _backward(ce)
        self.exchange.start(self.buckets[1 + ind])       # head first: final before the encoder's
        self.exchange.start(self.buckets[0])
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world, groups=(0, 1 + ind))
        return ce.detach()


class FineTuneEngine:
    """Fine-tuning / feature-extraction baselines (main_ftp1_enc_newbn.py:224-244,
    main_FT2_flexible_new.py:215-235): one train-mode forward through ``decoder_new``, weighted CE,
    backward, Adam(5e-4) over encoder + new decoder (``finetune=True``) or the new decoder only
    (feature extraction).  In feature-extraction mode the reference leaves the encoder's
    ``requires_grad`` on but never steps it; its gradients are not computed here (same results).
    The old decoders are frozen in both modes; the encoder's BN statistics keep updating."""

    def __init__(self, model, weight, finetune, forward_new, lr=5e-4, weight_decay=1e-4,
                 process_group=None):
        self.model, self.weight, self.forward_new = model, weight, forward_new
        broadcast_replicas([model], process_group)
        model.mask_generator = replica_mask_generator(weight.device, process_group)
        for n, p in model.named_parameters():
            if n.startswith("decoder_old"):
                p.requires_grad = False
            elif n.startswith("encoder") and not finetune:
                p.requires_grad = False
        params = (list(model.encoder.parameters()) if finetune else []) + \
            list(model.decoder_new.parameters())
        self.optimizer = FlatAdam([{"params": params}], lr, (0.9, 0.999), 1e-8, weight_decay)
        self.exchange = GradExchange(process_group)
        self.world = self.exchange.world

    def iteration(self, images, targets):
        if not self.model.training:
            self.model.train()
        outputs = self.forward_new(images)
        self.last_outputs = outputs.detach()
        ce = ops.cross_entropy2d(outputs, targets[:, 0], self.weight)
        self.optimizer.zero_grad()
        _backward(ce)
        self.exchange.start(self.optimizer.flat_grad)
        self.exchange.join()
        self.optimizer.step(grad_scale=1.0 / self.world)
        return ce.detach()
