"""Training-step driver: what the reference runs between optimizer.zero_grad() and optimizer.step()
(MyTrain_med.py:59-86) as one engine pass — forward, the 4-pair structure loss, backward, element-wise
gradient clamp and Adam — over flat parameter / gradient / moment arenas.

Data parallelism (one process per GPU): gradients are summed with RCCL all-reduce (torch.distributed backend
"nccl") in buckets that are launched as soon as backward has produced every gradient they contain, so the
collective overlaps the remaining backward kernels; the 1/world scaling is folded into the clamp+Adam kernel.
BatchNorm statistics stay per replica, exactly as nn.DataParallel + nn.BatchNorm2d would do in the reference.
"""
import ctypes as C
import os

import torch

from .capi import call, F32
from .engine import Engine, Act, PackCache, GradQueue, StepArena, TUNER, _p, _stream
from .graph import get_compute_dtype
from .dp import GradBuckets
from . import loss as L


def _r4(n):
    return (n + 3) // 4 * 4


# "thread_local": other threads of the process (RCCL's watchdog polls events while a rank captures) do not invalidate the capture
CAPTURE_MODE = os.environ.get("PN2_CAPTURE_MODE", "thread_local")
DP_SEGMENTS = os.environ.get("PN2_DP_SEGMENTS", "1") == "1"          # data-parallel replay: one hipGraph per gradient-bucket boundary, all-reduce overlapped (0: one graph, reduce after it)
# data-parallel replay on RCCL: the bucket all-reduces are captured INTO the step's hipGraph (forked onto RCCL's stream by the events c10d records, joined before
# the optimizer) - a replay is ONE hipGraphLaunch.  Measured on a one-rank communicator (tools/dp_probe.py): 14.85 ms local, 15.31 ms with c10d's asynchronous
# collectives between graph segments (the cross-stream event traffic costs 0.37 ms per step plus ~0.09 ms per cut), back to local speed when captured.  A backend
# that cannot be captured (gloo) or a failing capture falls back to the chain of graph segments - on EVERY rank: the outcome is agreed with a MIN all-reduce
# (Trainer._agree), a rank never replays captured collectives while another issues eager ones.
# PN2_DP_CAPTURE: "1" always try, "0" never, unset = only on a one-rank communicator (verified on hardware, bench.py --dp1).  Multi-rank runs default to the
# segment chain - c10d's ordinary asynchronous collectives, the path every DDP job takes - until captured multi-rank RCCL has run on an 8-GPU node.
_DP_CAPTURE_ENV = os.environ.get("PN2_DP_CAPTURE", "")


def dp_capture_wanted(world):
    return _DP_CAPTURE_ENV == "1" or (_DP_CAPTURE_ENV == "" and world == 1)


_CAP_GROUPS = {}          # main process group -> (communicator for captured collectives, time of its one eager collective): shared by the trainers of a process


class Trainer:
    def __init__(self, model, lr=1e-4, clip=0.5, betas=(0.9, 0.999), eps=1e-8, dtype=None, process_group=None, bucket_bytes=100 << 20,
                 loss="structure", loss_weights=(0.5, 0.7, 0.3), weight_decay=0.0, hot=None, force_dp=False):
        """loss: "structure" - the 4-pair structure loss of MyTrain_med.py:78-82 on (images, masks), Adam + clip_gradient (binary_seg);
                 "mutation"  - the 15-subset CE + Dice + BCE loss of EMCAD/trainer.py:106-140 on (images, (label, bg_mask)) with the 8 maps of a
                               dual EMCADNet; pass clip=None and weight_decay=1e-4 for its AdamW (trainer.py:75).
        hot: the parameters the step trains (default model.hot_parameters()).
        force_dp: run the data-parallel machinery (bucket hooks, graph segments, asynchronous all-reduce) even on a 1-rank process group - how the
                 RCCL path is exercised on a single GPU (tests/test_gpu_dp.py, bench.py --dp1; also PN2_DP_FORCE=1)."""
        self.model = model
        self.loss_kind, self.loss_weights, self.weight_decay = loss, tuple(float(v) for v in loss_weights), float(weight_decay)
        if loss not in ("structure", "mutation"):
            raise ValueError(f"unknown loss {loss!r}")
        clip = 3.0e38 if clip is None else clip
        self.lr, self.clip, self.betas, self.eps = lr, clip, betas, eps
        self.dtype = get_compute_dtype() if dtype is None else dtype
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.dp = process_group is not None and (self.world > 1 or force_dp or os.environ.get("PN2_DP_FORCE", "0") == "1")
        hot = list(model.hot_parameters()) if hot is None else list(hot)
        hot_ids = {id(p) for p in hot}
        cold = [p for p in model.parameters() if id(p) not in hot_ids]
        dev = hot[0].device
        if dev.type != "cuda":
            raise RuntimeError("Trainer needs the model on the GPU (no CPU path)")
        self.n_hot = sum(_r4(p.numel()) for p in hot)
        n_all = self.n_hot + sum(_r4(p.numel()) for p in cold)
        self.flat = torch.zeros(n_all, dtype=torch.float32, device=dev)
        self.gflat = torch.zeros(self.n_hot, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.n_hot, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(self.n_hot, dtype=torch.float32, device=dev)
        # [0..3] Adam bias corrections (pn2_adam_tick); [4..6] lr, clip, weight decay READ BY THE KERNEL (flag [7]): a captured step follows set_lr()
        self.bias_corr = torch.tensor([0.0, 0.0, 1.0, 1.0, lr, clip, float(weight_decay), 1.0], dtype=torch.float32, device=dev)
        self.off = {}
        o = 0
        for p in hot + cold:
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            self.off[id(p)] = (o, n)
            o += _r4(n)
        self.hot = hot
        # gradient buckets in arena order (backward completes them from the tail)
        bucket_bytes = int(os.environ.get("PN2_DP_BUCKET_MB", "0")) << 20 or bucket_bytes
        self.buckets = GradBuckets(self.gflat, [(id(p), self.off[id(p)][0], _r4(self.off[id(p)][1])) for p in hot], bucket_bytes, process_group,
                                   wire_dtype=torch.bfloat16 if os.environ.get("PN2_DP_WIRE", "fp32") == "bf16" else None)
        # The collectives that are captured INTO a step graph run on a communicator of their own (see _capture_with_collectives): its stream has carried exactly one
        # eager collective - the warm-up below, retired by c10d's watchdog long before any capture() - so no watchdog ever polls an event of a capturing stream.
        self.pg_cap, self._cap_warm_t = None, 0.0
        if self.dp and dp_capture_wanted(self.world):
            import time
            import torch.distributed as dist
            ranks = dist.get_process_group_ranks(process_group)
            if dist.get_backend(process_group) == "nccl" and len(ranks) == dist.get_world_size():      # new_group is a collective over the default group
                if process_group not in _CAP_GROUPS:
                    cap = dist.new_group(ranks=ranks, backend="nccl")
                    warm = torch.zeros(4, dtype=torch.float32, device=dev)
                    dist.all_reduce(warm, group=cap)          # creates the communicator and its stream outside any capture
                    torch.cuda.synchronize()
                    _CAP_GROUPS[process_group] = (cap, time.monotonic())
                self.pg_cap, self._cap_warm_t = _CAP_GROUPS[process_group]
        self._seg = None                # capture of a data-parallel step in progress (see _capture_segments)
        self._expected = None           # id(p) -> gradient contributions per step, learned from the first backward pass (see _backward)
        self.last_outs = None
        self.pack_cache = PackCache()
        # Everything shape-dependent (step arena, deferred-launch tables, captured graphs) lives in a per-shape state, so the
        # 0.75x / 1x / 1.25x multi-scale schedule of MyTrain_med.py:55,70-73 alternates between three warm states.
        self._states = {}
        self._cur = None
        self.fuse_tail = os.environ.get("PN2_FUSED_TAIL", "1") == "1"
        # per-shape conv kernel/tile choices, filled by timing the candidates during the first (eager) step; the table is
        # per process (= per GPU) so every trainer in the process launches identical kernels (bit-reproducible runs)
        self.tuner = TUNER if os.environ.get("PN2_AUTOTUNE", "1") == "1" else None
        self._tuned = len(self.tuner) if self.tuner is not None else 0

    # ------------------------------------------------------------------ optimizer surface (utils.adjust_lr / clip_gradient of MyTrain_med.py:85,155)
    def set_lr(self, lr):
        """Takes effect at the next step, captured hipGraphs included (the kernel reads lr from the device)."""
        self.lr = float(lr)
        self.bias_corr[4] = self.lr

    def set_clip(self, clip):
        self.clip = 3.0e38 if clip is None else float(clip)
        self.bias_corr[5] = self.clip

    @property
    def param_groups(self):
        """torch.optim-style view so that utils.adjust_lr(trainer, ...) works: param_groups[0]['lr'] *= decay."""
        tr = self

        class _Group(dict):
            def __setitem__(g, k, v):
                dict.__setitem__(g, k, v)
                if k == 'lr':
                    tr.set_lr(v)
        return [_Group(lr=self.lr, params=self.hot, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay)]

    # ------------------------------------------------------------------ pieces
    def _grad_view(self, p):
        off, n = self.off[id(p)]
        return self.gflat[off:off + n].view(p.shape)

    def _state(self, images, size=None):
        N, _, H, W = images.shape
        key = (N, H, W, size if size is not None else H)
        st = self._states.get(key)
        if st is None:
            from types import SimpleNamespace
            # PN2_DEFER_WGRAD: 2 (default) wgrad + slab reduction deferred into table-driven launches, 1 only the reductions, 0 neither
            mode = os.environ.get("PN2_DEFER_WGRAD", "2")
            st = self._states[key] = SimpleNamespace(
                key=key, steps_run=0, graph=None, graph_opt=None, segments=None, s_images=None, s_gts=None, s_loss=None,
                lock_cache={},
                grad_queue=GradQueue(defer_wgrad=mode == "2") if mode in ("1", "2") else None,
                arena=StepArena() if os.environ.get("PN2_STEP_ARENA", "1") == "1" else None)
        self._cur = st
        return st

    # the state of the most recent step (tools / tests look at these)
    arena = property(lambda self: self._cur.arena if self._cur else None)
    grad_queue = property(lambda self: self._cur.grad_queue if self._cur else None)
    steps_run = property(lambda self: self._cur.steps_run if self._cur else 0)

    def forward_backward(self, images, gts, reduce_hook=True, size=None):
        """forward + loss + backward; returns loss[5] = (l2, l3, l4, l5 pair losses, total) on device.
        size: train at size x size — images and masks are resized on the device first, bilinear with align_corners=True, exactly the
        multi-scale rescale of MyTrain_med.py:70-73."""
        st = self._state(images, size)
        self.pack_cache.refresh()
        if st.arena is not None:
            st.arena.begin_step(self.flat.device)
        st.steps_run += 1
        eng = Engine(self.dtype, True, grad_provider=self._grad_view, need_grad=True, pack_cache=self.pack_cache, tuner=self.tuner, grad_queue=st.grad_queue, arena=st.arena,
                     lock_cache=st.lock_cache if (st.arena is not None and st.arena.buf is not None) else None)    # recorded launches hold raw pointers:
        # only with the bump arena (nothing is recycled inside a step) do the buffers of a lane outlive the deferred emission
        eng.fuse_tail = self.fuse_tail and self.loss_kind == "structure"
        if size is not None and (size != images.shape[2] or size != images.shape[3]):
            x = eng.cast(eng.resize_to(eng.from_nchw(images, dt=F32), size, size, align_corners=True), self.dtype)
            n_, _, h_, w_ = gts.shape
            g = Act(eng, gts.reshape(n_, h_, w_, 1).float().contiguous(), 1, 1, 1, F32, requires_grad=False)
            gts = eng.resize_to(g, size, size, align_corners=True).t
        else:
            x = eng.from_nchw(images)
        outs = self.model._build(eng, x)
        eng.finish_forward()
        if self.loss_kind == "mutation":
            label, bg_mask = gts
            loss = L.mutation_forward_backward(eng, outs, label, bg_mask, self.loss_weights)
            return self._backward(eng, st, loss, None, reduce_hook)
        N, H, W = outs[0].N, outs[0].H, outs[0].W
        P = len(outs) // 2
        lat = eng.lateral_block()
        if lat is None or any(o.t.data_ptr() != lat[j].data_ptr() for j, o in enumerate(outs)):
            lat = torch.stack([o.t for o in outs])
        HW = H * W
        mask = gts.reshape(N, HW).float().contiguous()
        fused = len(eng.tail) == len(outs) and outs[0].C == 1
        if eng.tail and not fused:
            raise RuntimeError("only some lateral maps were deferred to the fused DSRA tail")
        if fused:       # up-sampling + loss in one pass; the backward goes straight to the low-res maps
            loss = L.tail_forward_backward(eng, eng.tail, lat, P, mask, N, H, W) if os.environ.get("PN2_TAIL_FUSED_ENTRY", "1") == "1" else None
            if loss is None:
                loss, saved = L.tail_forward(eng, eng.tail, lat, P, mask, N, H, W)
                L.tail_backward(eng, eng.tail, P, mask, saved, 1.0)
        else:
            loss, saved = L.loss_forward(lat, P, mask, N, HW, H, W)
            dlat = eng.alloc(lat.shape, lat.dtype)
            L.loss_backward(lat, dlat, P, mask, saved, N, HW, 1.0)
            for j, o in enumerate(outs):
                o.grad = dlat[j]
                o.grad_written = True
        return self._backward(eng, st, loss, lat, reduce_hook)

    def _backward(self, eng, st, loss, lat, reduce_hook):
        if self.dp:
            self.buckets.reset()
        rq = st.grad_queue
        if rq is not None:
            rq.begin_step()

        def grads_complete():       # everything queued so far must be launched before a bucket is sent
            eng.flush_colsum()
            if rq is not None:
                rq.flush()
        # A bucket may leave as soon as every gradient in it is COMPLETE.  "Has been written" is not enough: a weight applied k times per step
        # (CAB's shared fc1 / fc2, EMCAD_dual's single sab conv) receives k contributions from different tape entries.  The first backward pass
        # of a trainer therefore only counts contributions per parameter (its buckets all leave at the end); later passes launch a bucket when
        # every count has reached that number, and a contribution that arrives after its bucket was sent is an error, not a silent corruption.
        hook = self.dp and reduce_hook
        expected = self._expected
        if hook and expected is not None:
            self.buckets.expect(expected)

            def late(k, c, bk=self.buckets):
                if k in bk.launched_keys:
                    raise RuntimeError("a gradient contribution arrived after its bucket was all-reduced (the model's use of its parameters changed "
                                       "between steps): build a new Trainer")
                bk.contribution(k)
            eng.pgrads.on_sink = late
        for fn in reversed(eng.tape):
            fn()
            if hook and expected is not None:
                self.buckets.launch_ready(eng.pgrads.counts, before_launch=grads_complete, expected=expected)
                if self._seg is not None and self.buckets.record:
                    self._cut_segment()
        eng.tape = []
        if eng._pending_pool:
            raise RuntimeError("a deferred AvgPool2d backward was never applied (see Engine.backward)")
        eng.pgrads.on_sink = None
        if len(eng.pgrads.written) != len(self.hot):
            # torch.optim skips parameters whose grad is None; the fused kernel steps the whole arena, so a trained parameter without a gradient
            # would be stepped with LAST step's gradient.  Fail loudly instead (pass hot= without it, e.g. hot_parameters(one_channel=False)).
            missing = [n for n, p in self.model.named_parameters() if id(p) in {id(q) for q in self.hot} and id(p) not in eng.pgrads.written]
            raise RuntimeError(f"{len(missing)} trained parameters received no gradient this step (first: {missing[:3]}): exclude them from `hot`")
        grads_complete()
        if self.dp:
            if reduce_hook and expected is None:
                self._expected = dict(eng.pgrads.counts)
            self.buckets.finish()
        self.last_outs = lat
        # break the reference cycles engine <-> activations <-> closures now: the engine holds this trainer (grad_provider), and a trainer that only
        # dies when the cyclic collector gets to it would destroy its hipGraphs at an arbitrary later time - e.g. inside another graph's capture
        eng.tail.clear()
        eng._lat = None
        eng.pgrads.provider = None
        eng.lock_cache = eng.grad_queue = eng.arena = eng.pack_cache = None
        if self.tuner is not None and len(self.tuner) != self._tuned and os.environ.get("PN2_TUNE_CACHE"):
            from .engine import save_tuner
            save_tuner(os.environ["PN2_TUNE_CACHE"])
        self._tuned = len(self.tuner) if self.tuner is not None else 0
        return loss

    def optimizer_step(self):
        st = _stream()
        call.pn2_adam_tick(_p(self.bias_corr), self.betas[0], self.betas[1], st)
        call.pn2_clamp_adam(_p(self.flat), _p(self.gflat), _p(self.exp_avg), _p(self.exp_avg_sq), self.n_hot, self.lr, self.betas[0], self.betas[1],
                            self.eps, self.clip, 1.0 / self.world, _p(self.bias_corr), self.weight_decay, st)

    def step(self, images, gts, size=None):
        """One MyTrain_med.py:59-86 iteration (at `size` x `size` when given).  Returns the device tensor [loss2, loss3, loss4, loss5, total]."""
        loss = self.forward_backward(images, gts, size=size)
        self.optimizer_step()
        return loss

    # ------------------------------------------------------------------ checkpoint / resume (the reference saves weights only, MyTrain_med.py:99-103)
    def state_dict(self):
        """Everything a resumed run needs: the model's state_dict (weights + BN running statistics) and the optimizer state
        (Adam moments over the flat arena, bias-correction powers = step count, hyper-parameters)."""
        return {"model": {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                "optimizer": {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "bias_corr": self.bias_corr.clone(),
                              "lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "clip": self.clip, "weight_decay": self.weight_decay,
                              "layout": [(n, self.off[id(p)][0], p.numel()) for n, p in self.model.named_parameters() if id(p) in self.off]}}

    def load_state_dict(self, sd):
        """Restore a state_dict() of a Trainer built over the same model class; the next step continues bit for bit."""
        layout = [(n, self.off[id(p)][0], p.numel()) for n, p in self.model.named_parameters() if id(p) in self.off]
        o = sd["optimizer"]
        if [tuple(x) for x in o["layout"]] != layout:
            raise RuntimeError("optimizer state was saved for a different parameter layout")
        self.model.load_state_dict(sd["model"], strict=True)          # copies into the arena views in place
        with torch.no_grad():
            self.exp_avg.copy_(o["exp_avg"]); self.exp_avg_sq.copy_(o["exp_avg_sq"]); self.bias_corr[:4].copy_(o["bias_corr"][:4])
        self.lr, self.betas, self.eps, self.clip, self.weight_decay = o["lr"], tuple(o["betas"]), o["eps"], o["clip"], o["weight_decay"]
        with torch.no_grad():
            self.bias_corr[4:7] = torch.tensor([self.lr, self.clip, self.weight_decay], device=self.bias_corr.device)
        for st in self._states.values():                               # captured graphs baked the old hyper-parameters in
            st.graph = st.graph_opt = st.segments = None
        return self

    # ------------------------------------------------------------------ hipGraph replay of the whole step
    def capture(self, images, gts, warmup=3, size=None):
        """Capture forward+loss+backward(+Adam) into hipGraphs and replay them with `replay(images, gts)`.
        With data parallelism the bucket all-reduces are either captured into the step graph as forked branches (_capture_with_collectives; default on a one-rank
        communicator, PN2_DP_CAPTURE=1 elsewhere) or issued by c10d between a chain of graph segments cut where buckets leave (_capture_segments); all ranks take the
        same form (_agree).  One set of graphs per (batch shape, train size): call once per scale of a multi-scale schedule."""
        st = self._state(images, size)
        if st.steps_run + warmup < 2:
            # step 1 measures the arena, step 2 builds the deferred-launch tables on the arena addresses the graph will replay
            raise RuntimeError("capture() needs at least 2 eager steps before the captured one (warmup >= 2 on a fresh Trainer)")
        st.s_images = images.clone()
        st.s_gts = tuple(g.clone() for g in gts) if isinstance(gts, (tuple, list)) else gts.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self.step(st.s_images, st.s_gts, size=size)
            if self.dp and not DP_SEGMENTS:      # that captured pass has no bucket hooks: let it build its own (single) reduce table eagerly
                self.forward_backward_local(st.s_images, st.s_gts, size=size)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        import gc
        st.graph = torch.cuda.CUDAGraph()
        gc_was = gc.isenabled()
        gc.collect()
        gc.disable()                   # no collector runs inside a capture: a destructor that touches the HIP runtime there aborts the process
        try:
            if not self.dp:
                with torch.cuda.graph(st.graph, capture_error_mode=CAPTURE_MODE):
                    st.s_loss = self.step(st.s_images, st.s_gts, size=size)
                st.graph_opt = None
            elif self._agree(self._capture_with_collectives(st, size)):
                st.graph_opt, st.segments = None, None
            else:
                st.graph = torch.cuda.CUDAGraph()
                if DP_SEGMENTS and self._expected is not None:
                    st.segments = self._capture_segments(st, size)
                    st.graph = st.segments[0][0]
                else:
                    st.segments = None
                    with torch.cuda.graph(st.graph, capture_error_mode=CAPTURE_MODE):
                        st.s_loss = self.forward_backward_local(st.s_images, st.s_gts, size=size)
                st.graph_opt = torch.cuda.CUDAGraph()
                with torch.cuda.graph(st.graph_opt, capture_error_mode=CAPTURE_MODE):
                    self.optimizer_step()
        finally:
            if gc_was:
                gc.enable()
        return self

    def _capture_with_collectives(self, st, size):
        """The whole data-parallel step - forward, loss, backward with the bucket all-reduces launched where the eager step launches them, the wait for them,
        the optimizer - as ONE hipGraph.  c10d runs a collective on its own stream behind an event of the capturing stream and joins it back in Work.wait():
        inside a capture those become a forked branch of the graph, so the all-reduce of a bucket overlaps the backward kernels that follow it exactly as in
        the eager step, without any host-side stream traffic at replay.  Returns False (nothing captured) when the backend cannot be captured."""
        if self.pg_cap is None or self._expected is None:
            return False
        # c10d's watchdog thread polls the events of EAGER collectives (every 100 ms) until it has seen them complete.  Once a capture pulls a communicator's
        # stream in, HIP refuses hipEventQuery on events of that stream ("operation not permitted on an event last recorded in a capturing stream") and the
        # watchdog takes the process down - round 4 hit that when a capture started within a poll interval of the last eager collective of the warm-up steps and
        # papered over it with a sleep.  Now the captured collectives have their own communicator (self.pg_cap): the eager steps - warm-up, fall-back, the
        # agreement all-reduce - run on self.pg, whose stream is never captured, and pg_cap's single eager collective dates from the constructor, at least two
        # eager training steps ago (capture() insists on them).  The guard below only ever waits in a test that builds and captures within the same 0.5 s.
        import time
        torch.cuda.synchronize()
        dt = 0.5 - (time.monotonic() - self._cap_warm_t)
        if dt > 0:
            time.sleep(dt)
        g = torch.cuda.CUDAGraph()
        main_pg, self.buckets.pg = self.buckets.pg, self.pg_cap
        try:
            with torch.cuda.graph(g, capture_error_mode=CAPTURE_MODE):
                st.s_loss = self.step(st.s_images, st.s_gts, size=size)
        except Exception as e:          # noqa: BLE001   (the context manager has ended the capture; fall back to graph segments)
            import warnings
            warnings.warn(f"capturing the RCCL collectives into the step graph failed ({type(e).__name__}: {e}); falling back to graph segments")
            self.buckets.works = []
            torch.cuda.synchronize()
            return False
        finally:
            self.buckets.pg = main_pg
        st.graph = g
        return True

    def _agree(self, ok):
        """Every rank must replay the same form of the step: a rank whose capture of the collectives failed would issue eager all-reduces between graph segments
        while the others replay captured ones - different collective sequences on the wire, a hang.  MIN over the ranks of the local outcome, eagerly, on the main
        communicator (nothing is capturing here).  One rank: nothing to agree."""
        if self.world > 1:
            import torch.distributed as dist
            torch.cuda.synchronize()
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.flat.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.pg)
            agreed = bool(int(flag.item()))
            if ok and not agreed:
                import warnings
                warnings.warn("another rank could not capture the RCCL collectives into its step graph; every rank falls back to graph segments")
            return agreed
        return ok

    def _cut_segment(self):
        """Capture of a data-parallel step: one or more gradient buckets just became complete - close the hipGraph segment that produced them
        and open the next one; replay() starts their all-reduce right after enqueueing this segment, next to the segments that follow."""
        seg = self._seg
        seg["g"].capture_end()
        seg["list"].append((seg["g"], list(self.buckets.record)))
        del self.buckets.record[:]
        seg["g"] = torch.cuda.CUDAGraph()
        seg["g"].capture_begin(pool=seg["pool"], capture_error_mode=CAPTURE_MODE)

    def _capture_segments(self, st, size):
        """forward+loss+backward of a data-parallel rank as a CHAIN of hipGraphs cut where buckets leave (same places as in the eager step, so
        the deferred weight-gradient tables are the ones the eager steps built).  All segments share one memory pool and are replayed in capture
        order.  Returns [(graph, [bucket, ...]), ...]."""
        segs = []
        cs = torch.cuda.Stream()
        cs.wait_stream(torch.cuda.current_stream())
        torch.cuda.synchronize()
        with torch.cuda.stream(cs):
            g = torch.cuda.CUDAGraph()
            self._seg = {"g": g, "pool": torch.cuda.graph_pool_handle(), "list": segs}
            self.buckets.record = []
            g.capture_begin(pool=self._seg["pool"], capture_error_mode=CAPTURE_MODE)
            try:
                st.s_loss = self.forward_backward(st.s_images, st.s_gts, size=size)
                self._seg["g"].capture_end()
                segs.append((self._seg["g"], list(self.buckets.record)))
            except BaseException:
                # an error inside the hooked backward (late contribution, a kernel status, ...) must not leave the stream in capture mode: every
                # later HIP call of the process would fail with an unrelated capture-invalidated error.  End the open segment, drop the partial chain.
                try:
                    self._seg["g"].capture_end()
                except Exception:
                    pass
                del segs[:]
                st.segments = st.graph = None
                raise
            finally:
                self._seg = None
                self.buckets.record = None
        torch.cuda.current_stream().wait_stream(cs)
        torch.cuda.synchronize()
        return segs

    def forward_backward_local(self, images, gts, size=None):
        w, self.dp = self.dp, False
        try:
            return self.forward_backward(images, gts, size=size)
        finally:
            self.dp = w

    def replay(self, images=None, gts=None, size=None):
        """Replay the graphs captured for this batch shape / train size (the most recently used state when called without arguments)."""
        st = self._cur if images is None else self._state(images, size)
        if st is None or st.graph is None:
            raise RuntimeError("capture() this batch shape / train size first")
        if images is not None:
            st.s_images.copy_(images, non_blocking=True)
            if isinstance(gts, (tuple, list)):
                for d_, s_ in zip(st.s_gts, gts):
                    d_.copy_(s_, non_blocking=True)
            else:
                st.s_gts.copy_(gts, non_blocking=True)
        if st.graph_opt is None:
            st.graph.replay()
        elif st.segments:
            # data parallel: every segment ends where gradient buckets are complete; their all-reduce runs next to the segments that follow
            self.buckets.reset()
            for g, bs in st.segments:
                g.replay()
                self.buckets.launch_async(bs)
            self.buckets.wait()
            st.graph_opt.replay()
        else:
            st.graph.replay()
            self.buckets.reduce_all()
            st.graph_opt.replay()
        return st.s_loss
