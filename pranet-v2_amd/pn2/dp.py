"""Gradient buckets for one-process-per-GPU data parallelism (device agnostic: RCCL on GPUs, gloo in CPU tests).

The flat gradient arena is cut into contiguous buckets in parameter order.  Backward produces gradients from the tail
of the arena towards its head; `launch_ready()` is called after every tape entry and starts an asynchronous all-reduce
(sum) for each bucket whose gradients are all present, so the collective overlaps the remaining backward kernels.
Averaging (1/world) is not done here — it is folded into the fused clamp+Adam kernel (grad_scale).
"""
import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, gflat, spans, bucket_bytes=100 << 20, process_group=None, wire_dtype=None, head_bytes=None):
        """spans: ordered list of (key, offset, length) of the parameters inside `gflat` (elements).
        Buckets are cut FROM THE TAIL of the arena - the order backward completes gradients in: full buckets of `bucket_bytes`, whose all-reduce overlaps
        the backward kernels that follow, and the bucket that leaves last (the head of the arena: nothing is left to overlap its all-reduce with) holds at
        most `head_bytes` (default bucket_bytes / 5).  Every cut costs a flush of the deferred weight-gradient tables (0.25 ms on an MI355X: two sets of smaller, less balanced
        tables instead of one), every byte of the last bucket is exposed wire time: 100 MB gives PraNet-V2's 122 MB of gradients ONE cut - a 101 MB bucket that leaves with a fifth of
        the backward pass still to run, and a 16 MB head (one-rank RCCL, ms per step: local 14.22, no cut 14.22, one cut 14.47, two cuts 14.71).
        wire_dtype: None = the buckets travel as they are (fp32, what the reference's DDP does); torch.bfloat16 = each bucket is rounded to bf16
        for the exchange (half the bytes on xGMI) and widened back into the fp32 arena after it - the sum itself is then a bf16 sum, so this is
        a bandwidth / precision trade the caller has to ask for (PN2_DP_WIRE=bf16)."""
        self.gflat, self.pg, self.wire = gflat, process_group, wire_dtype
        es = gflat.element_size()
        head_bytes = bucket_bytes // 5 if head_bytes is None else head_bytes
        cuts = []                # bucket boundaries as span indices, from the tail: bucket = spans[i0:i1]
        i1, nbytes = len(spans), 0
        for i in range(len(spans) - 1, -1, -1):
            nbytes += spans[i][2] * es
            if nbytes >= bucket_bytes:
                cuts.append((i, i1)); i1, nbytes = i, 0
        if i1 > 0:               # what is left at the head of the arena: the last bucket to leave - at most head_bytes, the rest a bucket of its own
            i0, nb = 0, 0
            while i0 < i1 and nb + spans[i0][2] * es <= head_bytes:
                nb += spans[i0][2] * es; i0 += 1
            if 0 < i0 < i1:
                cuts.append((i0, i1)); i1 = i0
            cuts.append((0, i1))
        self.buckets = []        # (start, end, frozenset(keys)), in arena order
        for i0, i1 in reversed(cuts):
            self.buckets.append((spans[i0][1], spans[i1 - 1][1] + spans[i1 - 1][2], frozenset(k for k, _, _ in spans[i0:i1])))
        end = spans[-1][1] + spans[-1][2] if spans else 0
        self.total = end
        self.record = None       # a list while a step is being CAPTURED: launches are noted (bucket indices), not issued (Trainer.capture)
        self.reset()

    def reset(self):
        self.pending = list(range(len(self.buckets)))
        self.works = []
        self.order = []          # bucket indices in launch order (for tests)
        self.launched_keys = set()
        self._remaining = None   # per-bucket outstanding contributions (expect() / contribution()), None: launch_ready() scans
        self._ready = []

    def expect(self, expected):
        """Arm the O(1) bookkeeping of a backward pass: `expected` maps key -> contributions a full pass delivers (learned by the trainer's first
        pass).  Every ParamGrads.sink then calls contribution(key); a bucket whose outstanding count reaches 0 is queued for the next
        launch_ready() - instead of rescanning every key of every pending bucket after every tape entry (~500 x ~500 dictionary look-ups per
        eager step).  Buckets none of whose keys is expected stay pending until finish(), as in the scanning form."""
        self._key_bucket = {}
        self._remaining = []
        for b, (_, _, keys) in enumerate(self.buckets):
            n = 0
            for k in keys:
                if k in expected:
                    self._key_bucket[k] = b
                    n += expected[k]
            self._remaining.append(n if n > 0 else -1)
        self._ready = []

    def contribution(self, key):
        b = self._key_bucket.get(key)
        if b is None:
            return
        self._remaining[b] -= 1
        if self._remaining[b] == 0:
            self._ready.append(b)

    def _launch(self, b):
        a, e, keys = self.buckets[b]
        self.launched_keys |= keys
        if self.record is not None:
            self.record.append(b)
        else:
            self._issue(a, e)
        self.order.append(b)

    def _issue(self, a, e):
        if self.wire is None:
            self.works.append((dist.all_reduce(self.gflat[a:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True), None, a, e))
        else:
            buf = self.gflat[a:e].to(self.wire)
            self.works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), buf, a, e))

    def launch_async(self, bs):
        """Replay side of a captured step: start the all-reduce of buckets `bs` now (the graph segment that completed them was just enqueued on
        the current stream; the collective's stream waits for it and then runs next to the following segment)."""
        for b in bs:
            a, e, _ = self.buckets[b]
            self._issue(a, e)

    def wait(self):
        for w, buf, a, e in self.works:
            w.wait()
            if buf is not None:
                self.gflat[a:e].copy_(buf)
        self.works = []

    def launch_ready(self, written, before_launch=None, expected=None):
        """Start the all-reduce of every pending bucket whose gradients are COMPLETE.
        written: the set of keys that have received a gradient, or (with `expected`) a dict key -> contributions received so far this step.
        expected: dict key -> contributions a full backward pass delivers to that parameter (a weight applied k times per step - CAB's shared
        fc1 / fc2, EMCAD_dual's single sab conv - receives k of them, in different tape entries): a key is complete only when its count has
        reached that number; keys missing from `expected` never receive a gradient and do not hold a bucket back... until finish().
        before_launch: called once if anything is about to be launched (e.g. join a side stream that produced the gradients)."""
        if self._remaining is not None and expected is not None:
            ready, self._ready = sorted(self._ready), []          # armed by expect(): O(1) per call; same launch order as the scan below
        elif expected is None:
            ready = [b for b in self.pending if self.buckets[b][2] <= written]
        else:
            ready = [b for b in self.pending if all(written.get(k, 0) >= expected[k] for k in self.buckets[b][2] if k in expected)
                     and any(k in expected for k in self.buckets[b][2])]
        if ready and before_launch is not None:
            before_launch()
        for b in ready:
            self._launch(b)
            self.pending.remove(b)

    def finish(self):
        """Launch whatever is left (parameters that got no gradient this step) and wait for all collectives."""
        for b in list(self.pending):
            self._launch(b)
        self.pending = []
        self.wait()

    def reduce_all(self):
        self.reset()
        self.finish()
