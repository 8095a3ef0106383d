"""Gradient buckets for one-process-per-GPU data parallelism (device agnostic: RCCL on GPUs, gloo in CPU tests).

The flat gradient arena is cut into contiguous buckets in parameter order.  Backward produces gradients from the tail
of the arena towards its head; `launch_ready()` is called after every tape entry and starts an asynchronous all-reduce
(sum) for each bucket whose gradients are all present, so the collective overlaps the remaining backward kernels.
Averaging (1/world) is not done here — it is folded into the fused clamp+Adam kernel (grad_scale).
"""
import torch
import torch.distributed as dist


class GradBuckets:
    def __init__(self, gflat, spans, bucket_bytes=32 << 20, process_group=None):
        """spans: ordered list of (key, offset, length) of the parameters inside `gflat` (elements)."""
        self.gflat, self.pg = gflat, process_group
        self.buckets = []        # (start, end, frozenset(keys))
        cur, start, nbytes = [], 0, 0
        end = 0
        for key, off, n in spans:
            cur.append(key)
            end = off + n
            nbytes += n * gflat.element_size()
            if nbytes >= bucket_bytes:
                self.buckets.append((start, end, frozenset(cur)))
                cur, start, nbytes = [], end, 0
        if cur:
            self.buckets.append((start, end, frozenset(cur)))
        self.total = end
        self.reset()

    def reset(self):
        self.pending = list(range(len(self.buckets)))
        self.works = []
        self.order = []          # bucket indices in launch order (for tests)

    def _launch(self, b):
        a, e, _ = self.buckets[b]
        self.works.append(dist.all_reduce(self.gflat[a:e], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))
        self.order.append(b)

    def launch_ready(self, written, before_launch=None):
        """Start the all-reduce of every pending bucket whose parameter keys are all in `written`.
        before_launch: called once if anything is about to be launched (e.g. join a side stream that produced the gradients)."""
        ready = [b for b in self.pending if self.buckets[b][2] <= written]
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
        for w in self.works:
            w.wait()
        self.works = []

    def reduce_all(self):
        self.reset()
        self.finish()
