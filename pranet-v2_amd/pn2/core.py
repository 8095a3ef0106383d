"""Host engine, shared part: NHWC activations, parameter-gradient sinks, the persistent caches (packed panels, folded BatchNorm rows, step arena,
deferred weight-gradient queue), the tuning table and the behaviour switches.  pn2/engine.py builds the Engine class on top of it from the op
mix-ins ops_conv.py / ops_encoder.py / ops_spatial.py.

NHWC activations, a reverse-mode tape, and the op set the PraNet models are written in.

Everything here is plumbing around the C ABI (capi.py): PyTorch supplies device memory and the
current HIP stream; every arithmetic pass over an activation is one of the gfx950 kernels in csrc/.
There is deliberately no CPU implementation: ops raise on non-GPU tensors.

Layout: activations are NHWC with *physical* channels.  A tensor whose logical channels come in
groups of `gw` (Res2Net's 26/52-wide splits, K-channel heads) stores each group in `gwp` = gw rounded
up to 8 slots, the pad slots holding exact zeros; weights are packed with matching zero rows/columns,
so the arithmetic is unchanged while every pixel row stays 16-byte aligned.
"""
import ctypes as C
import math
import os

import torch

from . import capi
from .capi import call, F32, BF16

TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}
_PERMS = {}     # channel_shuffle permutations (device int32 tensors) by (channels, groups)
TUNER = {}      # process-wide conv shape -> tuned kernel/tile code (see Engine._tune_gemm)


def load_tuner(path):
    """Merge a saved tuning table (PN2_TUNE_CACHE=<file>) so that a run does not have to time the candidates again."""
    import ast
    import json
    try:
        with open(path) as f:
            for k, v in json.load(f).items():
                TUNER.setdefault(ast.literal_eval(k), tuple(v) if isinstance(v, list) else v)
    except (OSError, ValueError):
        pass


def save_tuner(path):
    """Rank 0 only (every DP rank would otherwise race on the same file), through a temporary file + os.replace (no torn reads)."""
    import json
    if int(os.environ.get("RANK", "0")) != 0:
        return
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as f:
        json.dump({repr(k): v for k, v in TUNER.items()}, f)
    os.replace(tmp, path)


TUNE_REPS = int(os.environ.get("PN2_TUNE_REPS", "3"))            # timed repetitions per tuning candidate (the minimum counts)
# Shipped tuning table: the (kernel, tile) and wgrad (kernel, pixel splits) choices for the conv shapes of the BASELINE configurations on an MI355X,
# produced by the tuner itself (PN2_TUNE_REPS=7 PN2_TUNE_CACHE=... python bench.py per configuration).  Keys carry the complete shape, so a table
# entry only ever applies to exactly the launch it was timed for; shapes not in the table are tuned at first use as before.  PN2_TUNE_TABLE=0 ignores it.
if os.environ.get("PN2_TUNE_CACHE"):
    load_tuner(os.environ["PN2_TUNE_CACHE"])          # (first entry wins: an explicit cache overrides the shipped table)
if os.environ.get("PN2_TUNE_TABLE", "1") != "0":          # "0": no table; "1" (default): the shipped one; anything else: path of another table (A/B of tuning runs)
    _tt = os.environ.get("PN2_TUNE_TABLE", "1")
    load_tuner(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json") if _tt == "1" else _tt)


def rup(v, m):
    return (v + m - 1) // m * m


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _w4(w):
    """OIHW shape of a conv weight; an nn.Linear weight [out, in] is a 1x1 conv weight with the same memory layout."""
    return tuple(w.shape) if w.dim() == 4 else (w.shape[0], w.shape[1], 1, 1)


class _LinearAsConv:
    """nn.Linear over the channel axis of NHWC tokens == 1x1 convolution (pvtv2.py:19,22,62-65)."""
    __slots__ = ("weight", "stride", "padding", "dilation", "groups")

    def __init__(self, lin):
        self.weight, self.stride, self.padding, self.dilation, self.groups = lin.weight, (1, 1), (0, 0), (1, 1), 1


class Act:
    """NHWC activation view.  t: torch tensor (N,H,W,Cp) whose last dim is contiguous; ld = pixel stride."""
    __slots__ = ("eng", "t", "N", "H", "W", "C", "gw", "gwp", "dt", "grad", "_written", "child_written", "requires_grad", "parent", "c0", "lat", "galias",
                 "bnb", "bstats", "sum_of", "dual_done", "_sealed", "grad_masked", "colparts", "pool_prior")

    def __init__(self, eng, t, C_, gw=None, gwp=None, dt=None, requires_grad=True):
        self.eng, self.t = eng, t
        self.N, self.H, self.W = t.shape[0], t.shape[1], t.shape[2]
        self.C = C_
        self.gw = gw if gw is not None else t.shape[3]
        self.gwp = gwp if gwp is not None else t.shape[3]
        self.dt = dt if dt is not None else (F32 if t.dtype == torch.float32 else BF16)
        self.grad, self._written, self.child_written, self.requires_grad = None, False, False, requires_grad
        self.parent, self.c0 = None, 0
        self.lat = None                 # index of the full-resolution lateral output slot this Act is (Engine.lateral_out)
        self.galias = None              # Act whose gradient storage this one shares (Engine.binary(..., grad_alias=True))
        self.bnb = None                 # Bnb: the train-mode BatchNorm this activation is the output of (statistics of its gradient can be taken in a dgrad epilogue)
        self.bstats = None              # [(c0, ncols, p1, p2, nblk, ldp)] BatchNorm-backward partial sums left by dgrad epilogues, by physical column range
        self.sum_of = None              # (u, v): this Act is u + v written by u's BN-apply pass (conv_bn_act(sum_with=v)); its gradient aliases v's
        self.dual_done = False          # the sum's consumer wrote the gradient of BOTH operands (dual-target dgrad epilogue)
        self._sealed = False            # a dgrad that declared itself the last contribution has written this gradient
        self.grad_masked = False        # that dgrad stored dz = dy * [y > 0] (PN2_BNB_STORE_MASKED): the gradient buffer already carries the ReLU mask
        self.colparts = None            # (partial rows, nblk, gradient tensor): column sums of this Act's gradient left by the kernel that wrote it (EncoderOps.dwconv_gelu)
        self.pool_prior = None          # gradient of AvgPool2d(2, 2)(this Act), not yet applied: the dgrad that completes this Act's gradient adds 1/4 of it in its epilogue (SpatialOps.avgpool(fold_bwd=True))

    @property
    def grad_written(self):
        # a slice of a buffer that was written as a whole (e.g. dgrad into a concat buffer) counts as written
        return self._written or (self.parent is not None and self.parent.grad_written)

    @grad_written.setter
    def grad_written(self, v):
        self._written = v
        if v and self.parent is not None:
            self.parent.child_written = True

    Cp = property(lambda s: s.t.shape[3])
    ld = property(lambda s: s.t.stride(2))
    M = property(lambda s: s.N * s.H * s.W)
    ptr = property(lambda s: C.c_void_p(s.t.data_ptr()))

    def slice(self, c0, c1, C_=None, gw=None, gwp=None):
        """Channel-slice view (physical channel range); its gradient is the same slice of this grad."""
        a = Act(self.eng, self.t[..., c0:c1], C_ if C_ is not None else c1 - c0, gw, gwp, self.dt, self.requires_grad)
        a.parent, a.c0 = self, c0
        if self.bnb is not None and self.bnb.split == 0:
            a.bnb = self.bnb.cols(c0, c1)
        return a

    def root(self):
        """-> (outermost parent, this view's first physical column inside it)"""
        a, off = self, 0
        while a.parent is not None:
            off += a.c0
            a = a.parent
        return a, off

    def add_bstats(self, c0, ncols, p1, p2, nblk, ldp):
        r, off = self.root()
        if r.bstats is None:
            r.bstats = []
        r.bstats.append((off + c0, ncols, p1, p2, nblk, ldp))

    def find_bstats(self):
        """Segments [(c0 (relative), ncols, p1, p2, nblk, ldp)] that cover this view's columns, newest entry first; None where nothing covers."""
        r, off = self.root()
        segs, c = [], 0
        have = r.bstats or []
        while c < self.Cp:
            hit = None
            for (s0, n, p1, p2, nblk, ldp) in reversed(have):
                if s0 <= off + c < s0 + n:
                    hit = (s0, n, p1, p2, nblk, ldp)
                    break
            if hit is None:
                # uncovered run up to the next covered column
                nxt = min([s0 - off for (s0, n, *_r) in have if s0 - off > c] + [self.Cp])
                segs.append((c, nxt - c, None, None, 0, 0))
                c = nxt
            else:
                s0, n, p1, p2, nblk, ldp = hit
                skip = off + c - s0
                take = min(n - skip, self.Cp - c)
                segs.append((c, take, p1[:, skip:], p2[:, skip:], nblk, ldp))
                c += take
        return segs

    def grad_buf(self):
        """Gradient storage (allocated on first use, uninitialised)."""
        if self.grad is None:
            if self.galias is not None:
                self.grad = self.galias.grad_buf()
            elif self.parent is not None:
                self.grad = self.parent.grad_buf()[..., self.c0:self.c0 + self.Cp]
            else:
                self.grad = self.eng.alloc(self.t.shape, self.t.dtype)
        return self.grad

    def grad_sink(self):
        """-> (tensor, accumulate_flag) for a backward op that contributes to this activation's gradient."""
        if self._sealed:
            raise RuntimeError("a gradient contribution arrived after the dgrad that was declared the last one (x_last=True)")
        g = self.grad_buf()
        acc = 1 if self.grad_written else 0
        if acc:
            self.colparts = None          # column sums a kernel left for an earlier, complete state of this gradient are stale once another consumer accumulates into it
        self.grad_written = True
        return g, acc


class Bnb:
    """What a dgrad epilogue needs to take the BatchNorm-backward statistics of the gradient it produces (pn2_conv_gemm_ep):
    raw: the BN's input (raw conv output) as a [N,H,W,C] view; par: [4][C] rows scale, shift, mean, invstd (a view: row stride = par.stride(0));
    relu: the activation behind the BN; ymask: the stored output (tensor view) when the ReLU mask cannot be recomputed from raw (BN + residual + ReLU).
    split / raw2 / par2 / tail: a concat buffer whose columns >= split are (a copy of) another BatchNorm's output `tail` (raw2 / par2 indexed by the
    same local column; par2 None = those columns carry no BatchNorm)."""
    __slots__ = ("raw", "par", "relu", "ymask", "split", "raw2", "par2", "tail", "res")

    def __init__(self, raw, par, relu, ymask=None, split=0, raw2=None, par2=None, tail=None):
        self.raw, self.par, self.relu, self.ymask, self.split, self.raw2, self.par2, self.tail = raw, par, relu, ymask, split, raw2, par2, tail
        self.res = None          # Act: the residual operand when it is itself a train-mode BatchNorm output without activation (Bottle2neck's downsample): its gradient is this BN's masked dz

    def cols(self, c0, c1):
        return Bnb(self.raw[..., c0:c1], self.par[:, c0:c1], self.relu, self.ymask[..., c0:c1] if self.ymask is not None else None)


class ParamGrads:
    """Where parameter gradients go.  Default: fresh fp32 tensors (autograd mode).  The trainer swaps in
    views of its flat gradient arena so the fused clamp+Adam kernel sees one contiguous buffer."""

    def __init__(self, provider=None):
        self.provider = provider
        self.bufs = {}
        self.written = set()
        self.counts = {}            # id(p) -> contributions received this step (a weight applied k times per step receives k)
        self.on_sink = None         # optional hook(key, count): the data-parallel trainer checks that no contribution follows a sent bucket

    def sink(self, p):
        k = id(p)
        if k not in self.bufs:
            self.bufs[k] = self.provider(p) if self.provider else torch.empty_like(p, dtype=torch.float32)
        acc = 1 if k in self.written else 0
        self.written.add(k)
        self.counts[k] = self.counts.get(k, 0) + 1
        if self.on_sink is not None:
            self.on_sink(k, self.counts[k])
        return self.bufs[k], acc

    def get(self, p):
        return self.bufs.get(id(p)) if id(p) in self.written else None


class PackCache:
    """Persistent packed weight panels + the device job table that refreshes all of them in one launch."""

    def __init__(self):
        self.entries = {}          # key -> (panel tensor, PackDesc)
        self.jobs = []             # capi.PackJob (host copies)
        self.keep = []             # weights referenced by the table (pointers must stay valid)
        self.table = None
        self.retired = []          # tables replaced by add(): a hipGraph captured earlier still launches them (their jobs, panels and weights all stay alive) - never freed
        self.version = 0           # bumped by add(): holders of captured graphs can tell that the job set has grown since their capture
        self.dt = None

    # a cache hangs off a module parameter on the nn.Module surface: pickling / deep-copying the module must not drag device job tables along
    def __reduce__(self):
        return (PackCache, ())

    def __deepcopy__(self, memo):
        return PackCache()

    def add(self, key, w, wp, d):
        self.entries[key] = (wp, d)
        j = capi.PackJob()
        j.w, j.wp = w.data_ptr(), wp.data_ptr()
        C.memmove(C.byref(j.d), C.byref(d), C.sizeof(capi.PackDesc))
        self.jobs.append(j)
        self.keep.append(w)
        if self.table is not None:
            self.retired.append((self.table, self.bstart))
        self.table = None
        self.version += 1
        self.dt = key[4]

    def refresh(self):
        """Repack every cached panel from the current fp32 master weights (call once per step, before forward)."""
        if not self.jobs:
            return
        if self.table is None:
            self.table, self.bstart, self.nblocks = _job_table(capi.PackJob, self.jobs, [call.pn2_pack_blocks(C.byref(j.d)) for j in self.jobs])
        call.pn2_pack_weights_multi(self.dt, _p(self.table), _p(self.bstart), len(self.jobs), self.nblocks, _stream())


class BnFoldCache:
    """Folded eval-mode BatchNorm rows (scale, shift) of a model's layers, persistent across forwards and refreshed from the live gamma / beta / running
    statistics by ONE table-driven launch per forward (pn2_bn_eval_prepare_multi) - inside a captured inference graph that is one node instead of one per layer."""

    def __init__(self):
        self.entries = {}          # (id(bn), Cp, gw, gwp) -> [2][Cp] fp32 rows scale, shift
        self.jobs, self.keep, self.table = [], [], None
        self.retired, self.version = [], 0          # as in PackCache: replaced tables stay allocated for the graphs that captured them

    def __reduce__(self):
        return (BnFoldCache, ())

    def __deepcopy__(self, memo):
        return BnFoldCache()

    def stale(self):
        return any(j.gamma != bn.weight.data_ptr() or j.running_mean != bn.running_mean.data_ptr() for j, bn in zip(self.jobs, self.keep))

    def add(self, key, bn, par, bd, off=0):
        """register `bn` (rows par[0][off:], par[1][off:]); `key` -> par for the lookup (several BatchNorms may share one [2][sum C] block)"""
        self.entries[key] = par
        j = capi.BnPrepJob()
        j.gamma, j.beta, j.running_mean, j.running_var = bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr()
        j.scale, j.shift = par[0][off:].data_ptr(), par[1][off:].data_ptr()
        C.memmove(C.byref(j.d), C.byref(bd), C.sizeof(capi.BnDesc))
        self.jobs.append(j)
        self.keep.append(bn)
        if self.table is not None:
            self.retired.append((self.table, self.bstart))
        self.table = None
        self.version += 1

    def refresh(self):
        if not self.jobs:
            return
        if self.table is None:
            self.table, self.bstart, self.nblocks = _job_table(capi.BnPrepJob, self.jobs, [(j.d.Cp + 255) // 256 for j in self.jobs])
        call.pn2_bn_eval_prepare_multi(_p(self.table), _p(self.bstart), len(self.jobs), self.nblocks, _stream())


def _job_table(struct, jobs, blocks):
    """-> (device copy of the job array, device prefix sums of the per-job workgroup counts, total workgroups)."""
    if min(blocks) < 1:
        raise RuntimeError("job with an unsupported geometry in a table-driven launch")
    arr = (struct * len(jobs))(*jobs)
    table = torch.frombuffer(bytearray(bytes(memoryview(arr).cast("B"))), dtype=torch.uint8).cuda()
    start = [0]
    for b in blocks:
        start.append(start[-1] + b)
    return table, torch.tensor(start, dtype=torch.int32).cuda(), start[-1]


class StepArena:
    """Bump allocator for everything an Engine allocates during one training step.  The first step runs on the torch allocator
    and measures the footprint; later steps carve the same sequence of buffers out of one persistent block, so every activation,
    gradient and scratch buffer has the SAME address in every step (eager or inside a captured hipGraph).  That is what lets the
    deferred, table-driven launches (GradQueue) reuse their device job tables, and it takes the allocator off the eager path.
    Sized for a 288 GB part: nothing is recycled inside a step."""

    def __init__(self):
        self.buf, self.off, self.need = None, 0, 0

    def begin_step(self, dev):
        want = self.need
        if want and (self.buf is None or self.buf.numel() < want) and not torch.cuda.is_current_stream_capturing():
            self.buf = None
            self.buf = torch.empty(want, dtype=torch.uint8, device=dev)
        self.off, self.need = 0, 0

    def alloc(self, shape, dtype, dev):
        n = dtype.itemsize
        for d in shape:
            n *= d
        na = (n + 255) // 256 * 256
        self.need += na
        if self.buf is not None and self.off + na <= self.buf.numel():
            t = self.buf[self.off:self.off + n].view(dtype).view(shape)
            self.off += na
            return t
        return torch.empty(shape, dtype=dtype, device=dev)


TUNE_COLD = os.environ.get("PN2_TUNE_COLD", "1") == "1"             # the tuners time every candidate behind a cache-evicting fill (tests/conftest.py switches it off)
WGRAD_WGS = 640               # pixel splits: workgroups a single wgrad aims at ...
WGRAD_SLAB_MB = 24            # ... within this many MB of fp32 slabs
WGRAD_MIX = float(os.environ.get("PN2_WGRAD_MIX", "2"))          # > 0: the pointwise and the k x k weight-gradient jobs of the 128 x 256 tile share ONE table-driven launch (GradQueue._build; the value = how much faster the k x k list is consumed); 0: two launches
WGRAD_SLAB_CAP = float(os.environ.get("PN2_WGRAD_SLAB_CAP", "0.35"))   # table-driven wgrad: fp32 slab bytes of a job <= this x its operand bytes (GradQueue.table_splits; 0: off)
WGRAD_ROTATE = os.environ.get("PN2_WGRAD_ROTATE", "1") == "1"      # table-driven wgrad: a job's pixel splits start on the XCD after the previous job's last one
_THRASH = {}


def _thrash():
    """Overwrite 512 MB (more than the L2s and the 256 MB memory-side cache) so that the next kernel starts from HBM."""
    dev = torch.cuda.current_device()
    t = _THRASH.get(dev)
    if t is None:
        t = _THRASH[dev] = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    t.fill_(1)


DEFER_COLSUM = os.environ.get("PN2_DEFER_COLSUM", "1") == "1"
SMALL_CIN_DGRAD = os.environ.get("PN2_SMALL_CIN_DGRAD", "1") == "1"  # strided convs with <= 4 input channels: per-pixel data gradient
GRAD_ALIAS = os.environ.get("PN2_GRAD_ALIAS", "1") == "1"             # sums whose second operand has no other consumer share its gradient storage
SPLITK = os.environ.get("PN2_SPLITK", "1") == "1"                     # split-K for few-row / long-contraction convs
TUNE_LOG = None          # debugging: a list that receives (key, code, source) of every conv-tile lookup
KS2 = os.environ.get("PN2_KS2", "1") == "1"                           # tuner candidates with intra-workgroup split-K (conv_dma_gemm_ks2): one wave of tiles, long K loop
KSPLIT_MINK = 4096            # shortest contraction that is split (M <= 4096 rows; shorter ones lose to the partial-tile traffic)
PATCH_DGRAD = os.environ.get("PN2_PATCH_DGRAD", "1") == "1"         # kernel == stride convs: data gradient as GEMM + depth-to-space
FUSE_BIAS = os.environ.get("PN2_FUSE_BIAS", "1") == "1"             # bias of BN-less convs / nn.Linear in the GEMM epilogue (PN2_CONV_BIAS)
BNB_EPILOGUE = os.environ.get("PN2_BNB_EPILOGUE", "1") == "1"       # BatchNorm-backward statistics in the epilogue of the dgrad GEMM that completes dy
LOCKSTEP = os.environ.get("PN2_LOCKSTEP", "1") == "1"               # independent chains (RFB branches, stage-block branches) share table-driven launches
DW_COLSUM = os.environ.get("PN2_DW_COLSUM", "1") == "1"             # PVTv2 Mlp: fc1's bias gradient from the depth-wise conv's data-gradient walk (no second read of that gradient)
MASKED_STORE = os.environ.get("PN2_MASKED_STORE", "1") == "1"       # ... which then stores dy * [y > 0] for BN + residual + ReLU outputs (residual gradient aliases it)
POOL_BWD_QUAD = os.environ.get("PN2_POOL_BWD_QUAD", "1") == "1"  # ... and its backward without the full-resolution gradient tensor (pn2_pool_bn_bwd_reduce / _apply); 0: pool-backward launch + the generic BatchNorm passes
RES_STATS = os.environ.get("PN2_RES_STATS", "1") == "1"          # the downsample BatchNorm's backward sums from the dgrad epilogue that forms bn3's masked gradient (pn2_conv_ep.c): no reduce pass
MUL_BWD = os.environ.get("PN2_MUL_BWD", "1") == "1"              # backward of a product (the aggregation's): both operand gradients from one pass (pn2_mul_bwd); 0: two pn2_binary launches
POOL_FOLD = os.environ.get("PN2_POOL_FOLD", "1") == "1"          # Bottle2neck stage blocks: the backward of the downsample branch's AvgPool2d(2, 2) rides in conv1's dgrad epilogue (pn2_conv_ep.pool) - no pool-backward launch
POOL_FUSE = os.environ.get("PN2_POOL_FUSE", "1") == "1"          # the stem's bn1 -> ReLU -> MaxPool as one op: the 176 x 176 BatchNorm output is never written (conv_bn_act(pool=True))
TEE_CONCAT = os.environ.get("PN2_TEE_CONCAT", "1") == "1"         # Bottle2neck: conv1 + bn1 + ReLU writes its pass-through slice into the concat buffer as well (pn2_affine_act_tee); False: a copy launch
EVAL_FUSE = True          # eval mode: conv + BatchNorm (+ ReLU) (+ residual) in ONE launch (pn2_conv_gemm_affine); tests switch it off to compare with the two-launch path
ZERO_CROP_SKIP = os.environ.get("PN2_ZERO_CROP_SKIP", "1") == "1"   # K = 1 DSRA: the crop maps' gradient is identically zero - skip the adjoints of the resamples that made them


class GradQueue:
    """Deferred weight-gradient work of one training step.  A conv's wgrad and the split-K slab reduction that follows it only
    feed the optimizer, so the backward pass queues them (dy / x stay alive in the step arena) and `flush()` runs them as a few
    table-driven launches: one pn2_conv_wgrad_multi per kernel instantiation, then one pn2_wgrad_reduce_multi.  Device job tables
    are cached per flush segment and reused for as long as the queued pointers are unchanged (always, with a StepArena)."""

    def __init__(self, defer_wgrad=True, slab_cap=None):
        self.defer_wgrad = defer_wgrad
        self.slab_cap = slab_cap          # None: WGRAD_SLAB_CAP; 0: the tuner's split counts
        self.slabs = {}
        self.cache = {}                   # segment index -> (signature, launches)
        self.ccache = {}                  # same for the engine's queued column-sum finalisations
        self.begin_step()

    def begin_step(self):
        self.seg = 0
        self.wjobs, self.rjobs, self.keep = [], [], []
        self.uses, self.levels = {}, {}

    def slab(self, key, shape, dev):
        # a weight applied twice in one step (CAB's shared fc1 / fc2 on the average- and max-pooled vectors) needs a slab per use: the
        # deferred wgrads of both uses run before either reduction
        n = self.uses.get(key, 0)
        self.uses[key] = n + 1
        key = key + (n,)
        t = self.slabs.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            if t is not None:
                raise RuntimeError("wgrad slab geometry changed between steps; build a new Trainer for a new input shape")
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("run eager steps before capturing (persistent wgrad slabs are allocated then)")
            t = self.slabs[key] = torch.empty(shape, dtype=torch.float32, device=dev)
        return t

    def table_splits(self, nsplit, M, chans, wd, esize=2):
        """Pixel splits of a wgrad that runs inside a table-driven launch.  The tuner times a conv ALONE, where splits are what fills the chip; inside a table the
        other jobs do that, and every split costs a fp32 slab (Rp x Kp x 4 B written, then re-read by the reduce: 3.9 GB of the step's 46 GB with the tuned counts).
        The slabs of a job are therefore capped at WGRAD_SLAB_CAP (0.35) x the bytes of its own operands (dy + x): the long-contraction / few-pixel layers go from
        3..16 splits to 1..2, the many-pixel layers keep theirs.  Sweep at bs = 32 (ms per step): no cap 14.51, 1: 14.27, 0.5: 14.22, 0.35: 14.19, 0.25: 14.19, 0.18: 14.18,
        0.125: 14.53 (the greedy XCD rotation of _build is what makes few splits pay: without it they pile onto the first XCDs)."""
        cap = self.slab_cap if self.slab_cap is not None else WGRAD_SLAB_CAP
        if cap <= 0:
            return nsplit
        # esize: bytes per operand element (round 6: the fp32 paths passed through here with bf16's 2 - half the cap they were meant to have; fp32fast 42.3 -> 41.7 ms)
        return min(nsplit, max(1, int(cap * M * chans * esize // (wd.Rp * wd.Kp * 4))))

    def add_wgrad(self, dt, dy, x_ptr, x_keep, slab, wd, nsplit, flops=0):
        self.wjobs.append((dt, dy.data_ptr(), x_ptr.value, slab.data_ptr(), wd, nsplit, flops))
        self.keep.append((dy, x_keep))

    def add_reduce(self, slab, gw, rd, nsplit, accumulate):
        # a weight applied more than once in a step (CAB's shared fc1 / fc2): the k-th contribution to a gradient goes into reduction level k of
        # this segment - one pn2_wgrad_reduce_multi per level, launched in order - instead of cutting the segment (the wgrads themselves
        # write private slabs and need no order)
        lvl = self.levels.get(gw.data_ptr(), 0)
        if accumulate and lvl == 0:
            self.flush()          # the earlier contribution was not queued here (an immediate kernel): it must be finished first
            lvl = 0
        self.levels[gw.data_ptr()] = lvl + 1
        self.rjobs.append((slab.data_ptr(), gw.data_ptr(), rd, nsplit, accumulate, lvl))

    def _build(self):
        launches = []
        groups = {}
        for dt, dy, x, slab, wd, ns, fl in self.wjobs:
            v = call.pn2_conv_wgrad_variant(dt, C.byref(wd))
            if v < 0:
                raise RuntimeError("unsupported wgrad geometry")
            groups.setdefault((dt, v), []).append((dy, x, slab, wd, ns, fl))
        order = lambda js: sorted(js, key=lambda j: -((j[3].N * j[3].OH * j[3].OW + j[4] - 1) // j[4]) * j[3].KH * j[3].KW)
        mixed = {}
        if WGRAD_MIX > 0:
            # the k x k (variant 12) and pointwise (13) jobs of the 128 x 256 tile in ONE table (14): chains of ~121 stages that leave the memory system idle next to jobs that stream
            # at the HBM rate.  Job ranges interleaved so that a CU holds a workgroup of each kind: the k x k list is consumed WGRAD_MIX times as fast as the pointwise list
            # (its chains must not start late), both in their longest-first order.
            for dt in {k[0] for k in groups}:
                a, b = groups.get((dt, 12)), groups.get((dt, 13))
                if a and b:
                    a, b = order(a), order(b)
                    real = lambda j: j[4] * call.pn2_conv_wgrad_blocks(C.byref(j[3]), j[4]) // (8 * ((j[4] + 7) // 8))          # workgroups that do work (the rest of a tile's 8 XCD slots exit)
                    ta, tb = sum(real(j) for j in a) or 1, sum(real(j) for j in b) or 1
                    out, ia, ib, ca, cb = [], 0, 0, 0, 0
                    while ia < len(a) or ib < len(b):
                        if ib >= len(b) or (ia < len(a) and ca / (ta / WGRAD_MIX) <= cb / tb):
                            out.append(a[ia]); ca += real(a[ia]); ia += 1
                        else:
                            out.append(b[ib]); cb += real(b[ib]); ib += 1
                    del groups[(dt, 12)], groups[(dt, 13)]
                    groups[(dt, 14)] = out
                    mixed[(dt, 14)] = True
        for (dt, v), js in sorted(groups.items()):
            # longest workgroups first (pixels per split x taps): the hardware hands out workgroups in index order, so the short jobs fill the
            # tail of the launch instead of the long ones stretching it
            if (dt, v) not in mixed:
                js = order(js)
            arr, load = [], [0] * 8
            for dy, x, slab, wd, ns, fl in js:
                j = capi.WgradJob()
                j.dy, j.x, j.slab, j.nsplit = dy, x, slab, ns
                if WGRAD_ROTATE:
                    # split s of a job runs on XCD (s + rot) % 8 (all its tiles: one L2 fetches that split's dy / x slice once).  Jobs arrive longest first; each takes
                    # the rotation that keeps the most loaded XCD lowest (work of a split ~ its steps x the job's tiles; equal tile shape inside one table)
                    w = ((wd.N * wd.OH * wd.OW + 31) // 32 + ns - 1) // ns * (call.pn2_conv_wgrad_blocks(C.byref(wd), ns) // (8 * ((ns + 7) // 8)))
                    per = [(ns - x_ + 7) // 8 for x_ in range(8)]              # splits on logical XCD slot x_
                    best = min(range(8), key=lambda r: (max(load[(x_ + r) & 7] + per[x_] * w for x_ in range(8)), r))
                    for x_ in range(8):
                        load[(x_ + best) & 7] += per[x_] * w
                    j.rot = best
                C.memmove(C.byref(j.d), C.byref(wd), C.sizeof(capi.WgradDesc))
                arr.append(j)
            table, bstart, nblocks = _job_table(capi.WgradJob, arr, [call.pn2_conv_wgrad_blocks(C.byref(j.d), j.nsplit) for j in arr])
            launches.append(("w", dt, v, table, bstart, len(arr), nblocks, sum(j[5] for j in js)))
        for lvl in sorted({j[5] for j in self.rjobs}):
            arr = []
            for slab, gw, rd, ns, acc, l_ in self.rjobs:
                if l_ != lvl:
                    continue
                j = capi.ReduceJob()
                j.slab, j.gw, j.nsplit, j.accumulate = slab, gw, ns, acc
                C.memmove(C.byref(j.d), C.byref(rd), C.sizeof(capi.PackDesc))
                arr.append(j)
            table, bstart, nblocks = _job_table(capi.ReduceJob, arr, [call.pn2_wgrad_reduce_blocks(C.byref(j.d)) for j in arr])
            launches.append(("r", 0, 0, table, bstart, len(arr), nblocks, 0))
        return launches

    def flush(self):
        """Launch what was queued since the previous flush."""
        if not self.wjobs and not self.rjobs:
            return
        sig = (tuple(j[:4] + (j[5],) for j in self.wjobs), tuple(j[:2] + j[3:] for j in self.rjobs))
        hit = self.cache.get(self.seg)
        if hit is None or hit[0] != sig:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("run two eager steps before capturing (the deferred-launch tables are built then)")
            hit = self.cache[self.seg] = (sig, self._build())
        st = _stream()
        for kind, dt, v, table, bstart, njobs, nblocks, flops in hit[1]:
            if kind == "w":
                capi.WORK.update(flops=flops, tag="", shape=f"variant{v} jobs{njobs}")
                call.pn2_conv_wgrad_multi(dt, v, _p(table), _p(bstart), njobs, nblocks, st)
            else:
                call.pn2_wgrad_reduce_multi(_p(table), _p(bstart), njobs, nblocks, st)
        self.seg += 1
        self.wjobs, self.rjobs, self.keep = [], [], []
        self.levels = {}
