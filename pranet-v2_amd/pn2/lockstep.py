"""Lock-step execution of independent chains: one table-driven launch per kernel kind and position instead of one launch per chain.

PraNet-V2 is full of small independent chains of identical structure - the three RFB modules and their three branches each (pranet.py:46-83),
the three parallel 3x3 convs of a Res2Net stage block (Res2Net_v1b.py:66-69).  Each link is a 5-10 us kernel on a tensor of a few MB, so nine
chains launched one after the other cost nine times the latency of one.  `Engine.lockstep(key, fns)` runs the chains (`lanes`) in RECORDING mode:
the C-ABI launches they make are not issued but turned into job structs; afterwards the recorder walks the positions - launch 0 of every lane,
launch 1 of every lane, ... - and issues the jobs of equal kind at a position as ONE pn2_*_multi launch from a device job table (tables are cached
per (region, position, kind) and reused while the recorded pointers are unchanged: always, with a step arena).  A lane's own order is preserved and
lanes are independent, so the arithmetic is bit-identical to the unbatched run.  Launches without a table-driven form are issued one by one at
their position.  The backward closures of a lock-step region run in lock step as well.
"""
import ctypes as C

import torch

from . import capi
from .capi import call


def _v(p):
    """ctypes pointer argument -> integer address (None / NULL -> None)"""
    if p is None:
        return None
    v = p.value if isinstance(p, C.c_void_p) else p
    return v or None


def _copy_struct(dst, src_byref):
    C.memmove(C.byref(dst), src_byref, C.sizeof(dst))


# ---------------------------------------------------------------------------------------------- converters: recorded call -> (kind, job struct) or None
def _conv_job(a, with_ep):
    res = None
    if with_ep == 2:               # pn2_conv_gemm_affine: scale / shift ride in psum / psq, the residual in ep.a.y
        dt, in_, wp, out, psum, psq, res, ld_res, d, _st = a
        ep = None
    elif with_ep:
        dt, in_, wp, out, d, ep, _st = a
        psum = psq = None
    else:
        dt, in_, wp, out, psum, psq, d, _st = a
        ep = None
    j = capi.ConvJob()
    j.in_, j.wp, j.out, j.psum, j.psq = _v(in_), _v(wp), _v(out), _v(psum), _v(psq)
    _copy_struct(j.d, d)
    if ep is not None:
        _copy_struct(j.ep, ep)
    if _v(res) is not None:
        j.ep.a.y, j.ep.a.ld_y = _v(res), ld_res
    with_ep = 1 if with_ep == 1 else 0
    tile = call.pn2_conv_gemm_tile(dt, C.byref(j.d))
    if tile < 0:
        return None
    bm, bn = tile >> 8, tile & 255
    nb = call.pn2_conv_gemm_job_blocks(dt, C.byref(j), bm, bn)
    if nb < 1:
        return None
    return ("conv", dt, bm, bn, 1 if with_ep else 0), j, nb


def _bnfin_job(a):
    psum, psq, nblk, d, gamma, beta, rm, rv, scale, shift, mean, invstd, _st = a
    j = capi.BnFinJob()
    j.psum, j.psq, j.gamma, j.beta, j.running_mean, j.running_var = _v(psum), _v(psq), _v(gamma), _v(beta), _v(rm), _v(rv)
    j.scale, j.shift, j.mean, j.invstd, j.nblk = _v(scale), _v(shift), _v(mean), _v(invstd), nblk
    _copy_struct(j.d, d)
    nb = call.pn2_bn_finalize_job_blocks(C.byref(j))
    return (("bnfin",), j, nb) if nb >= 1 else None


def _affine_job(a, with_sum):
    j = capi.AffineJob()
    if with_sum:
        dt, x, ld_x, y, ld_y, M, Cc, scale, shift, relu, add, ld_add, y2, ld_y2, _st = a
        j.add, j.ld_add, j.y2, j.ld_y2 = _v(add), ld_add, _v(y2), ld_y2
        res, ld_res = None, 0
    else:
        dt, x, ld_x, dt_out, y, ld_y, M, Cc, scale, shift, res, ld_res, relu, _st = a
        if dt == capi.BF16 and dt_out == capi.F32 and MIXED:          # K-channel head maps: the element-wise kernel in table form
            j.x, j.ld_x, j.y, j.ld_y, j.M, j.C, j.scale, j.shift, j.res, j.ld_res, j.relu = _v(x), ld_x, _v(y), ld_y, M, Cc, _v(scale), _v(shift), _v(res), ld_res, relu
            nb = call.pn2_affine_job_blocks(dt | MULTI_F32, C.byref(j))
            return (("affine", dt | MULTI_F32), j, nb) if nb >= 1 else None
        if dt_out != dt or _v(scale) is None or _v(shift) is None:
            return None
    j.x, j.ld_x, j.y, j.ld_y, j.M, j.C, j.scale, j.shift, j.res, j.ld_res, j.relu = _v(x), ld_x, _v(y), ld_y, M, Cc, _v(scale), _v(shift), _v(res), ld_res, relu
    nb = call.pn2_affine_job_blocks(dt, C.byref(j))
    return (("affine", dt), j, nb) if nb >= 1 else None


def _bnbfin_job(a, seg):
    j = capi.BnBFinJob()
    if seg:
        segs, d, gamma, invstd, dgamma, dbeta, acc, coef, _st = a
        _copy_struct(j.sg, segs)
        _copy_struct(j.d, d)
    else:
        p1, p2, nblk, d, gamma, invstd, dgamma, dbeta, acc, coef, _st = a
        _copy_struct(j.d, d)
        j.sg.nseg = 1
        j.sg.c0[0], j.sg.nblk[0], j.sg.ldp[0], j.sg.p1[0], j.sg.p2[0] = 0, nblk, (j.d.ldp or j.d.Cp), _v(p1), _v(p2)
    j.gamma, j.invstd, j.dgamma, j.dbeta, j.coef, j.accumulate = _v(gamma), _v(invstd), _v(dgamma), _v(dbeta), _v(coef), acc
    nb = call.pn2_bn_bwd_finalize_job_blocks(C.byref(j))
    return (("bnbfin",), j, nb) if nb >= 1 else None


def _bnapply_job(a):
    dt, dt_dy, dy, ld_dy, Cdy, y, ld_y, dt_y, x, ld_x, M, Cp, mean, invstd, coef, dx, ld_dx, dres, ld_dres, dres_accum, msc, msh, r6, _st = a
    if dt == capi.BF16 and dt_dy == capi.F32 and MIXED and _v(msc) is None and (_v(y) is None or dt_y == dt):          # fp32 gradient of a bf16 layer (K-channel heads)
        j = capi.BnApplyJob()
        j.dy, j.ld_dy, j.y, j.ld_y, j.x, j.ld_x, j.M, j.Cp, j.pad_ = _v(dy), ld_dy, _v(y), ld_y, _v(x), ld_x, M, Cp, Cdy
        j.mean, j.invstd, j.coef, j.dx, j.ld_dx, j.dres, j.ld_dres, j.dres_accum, j.r6 = _v(mean), _v(invstd), _v(coef), _v(dx), ld_dx, _v(dres), ld_dres, dres_accum, r6
        nb = call.pn2_bn_bwd_apply_job_blocks(dt | MULTI_F32, C.byref(j))
        return (("bnapply", dt | MULTI_F32), j, nb) if nb >= 1 else None
    if dt != dt_dy or Cdy != Cp or (_v(y) is not None and dt_y != dt):
        return None
    j = capi.BnApplyJob()
    j.dy, j.ld_dy, j.y, j.ld_y, j.x, j.ld_x, j.M, j.Cp = _v(dy), ld_dy, _v(y), ld_y, _v(x), ld_x, M, Cp
    j.mean, j.invstd, j.coef, j.dx, j.ld_dx, j.dres, j.ld_dres, j.dres_accum = _v(mean), _v(invstd), _v(coef), _v(dx), ld_dx, _v(dres), ld_dres, dres_accum
    j.msc, j.msh, j.r6 = _v(msc), _v(msh), r6
    nb = call.pn2_bn_bwd_apply_job_blocks(dt, C.byref(j))
    lean = not j.y and not j.dres           # jobs without a stored-activation mask / residual gradient share the register-lean launch (PN2_MULTI_LEAN)
    return (("bnapply", dt | (0x100 if lean else 0)), j, nb) if nb >= 1 else None


def _bnreduce_job(a):
    dt, dt_dy, dy, ld_dy, Cdy, y, ld_y, dt_y, x, ld_x, M, Cp, mean, invstd, p1, p2, nblk, msc, msh, r6, _st = a
    if dt == capi.BF16 and dt_dy == capi.F32 and MIXED and (_v(y) is None or dt_y == dt):
        j = capi.BnReduceJob()
        j.dy, j.ld_dy, j.y, j.ld_y, j.x, j.ld_x, j.M, j.Cp, j.pad_ = _v(dy), ld_dy, _v(y), ld_y, _v(x), ld_x, M, Cp, Cdy
        j.mean, j.invstd, j.p1, j.p2, j.nblk, j.msc, j.msh, j.r6 = _v(mean), _v(invstd), _v(p1), _v(p2), nblk, _v(msc), _v(msh), r6
        nb = call.pn2_bn_bwd_reduce_job_blocks(dt | MULTI_F32, C.byref(j))
        return (("bnreduce", dt | MULTI_F32), j, nb) if nb >= 1 else None
    if dt != dt_dy or Cdy != Cp or (_v(y) is not None and dt_y != dt):
        return None
    j = capi.BnReduceJob()
    j.dy, j.ld_dy, j.y, j.ld_y, j.x, j.ld_x, j.M, j.Cp = _v(dy), ld_dy, _v(y), ld_y, _v(x), ld_x, M, Cp
    j.mean, j.invstd, j.p1, j.p2, j.nblk, j.msc, j.msh, j.r6 = _v(mean), _v(invstd), _v(p1), _v(p2), nblk, _v(msc), _v(msh), r6
    nb = call.pn2_bn_bwd_reduce_job_blocks(dt, C.byref(j))
    return (("bnreduce", dt), j, nb) if nb >= 1 else None


def _copy_job(a):
    dt_in, src, ld_s, dt_out, dst, ld_d, M, Cc, acc, _st = a
    if dt_in != dt_out:
        return None
    j = capi.CopyJob()
    j.src, j.dst, j.ld_s, j.ld_d, j.M, j.C, j.accumulate = _v(src), _v(dst), ld_s, ld_d, M, Cc, acc
    nb = call.pn2_copy_job_blocks(dt_in, C.byref(j))
    return (("copy", dt_in), j, nb) if nb >= 1 else None


CONVERT = {
    "pn2_copy": _copy_job,
    "pn2_conv_gemm": lambda a: _conv_job(a, False),
    "pn2_conv_gemm_ep": lambda a: _conv_job(a, True),
    "pn2_conv_gemm_affine": lambda a: _conv_job(a, 2),
    "pn2_bn_finalize": _bnfin_job,
    "pn2_affine_act": lambda a: _affine_job(a, False),
    "pn2_affine_act_sum": lambda a: _affine_job(a, True),
    "pn2_bn_bwd_finalize": lambda a: _bnbfin_job(a, False),
    "pn2_bn_bwd_finalize_seg": lambda a: _bnbfin_job(a, True),
    "pn2_bn_bwd_apply": _bnapply_job,
    "pn2_bn_bwd_reduce": _bnreduce_job,
}
# every other launch a lane may make is issued on its own, at its position in the lane (value-returning helpers and the table launches themselves excepted)
SINGLE = tuple(n for n in capi.SIGNATURES if n not in CONVERT and n not in capi._VALUE_FUNCS and not n.endswith("_multi"))

_ACTIVE = []        # stack of recording Lockstep objects (innermost last)
import os
_NOBATCH = set(os.environ.get("PN2_LOCKSTEP_NOBATCH", "").split(","))      # debugging: launch names that are never batched
MULTI_F32 = 0x200                                                             # PN2_MULTI_F32DY / PN2_MULTI_F32OUT (pn2.h)
MIXED = os.environ.get("PN2_LOCKSTEP_MIXED", "1") == "1"                    # bf16 layers with fp32 K-channel maps (the head convs of the DSRA stages) join table-driven launches too
COALESCE = os.environ.get("PN2_LOCKSTEP_COALESCE", "1") == "1"              # runs of BatchNorm finalize launches of one lane share a position (see Lockstep.emit)


def pause():
    """Context manager: launches made inside are issued immediately (the tile tuner times real launches)."""
    class _P:
        def __enter__(s):
            s.saved = list(_ACTIVE)
            for r in s.saved:
                r._unpatch()
            del _ACTIVE[:]

        def __exit__(s, *exc):
            for r in s.saved:
                r._patch()
            _ACTIVE.extend(s.saved)
            return False
    return _P()


class Lockstep:
    def __init__(self, key, cache):
        self.key, self.cache = key, cache
        self.lanes = []
        self._saved = None

    # ------------------------------------------------------------------ recording
    def _patch(self):
        self._saved = {}
        for name in list(CONVERT) + list(SINGLE):
            self._saved[name] = call.__dict__.get(name)
            orig = getattr(call, name)                      # the checked C call (or the profiler's wrapper around it)

            def rec(*a, _name=name, _orig=orig):
                work = dict(capi.WORK)                       # profiling annotation of this launch (algorithmic flops / tag), consumed at emission
                capi.WORK.clear()
                self.lanes[-1].append((_name, a, _orig, work))
            setattr(call, name, rec)

    def _unpatch(self):
        for name, old in self._saved.items():
            if old is None:
                call.__dict__.pop(name, None)
            else:
                setattr(call, name, old)
        self._saved = None

    def lane(self):
        ls = self

        class _Lane:
            def __enter__(s):
                ls.lanes.append([])
                ls._patch()
                _ACTIVE.append(ls)

            def __exit__(s, *exc):
                _ACTIVE.remove(ls)
                ls._unpatch()
                return False
        return _Lane()

    # ------------------------------------------------------------------ emission
    def emit(self):
        from .engine import _job_table, _p, _stream
        # a lane's position = ONE recorded launch, or a run of consecutive BatchNorm finalize launches of distinct BatchNorms (the six finalize calls behind the fused
        # 1x1 reducer GEMM of an RFB module, forward and backward: each depends on the GEMM / reduce in front of the run, none on another member of it) - the run shares
        # one position, i.e. one table-driven launch for 3 lanes x 6 jobs instead of six launches of 3 jobs (round 6: -10 launches, -0.07 ms per step)
        lanes = [self._coalesce(l) for l in self.lanes] if COALESCE else [[[e] for e in l] for l in self.lanes]
        npos = max((len(l) for l in lanes), default=0)
        st = _stream()
        capturing = torch.cuda.is_current_stream_capturing()
        for pos in range(npos):
            groups, singles = {}, []
            for lane in lanes:
                if pos >= len(lane):
                    continue
                for name, a, orig, work in lane[pos]:
                    conv = CONVERT.get(name) if name not in _NOBATCH else None
                    job = conv(a) if conv is not None else None
                    if job is None:
                        singles.append((a, orig, work))
                    else:
                        groups.setdefault(job[0], []).append((job[1], job[2], a, orig, work, name))
            for a, orig, work in singles:
                capi.WORK.clear(); capi.WORK.update(work)
                orig(*a)
            for kind, jobs in groups.items():
                if len(jobs) == 1:
                    capi.WORK.clear(); capi.WORK.update(jobs[0][4])
                    jobs[0][3](*jobs[0][2])
                    continue
                structs = [j[0] for j in jobs]
                sig = b"".join(bytes(memoryview(s_).cast("B")) for s_ in structs)
                ck = (self.key, pos, kind)
                hit = self.cache.get(ck)
                if hit is None or hit[0] != sig:
                    if capturing:
                        raise RuntimeError("run two eager steps before capturing (the lock-step launch tables are built then)")
                    table, bstart, total = _job_table(type(structs[0]), structs, [j[1] for j in jobs])
                    hit = self.cache[ck] = (sig, table, bstart, total)
                _, table, bstart, total = hit
                n = len(jobs)
                if kind[0] == "conv":
                    capi.WORK.clear()
                    capi.WORK.update(flops=sum(j[4].get("flops", 0) for j in jobs), tag=jobs[0][4].get("tag", ""), shape=f"lockstep x{n} tile {kind[2]}x{kind[3]}",
                                     shapes=[j[4].get("shape", "") for j in jobs])
                    epbits = kind[4]
                    if epbits:       # what the BatchNorm-backward epilogues of these jobs need in LDS (pn2.h: bits 1..4 of `ep`)
                        for s_ in structs:
                            sa = s_.ep.a.mode & capi.BNB_STATS
                            epbits |= (2 if sa else 0) | (4 if sa and (s_.ep.a.mode & capi.BNB_MASK_Y) else 0) | (8 if s_.d.flags & capi.CONV_ACCUM else 0) \
                                      | (16 if s_.ep.b.out and (s_.ep.b.mode & capi.BNB_STATS) else 0)
                    call.pn2_conv_gemm_multi(kind[1], kind[2], kind[3], epbits, _p(table), _p(bstart), n, total, st)
                    continue
                capi.WORK.clear()
                capi.WORK.update(bytes=self._bytes(jobs))
                if kind[0] == "bnfin":
                    call.pn2_bn_finalize_multi(_p(table), _p(bstart), n, total, st)
                elif kind[0] == "affine":
                    call.pn2_affine_multi(kind[1], _p(table), _p(bstart), n, total, st)
                elif kind[0] == "bnbfin":
                    call.pn2_bn_bwd_finalize_multi(_p(table), _p(bstart), n, total, st)
                elif kind[0] == "bnapply":
                    call.pn2_bn_bwd_apply_multi(kind[1], _p(table), _p(bstart), n, total, st)
                elif kind[0] == "bnreduce":
                    call.pn2_bn_bwd_reduce_multi(kind[1], _p(table), _p(bstart), n, total, st)
                elif kind[0] == "copy":
                    call.pn2_copy_multi(kind[1], _p(table), _p(bstart), n, total, st)
        self.lanes = []

    _RUN_OUT = {"pn2_bn_finalize": 8, "pn2_bn_bwd_finalize": 9, "pn2_bn_bwd_finalize_seg": 7}          # index of the launch's private output row (scale / coef)

    @classmethod
    def _coalesce(cls, lane):
        out = []
        for e in lane:
            k = cls._RUN_OUT.get(e[0])
            if (k is not None and out and out[-1][0][0] == e[0] and e[0] not in _NOBATCH
                    and all(_v(p[1][k]) != _v(e[1][k]) for p in out[-1])):
                out[-1].append(e)
            else:
                out.append([e])
        return out

    def _bytes(self, jobs):
        """minimum HBM bytes of a batched streaming launch for the profiler (sum over its jobs, same accounting as pn2.profile._bytes)"""
        from .profile import _bytes
        return sum(_bytes(name, a) for _j, _nb, a, _orig, _w, name in jobs)
