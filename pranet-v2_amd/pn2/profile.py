"""Live per-kernel timing for bench.py: one instrumented eager step with HIP events around every C-ABI launch.

The kernels are launched on torch's current stream, so torch.cuda.Event (a hipEvent recorded on that stream) brackets
exactly the launch it surrounds.  Work is *algorithmic*: logical conv FLOPs (2*M*Cout*Cin*KH*KW for forward, dgrad and
wgrad alike) and minimum tensor bytes for the streaming kernels — not padded/physical counts.
"""
import time

import torch

from . import capi
from .capi import F32

_EL = {0: 4, 1: 2}
ACTUAL = {}     # bytes the fused DSRA tail kernels really move (next to SURVEY's algorithmic 17*S figure)


def _bytes(name, a):
    """minimum HBM bytes of one launch from its C arguments (streaming kernels only)"""
    try:
        if name == "pn2_affine_act":
            dt_in, _, _, dt_out, _, _, M, C, _, _, res = a[:11]
            return M * C * (_EL[dt_in] * (2 if res.value else 1) + _EL[dt_out])
        if name == "pn2_bn_bwd_reduce":
            dt, dt_dy, _, _, Cdy, y = a[0], a[1], a[2], a[3], a[4], a[5]
            M, Cp = a[10], a[11]
            return M * (Cdy * _EL[dt_dy] + Cp * _EL[dt] * (2 if y.value else 1))
        if name == "pn2_bn_bwd_apply":
            dt, dt_dy, Cdy, y, M, Cp, dres = a[0], a[1], a[4], a[5], a[10], a[11], a[17]
            return M * (Cdy * _EL[dt_dy] + Cp * _EL[dt] * (2 + (1 if y.value else 0) + (1 if dres.value else 0)))
        if name == "pn2_affine_act_sum":
            dt, M, C = a[0], a[5], a[6]
            return M * C * _EL[dt] * 4                      # raw + the other operand in, y and y + operand out
        if name in ("pn2_bilinear_fwd", "pn2_bilinear_bwd", "pn2_avgpool_fwd", "pn2_avgpool_bwd"):
            dt, N, H, W, C, OH, OW = a[0], a[5], a[6], a[7], a[8], a[9], a[10]
            return N * C * _EL[dt] * (H * W + OH * OW)
        if name in ("pn2_maxpool3x3s2_fwd", "pn2_maxpool3x3s2_bwd"):
            dt, N, H, W, C, OH, OW = a[0], a[6], a[7], a[8], a[9], a[10], a[11]
            return N * C * (_EL[dt] * (H * W + OH * OW) + OH * OW)
        if name in ("pn2_dsra_tail_fwd", "pn2_dsra_tail_bwd"):
            # SURVEY 8(d) fully-fused figure, 17*S per image for K=1 (S = OH*OW*4 B): forward writes the 2P maps and reads the mask,
            # backward accounts for re-reading the 2P maps (this implementation recomputes them from the low-res logits instead)
            d = a[0]._obj
            S = d.N * d.OH * d.OW * 4
            ACTUAL["tail"] = ACTUAL.get("tail", 0) + (S * (2 * d.P + 2) if name.endswith("fwd") else S * 2)   # what the kernels really move: fwd writes 2P maps +
            return S * (2 * d.P + 1) if name.endswith("fwd") else S * 2 * d.P                                  # reads mask, weit; bwd reads mask, weit only
        if name == "pn2_dsra_tail_fwd_bwd":          # the two above in one call: 17*S per image at P = 4
            d = a[0]._obj
            S = d.N * d.OH * d.OW * 4
            ACTUAL["tail"] = ACTUAL.get("tail", 0) + S * (2 * d.P + 4)
            return S * (4 * d.P + 1)
        if name == "pn2_structure_loss_fwd":
            P, N, HW = a[2], a[9], a[10]
            return N * HW * 4 * (2 * P + 2)
        if name == "pn2_structure_loss_bwd":
            P, N, HW = a[3], a[9], a[10]
            return N * HW * 4 * (4 * P + 2)
        if name == "pn2_clamp_adam":
            return int(a[4]) * 4 * 7
        if name == "pn2_binary":
            dt, M, C = a[0], a[8], a[9]
            return M * C * _EL[dt] * 3
        if name == "pn2_copy":
            return a[6] * a[7] * (_EL[a[0]] + _EL[a[3]])
        if name == "pn2_nchw_to_nhwc":
            return a[4] * a[6] * (a[5] * 4 + a[7] * _EL[a[0]])
    except Exception:
        return 0
    return 0


def _shape(name, a):
    try:
        if name == "pn2_affine_act":
            return f"M{a[6]} C{a[7]} ldx{a[2]} ldy{a[5]} res{1 if a[10].value else 0}"
        if name in ("pn2_bn_bwd_reduce", "pn2_bn_bwd_apply"):
            return f"M{a[10]} C{a[11]} lddy{a[3]} ldx{a[9]} y{1 if a[5].value else 0}"
        if name == "pn2_wgrad_reduce":
            d = a[2]._obj
            return f"{d.Cin}->{d.Cout} k{d.KH}x{d.KW} Rp{d.Rp} Kp{d.Kp} ns{a[3]}"
        if name in ("pn2_bilinear_fwd", "pn2_bilinear_bwd", "pn2_avgpool_fwd", "pn2_avgpool_bwd"):
            return f"dt{a[0]} N{a[5]} {a[6]}x{a[7]} C{a[8]} -> {a[9]}x{a[10]}"
        if name == "pn2_binary":
            return f"dt{a[0]} op{a[1]} M{a[8]} C{a[9]} acc{a[10]}"
        if name == "pn2_copy":
            return f"dt{a[0]}->{a[3]} M{a[6]} C{a[7]} lds{a[2]} ldd{a[5]} acc{a[8]}"
        if name in ("pn2_bn_finalize", "pn2_bn_bwd_finalize"):
            return f"nblk{a[2]} Cp{a[3]._obj.Cp}"
        if name == "pn2_dwconv":
            return f"{a[4]}x{a[5]}x{a[6]}x{a[7]} k{a[8]} flip{a[9]} acc{a[10]} stats{1 if a[11].value else 0}"
        if name == "pn2_dwconv_wgrad":
            return f"{a[4]}x{a[5]}x{a[6]}x{a[7]} k{a[8]}"
        if name == "pn2_dwconv3x3":
            return f"{a[6]}x{a[7]}x{a[8]}x{a[9]} flip{a[10]} acc{a[11]} gelu{1 if a[5].value else 0}"
        if name == "pn2_dwconv3x3_wgrad":
            return f"{a[5]}x{a[6]}x{a[7]}x{a[8]} gelu{1 if a[9].value else 0}"
        if name == "pn2_attn_fwd":
            return f"B{a[8]} Nq{a[9]} Nkv{a[10]} h{a[11]}"
        if name == "pn2_attn_bwd":
            return f"B{a[16]} Nq{a[17]} Nkv{a[18]} h{a[19]}"
        if name in ("pn2_layernorm_fwd", "pn2_layernorm_bwd"):
            return f"M{a[5]} C{a[6]}"
        if name == "pn2_colsum":
            return f"M{a[3]} C{a[4]}"
        if name == "pn2_colsum_finalize":
            return f"nblk{a[1]} C{a[2]}"
        if name == "pn2_gather_sum":
            return f"n{a[6]} k{a[7]}"
        if name in ("pn2_upsample_nearest2x", "pn2_upsample_nearest2x_bwd"):
            return f"{a[3]}x{a[4]}x{a[5]}x{a[6]}"
    except Exception:
        pass
    return ""


class Recorder:
    def __init__(self):
        self.rows = []          # (name, flops, bytes, e0, e1)
        self.saved = {}

    def __enter__(self):
        lib = capi.load()
        for name in capi.SIGNATURES:
            if name in capi._VALUE_FUNCS:
                continue
            fn = getattr(lib, name)

            def wrapped(*a, _fn=fn, _name=name):
                by_ = 0
                if _name in ("pn2_conv_gemm", "pn2_conv_gemm_ep", "pn2_conv_gemm_multi", "pn2_conv_wgrad", "pn2_conv_wgrad_multi"):
                    fl, tag, shape = capi.WORK.pop("flops", 0), capi.WORK.pop("tag", ""), capi.WORK.pop("shape", "")
                    es = 4 if int(a[0]) == capi.F32 else 2
                    by_ = sum(_conv_bytes(s_, es) for s_ in (capi.WORK.pop("shapes", None) or [shape]))
                else:
                    fl, tag, shape = 0, "", _shape(_name, a)
                    by_ = capi.WORK.pop("bytes", 0) if _name.endswith("_multi") else _bytes(_name, a)      # lock-step launches: summed over their jobs
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                if _name in capi._MMA_FUNCS and int(a[0]) == capi.F32:          # fp32fast: the translation capi.call applies at the ABI boundary
                    a = (capi.F32_MMA,) + tuple(a[1:])
                e0.record()
                rc = _fn(*a)
                e1.record()
                if rc != 0:
                    raise RuntimeError(f"{_name} failed with status {rc}")
                if _name == "pn2_conv_gemm_ep":
                    _name = "pn2_conv_gemm"           # same kernel symbols; the tag (:dgrad) keeps the family together
                self.rows.append((_name + tag, fl, by_, e0, e1, shape))
            self.saved[name] = capi.call.__dict__.get(name)
            setattr(capi.call, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, old in self.saved.items():
            if old is None:
                capi.call.__dict__.pop(name, None)
            else:
                setattr(capi.call, name, old)

    def summary(self, detail=False):
        torch.cuda.synchronize()
        agg = {}
        for name, fl, by, e0, e1, shape in self.rows:
            if detail and shape:
                name = name + " " + shape
            d = agg.setdefault(name, {"ms": 0.0, "launches": 0, "flops": 0, "bytes": 0})
            d["ms"] += e0.elapsed_time(e1); d["launches"] += 1; d["flops"] += fl; d["bytes"] += by if not fl else 0
        return agg


def _conv_bytes(shape, es):
    """algorithmic bytes of one conv GEMM launch from its shape string 'Cin->Cout kHxW sS dD NxOHxOW': input + output activations + weights, each
    moved once (the dgrad of the same conv moves the same tensors the other way)"""
    try:
        io, k, s_, _, dims = shape.split(" ")[:5]
        cin = int(io.split("->")[0]); cout = sum(int(v) for v in io.split("->")[1].split("+"))
        kh, kw = (int(v) for v in k[1:].split("x")); st = int(s_[1:])
        n, oh, ow = (int(v) for v in dims.split("x"))
        return es * (n * oh * st * ow * st * cin + n * oh * ow * cout + cin * cout * kh * kw)
    except Exception:
        return 0


PMC_FILE = "r06_pmc_hbm_traffic.json"       # the current round's summary (tools/prof_pmc.sh)


def _pmc_traffic(symbols, config):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC summary (profiles/, made by tools/pmc_summary.py from separate
    FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-graph`).  Only reported when that summary was recorded for THIS workload (model, batch,
    size, dtype); None otherwise - a number measured on another configuration would describe a different run."""
    import json, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles", PMC_FILE)
    try:
        doc = json.load(open(path))
        ks = doc["kernels"]
    except Exception:
        return None
    if any(doc.get("config", {}).get(k) != v for k, v in config.items()):
        return None
    rows = [ks[k] for k in symbols if k in ks]
    if not rows:
        return None
    n = sum(r["launches"] for r in rows)
    return {"bytes_per_launch": int(sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows) / n), "source": "profiles/" + PMC_FILE,
            "commit": doc.get("commit"), "step_total_bytes": doc.get("step_total_bytes")}


def _conv_class(shape):
    """layer class of a conv launch from its shape string 'Cin->Cout kHxW sS dD NxHxW' (per-class TF/s)"""
    try:
        io, k = shape.split(" ")[0], shape.split(" ")[1]
        cin = int(io.split("->")[0]); cout = sum(int(v) for v in io.split("->")[1].split("+"))
        if k == "k1x1":
            return "1x1 wide (>= 416 ch)" if max(cin, cout) >= 416 else "1x1 narrow (< 416 ch)"
        if k == "k3x3" and cin == cout and cin in (26, 52, 104, 208):
            return "3x3 Res2Net branch (26/52/104/208 ch)"
        if k == "k5x5":
            return "5x5 (ra4)"
        return "3x3 / 1xk / kx1 other (stem, RFB, aggregation, heads)"
    except Exception:
        return "other"


def measure_step(trainer, x, m, dtype, config=None):
    peak_tf = 2500.0 if dtype == "bf16" else 157.3
    ACTUAL.clear()
    trainer.step(x, m)                       # eager warm-up (allocator, caches)
    # Keep the GPU queue full during the instrumented step: a spin kernel first, so the host runs ahead and every event marker
    # executes back to back with the kernel it brackets (otherwise each elapsed time would include the idle-queue dispatch latency).
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); torch.cuda._sleep(10_000_000); e1.record(); torch.cuda.synchronize()
    per_cycle_ms = e0.elapsed_time(e1) / 1e7
    t0 = time.perf_counter(); trainer.step(x, m); host_ms = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
    with Recorder() as rec:
        torch.cuda._sleep(int(min(1.5 * host_ms + 20.0, 2000.0) / max(per_cycle_ms, 1e-9)))
        trainer.step(x, m)
    agg = rec.summary()
    detail = rec.summary(detail=True)
    kernels = {}
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1]["ms"]):
        e = {"ms": round(d["ms"], 3), "launches": d["launches"]}
        if d["flops"]:
            e["TFLOPs"] = round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 2)
        elif d["bytes"]:
            e["GBps"] = round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1)
        kernels[name] = e
    # dominant MFMA kernel family = the implicit-GEMM conv kernels (forward and dgrad launch the same kernel symbols, conv_dma_gemm<...> /
    # conv_gather_gemm<...>, so the rocprofv3 kernel stats of those symbols are what this line has to agree with)
    fam = [d for n, d in agg.items() if n.startswith("pn2_conv_gemm")]
    fl, ms, nl = sum(d["flops"] for d in fam), sum(d["ms"] for d in fam), sum(d["launches"] for d in fam)
    ach = fl / (ms * 1e-3) / 1e12
    roofline = {"kernel": "pn2_conv_gemm + pn2_conv_gemm_multi (fwd+dgrad incl. the BatchNorm-backward statistics epilogues; symbols conv_dma_gemm[_tab|_ks|_tab_ks2]<*>, conv_gather_gemm[_tab]<*>)", "bound": "mfma", "achieved": round(ach, 2),
                "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach / peak_tf, 4), "traffic": None, "launches": nl,
                "avg_launch_us": round(1e3 * ms / nl, 2), "algorithmic_gflop_per_launch": round(fl / nl / 1e9, 3)}
    # the same family against the roofline of EACH launch: a launch cannot finish before max(flops / MFMA peak, algorithmic bytes / HBM peak) - the
    # 1x1 convs of the stem / layer1 / layer2 (247 808 .. 991 232 rows, <= 256 channels: 30-80 flop per byte against a machine balance of 312) are
    # bound by HBM, not by the matrix cores, so the MFMA fraction alone understates how close those launches are to what the chip allows
    t_floor = t_mfma = t_hbm = 0.0
    n_hbm = 0
    for name, fl_, by_, _e0, _e1, _shape in rec.rows:
        if name.startswith("pn2_conv_gemm") and fl_:
            a_, b_ = fl_ / (peak_tf * 1e12), by_ / 8e12
            t_floor += max(a_, b_); t_mfma += a_; t_hbm += b_; n_hbm += b_ > a_
    roofline["per_launch_roofline"] = {"frac": round(t_floor / (ms * 1e-3), 4), "floor_ms": round(t_floor * 1e3, 3), "mfma_floor_ms": round(t_mfma * 1e3, 3),
                                       "hbm_floor_ms": round(t_hbm * 1e3, 3), "hbm_bound_launches": n_hbm, "measured_ms": round(ms, 3),
                                       "note": "sum over launches of max(flops/2.5 PFLOP/s, (in+out+weights bytes)/8 TB/s) / measured time"}
    # forward and dgrad launches apart: the dgrad launches carry the BatchNorm-backward statistics (extra operand reads, epilogue arithmetic) that
    # used to be separate pn2_bn_bwd_reduce passes, so the family's combined rate is not comparable with a round in which they did not
    for tag_ in (":fwd", ":dgrad"):
        sel = [d_ for n_, d_ in agg.items() if n_.startswith("pn2_conv_gemm") and n_.endswith(tag_)]
        if sel:
            fl_, ms_ = sum(d_["flops"] for d_ in sel), sum(d_["ms"] for d_ in sel)
            roofline["forward" if tag_ == ":fwd" else "dgrad_with_bn_statistics"] = {
                "achieved": round(fl_ / (ms_ * 1e-3) / 1e12, 2), "frac": round(fl_ / (ms_ * 1e-3) / 1e12 / peak_tf, 4), "ms": round(ms_, 3),
                "launches": sum(d_["launches"] for d_ in sel)}
    traffic = _pmc_traffic(("conv_dma_gemm", "conv_gather_gemm", "conv_dma_gemm_tab", "conv_gather_gemm_tab", "conv_dma_gemm_ks", "conv_dma_gemm_tab_ks2"), config or {})
    if traffic is not None:
        roofline["traffic"] = traffic["bytes_per_launch"]
        roofline["traffic_source"] = traffic["source"]
        roofline["traffic_commit"] = traffic["commit"]
        roofline["step_hbm_bytes_pmc"] = traffic["step_total_bytes"]
    # per layer class (forward + dgrad launches of the conv GEMM family)
    cls = {}
    for name, d in detail.items():
        if name.startswith("pn2_conv_gemm") and d["flops"]:
            c = cls.setdefault(_conv_class(name.split(" ", 1)[1] if " " in name else ""), [0.0, 0, 0])
            c[0] += d["ms"]; c[1] += d["launches"]; c[2] += d["flops"]
    roofline["conv_classes"] = {k: {"ms": round(v[0], 3), "launches": v[1], "TFLOPs": round(v[2] / (v[0] * 1e-3) / 1e12, 1), "frac": round(v[2] / (v[0] * 1e-3) / 1e12 / peak_tf, 4)}
                                for k, v in sorted(cls.items(), key=lambda kv: -kv[1][0])}
    # the BatchNorm family: HBM-bound streaming passes + the per-layer finalisation launches
    bn_names = ("pn2_affine_act", "pn2_affine_act_sum", "pn2_affine_multi", "pn2_bn_finalize", "pn2_bn_finalize_multi", "pn2_bn_bwd_reduce", "pn2_bn_bwd_reduce_multi",
                "pn2_bn_bwd_finalize", "pn2_bn_bwd_finalize_seg", "pn2_bn_bwd_finalize_multi", "pn2_bn_bwd_apply", "pn2_bn_bwd_apply_multi")
    bn = [agg[k] for k in bn_names if k in agg]
    if bn:
        by, ms_, nl_ = sum(t["bytes"] for t in bn), sum(t["ms"] for t in bn), sum(t["launches"] for t in bn)
        roofline["hbm_batchnorm"] = {"kernels": [k for k in bn_names if k in agg], "bound": "hbm", "achieved": round(by / (ms_ * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(by / (ms_ * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_MB": round(by / 1e6, 1), "ms": round(ms_, 3), "launches": nl_}
    wg = [d for n, d in agg.items() if n.startswith("pn2_conv_wgrad")]
    if wg:
        wfl, wms, wnl = sum(d["flops"] for d in wg), sum(d["ms"] for d in wg), sum(d["launches"] for d in wg)
        roofline["wgrad"] = {"kernel": "pn2_conv_wgrad_multi (symbols conv_wgrad_tab<*>, conv_wgrad_dma_tab<*>)", "bound": "mfma",
                             "achieved": round(wfl / (wms * 1e-3) / 1e12, 2), "peak": peak_tf, "unit": "TFLOP/s",
                             "frac": round(wfl / (wms * 1e-3) / 1e12 / peak_tf, 4), "launches": wnl, "avg_launch_us": round(1e3 * wms / wnl, 2)}
    names = (("pn2_dsra_tail_fwd_bwd",) if "pn2_dsra_tail_fwd_bwd" in agg else
             ("pn2_dsra_tail_fwd", "pn2_dsra_tail_bwd") if "pn2_dsra_tail_fwd" in agg else ("pn2_structure_loss_fwd", "pn2_structure_loss_bwd"))
    tail = [agg[k] for k in names if k in agg]
    if tail:
        by = sum(t["bytes"] for t in tail); ms = sum(t["ms"] for t in tail)
        roofline["hbm_dsra_tail"] = {"kernels": list(names), "bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                     "frac": round(by / (ms * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_MB": round(by / 1e6, 1), "us": round(ms * 1e3, 1),
                                     "actual_MB": round(ACTUAL.get("tail", 0) / 1e6, 1), "actual_frac": round(ACTUAL.get("tail", 0) / (ms * 1e-3) / 1e9 / 8000.0, 4)}
    return {"roofline": roofline, "kernels": kernels}
