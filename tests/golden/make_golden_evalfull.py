#!/usr/bin/env python3
"""Golden vectors for the remaining metrics of the reference's eval_for_testAllInOne (binary_seg/eval.py:18-66): Sm, wFm, meanEm next to meanDic / meanIoU /
mae, on small synthetic prediction / ground-truth pairs.  Runs ONLY in the build container (imports /root/reference); only data is written (eval_full.npz).
Also stored for the tie-breaking of the exact Euclidean feature transform: scipy's own (distance, row index, column index) on a map with many equidistant
sites, and the reference's `Et` (error at the nearest foreground pixel) on the blob case.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_evalfull.py
"""
import os, sys, warnings
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE); sys.path.insert(0, ROOT)
from _ref_import import import_reference            # noqa: E402

R = import_reference()
import importlib                                     # noqa: E402
cwd = os.getcwd(); os.chdir("/root/reference/binary_seg")
try:
    ev = importlib.import_module("eval")
    ef = importlib.import_module("utils.eval_functions")
finally:
    os.chdir(cwd)
from scipy.ndimage import distance_transform_edt    # noqa: E402

METRICS = ["meanDic", "meanIoU", "wFm", "Sm", "meanEm", "mae"]


def cases():
    rng = np.random.default_rng(5)
    H, Wd = 64, 80
    yy, xx = np.mgrid[0:H, 0:Wd]
    blob = (((yy - 30) / 14.0) ** 2 + ((xx - 42) / 22.0) ** 2 < 1).astype(np.float64)
    smooth = np.clip(255 * (0.75 * blob + 0.25 * rng.random((H, Wd))) + 12 * rng.standard_normal((H, Wd)), 0, 255).astype(np.uint8)
    out = {"blob": (smooth, blob), "zero_pred": (np.zeros((H, Wd), np.uint8), blob), "exact": ((255 * blob).astype(np.uint8), blob),
           "full": (np.full((H, Wd), 255, np.uint8), np.ones((H, Wd)))}
    # two blobs + isolated pixels on a non-square map, a soft prediction: many equidistant nearest pixels, all four S-measure quadrants populated
    H2, W2 = 100, 77
    yy, xx = np.mgrid[0:H2, 0:W2]
    g2 = ((((yy - 25) / 11.0) ** 2 + ((xx - 20) / 9.0) ** 2 < 1) | (((yy - 70) / 15.0) ** 2 + ((xx - 55) / 13.0) ** 2 < 1)).astype(np.float64)
    g2[5, 70] = 1; g2[95, 3] = 1; g2[50, 38] = 1
    p2 = np.clip(255 * (0.6 * g2 + 0.4 * rng.random((H2, W2))) , 0, 255).astype(np.uint8)
    out["two"] = (p2, g2)
    # headline test size with a random prediction
    g3 = (rng.random((352, 352)) < 0.02).astype(np.float64)
    g3[100:180, 120:260] = 1
    p3 = rng.integers(0, 256, (352, 352), dtype=np.uint8)
    out["rand352"] = (p3, g3)
    return out


if __name__ == "__main__":
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag, (pred, gt) in cases().items():
            vals = ev.eval_for_testAllInOne({"metrics": METRICS}, pred, (gt * 255).astype(np.uint8) if gt.max() > 0 else gt.astype(np.uint8))
            # (eval.py:24-26: gt arrives as the mask image's 0/255 array and is binarised with > 0.5)
            out[tag + "_pred"] = pred; out[tag + "_gt"] = gt.astype(np.float32)
            out[tag + "_vals"] = np.array([float(v) for v in vals], dtype=np.float64)
            pm, gm = pred.astype(np.float64) / 255, (gt > 0.5).astype(np.float64)
            thr = np.linspace(1, 0, 256)
            Ecurve = []
            for t in thr:
                b = np.zeros_like(pm); b[pm >= t] = 1
                Ecurve.append(ef.EnhancedMeasure(b, gm))
            out[tag + "_E"] = np.array(Ecurve)
            print(tag, dict(zip(METRICS, out[tag + "_vals"])))
    # scipy's feature transform on a tie-rich map (tests/test_oracle_golden.py pins oracle.edt_nearest to it index by index)
    rng = np.random.default_rng(9)
    tie = np.zeros((40, 53))
    tie[::7, ::9] = 1; tie[20, :] = 0; tie[3:6, 30:33] = 1
    tie[rng.integers(0, 40, 12), rng.integers(0, 53, 12)] = 1
    dst, idx = distance_transform_edt(1 - tie, return_indices=True)
    out["tie_gt"] = tie.astype(np.float32); out["tie_dst"] = dst; out["tie_ri"] = idx[0].astype(np.int32); out["tie_rj"] = idx[1].astype(np.int32)
    p, g = cases()["blob"]
    E = np.abs(p.astype(np.float64) / 255 - g)
    d2, i2 = distance_transform_edt(1 - g, return_indices=True)
    Et = E.copy(); Et[g != 1] = Et[i2[:, g != 1][0], i2[:, g != 1][1]]
    out["blob_Et"] = Et
    np.savez_compressed(os.path.join(HERE, "eval_full.npz"), **out)
    print("wrote eval_full.npz", len(out), "arrays", os.path.getsize(os.path.join(HERE, "eval_full.npz")) // 1024, "KiB")
