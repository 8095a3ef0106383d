"""PVTv2 encoder entry points (reference: lib/pvtv2.py:197-436).

Not built yet: the transformer encoder (LayerNorm, spatial-reduction attention, depth-wise conv MLP) is the
next scope row after the Res2Net path.  The factories exist so that
`from lib.pvtv2 import pvt_v2_b2` resolves; calling them raises instead of silently using a PyTorch fallback.
"""


def _todo(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name}: PVTv2 encoder kernels are not implemented yet (next scope row); no PyTorch fallback by design")
    f.__name__ = name
    return f


pvt_v2_b0, pvt_v2_b1, pvt_v2_b2, pvt_v2_b3, pvt_v2_b4, pvt_v2_b5 = (_todo(f"pvt_v2_b{i}") for i in range(6))
