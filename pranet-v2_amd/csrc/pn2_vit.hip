// pn2_vit.hip — the non-GEMM kernels of the PVTv2 encoder (reference: /root/reference/binary_seg/lib/pvtv2.py):
//   LayerNorm forward / backward            nn.LayerNorm at pvtv2.py:71,119,126,169,224-247 (eps 1e-5 / 1e-6)
//   column sums                             bias gradients of nn.Linear / biased nn.Conv2d (:19,22,62-65,70,167)
//   depth-wise 3x3 conv (+bias, +GELU)      DWConv :363-374 followed by nn.GELU in Mlp.forward :42-49
//   spatial-reduction attention             Attention.forward :90-111 (head_dim 64, <= 256 reduced key/value tokens)
// Tokens [B, N, C] of the reference are NHWC pixels here (N = H*W), so no transposes are needed anywhere.
// The Linear layers themselves run on the implicit-GEMM conv kernels (1x1) of pn2_conv.hip.
// All kernels are deterministic (fixed-order reductions, no floating-point atomics).
#include <cstdint>
#include <cstdlib>
#include "pn2_common.h"
#include "pn2_dw.h"
#include "../../include/pn2.h"

namespace {

template <typename T> __device__ __forceinline__ void ldv(const T* p, float* f) { TT<T>::unpack(*reinterpret_cast<const uint4*>(p), f); }
template <typename T> __device__ __forceinline__ void stv(T* p, const float* f) { *reinterpret_cast<uint4*>(p) = TT<T>::pack(f); }

// ------------------------------------------------------------------------------------------ LayerNorm
// A row (token) is handled by LPR lanes of one wave (LPR = power of two >= C/VEC, <= 64); a lane owns up to NV channel vectors.
constexpr int LN_NV = 4;

template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_k(const T* __restrict__ x, int ld_x, T* __restrict__ y, int ld_y, int M, int C, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float eps, float* __restrict__ mean, float* __restrict__ rstd, int LPR) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V, rpw = 64 / LPR, lane = threadIdx.x & 63, lr = lane % LPR, slot = lane / LPR;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * rpw + slot;
    const bool live = row < M;
    float v[LN_NV][V];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < LN_NV; ++k) {
        const int cv = lr + k * LPR;
        if (live && cv < CV) {
            ldv<T>(x + (size_t)row * ld_x + cv * V, v[k]);
#pragma unroll
            for (int e = 0; e < V; ++e) s += v[k][e];
        }
    }
    for (int o = 1; o < LPR; o <<= 1) s += __shfl_xor(s, o);
    const float mu = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < LN_NV; ++k) {
        const int cv = lr + k * LPR;
        if (live && cv < CV) {
#pragma unroll
            for (int e = 0; e < V; ++e) { const float d = v[k][e] - mu; q += d * d; }
        }
    }
    for (int o = 1; o < LPR; o <<= 1) q += __shfl_xor(q, o);
    const float rs = rsqrtf(q / (float)C + eps);
    if (live && lr == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
    for (int k = 0; k < LN_NV; ++k) {
        const int cv = lr + k * LPR;
        if (live && cv < CV) {
            float o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = (v[k][e] - mu) * rs * gamma[cv * V + e] + beta[cv * V + e];
            stv<T>(y + (size_t)row * ld_y + cv * V, o);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma ;  partial dgamma / dbeta rows per block
template <typename T, int NV>
__global__ __launch_bounds__(256) void ln_bwd_k(const T* __restrict__ dy, int ld_dy, const T* __restrict__ x, int ld_x, int M, int C, const float* __restrict__ gamma,
                                                const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx, int ld_dx, int acc_dx,
                                                float* __restrict__ pg, float* __restrict__ pb, int rows_per_blk, int LPR) {
    constexpr int V = TT<T>::VEC;
    extern __shared__ float sh[];            // [2][slots][C] for the cross-slot reduction
    const int CV = C / V, rpw = 64 / LPR, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane % LPR, slot = wid * rpw + lane / LPR;
    const int nslot = 4 * rpw;
    const int r0 = blockIdx.x * rows_per_blk;
    int r1 = r0 + rows_per_blk; if (r1 > M) r1 = M;
    float ag[NV][V], ab[NV][V], gm[NV][V];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; const int c = (lr + k * LPR) * V + e; gm[k][e] = c < C ? gamma[c] : 0.f; }
    // the next row's vectors are fetched before the current row is reduced: the rows of a slot form a dependent chain (load -> two lane
    // reductions -> store), and without the prefetch every link pays the full memory latency
    uint4 nd[NV], nx[NV];
    float nmu = 0.f, nrs = 0.f;
    {
        const int row = r0 + slot;
        if (row < r1) {
            nmu = mean[row]; nrs = rstd[row];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int cv = lr + k * LPR;
                if (cv < CV) { nd[k] = *reinterpret_cast<const uint4*>(dy + (size_t)row * ld_dy + cv * V); nx[k] = *reinterpret_cast<const uint4*>(x + (size_t)row * ld_x + cv * V); }
            }
        }
    }
    for (int rb = r0; rb < r1; rb += nslot) {           // uniform trip count: the shuffles below need every lane
        const int row = rb + slot;
        const bool live = row < r1;
        uint4 cd[NV], cx[NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) { cd[k] = nd[k]; cx[k] = nx[k]; }
        const float mu = live ? nmu : 0.f, rs = live ? nrs : 0.f;
        {
            const int rown = row + nslot;
            if (rown < r1) {
                nmu = mean[rown]; nrs = rstd[rown];
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int cv = lr + k * LPR;
                    if (cv < CV) { nd[k] = *reinterpret_cast<const uint4*>(dy + (size_t)rown * ld_dy + cv * V); nx[k] = *reinterpret_cast<const uint4*>(x + (size_t)rown * ld_x + cv * V); }
                }
            }
        }
        float g[NV][V], xh[NV][V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int cv = lr + k * LPR;
            if (live && cv < CV) {
                float d[V], xv[V];
                TT<T>::unpack(cd[k], d);
                TT<T>::unpack(cx[k], xv);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    xh[k][e] = (xv[e] - mu) * rs; g[k][e] = d[e] * gm[k][e];
                    s1 += g[k][e]; s2 += g[k][e] * xh[k][e];
                    ag[k][e] += d[e] * xh[k][e]; ab[k][e] += d[e];
                }
            }
        }
        for (int o = 1; o < LPR; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
        const float c1 = s1 / (float)C, c2 = s2 / (float)C;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int cv = lr + k * LPR;
            if (live && cv < CV) {
                float o[V];
                T* d = dx + (size_t)row * ld_dx + cv * V;
                if (acc_dx) ldv<T>(d, o);
#pragma unroll
                for (int e = 0; e < V; ++e) { const float t = rs * (g[k][e] - c1 - xh[k][e] * c2); o[e] = acc_dx ? o[e] + t : t; }
                stv<T>(d, o);
            }
        }
    }
    // block partial of dgamma / dbeta: every slot owns the same channels -> sum the slots in a fixed order through LDS
    float* sg = sh; float* sb = sh + nslot * C;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int cv = lr + k * LPR;
        if (cv < CV) {
#pragma unroll
            for (int e = 0; e < V; ++e) { sg[slot * C + cv * V + e] = ag[k][e]; sb[slot * C + cv * V + e] = ab[k][e]; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f, b = 0.f;
        for (int s_ = 0; s_ < nslot; ++s_) { a += sg[s_ * C + c]; b += sb[s_ * C + c]; }
        pg[(size_t)blockIdx.x * C + c] = a; pb[(size_t)blockIdx.x * C + c] = b;
    }
}

// out[c] (+)= sum_b p[b*ld + c]   (fixed order; double accumulation).  32 columns x 8 row lanes per block, 4 loads in flight per thread.
__device__ __forceinline__ void colsum_finalize_body(const float* __restrict__ p, int nblk, int C, int ld, float* __restrict__ out, int accumulate, int bloc) {
    __shared__ double sh[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5, c = bloc * 32 + cl;
    double s = 0.0;
    if (c < C) {
        int r = rl;
        for (; r + 24 < nblk; r += 32) {
            const float a0 = p[(size_t)r * ld + c], a1 = p[(size_t)(r + 8) * ld + c], a2 = p[(size_t)(r + 16) * ld + c], a3 = p[(size_t)(r + 24) * ld + c];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        }
        for (; r < nblk; r += 8) s += (double)p[(size_t)r * ld + c];
    }
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += sh[k][cl];
        out[c] = accumulate ? out[c] + (float)t : (float)t;
    }
}
__global__ __launch_bounds__(256) void colsum_finalize_k(const float* __restrict__ p, int nblk, int C, int ld, float* __restrict__ out, int accumulate) {
    colsum_finalize_body(p, nblk, C, ld, out, accumulate, blockIdx.x);
}
// many finalisations in one launch, from a device job table (the parameter-gradient sums of a step only feed the optimizer, so the
// backward pass queues them): block b works on job j = find_job(bstart, b), local block b - bstart[j]
__global__ __launch_bounds__(256) void colsum_finalize_multi_k(const pn2_colsum_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_colsum_job j = jobs[jb];
    colsum_finalize_body(j.partial, j.nblk, j.C, j.ld, j.out, j.accumulate, blockIdx.x - bstart[jb]);
}

// partial[blk][C] = sum over the block's rows of dy[row][c]
template <typename T>
__device__ __forceinline__ void colsum_body(const T* __restrict__ dy, int ld, int M, int C, float* __restrict__ partial, int rows_per_blk, int CVP, int bloc) {
    constexpr int V = TT<T>::VEC;
    extern __shared__ float sh[];            // [R][CVP*V]
    const int CV = C / V, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int r0 = bloc * rows_per_blk;
    int r1 = r0 + rows_per_blk; if (r1 > M) r1 = M;
    for (int cvb = 0; cvb < CV; cvb += CVP) {
        const int cv = cvb + cvl;
        float a[V];
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] = 0.f;
        if (cv < CV) {
            for (int m = r0 + rl; m < r1; m += R) {
                float d[V];
                ldv<T>(dy + (size_t)m * ld + cv * V, d);
#pragma unroll
                for (int e = 0; e < V; ++e) a[e] += d[e];
            }
        }
#pragma unroll
        for (int e = 0; e < V; ++e) sh[(rl * CVP + cvl) * V + e] = a[e];
        __syncthreads();
        if (rl == 0 && cv < CV) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float s = 0.f;
                for (int r = 0; r < R; ++r) s += sh[(r * CVP + cvl) * V + e];
                partial[(size_t)bloc * C + cv * V + e] = s;
            }
        }
        __syncthreads();
    }
}
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T* __restrict__ dy, int ld, int M, int C, float* __restrict__ partial, int rows_per_blk, int CVP) {
    colsum_body<T>(dy, ld, M, C, partial, rows_per_blk, CVP, blockIdx.x);
}
// the column sums of many tensors in one launch (device job table, as colsum_finalize_multi_k): the bias gradients of a step
template <typename T>
__global__ __launch_bounds__(256) void colsum_multi_k(const pn2_colsum_in_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_colsum_in_job j = jobs[jb];
    colsum_body<T>((const T*)j.dy, j.ld, j.M, j.C, j.partial, j.rows, j.cvp, blockIdx.x - bstart[jb]);
}

// ------------------------------------------------------------------------------------------ depth-wise 3x3 (+bias, +GELU)
// erf without branches, on channel PAIRS (v_pk_fma_f32 / v_pk_mul_f32: one instruction per two channels).  The device library's erff is ~150 instructions
// with data-dependent branches; two of them per element made the depth-wise conv + GELU walks (and the weight-gradient walk that forms dz = dy * gelu'(z))
// instruction-bound.  Two polynomial pieces, both evaluated, one selected:
//   |a| <= 0.921875: erf(a) = a + a * p(a^2)                       (degree 5 in a^2)
//   |a| >  0.921875: erf(a) = sign(a) * (1 - exp(-(t + t * q(t)))), t = min(|a|, 4)   (degree 7: -log(erfc(t)) / t - 1 on [0.921875, 4])
// Chebyshev fits (coefficients rounded to fp32); max |error| 7.2e-8, max relative error 8.4e-8 against float64 erf over [-6, 6] - the last bit of fp32,
// as the library function.  NaN propagates through the first piece, +-inf saturates through the second.  Every operation is written out (no contraction
// left to the compiler), so the scalar wrappers below return the bits of the packed form.
typedef float dwf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dwf2 dw2(float v) { return dwf2{v, v}; }
__device__ __forceinline__ dwf2 dwfma(dwf2 a, dwf2 b, dwf2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ dwf2 dw_exp2(dwf2 v) { return dwf2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; }
__device__ __forceinline__ dwf2 erf_nb2(dwf2 a) {
#pragma clang fp contract(off)
    const dwf2 t = dwf2{fabsf(a.x), fabsf(a.y)}, s = a * a;
    dwf2 p = dw2(-0x1.3b07e8p-11f);
    p = dwfma(p, s, dw2(0x1.477bf2p-8f)); p = dwfma(p, s, dw2(-0x1.b69768p-6f)); p = dwfma(p, s, dw2(0x1.ce1b58p-4f));
    p = dwfma(p, s, dw2(-0x1.8126ecp-2f)); p = dwfma(p, s, dw2(0x1.06eba6p-3f));
    const dwf2 ra = dwfma(p, a, a);
    const dwf2 tc = dwf2{fminf(t.x, 4.f), fminf(t.y, 4.f)};
    dwf2 q = dw2(-0x1.020fbcp-20f);
    q = dwfma(q, tc, dw2(0x1.142290p-15f)); q = dwfma(q, tc, dw2(-0x1.fff3bcp-12f)); q = dwfma(q, tc, dw2(0x1.176048p-8f)); q = dwfma(q, tc, dw2(-0x1.9a65fcp-6f));
    q = dwfma(q, tc, dw2(0x1.b94d44p-4f)); q = dwfma(q, tc, dw2(0x1.44b95ap-1f)); q = dwfma(q, tc, dw2(0x1.07f2d6p-3f));
    q = dwfma(q, tc, tc);
    const dwf2 e = dw2(1.f) - dw_exp2(q * dw2(-1.4426950408889634f));          // 1 - exp(-q)
    return dwf2{t.x > 0.921875f ? copysignf(e.x, a.x) : ra.x, t.y > 0.921875f ? copysignf(e.y, a.y) : ra.y};
}
__device__ __forceinline__ dwf2 gelu_f2(dwf2 x) {           // x * Phi(x)
#pragma clang fp contract(off)
    return (dw2(0.5f) * x) * (dw2(1.f) + erf_nb2(x * dw2(0.70710678118654752f)));
}
__device__ __forceinline__ dwf2 gelu_grad2(dwf2 x) {        // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
#pragma clang fp contract(off)
    const dwf2 ph = (x * dw2(0.3989422804014327f)) * dw_exp2(((dw2(-0.5f) * x) * x) * dw2(1.4426950408889634f));
    return dwfma(dw2(0.5f), dw2(1.f) + erf_nb2(x * dw2(0.70710678118654752f)), ph);
}
__device__ __forceinline__ float gelu_f(float x) { return gelu_f2(dw2(x)).x; }
__device__ __forceinline__ float gelu_grad(float x) { return gelu_grad2(dw2(x)).x; }

// z = sum_taps w[c][tap] * x[pixel + tap][c] (+ b[c]) ; y = gelu(z) (optional).  flip: correlate with the mirrored kernel (data gradient).
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_k(const T* __restrict__ dy, const T* __restrict__ z, T* __restrict__ dz, size_t nvec) {
    constexpr int V = TT<T>::VEC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float d[V], zz[V];
        ldv<T>(dy + i * V, d); ldv<T>(z + i * V, zz);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] *= gelu_grad(zz[e]);
        stv<T>(dz + i * V, d);
    }
}

// partial[blk][C*10]: [c*9 + tap] = sum_pixels dz[p][c] * x[p + tap][c] ; [C*9 + c] = sum_pixels dz[p][c]
// ---- sliding-window walk: a thread owns VT channels and walks a row segment of SEG output pixels, keeping the 3x3 input window packed in
// registers — 3 new loads per output pixel instead of 9 (a 9-load walk ran at ~1.5 TB/s, bound by the L1/TA requests in flight).
// VT = 8 / 4 / 2 channels per thread (16 / 8 / 4-byte loads for bf16): the narrow variants trade load width for 2-4x more waves, which is
// what the small late-stage tensors need (they are latency-bound, one short segment per thread).
template <typename T, int VT>
__device__ __forceinline__ void dw_load_col(const T* const (&rowp)[3], const bool (&vy)[3], int ix, int W, int C, DwVec<T, VT> (&col)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        col[r].zero();
        if (vy[r] && (unsigned)ix < (unsigned)W) col[r].load(rowp[r] + (size_t)ix * C);
    }
}

template <typename T, int VT>
__device__ __forceinline__ void dw_rows(const T* x, int row, int oy, int H, int W, int C, int cv, const T* (&rowp)[3], bool (&vy)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        vy[r] = (unsigned)(oy + r - 1) < (unsigned)H;
        rowp[r] = x + ((ptrdiff_t)(row + r - 1) * W) * C + cv * VT;
    }
}

template <typename T, int VT>
__global__ __launch_bounds__(256) void dwconv3x3_row_k(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, T* __restrict__ z, T* __restrict__ y,
                                                       int N, int H, int W, int C, int flip, int accumulate, int SEG, int SPR, int CVP, int walign) {
    typedef DwVec<T, VT> Vec;
    const int CV = C / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);          // consecutive segments (and their halo rows) share an XCD's L2
    const int s = bid * R + rl, cv = blockIdx.y * CVP + cvl;
    if (s >= N * H * SPR || cv >= CV) return;
    const int sx = s % SPR, row = s / SPR, oy = row % H;
    const int x0 = sx * SEG, x1 = min(W, x0 + SEG);
    float wr[9][VT], br[VT];
    {
        float wf[9 * VT];                                       // the 9*VT weights of this channel group are contiguous
        const float* wp = w + (size_t)cv * VT * 9;
        if constexpr (VT % 4 == 0) {
            if (walign) {
#pragma unroll
                for (int i = 0; i < 9 * VT / 4; ++i) {
                    const float4 q = reinterpret_cast<const float4*>(wp)[i];
                    wf[4 * i] = q.x; wf[4 * i + 1] = q.y; wf[4 * i + 2] = q.z; wf[4 * i + 3] = q.w;
                }
            } else {
#pragma unroll
                for (int i = 0; i < 9 * VT; ++i) wf[i] = wp[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 9 * VT; ++i) wf[i] = wp[i];
        }
#pragma unroll
        for (int e = 0; e < VT; ++e) {
            br[e] = b ? b[cv * VT + e] : 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t) wr[t][e] = flip ? wf[e * 9 + 8 - t] : wf[e * 9 + t];
        }
    }
    const T* rowp[3]; bool vy[3];
    dw_rows<T, VT>(x, row, oy, H, W, C, cv, rowp, vy);
    Vec c0[3], c1[3], c2[3], n1[3], n2[3];
    dw_load_col<T, VT>(rowp, vy, x0 - 1, W, C, c0); dw_load_col<T, VT>(rowp, vy, x0, W, C, c1); dw_load_col<T, VT>(rowp, vy, x0 + 1, W, C, c2);
    dw_load_col<T, VT>(rowp, vy, x0 + 2 < x1 + 1 ? x0 + 2 : -1, W, C, n1);
    for (int ox = x0; ox < x1; ++ox) {
        dw_load_col<T, VT>(rowp, vy, ox + 3 < x1 + 1 ? ox + 3 : -1, W, C, n2);      // columns past the segment's halo are never needed
        float a[VT], xv[VT];
#pragma unroll
        for (int e = 0; e < VT; ++e) a[e] = br[e];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            c0[r].unpack(xv);
#pragma unroll
            for (int e = 0; e < VT; ++e) a[e] += wr[r * 3][e] * xv[e];
            c1[r].unpack(xv);
#pragma unroll
            for (int e = 0; e < VT; ++e) a[e] += wr[r * 3 + 1][e] * xv[e];
            c2[r].unpack(xv);
#pragma unroll
            for (int e = 0; e < VT; ++e) a[e] += wr[r * 3 + 2][e] * xv[e];
        }
        const size_t o = ((size_t)row * W + ox) * C + cv * VT;
        Vec ov;
        if (accumulate) {
            ov.load(z + o); ov.unpack(xv);
#pragma unroll
            for (int e = 0; e < VT; ++e) a[e] += xv[e];
        }
        ov.pack(a); ov.store(z + o);
        if (y) {
#pragma unroll
            for (int e = 0; e < VT; ++e) a[e] = gelu_f(a[e]);
            ov.pack(a); ov.store(y + o);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; c2[r] = n1[r]; n1[r] = n2[r]; }
    }
}

// ---- the 3x3 walks again, built for instruction count (round 5; bf16).  What the counters and three ablations said about dwconv3x3_row_k on the PVTv2 Mlp
// shapes (tools/dw_micro.py, tools/prof_dw_pmc.sh, tools/dw_dbg.py): it fetches 1.13x the algorithmic bytes (the halo rows hit the L2), keeping more loads
// in flight makes it slower, and with two of the three rows and all stores switched off it still takes 77 of 84 us - while a plain copy of the same tensors
// streams at 7 TB/s.  It is bound by vector instructions: every step re-unpacks the whole 3x3 window (72 shift / and per 8 channels), multiplies channel
// by channel, rotates the window through register moves (which also forces a vmcnt wait per step) and guards every load with a branch.  Here:
//   * the window lives UNPACKED in registers as channel pairs (3 columns x 3 rows x VT/2 float2): only the new column is unpacked per step;
//   * all arithmetic is packed (v_pk_fma_f32: two channels per instruction), GELU and its derivative too (erf_nb2);
//   * window slots and the ring of D = 3 prefetched (packed) columns are addressed by compile-time indices (the step loop is unrolled by 3 = a full
//     rotation of both), so nothing is moved and a column is fetched three steps before it is unpacked;
//   * halo rows, image border and segment end are buffer-descriptor range checks (offset with bit 31 set: loads return zeros, stores are dropped) - plain
//     integer arithmetic on byte offsets, no branches in the walk.  Tensors below 2 GB (host).
// Same order of the nine products per channel as dwconv3x3_row_k (row-major taps on top of the bias), every one a fused multiply-add; the compiler had left
// a few of the old walk's products unfused (v_pk_mul + v_pk_add), so the two agree to the last fp32 bit or two - 1-ulp flips of the bf16 outputs
// (tools/dw_check.py: both sit at the bf16 rounding error against a float64 conv).
// Measured (tools/dw_micro.py, 16 x 88 x 88 x 512): conv + GELU 130 -> 108 us, data gradient 84 -> 62, weight gradient 100 -> 71, with dz = dy * gelu'(z) 155 -> 122;
// PVT_PraNet_V2 bs 16: 10.93 -> 10.69 ms per step.  The instruction count fell 2.5-3x, the time by a quarter: with loads and stores switched off the walk still
// takes 52 of 55 us (tools/dw_dbg.py) - what is left is issue-bound at roughly twice the ideal 4 cycles per wave instruction.
typedef unsigned dw_u4 __attribute__((ext_vector_type(4)));
typedef unsigned dw_u2 __attribute__((ext_vector_type(2)));
template <typename T, int VT>
__device__ __forceinline__ void dw_bload(DwVec<T, VT>& v, __amdgpu_buffer_rsrc_t rs, unsigned off) {
    constexpr int NW = DwVec<T, VT>::NW;
    if constexpr (NW == 4) { const dw_u4 q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0); v.w[0] = q.x; v.w[1] = q.y; v.w[2] = q.z; v.w[3] = q.w; }
    else if constexpr (NW == 2) { const dw_u2 q = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0); v.w[0] = q.x; v.w[1] = q.y; }
    else v.w[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0);
}
template <typename T, int VT>
__device__ __forceinline__ void dw_bstore(const DwVec<T, VT>& v, __amdgpu_buffer_rsrc_t rs, unsigned off) {
    constexpr int NW = DwVec<T, VT>::NW;
    if constexpr (NW == 4) { dw_u4 q; q.x = v.w[0]; q.y = v.w[1]; q.z = v.w[2]; q.w = v.w[3]; __builtin_amdgcn_raw_buffer_store_b128(q, rs, off, 0, 0); }
    else if constexpr (NW == 2) { dw_u2 q; q.x = v.w[0]; q.y = v.w[1]; __builtin_amdgcn_raw_buffer_store_b64(q, rs, off, 0, 0); }
    else __builtin_amdgcn_raw_buffer_store_b32(v.w[0], rs, off, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dw_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)p >> 32)) << 32) |
                                                     (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)p)), 0, (int)0x80000000u, 0x00020000);
}
constexpr unsigned DW_INV = 0x80000000u;
// -DPN2_DW_ABLATE (tools/dw_dbg.py): bit 1 of `flip` switches the two halo rows off, bit 2 the stores - what the walk costs without its memory traffic
#ifdef PN2_DW_ABLATE
#define DW_ABL_ROW(f, r) (((f) & 2) && (r) != 1)
#define DW_ABL_ST(f) ((f) & 4)
#else
#define DW_ABL_ROW(f, r) false
#define DW_ABL_ST(f) false
#endif
constexpr int DW_D = 3;          // packed columns in flight ahead of the window (= the unroll of the step loop)
__device__ __forceinline__ unsigned dw_mask(bool ok) { unsigned m = ok ? 0u : DW_INV; asm volatile("" : "+v"(m)); return m; }      // (opaque: see above)
// bf16 pair -> two floats / back
__device__ __forceinline__ dwf2 dw_unpk(unsigned w) { return dwf2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }
template <int VT>
__device__ __forceinline__ void dw_unpack_col(const DwVec<bf16_t, VT> (&c)[3], dwf2 (&o)[3][VT / 2]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < VT / 2; ++i) o[r][i] = dw_unpk(c[r].w[i]);
}
// the VT weights of tap t as channel pairs, from the [c][9] fp32 master
template <int VT>
__device__ __forceinline__ void dw_weights(const float* __restrict__ w, int cv, int flip, int walign, dwf2 (&wr)[9][VT / 2]) {
    float wf[9 * VT];
    const float* wp = w + (size_t)cv * VT * 9;
    if (VT % 4 == 0 && walign) {
#pragma unroll
        for (int i = 0; i < 9 * VT / 4; ++i) {
            const float4 q = reinterpret_cast<const float4*>(wp)[i];
            wf[4 * i] = q.x; wf[4 * i + 1] = q.y; wf[4 * i + 2] = q.z; wf[4 * i + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 9 * VT; ++i) wf[i] = wp[i];
    }
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < VT / 2; ++i) wr[t][i] = flip ? dwf2{wf[(2 * i) * 9 + 8 - t], wf[(2 * i + 1) * 9 + 8 - t]} : dwf2{wf[(2 * i) * 9 + t], wf[(2 * i + 1) * 9 + t]};
}

// CS: the kernel also leaves the column sums of what it stores (as stored: the rounded values) per workgroup, cpart[gridDim.x][C] - the bias gradient of the
// nn.Linear in front (Mlp.fc1, whose dY this data gradient IS) without the second read of the largest gradient tensor of the block by the column-sum pass.
template <int VT, bool GELU, bool CS = false>
__global__ __launch_bounds__(256) void dwconv3x3_win_k(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, bf16_t* __restrict__ z,
                                                       bf16_t* __restrict__ y, int N, int H, int W, int C, int flip, int SEG, int SPR, int CVP, int walign,
                                                       float* __restrict__ cpart = nullptr) {
    typedef DwVec<bf16_t, VT> Vec;
    constexpr int NP = VT / 2;
    const int CV = C / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);          // consecutive segments (and their halo rows) share an XCD's L2
    int s = bid * R + rl;
    const int cv = blockIdx.y * CVP + cvl;
    bool live = true;
    if (s >= N * H * SPR || cv >= CV) {
        if constexpr (!CS) return;
        live = false; s = 0;          // (CS: every thread reaches the workgroup's column-sum exchange; a dead one walks no step)
    }
    const int cvx = live ? cv : 0;
    const int sx = s % SPR, row = s / SPR, oy = row % H;
    const int x0 = sx * SEG, x1 = live ? min(W, x0 + SEG) : x0, hi = min(x1, W - 1);          // columns x0 - 1 .. hi are read (the image's and the segment's halo)
    dwf2 wr[9][NP], br[NP];
    dw_weights<VT>(w, cvx, flip & 1, walign, wr);
#pragma unroll
    for (int i = 0; i < NP; ++i) br[i] = b ? dwf2{b[cvx * VT + 2 * i], b[cvx * VT + 2 * i + 1]} : dw2(0.f);
    const __amdgpu_buffer_rsrc_t rx = dw_rsrc(x), rz = dw_rsrc(z), ry = dw_rsrc(GELU ? y : z);
    const unsigned Cb = (unsigned)C * 2u, cvb = (unsigned)(cvx * VT) * 2u;
    unsigned rowb[3], rowm[3];          // byte offset of column 0 of the three input rows; bit 31 when the row lies outside the image
#pragma unroll
    for (int r = 0; r < 3; ++r) { rowb[r] = (unsigned)((row + r - 1) * W) * Cb + cvb; rowm[r] = dw_mask((unsigned)(oy + r - 1) < (unsigned)H && !DW_ABL_ROW(flip, r)); }
    const unsigned outb = (unsigned)(row * W) * Cb + cvb;
    Vec ring[DW_D][3];
    dwf2 win[3][3][NP];
    dwf2 csum[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) csum[i] = dw2(0.f);
    auto fetch = [&](Vec (&c)[3], int ix) {
        const unsigned co = (unsigned)ix * Cb, cm = dw_mask((unsigned)ix <= (unsigned)hi);
#pragma unroll
        for (int r = 0; r < 3; ++r) dw_bload<bf16_t, VT>(c[r], rx, (rowb[r] + co) | rowm[r] | cm);          // (the masks are OR-ed in: an add of two set bits 31 would carry out)
    };
    {
        Vec p0[3], p1[3];
        fetch(p0, x0 - 1); fetch(p1, x0);
#pragma unroll
        for (int k = 0; k < DW_D; ++k) fetch(ring[k], x0 + 1 + k);
        dw_unpack_col<VT>(p0, win[0]); dw_unpack_col<VT>(p1, win[1]);
    }
    for (int xb = x0; xb < x1; xb += 3) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int ox = xb + u;
            dw_unpack_col<VT>(ring[u], win[(u + 2) % 3]);          // column ox + 1 takes the slot of column ox - 2
            fetch(ring[u], ox + 1 + DW_D);
            const unsigned o = (outb + (unsigned)ox * Cb) | dw_mask(ox < x1 && !DW_ABL_ST(flip));
            dwf2 a[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) a[i] = br[i];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int i = 0; i < NP; ++i) a[i] = dwfma(wr[r * 3 + k][i], win[(u + k) % 3][r][i], a[i]);
            Vec ov;
#pragma unroll
            for (int i = 0; i < NP; ++i) ov.w[i] = TT<bf16_t>::cvt2(a[i].x, a[i].y);
            dw_bstore<bf16_t, VT>(ov, rz, o);
            if constexpr (CS) {
                if (ox < x1) {
#pragma unroll
                    for (int i = 0; i < NP; ++i) csum[i] += dw_unpk(ov.w[i]);
                }
            }
            if constexpr (GELU) {
#pragma unroll
                for (int i = 0; i < NP; ++i) { const dwf2 g = gelu_f2(a[i]); ov.w[i] = TT<bf16_t>::cvt2(g.x, g.y); }
                dw_bstore<bf16_t, VT>(ov, ry, o);
            }
            __builtin_amdgcn_sched_barrier(0);          // one step at a time: the scheduler otherwise gathers the waits of all three steps at the top of the loop body
        }
    }
    if constexpr (CS) {          // the R segment lanes of a channel group meet in LDS (fixed order); one partial row per workgroup
        __shared__ float cs_sh[256 * VT];
#pragma unroll
        for (int i = 0; i < NP; ++i) { cs_sh[(rl * CVP + cvl) * VT + 2 * i] = csum[i].x; cs_sh[(rl * CVP + cvl) * VT + 2 * i + 1] = csum[i].y; }
        __syncthreads();
        if (rl == 0 && cv < CV) {
#pragma unroll
            for (int e = 0; e < VT; ++e) {
                float t = 0.f;
                for (int r = 0; r < R; ++r) t += cs_sh[(r * CVP + cvl) * VT + e];
                cpart[(size_t)blockIdx.x * C + cv * VT + e] = t;
            }
        }
    }
}

// weight gradient on the same window: partial[chunk][C*10] as dwconv3x3_wgrad_row_k leaves it (same segments per lane, same order of the sums).
template <int VT, bool ZP>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_win_k(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ x, float* __restrict__ partial, int N, int H, int W, int C,
                                                             int SEG, int SPR, int SPC, int CVP, const bf16_t* __restrict__ zpre, bf16_t* __restrict__ dz_out) {
    typedef DwVec<bf16_t, VT> Vec;
    typedef DwVec<bf16_t, VT> Vec1;
    constexpr int NP = VT / 2;
    extern __shared__ float sh[];            // [R][CVP*VT] reused for each of the 10 sums
    const int CV = C / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int chunk = xcd_remap(blockIdx.x, gridDim.x), cv = blockIdx.y * CVP + cvl;
    const int nseg = N * H * SPR;
    dwf2 a[10][NP];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int i = 0; i < NP; ++i) a[t][i] = dw2(0.f);
    if (cv < CV) {
        const __amdgpu_buffer_rsrc_t rx = dw_rsrc(x), rd = dw_rsrc(dz), rzp = dw_rsrc(ZP ? zpre : dz), ro = dw_rsrc(ZP ? dz_out : (bf16_t*)nullptr);
        const unsigned Cb = (unsigned)C * 2u, cvb = (unsigned)(cv * VT) * 2u;
        const int s1 = min(nseg, (chunk + 1) * SPC);
        for (int s = chunk * SPC + rl; s < s1; s += R) {
            const int sx = s % SPR, row = s / SPR, oy = row % H;
            const int x0 = sx * SEG, x1 = min(W, x0 + SEG), hi = min(x1, W - 1);
            unsigned rowb[3], rowm[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { rowb[r] = (unsigned)((row + r - 1) * W) * Cb + cvb; rowm[r] = dw_mask((unsigned)(oy + r - 1) < (unsigned)H); }
            const unsigned outb = (unsigned)(row * W) * Cb + cvb;
            Vec ring[DW_D][3];
            Vec1 dring[DW_D], zring[DW_D];
            dwf2 win[3][3][NP];
            auto fetch = [&](Vec (&c)[3], int ix) {
                const unsigned co = (unsigned)ix * Cb, cm = dw_mask((unsigned)ix <= (unsigned)hi);
#pragma unroll
                for (int r = 0; r < 3; ++r) dw_bload<bf16_t, VT>(c[r], rx, (rowb[r] + co) | rowm[r] | cm);
            };
            auto fetch_d = [&](int k, int ox) {          // dz (and the GELU input) of output column ox; zeros past the segment
                const unsigned o = (outb + (unsigned)ox * Cb) | dw_mask(ox < x1);
                dw_bload<bf16_t, VT>(dring[k], rd, o);
                if constexpr (ZP) dw_bload<bf16_t, VT>(zring[k], rzp, o);
            };
            {
                Vec p0[3], p1[3];
                fetch(p0, x0 - 1); fetch(p1, x0);
#pragma unroll
                for (int k = 0; k < DW_D; ++k) { fetch(ring[k], x0 + 1 + k); fetch_d(k, x0 + k); }
                dw_unpack_col<VT>(p0, win[0]); dw_unpack_col<VT>(p1, win[1]);
            }
            for (int xb = x0; xb < x1; xb += 3) {
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int ox = xb + u;
                    dw_unpack_col<VT>(ring[u], win[(u + 2) % 3]);
                    fetch(ring[u], ox + 1 + DW_D);
                    dwf2 d[NP];
#pragma unroll
                    for (int i = 0; i < NP; ++i) d[i] = dw_unpk(dring[u].w[i]);
                    if constexpr (ZP) {
                        Vec1 pk;
#pragma unroll
                        for (int i = 0; i < NP; ++i) {
                            const dwf2 g = d[i] * gelu_grad2(dw_unpk(zring[u].w[i]));
                            pk.w[i] = TT<bf16_t>::cvt2(g.x, g.y);
                            d[i] = dw_unpk(pk.w[i]);                  // accumulate what the data-gradient pass will read (the rounded value)
                        }
                        dw_bstore<bf16_t, VT>(pk, ro, (outb + (unsigned)ox * Cb) | dw_mask(ox < x1));
                    }
                    fetch_d(u, ox + DW_D);
#pragma unroll
                    for (int i = 0; i < NP; ++i) a[9][i] += d[i];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int k = 0; k < 3; ++k)
#pragma unroll
                            for (int i = 0; i < NP; ++i) a[r * 3 + k][i] = dwfma(d[i], win[(u + k) % 3][r][i], a[r * 3 + k][i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int i = 0; i < NP; ++i) { sh[(rl * CVP + cvl) * VT + 2 * i] = a[t][i].x; sh[(rl * CVP + cvl) * VT + 2 * i + 1] = a[t][i].y; }
        __syncthreads();
        if (rl == 0 && cv < CV) {
#pragma unroll
            for (int e = 0; e < VT; ++e) {
                float s_ = 0.f;
                for (int r = 0; r < R; ++r) s_ += sh[(r * CVP + cvl) * VT + e];
                const int c = cv * VT + e;
                partial[(size_t)chunk * C * 10 + (t < 9 ? c * 9 + t : C * 9 + c)] = s_;
            }
        }
        __syncthreads();
    }
}

// weight gradient, same walk: partial[chunk][C*10]; a block = CVP channel groups x R segment lanes, a chunk = SPC segments.
// zpre non-null: dz = dy * gelu'(zpre) is formed here (and written to dz_out) instead of by a separate pn2_gelu_bwd pass.
template <typename T, int VT>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_row_k(const T* __restrict__ dz, const T* __restrict__ x, float* __restrict__ partial, int N, int H, int W, int C,
                                                             int SEG, int SPR, int SPC, int CVP, const T* __restrict__ zpre, T* __restrict__ dz_out) {
    typedef DwVec<T, VT> Vec;
    extern __shared__ float sh[];            // [R][CVP*VT] reused for each of the 10 sums
    const int CV = C / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int chunk = xcd_remap(blockIdx.x, gridDim.x), cv = blockIdx.y * CVP + cvl;
    const int nseg = N * H * SPR;
    float a[10][VT];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int e = 0; e < VT; ++e) a[t][e] = 0.f;
    if (cv < CV) {
        const int s1 = min(nseg, (chunk + 1) * SPC);
        for (int s = chunk * SPC + rl; s < s1; s += R) {
            const int sx = s % SPR, row = s / SPR, oy = row % H;
            const int x0 = sx * SEG, x1 = min(W, x0 + SEG);
            const T* rowp[3]; bool vy[3];
            dw_rows<T, VT>(x, row, oy, H, W, C, cv, rowp, vy);
            const size_t o0 = ((size_t)row * W) * C + cv * VT;
            Vec c0[3], c1[3], c2[3], n1[3], dn, zn;
            zn.zero();
            dw_load_col<T, VT>(rowp, vy, x0 - 1, W, C, c0); dw_load_col<T, VT>(rowp, vy, x0, W, C, c1); dw_load_col<T, VT>(rowp, vy, x0 + 1, W, C, c2);
            dn.load(dz + o0 + (size_t)x0 * C);
            if (zpre) zn.load(zpre + o0 + (size_t)x0 * C);
            for (int ox = x0; ox < x1; ++ox) {
                const Vec dc = dn, zc = zn;
                dw_load_col<T, VT>(rowp, vy, ox + 2 < x1 + 1 ? ox + 2 : -1, W, C, n1);
                if (ox + 1 < x1) {
                    dn.load(dz + o0 + (size_t)(ox + 1) * C);
                    if (zpre) zn.load(zpre + o0 + (size_t)(ox + 1) * C);
                }
                float d[VT], xv[VT];
                dc.unpack(d);
                if (zpre) {
                    zc.unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) d[e] *= gelu_grad(xv[e]);
                    Vec pk; pk.pack(d); pk.store(dz_out + o0 + (size_t)ox * C);
                    pk.unpack(d);                              // accumulate what the data-gradient pass will read (the rounded value)
                }
#pragma unroll
                for (int e = 0; e < VT; ++e) a[9][e] += d[e];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    c0[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) a[r * 3][e] += d[e] * xv[e];
                    c1[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) a[r * 3 + 1][e] += d[e] * xv[e];
                    c2[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) a[r * 3 + 2][e] += d[e] * xv[e];
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; c2[r] = n1[r]; }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) {
#pragma unroll
        for (int e = 0; e < VT; ++e) sh[(rl * CVP + cvl) * VT + e] = a[t][e];
        __syncthreads();
        if (rl == 0 && cv < CV) {
#pragma unroll
            for (int e = 0; e < VT; ++e) {
                float s = 0.f;
                for (int r = 0; r < R; ++r) s += sh[(r * CVP + cvl) * VT + e];
                const int c = cv * VT + e;
                partial[(size_t)chunk * C * 10 + (t < 9 ? c * 9 + t : C * 9 + c)] = s;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ spatial-reduction attention
// q   [B][Nq][heads*64]                      (Attention.q)
// kv  [B][Nkv][2*heads*64]: k of head h at columns h*64.., v at heads*64 + h*64..   (Attention.kv reshaped (B,-1,2,heads,64), :98-101)
// out [B][Nq][heads*64]  = softmax(q k^T * scale) v   with heads concatenated (:107)
// One block = 4 waves = 4 queries in flight for one (b, head); K^T, K, V^T, V of that head live in LDS (<= 256 keys).  Lane d of a wave
// owns dimension d of q / out; lane l owns keys l, l+64, ... for the score / softmax part.
constexpr int AT_MAXK = 4;       // keys per lane -> Nkv <= 256

template <typename T>
__device__ __forceinline__ void attn_stage_kv(const T* __restrict__ kv, int ld_kv, int Nkv, int NP, int heads, int h, float* Kt, float* Vt) {
    // Kt/Vt: [64 dims][NP + 1]: the odd row stride makes both walks conflict-free — lanes over keys (row d, consecutive l) and lanes
    // over dims (column l, stride NP + 1); pad keys hold zeros
    const int RS = NP + 1;
    for (int i = threadIdx.x; i < NP * 64; i += 256) {
        const int l = i >> 6, d = i & 63;
        float k = 0.f, v = 0.f;
        if (l < Nkv) { k = TT<T>::ld(kv + (size_t)l * ld_kv + h * 64 + d); v = TT<T>::ld(kv + (size_t)l * ld_kv + heads * 64 + h * 64 + d); }
        Kt[d * RS + l] = k;
        Vt[d * RS + l] = v;
    }
}

__device__ __forceinline__ float rdlane(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }

constexpr int AT_QB = 4;         // queries a wave works on at once: every K / V value read from LDS is used AT_QB times

template <typename T, int NK>
__global__ __launch_bounds__(256) void attn_fwd_k(const T* __restrict__ q, int ld_q, const T* __restrict__ kv, int ld_kv, T* __restrict__ out, int ld_o,
                                                  float* __restrict__ lse, int Nq, int Nkv, int heads, float scale, int q_per_blk) {
    extern __shared__ float lds[];
    constexpr int NP = NK * 64, RS = NP + 1;
    float* Kt = lds; float* Vt = lds + 64 * RS;
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    attn_stage_kv<T>(kv + (size_t)b * Nkv * ld_kv, ld_kv, Nkv, NP, heads, h, Kt, Vt);
    __syncthreads();
    const int q0 = blockIdx.x * q_per_blk;
    int q1 = q0 + q_per_blk; if (q1 > Nq) q1 = Nq;
    for (int qb = q0 + wid * AT_QB; qb < q1; qb += 4 * AT_QB) {
        float qd[AT_QB], s[AT_QB][NK];
#pragma unroll
        for (int i = 0; i < AT_QB; ++i) {
            const int qi = qb + i < q1 ? qb + i : q1 - 1;
            qd[i] = TT<T>::ld(q + ((size_t)b * Nq + qi) * ld_q + h * 64 + lane) * scale;
#pragma unroll
            for (int k = 0; k < NK; ++k) s[i][k] = 0.f;
        }
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            float kk[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) kk[k] = Kt[d * RS + k * 64 + lane];
#pragma unroll
            for (int i = 0; i < AT_QB; ++i) {
                const float qq = rdlane(qd[i], d);
#pragma unroll
                for (int k = 0; k < NK; ++k) s[i][k] += qq * kk[k];
            }
        }
        float inv[AT_QB], o_[AT_QB];
#pragma unroll
        for (int i = 0; i < AT_QB; ++i) {
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < NK; ++k) if (k * 64 + lane < Nkv) mx = fmaxf(mx, s[i][k]);
            for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) { s[i][k] = (k * 64 + lane < Nkv) ? expf(s[i][k] - mx) : 0.f; sum += s[i][k]; }
            sum = wave_sum(sum);
            inv[i] = 1.f / sum; o_[i] = 0.f;
            if (lane == 0 && qb + i < q1) lse[((size_t)b * heads + h) * Nq + qb + i] = mx + logf(sum);
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int l = 0; l < 64; ++l) {
                const float vv = Vt[lane * RS + k * 64 + l];
#pragma unroll
                for (int i = 0; i < AT_QB; ++i) o_[i] += rdlane(s[i][k], l) * vv;
            }
        }
#pragma unroll
        for (int i = 0; i < AT_QB; ++i)
            if (qb + i < q1) TT<T>::st(out + ((size_t)b * Nq + qb + i) * ld_o + h * 64 + lane, o_[i] * inv[i]);
    }
}

// backward.  A block owns a chunk of queries of one (b, head).  Per group of 4 x QB queries (QB per wave): phase A recomputes
// p = exp(s - lse), dP = dO V^T, dS = p (dP - sum p dP) and dq = scale dS K exactly like the forward walks K / V; P, dS, q, dO of the
// group go to LDS tiles.  Phase B: every wave owns NP/4 keys and accumulates dK[l][d] += dS[q][l] q[q][d], dV[l][d] += P[q][l] dO[q][d]
// in registers (lane = d, broadcast LDS reads).  Block partials [2][NP][64] are summed over the query chunks by attn_bwd_kv_reduce_k.
constexpr int AT_QCHUNK = 256;

template <typename T, int NK, int QB>
__global__ __launch_bounds__(256) void attn_bwd_k(const T* __restrict__ q, int ld_q, const T* __restrict__ kv, int ld_kv, const T* __restrict__ dout, int ld_do,
                                                  const float* __restrict__ lse, T* __restrict__ dq, int ld_dq, float* __restrict__ part,
                                                  int Nq, int Nkv, int heads, float scale) {
    extern __shared__ float lds[];
    constexpr int NP = NK * 64, RS = NP + 1, KPW = NP / 4;
    float* Kt = lds; float* Vt = Kt + 64 * RS;
    constexpr int GQ = 4 * QB;          // queries per group
    float* Pt = Vt + 64 * RS; float* dSt = Pt + GQ * NP; float* qt = dSt + GQ * NP; float* dot = qt + GQ * 64;
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    attn_stage_kv<T>(kv + (size_t)b * Nkv * ld_kv, ld_kv, Nkv, NP, heads, h, Kt, Vt);
    __syncthreads();
    const int q0 = blockIdx.x * AT_QCHUNK;
    int q1 = q0 + AT_QCHUNK; if (q1 > Nq) q1 = Nq;
    float ak[KPW], av[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j) { ak[j] = 0.f; av[j] = 0.f; }
    for (int g0 = q0; g0 < q1; g0 += 4 * QB) {
        const int qb = g0 + wid * QB;
        float qd[QB], dod[QB], L[QB], s[QB][NK], dp[QB][NK];
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const bool ok = qb + i < q1;
            const int qi = ok ? qb + i : q1 - 1;
            const size_t qo = (size_t)b * Nq + qi;
            qd[i] = ok ? TT<T>::ld(q + qo * ld_q + h * 64 + lane) * scale : 0.f;
            dod[i] = ok ? TT<T>::ld(dout + qo * ld_do + h * 64 + lane) : 0.f;
            L[i] = lse[((size_t)b * heads + h) * Nq + qi];
#pragma unroll
            for (int k = 0; k < NK; ++k) { s[i][k] = 0.f; dp[i][k] = 0.f; }
        }
#pragma unroll
        for (int d = 0; d < 64; ++d) {
            float kk[NK], vv[NK];
#pragma unroll
            for (int k = 0; k < NK; ++k) { kk[k] = Kt[d * RS + k * 64 + lane]; vv[k] = Vt[d * RS + k * 64 + lane]; }
#pragma unroll
            for (int i = 0; i < QB; ++i) {
                const float qq = rdlane(qd[i], d), gg = rdlane(dod[i], d);
#pragma unroll
                for (int k = 0; k < NK; ++k) { s[i][k] += qq * kk[k]; dp[i][k] += gg * vv[k]; }
            }
        }
        float dqd[QB];
#pragma unroll
        for (int i = 0; i < QB; ++i) {
            const bool ok = qb + i < q1;
            float delta = 0.f;
#pragma unroll
            for (int k = 0; k < NK; ++k) { s[i][k] = (ok && k * 64 + lane < Nkv) ? expf(s[i][k] - L[i]) : 0.f; delta += s[i][k] * dp[i][k]; }
            delta = wave_sum(delta);
            const int row = wid * QB + i;
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                dp[i][k] = s[i][k] * (dp[i][k] - delta);          // dS
                Pt[row * NP + k * 64 + lane] = s[i][k]; dSt[row * NP + k * 64 + lane] = dp[i][k];
            }
            qt[row * 64 + lane] = qd[i]; dot[row * 64 + lane] = dod[i];
            dqd[i] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < NK; ++k) {
#pragma unroll
            for (int l = 0; l < 64; ++l) {
                const float kk = Kt[lane * RS + k * 64 + l];
#pragma unroll
                for (int i = 0; i < QB; ++i) dqd[i] += rdlane(dp[i][k], l) * kk;
            }
        }
#pragma unroll
        for (int i = 0; i < QB; ++i)
            if (qb + i < q1) TT<T>::st(dq + ((size_t)b * Nq + qb + i) * ld_dq + h * 64 + lane, dqd[i] * scale);
        __syncthreads();
        // phase B: this wave's keys [wid*KPW, +KPW), all queries of the group
        for (int r = 0; r < GQ; ++r) {
            const float qv = qt[r * 64 + lane], gv = dot[r * 64 + lane];
            const float* pr = Pt + r * NP + wid * KPW; const float* dr = dSt + r * NP + wid * KPW;
#pragma unroll
            for (int j = 0; j < KPW; ++j) { ak[j] += dr[j] * qv; av[j] += pr[j] * gv; }
        }
        __syncthreads();
    }
    float* dst = part + ((((size_t)b * heads + h) * gridDim.x + blockIdx.x) * 2) * NP * 64;
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        dst[(size_t)(wid * KPW + j) * 64 + lane] = ak[j];                 // qd carried the softmax scale already
        dst[(size_t)NP * 64 + (size_t)(wid * KPW + j) * 64 + lane] = av[j];
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_kv_reduce_k(const float* __restrict__ part, T* __restrict__ dkv, int ld_dkv, int Nkv, int NP, int heads, int nqb) {
    const int l = blockIdx.x, h = blockIdx.y, b = blockIdx.z, d = threadIdx.x;
    const float* p = part + (((size_t)b * heads + h) * nqb * 2) * NP * 64 + (size_t)l * 64 + d;
    float sk = 0.f, sv = 0.f;
    for (int c = 0; c < nqb; ++c) { sk += p[(size_t)c * 2 * NP * 64]; sv += p[(size_t)c * 2 * NP * 64 + (size_t)NP * 64]; }
    T* dst = dkv + ((size_t)b * Nkv + l) * ld_dkv;
    TT<T>::st(dst + h * 64 + d, sk);
    TT<T>::st(dst + heads * 64 + h * 64 + d, sv);
}

// ------------------------------------------------------------------------------------------ bf16 MFMA attention
// Same math as attn_fwd_k / attn_bwd_k on v_mfma_f32_16x16x32_bf16.  A block = 4 waves = 64 queries of one (b, head); every wave owns 16 queries.
// Fragment layouts (lane l): A[l&15][(l>>4)*8 + j], B[(l>>4)*8 + j][l&15], C: column l&15, rows (l>>4)*4 + r.
// LDS tiles keep 16-byte fragment rows with an 8-element pad so the 16 lanes of a quarter-wave hit distinct banks.
typedef __attribute__((ext_vector_type(4))) float f32x4v;
__device__ __forceinline__ f32x4v mfma16(const uint4& a, const uint4& b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// Fragment (A[i][k] or B[k][i], i = lane & 15) of a 16-wide column block of a ROW-MAJOR LDS tile [k][col] whose rows are the contraction index,
// through the transposing LDS read: k-slot (g, j) <-> tile row (j >> 2) * 16 + g * 4 + (j & 3).  The other operand of the MFMA has to walk the
// contraction in the same order (perm_frag).  Conflict-free when the row stride is an odd multiple of 32 bytes (ATR = 80 elements).
constexpr int ATR = 80;
__device__ __forceinline__ uint4 tr_frag(const bf16_t* tile, int col0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    unsigned a = (unsigned)(size_t)(reinterpret_cast<const char*>(tile) + (g * 4 + (i >> 2)) * (ATR * 2) + (col0 + (i & 3) * 4) * 2);
    asm volatile("" : "+v"(a));            // keep the tile offset out of the instruction's immediate field
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a));
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(a + 16 * ATR * 2));
    const uint2 lo = __builtin_bit_cast(uint2, v0), hi = __builtin_bit_cast(uint2, v1);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}
// the matching fragment of an operand stored with the contraction index contiguous: row[k0 + g*4 .. +3] and row[k0 + 16 + g*4 .. +3]
__device__ __forceinline__ uint4 perm_frag(const bf16_t* row, int k0, int g) {
    const uint2 lo = *reinterpret_cast<const uint2*>(row + k0 + g * 4), hi = *reinterpret_cast<const uint2*>(row + k0 + 16 + g * 4);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// Forward.  A block = 8 waves = 128 queries per step; K and V rows of the (b, head) are staged once (16-byte copies, no transposes) and the block
// walks query tiles blockIdx.x, blockIdx.x + gridDim.x, ...  (the K/V staging of the one-tile-per-block version cost more than its MFMAs).
template <int NK>
__global__ __launch_bounds__(512) void attn_fwd_mfma_k(const bf16_t* __restrict__ q, int ld_q, const bf16_t* __restrict__ kv, int ld_kv, bf16_t* __restrict__ out, int ld_o,
                                                       float* __restrict__ lse, int Nq, int Nkv, int heads, float scale) {
    constexpr int NP = NK * 64, NT = NP / 16, KR = 72, PR = NP + 8;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* Ks = smem;                    // [NP][KR]   key rows                   -> B operand of S = Q K^T (lane = key, 16 bytes of d)
    bf16_t* Vs = Ks + NP * KR;            // [NP][ATR]  value rows                 -> B operand of O = P V through tr_frag (contraction = keys)
    bf16_t* Ps = Vs + NP * ATR;           // [8 waves][16][PR] probabilities       -> A operand of O = P V (perm_frag)
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const bf16_t* kvb = kv + (size_t)b * Nkv * ld_kv;
    for (int i = threadIdx.x; i < NP * 8; i += 512) {
        const int key = i >> 3, ch = i & 7;
        uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
        if (key < Nkv) { kk = *reinterpret_cast<const uint4*>(kvb + (size_t)key * ld_kv + h * 64 + ch * 8);
                         vv = *reinterpret_cast<const uint4*>(kvb + (size_t)key * ld_kv + heads * 64 + h * 64 + ch * 8); }
        *reinterpret_cast<uint4*>(Ks + key * KR + ch * 8) = kk;
        *reinterpret_cast<uint4*>(Vs + key * ATR + ch * 8) = vv;
    }
    __syncthreads();
    bf16_t* pw = Ps + wid * 16 * PR;
    const int ntile = (Nq + 127) / 128;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int q0 = tile * 128 + wid * 16;
        const int qa = min(q0 + l15, Nq - 1);                             // A-operand row of this lane
        const bf16_t* qp = q + ((size_t)b * Nq + qa) * ld_q + h * 64 + g * 8;
        const uint4 aq0 = *reinterpret_cast<const uint4*>(qp), aq1 = *reinterpret_cast<const uint4*>(qp + 32);
        f32x4v s[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16_t* kp = Ks + (nt * 16 + l15) * KR + g * 8;
            f32x4v c = {0.f, 0.f, 0.f, 0.f};
            c = mfma16(aq0, *reinterpret_cast<const uint4*>(kp), c);
            c = mfma16(aq1, *reinterpret_cast<const uint4*>(kp + 32), c);
            s[nt] = c;
        }
        float mx[4], sum[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m = -INFINITY;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { const float v = (nt * 16 + l15 < Nkv) ? s[nt][r] * scale : -INFINITY; s[nt][r] = v; m = fmaxf(m, v); }
            m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4)); m = fmaxf(m, __shfl_xor(m, 8));
            float t = 0.f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) { const float p = __expf(s[nt][r] - m); s[nt][r] = p; t += p; }
            t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
            mx[r] = m; sum[r] = t;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) pw[(g * 4 + r) * PR + nt * 16 + l15] = f2bf(s[nt][r]);
        __syncthreads();                                                  // (every wave walks the same number of tiles)
        f32x4v o[4];
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) o[nd] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NP / 32; ++ks) {
            const uint4 ap = perm_frag(pw + l15 * PR, ks * 32, g);
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) o[nd] = mfma16(ap, tr_frag(Vs + ks * 32 * ATR, nd * 16, lane), o[nd]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + g * 4 + r;
            if (qi < Nq) {
                const float inv = 1.f / sum[r];
                bf16_t* op = out + ((size_t)b * Nq + qi) * ld_o + h * 64 + l15;
#pragma unroll
                for (int nd = 0; nd < 4; ++nd) op[nd * 16] = f2bf(o[nd][r] * inv);
                if (l15 == 0) lse[((size_t)b * heads + h) * Nq + qi] = mx[r] + __logf(sum[r]);
            }
        }
        __syncthreads();                                                  // the next tile's probabilities overwrite Ps
    }
}

// delta[b][h][q] = sum_d dO[q][h*64 + d] * O[q][h*64 + d]   (= sum_keys P dP: lets the backward treat key ranges independently)
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_k(const T* __restrict__ dout, int ld_do, const T* __restrict__ o, int ld_o, float* __restrict__ delta, int B, int Nq, int heads) {
    const int lane = threadIdx.x & 63;
    const size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), total = (size_t)B * Nq * heads;
    if (item >= total) return;
    const int h = (int)(item % heads); const size_t bq = item / heads;          // bq = b * Nq + q
    float v = TT<T>::ld(dout + bq * ld_do + h * 64 + lane) * TT<T>::ld(o + bq * ld_o + h * 64 + lane);
    v = wave_sum(v);
    if (lane == 0) { const size_t b = bq / Nq, qi = bq % Nq; delta[(b * heads + h) * Nq + qi] = v; }
}

// backward for the keys [key0, key0 + NKB*64) of one (b, head); the block walks 64-query tiles blockIdx.x, blockIdx.x + gridDim.x, ...:
//   S = Q K^T, dP = dO V^T (MFMA) -> P = exp(S scale - lse), dS = P (dP - delta) scale
//   dQ (+)= dS K ; dK += dS^T Q, dV += P^T dO accumulate in registers over the tiles of the block and leave as ONE fp32 partial [2][NPT][64]
//   per block (summed by attn_bwd_kv_reduce_k).  K, V, Q and dO tiles are staged row-major with 16-byte copies; every product whose contraction
//   runs over tile rows (keys for dQ, queries for dK / dV) reads its B operand with the transposing LDS read (tr_frag), so nothing is transposed.
template <int NKB>
__global__ __launch_bounds__(256) void attn_bwd_mfma_k(const bf16_t* __restrict__ q, int ld_q, const bf16_t* __restrict__ kv, int ld_kv, const bf16_t* __restrict__ dout, int ld_do,
                                                       const float* __restrict__ lse, const float* __restrict__ delta, bf16_t* __restrict__ dq, int ld_dq, float* __restrict__ part,
                                                       int Nq, int Nkv, int heads, float scale, int key0, int NPT, int dq_acc) {
    constexpr int NP = NKB * 64, NT = NP / 16, KR = 72, TR = NP + 8, NKT = NT / 4;
    extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
    bf16_t* Ks = smem;                    // [NP][ATR] K rows: B of S = Q K^T (16-byte reads) and of dQ = dS K (tr_frag, contraction = keys)
    bf16_t* Vs = Ks + NP * ATR;           // [NP][KR]  V rows: B of dP = dO V^T
    bf16_t* Qs = Vs + NP * KR;            // [64][ATR] Q tile  [q][d]: A of S, B of dK (tr_frag, contraction = queries)
    bf16_t* Gs = Qs + 64 * ATR;           // [64][ATR] dO tile [q][d]: A of dP, B of dV
    bf16_t* Sa = Gs + 64 * ATR;           // [4][16][TR] dS, A layout per wave (perm_frag over keys)
    bf16_t* St = Sa + 64 * TR;            // [NP][KR]  dS transposed [key][q]  (perm_frag over queries)
    bf16_t* Pt = St + NP * KR;            // [NP][KR]  P transposed  [key][q]
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const bf16_t* kvb = kv + (size_t)b * Nkv * ld_kv;
    for (int i = threadIdx.x; i < NP * 8; i += 256) {
        const int kl = i >> 3, ch = i & 7, key = key0 + kl;
        uint4 kk = make_uint4(0, 0, 0, 0), vv = kk;
        if (key < Nkv) { kk = *reinterpret_cast<const uint4*>(kvb + (size_t)key * ld_kv + h * 64 + ch * 8);
                         vv = *reinterpret_cast<const uint4*>(kvb + (size_t)key * ld_kv + heads * 64 + h * 64 + ch * 8); }
        *reinterpret_cast<uint4*>(Ks + kl * ATR + ch * 8) = kk;
        *reinterpret_cast<uint4*>(Vs + kl * KR + ch * 8) = vv;
    }
    f32x4v ak[NKT][4], av[NKT][4];
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) { ak[i][nd] = f32x4v{0.f, 0.f, 0.f, 0.f}; av[i][nd] = ak[i][nd]; }
    bf16_t* sa = Sa + wid * 16 * TR;
    const int ntile = (Nq + 63) / 64;
    for (int tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int qt0 = tile * 64;
        __syncthreads();                                                  // the previous tile's readers of Qs / Gs / St / Pt are done (and K / V are in)
        for (int i = threadIdx.x; i < 64 * 8; i += 256) {
            const int ql = i >> 3, ch = i & 7, qi = qt0 + ql;
            uint4 qq = make_uint4(0, 0, 0, 0), gg = qq;
            if (qi < Nq) { qq = *reinterpret_cast<const uint4*>(q + ((size_t)b * Nq + qi) * ld_q + h * 64 + ch * 8);
                           gg = *reinterpret_cast<const uint4*>(dout + ((size_t)b * Nq + qi) * ld_do + h * 64 + ch * 8); }
            *reinterpret_cast<uint4*>(Qs + ql * ATR + ch * 8) = qq;
            *reinterpret_cast<uint4*>(Gs + ql * ATR + ch * 8) = gg;
        }
        const int q0 = qt0 + wid * 16;
        float L[4], D[4]; bool rok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + g * 4 + r;
            rok[r] = qi < Nq;
            const size_t li = ((size_t)b * heads + h) * Nq + (rok[r] ? qi : Nq - 1);
            L[r] = lse[li]; D[r] = delta[li];
        }
        __syncthreads();
        const bf16_t* qrow = Qs + (wid * 16 + l15) * ATR + g * 8; const bf16_t* grow = Gs + (wid * 16 + l15) * ATR + g * 8;
        const uint4 aq0 = *reinterpret_cast<const uint4*>(qrow), aq1 = *reinterpret_cast<const uint4*>(qrow + 32);
        const uint4 ag0 = *reinterpret_cast<const uint4*>(grow), ag1 = *reinterpret_cast<const uint4*>(grow + 32);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bf16_t* kp = Ks + (nt * 16 + l15) * ATR + g * 8; const bf16_t* vp = Vs + (nt * 16 + l15) * KR + g * 8;
            f32x4v s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = mfma16(aq0, *reinterpret_cast<const uint4*>(kp), s); s = mfma16(aq1, *reinterpret_cast<const uint4*>(kp + 32), s);
            dp = mfma16(ag0, *reinterpret_cast<const uint4*>(vp), dp); dp = mfma16(ag1, *reinterpret_cast<const uint4*>(vp + 32), dp);
            const bool kok = key0 + nt * 16 + l15 < Nkv;
            bf16_t pb[4], sb[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = (kok && rok[r]) ? __expf(s[r] * scale - L[r]) : 0.f;
                const float ds = p * (dp[r] - D[r]) * scale;
                pb[r] = f2bf(p); sb[r] = f2bf(ds);
                sa[(g * 4 + r) * TR + nt * 16 + l15] = sb[r];
            }
            // transposed tiles [key][query]: 4 consecutive queries of this lane -> one 8-byte store each
            const int key = nt * 16 + l15, qc = wid * 16 + g * 4;
            *reinterpret_cast<uint2*>(St + key * KR + qc) = make_uint2((unsigned)sb[0] | ((unsigned)sb[1] << 16), (unsigned)sb[2] | ((unsigned)sb[3] << 16));
            *reinterpret_cast<uint2*>(Pt + key * KR + qc) = make_uint2((unsigned)pb[0] | ((unsigned)pb[1] << 16), (unsigned)pb[2] | ((unsigned)pb[3] << 16));
        }
        __syncthreads();
        // dQ = dS K  (this wave's 16 queries x 64 dims, contraction over the keys of this range)
        f32x4v dqa[4];
#pragma unroll
        for (int nd = 0; nd < 4; ++nd) dqa[nd] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NP / 32; ++ks) {
            const uint4 a = perm_frag(sa + l15 * TR, ks * 32, g);
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) dqa[nd] = mfma16(a, tr_frag(Ks + ks * 32 * ATR, nd * 16, lane), dqa[nd]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (rok[r]) {
                bf16_t* dp_ = dq + ((size_t)b * Nq + q0 + g * 4 + r) * ld_dq + h * 64 + l15;
#pragma unroll
                for (int nd = 0; nd < 4; ++nd) dp_[nd * 16] = f2bf(dq_acc ? bf2f(dp_[nd * 16]) + dqa[nd][r] : dqa[nd][r]);
            }
        }
        // dK += dS^T Q, dV += P^T dO : this wave's key tiles (kt = wid, wid + 4, ...), contraction over the 64 queries of the tile
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
            const int kt = wid + 4 * i;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uint4 as = perm_frag(St + (kt * 16 + l15) * KR, ks * 32, g);
                const uint4 ap = perm_frag(Pt + (kt * 16 + l15) * KR, ks * 32, g);
#pragma unroll
                for (int nd = 0; nd < 4; ++nd) {
                    ak[i][nd] = mfma16(as, tr_frag(Qs + ks * 32 * ATR, nd * 16, lane), ak[i][nd]);
                    av[i][nd] = mfma16(ap, tr_frag(Gs + ks * 32 * ATR, nd * 16, lane), av[i][nd]);
                }
            }
        }
    }
    float* dst = part + ((((size_t)b * heads + h) * gridDim.x + blockIdx.x) * 2) * (size_t)NPT * 64;
#pragma unroll
    for (int i = 0; i < NKT; ++i) {
        const int kt = wid + 4 * i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const size_t row = (size_t)(key0 + kt * 16 + g * 4 + r) * 64 + l15;
#pragma unroll
            for (int nd = 0; nd < 4; ++nd) { dst[row + nd * 16] = ak[i][nd][r]; dst[(size_t)NPT * 64 + row + nd * 16] = av[i][nd][r]; }
        }
    }
}

// y[n][r][c] = x[n][r][c] * s[n]   (DropPath: per-sample keep mask / keep_prob, timm.models.layers.DropPath used at pvtv2.py:125,148-149)
template <typename T>
__global__ __launch_bounds__(256) void scale_samples_k(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ s, const T* __restrict__ res,
                                                       size_t vec_per_sample, size_t nvec) {
    constexpr int V = TT<T>::VEC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float v[V], r[V];
        ldv<T>(x + i * V, v);
        const float f = s[i / vec_per_sample];
        if (res) {                    // x + drop_path(f(x)) of Block.forward in one pass
            ldv<T>(res + i * V, r);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = v[e] * f + r[e];
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] *= f;
        }
        stv<T>(y + i * V, v);
    }
}

inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
inline int grid_for(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g)); }

}  // namespace

#define VIT_DISPATCH(dt, BODY) \
    if ((dt) == PN2_BF16) { typedef bf16_t T; BODY } else if ((dt) == PN2_F32) { typedef float T; BODY } else return -3;

extern "C" {

static int ln_lpr(int dt, int C) {
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -1;
    int lpr = pow2ceil(C / V);
    if (lpr < 8) lpr = 8;
    if (lpr > 64) lpr = 64;
    return ((C / V + lpr - 1) / lpr <= LN_NV) ? lpr : -1;
}

int pn2_layernorm_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, int M, int C, const float* gamma, const float* beta, float eps,
                      float* mean, float* rstd, void* stream) {
    if (!x || !y || !gamma || !beta || !mean || !rstd || M < 1) return -1;
    const int lpr = ln_lpr(dt, C);
    if (lpr < 0) return -2;
    const int rows_per_blk = 4 * (64 / lpr);
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(ln_fwd_k<T>, dim3((M + rows_per_blk - 1) / rows_per_blk), dim3(256), 0, (hipStream_t)stream, (const T*)x, ld_x, (T*)y, ld_y, M, C,
                                          gamma, beta, eps, mean, rstd, lpr); })
    PN2_CHECK_LAUNCH();
    return 0;
}

static int rows_for(int M, int unit) {          /* rows per block of the column-sum style reductions: ~ROWS_TARGET blocks, a multiple of `unit` */
    constexpr int target = 512;   // 2 workgroups per CU measured best (256: -1.7 %, 1024: -1.1 % on config 4)
    int rows = (M + target - 1) / target;
    rows = ((rows + unit - 1) / unit) * unit;
    return rows < unit ? unit : rows;
}

int pn2_rows_blocks(int M, int unit) { if (M < 1 || unit < 1) return -1; const int rows = rows_for(M, unit); return (M + rows - 1) / rows; }

int pn2_layernorm_bwd(int dt, const void* dy, int ld_dy, const void* x, int ld_x, int M, int C, const float* gamma, const float* mean, const float* rstd,
                      void* dx, int ld_dx, int accumulate_dx, float* pg, float* pb, int nblk, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !pg || !pb || M < 1 || nblk < 1) return -1;
    const int lpr = ln_lpr(dt, C);
    if (lpr < 0) return -2;
    const int nslot = 4 * (64 / lpr);
    const int rows = rows_for(M, nslot);
    if ((M + rows - 1) / rows != nblk) return -2;          // nblk must be pn2_rows_blocks(M, pn2_ln_slots(dt, C))
    const size_t lds = (size_t)2 * nslot * C * 4;
    if (lds > 64 * 1024) return -2;
    const bool nv1 = C / (dt == PN2_F32 ? 4 : 8) <= lpr;          // one channel vector per lane (C <= 512 bf16 / 256 fp32): the lean instantiation
    VIT_DISPATCH(dt, { if (nv1) hipLaunchKernelGGL((ln_bwd_k<T, 1>), dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const T*)dy, ld_dy, (const T*)x, ld_x, M, C, gamma, mean, rstd,
                                          (T*)dx, ld_dx, accumulate_dx, pg, pb, rows, lpr);
                       else hipLaunchKernelGGL((ln_bwd_k<T, LN_NV>), dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const T*)dy, ld_dy, (const T*)x, ld_x, M, C, gamma, mean, rstd,
                                          (T*)dx, ld_dx, accumulate_dx, pg, pb, rows, lpr); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_ln_slots(int dt, int C) { const int lpr = ln_lpr(dt, C); return lpr < 0 ? -1 : 4 * (64 / lpr); }

int pn2_colsum_finalize(const float* partial, int nblk, int C, int ld, float* out, int accumulate, void* stream) {
    if (!partial || !out || nblk < 1 || C < 1) return -1;
    hipLaunchKernelGGL(colsum_finalize_k, dim3((C + 31) / 32), dim3(256), 0, (hipStream_t)stream, partial, nblk, C, ld, out, accumulate);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_colsum_finalize_blocks(int C) { return C < 1 ? -1 : (C + 31) / 32; }

int pn2_colsum_finalize_multi(const pn2_colsum_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    hipLaunchKernelGGL(colsum_finalize_multi_k, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_colsum(int dt, const void* dy, int ld, int M, int C, float* partial, int nblk, void* stream) {
    if (!dy || !partial || M < 1 || nblk < 1) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V || ld % V) return -2;
    int cvp = pow2ceil(C / V); if (cvp > 256) cvp = 256;
    const int rows = rows_for(M, 256 / cvp);
    if ((M + rows - 1) / rows != nblk) return -2;          // nblk must be pn2_rows_blocks(M, pn2_colsum_unit(dt, C))
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(colsum_k<T>, dim3(nblk), dim3(256), 256 * TT<T>::VEC * 4, (hipStream_t)stream, (const T*)dy, ld, M, C, partial, rows, cvp); })
    PN2_CHECK_LAUNCH();
    return 0;
}

/* fills rows / cvp of a job for pn2_colsum_multi and returns its workgroup count (= rows of its partial buffer, as pn2_rows_blocks gives) */
int pn2_colsum_job_blocks(int dt, pn2_colsum_in_job* j) {
    if (!j || !j->dy || !j->partial || j->M < 1) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (j->C % V || j->ld % V) return -2;
    int cvp = pow2ceil(j->C / V); if (cvp > 256) cvp = 256;
    j->cvp = cvp; j->rows = rows_for(j->M, 256 / cvp);
    return (j->M + j->rows - 1) / j->rows;
}

int pn2_colsum_multi(int dt, const pn2_colsum_in_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(colsum_multi_k<T>, dim3(total_blocks), dim3(256), 256 * TT<T>::VEC * 4, (hipStream_t)stream, jobs_dev, block_start_dev, njobs); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_colsum_unit(int dt, int C) { const int V = dt == PN2_F32 ? 4 : 8; int cvp = pow2ceil(C / V); if (cvp > 256) cvp = 256; return 256 / cvp; }

// channels per thread (VT) and segment length of the depth-wise walks, from a sweep on MI355X (tools/dw_micro.py):
// kind 0 = conv + GELU (ALU-heavier: 8-byte vectors, more waves), 1 = plain conv / data gradient (16-byte vectors while there are threads to
// spare), 2 = weight gradient (80 accumulators per 8 channels: 2 channels per thread keeps 8 waves per SIMD).  Short segments for small tensors.
// PN2_DW_WIN=0: the round-3 walks (dwconv3x3_row_k / dwconv3x3_wgrad_row_k) instead of the window kernels (A/B; bf16 only - fp32 always takes them)
static int dw_win() { static const int v = [] { const char* e = getenv("PN2_DW_WIN"); return e ? atoi(e) : 1; }(); return v; }

static void dw_row_geometry(int dt, int kind, int N, int H, int W, int C, int& VT, int& SEG, int& SPR) {
    const int vmax = dt == PN2_F32 ? 4 : 8;
    auto threads = [&](int vt, int target) { return (long long)N * H * ((W + target - 1) / target) * (C / vt); };
    // window kernels: a segment costs ~150 instructions and two memory latencies before its first output (weights, the first five columns); 32-pixel segments
    // where the tensor still gives 4 resident waves per SIMD twice over (stage 1 of PVTv2-B2 at 352^2: 117 -> 108 us conv + GELU, 71 -> 62 us data gradient)
    static const int seg_env = [] { const char* e = getenv("PN2_DW_SEG"); return e ? atoi(e) : 32; }();
    int target = 16;
    if (kind == 2) VT = 2;
    else {
        VT = kind == 0 ? vmax / 2 : vmax;
        while (VT > 2 && ((C % VT) || (VT == vmax && kind == 1 && threads(VT, 16) < 2LL * 256 * 8 * 64))) VT >>= 1;
        if (threads(VT, 16) < 200000) target = 8;
    }
    if (seg_env > 0 && dt == PN2_BF16 && dw_win() && threads(4, seg_env) >= 2LL * 256 * 4 * 64 * 4) target = seg_env;
    SPR = (W + target - 1) / target; SEG = (W + SPR - 1) / SPR;
}

#define DW_VT(VT, BODY) switch (VT) { case 8: { constexpr int VT_ = 8; BODY } break; case 4: { constexpr int VT_ = 4; BODY } break; default: { constexpr int VT_ = 2; BODY } }

static int dwconv3x3_impl(int dt, const void* x, const float* w, const float* b, void* z, void* y_gelu, int N, int H, int W, int C, int flip, int accumulate, float* cpart, int cblk,
                          void* stream) {
    if (!x || !w || !z) return -1;
    if (C % 2 || (dt == PN2_BF16 && C % 2) || N < 1 || H < 1 || W < 1) return -2;
    int VT, SEG, SPR; dw_row_geometry(dt, y_gelu ? 0 : 1, N, H, W, C, VT, SEG, SPR);
    const int CV = C / VT, lanes = (dt == PN2_F32 ? 256 : 512) / VT;     // channel groups per block: 1 KB of one pixel
    const int cvp = CV >= lanes ? lanes : pow2ceil(CV), R = 256 / cvp, nseg = N * H * SPR;
    const int walign = ((uintptr_t)w & 15) == 0;
    const dim3 grid((nseg + R - 1) / R, (CV + cvp - 1) / cvp);
    if (dt == PN2_BF16 && dw_win() && !accumulate && (long long)N * H * W * C * 2 < 0x7fff0000LL) {          // the window walk (4 or 2 channels per thread)
        const int vt = C % 4 == 0 ? 4 : 2, cv4 = C / vt, ln = 512 / vt;
        const int cvp4 = cv4 >= ln ? ln : pow2ceil(cv4), R4 = 256 / cvp4;
        const dim3 g4((nseg + R4 - 1) / R4, (cv4 + cvp4 - 1) / cvp4);
#define PN2_DW_WIN(VT_, G_, CS_) hipLaunchKernelGGL((dwconv3x3_win_k<VT_, G_, CS_>), g4, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, b, (bf16_t*)z, (bf16_t*)y_gelu, N, H, W, C, flip, SEG, SPR, cvp4, walign, cpart)
        if (cblk == -1) return (int)g4.x;          // (pn2_dwconv3x3_colsum_blocks: rows of cpart)
        if (cpart) {
            if (y_gelu || cblk != (int)g4.x) return -2;
            if (vt == 4) PN2_DW_WIN(4, false, true); else PN2_DW_WIN(2, false, true);
        } else if (vt == 4) { if (y_gelu) PN2_DW_WIN(4, true, false); else PN2_DW_WIN(4, false, false); }
        else { if (y_gelu) PN2_DW_WIN(2, true, false); else PN2_DW_WIN(2, false, false); }
#undef PN2_DW_WIN
        PN2_CHECK_LAUNCH();
        return 0;
    }
    if (cpart || cblk == -1) return -2;          // column sums only ride on the window kernels
    if (dt == PN2_BF16) { DW_VT(VT, { hipLaunchKernelGGL((dwconv3x3_row_k<bf16_t, VT_>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, b, (bf16_t*)z, (bf16_t*)y_gelu,
                                                      N, H, W, C, flip, accumulate, SEG, SPR, cvp, walign); }) }
    else if (dt == PN2_F32) {
        if (VT == 4) hipLaunchKernelGGL((dwconv3x3_row_k<float, 4>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, w, b, (float*)z, (float*)y_gelu, N, H, W, C, flip, accumulate, SEG, SPR, cvp, walign);
        else hipLaunchKernelGGL((dwconv3x3_row_k<float, 2>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, w, b, (float*)z, (float*)y_gelu, N, H, W, C, flip, accumulate, SEG, SPR, cvp, walign);
    } else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_dwconv3x3(int dt, const void* x, const float* w, const float* b, void* z, void* y_gelu, int N, int H, int W, int C, int flip, int accumulate, void* stream) {
    return dwconv3x3_impl(dt, x, w, b, z, y_gelu, N, H, W, C, flip, accumulate, nullptr, 0, stream);
}
/* rows of `cpart` for pn2_dwconv3x3_colsum at this geometry (< 0: that entry does not serve it - run pn2_dwconv3x3 and a column-sum pass) */
int pn2_dwconv3x3_colsum_blocks(int dt, int N, int H, int W, int C) {
    if (dt != PN2_BF16 || C % 2 || N < 1 || H < 1 || W < 1) return -1;
    static const float dummy = 0.f;
    const int r = dwconv3x3_impl(dt, &dummy, &dummy, nullptr, (void*)&dummy, nullptr, N, H, W, C, 0, 0, nullptr, -1, nullptr);
    return r > 0 ? r : -1;
}
/* pn2_dwconv3x3 (no GELU, no accumulate) that also leaves cpart[nblk][C]: per-workgroup column sums of the stored result */
int pn2_dwconv3x3_colsum(int dt, const void* x, const float* w, const float* b, void* z, int N, int H, int W, int C, int flip, float* cpart, int nblk, void* stream) {
    if (!cpart || nblk < 1) return -1;
    return dwconv3x3_impl(dt, x, w, b, z, nullptr, N, H, W, C, flip, 0, cpart, nblk, stream);
}

int pn2_gelu_bwd(int dt, const void* dy, const void* z, void* dz, long long n, void* stream) {
    if (!dy || !z || !dz) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (n % V) return -2;
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(gelu_bwd_k<T>, dim3(grid_for((size_t)n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)z, (T*)dz, (size_t)n / V); })
    PN2_CHECK_LAUNCH();
    return 0;
}

static void dw_wgrad_geometry(int dt, int N, int H, int W, int C, int& VT, int& SEG, int& SPR, int& SPC, int& cvp, int& nchunk) {
    dw_row_geometry(dt, 2, N, H, W, C, VT, SEG, SPR);
    const bool win = dt == PN2_BF16 && dw_win() && (long long)N * H * W * C * 2 < 0x7fff0000LL;
    if (win) VT = C % 4 == 0 ? 4 : 2;          // window kernel: 4 channels per thread (20 packed accumulators), still 256 contiguous bytes of a pixel per block
    const int CV = C / VT, lanes = (dt == PN2_F32 ? 32 : 64) / VT * 2;          // 256 contiguous bytes of one pixel per block
    cvp = CV >= lanes ? lanes : pow2ceil(CV);
    const int R = 256 / cvp, gy = (CV + cvp - 1) / cvp, nseg = N * H * SPR;
    int want = 1536 / gy; if (want < 1) want = 1;            // ~1536 workgroups over (chunks x channel groups)
    SPC = (nseg + want - 1) / want;
    SPC = ((SPC + R - 1) / R) * R;
    nchunk = (nseg + SPC - 1) / SPC;
}

int pn2_dwconv3x3_wgrad_blocks(int dt, int N, int H, int W, int C) {
    if ((dt != PN2_F32 && dt != PN2_BF16) || C % 2 || N < 1 || H < 1 || W < 1) return -1;
    int VT, SEG, SPR, SPC, cvp, nchunk; dw_wgrad_geometry(dt, N, H, W, C, VT, SEG, SPR, SPC, cvp, nchunk);
    return nchunk;
}

int pn2_dwconv3x3_wgrad(int dt, const void* dz, const void* x, float* partial, int nblk, int N, int H, int W, int C, const void* zpre, void* dz_out, void* stream) {
    if (!dz || !x || !partial || nblk < 1 || (zpre && !dz_out)) return -1;
    if (C % 2) return -2;
    int VT, SEG, SPR, SPC, cvp, nchunk; dw_wgrad_geometry(dt, N, H, W, C, VT, SEG, SPR, SPC, cvp, nchunk);
    if (nchunk != nblk) return -2;                           // nblk must be pn2_dwconv3x3_wgrad_blocks(dt, N, H, W, C)
    const int CV = C / VT;
    const dim3 grid(nchunk, (CV + cvp - 1) / cvp);
    const size_t lds = (size_t)256 * VT * 4;
    if (dt == PN2_BF16 && dw_win() && (long long)N * H * W * C * 2 < 0x7fff0000LL) {
#define PN2_DW_WG(VT_, Z_) hipLaunchKernelGGL((dwconv3x3_wgrad_win_k<VT_, Z_>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dz, (const bf16_t*)x, partial, N, H, W, C, SEG, SPR, SPC, cvp, (const bf16_t*)zpre, (bf16_t*)dz_out)
        if (VT == 4) { if (zpre) PN2_DW_WG(4, true); else PN2_DW_WG(4, false); }
        else { if (zpre) PN2_DW_WG(2, true); else PN2_DW_WG(2, false); }
#undef PN2_DW_WG
        PN2_CHECK_LAUNCH();
        return 0;
    }
    if (dt == PN2_BF16) { DW_VT(VT, { hipLaunchKernelGGL((dwconv3x3_wgrad_row_k<bf16_t, VT_>), grid, dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dz, (const bf16_t*)x, partial,
                                                      N, H, W, C, SEG, SPR, SPC, cvp, (const bf16_t*)zpre, (bf16_t*)dz_out); }) }
    else if (dt == PN2_F32) {
        if (VT == 4) hipLaunchKernelGGL((dwconv3x3_wgrad_row_k<float, 4>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)dz, (const float*)x, partial, N, H, W, C, SEG, SPR, SPC, cvp, (const float*)zpre, (float*)dz_out);
        else hipLaunchKernelGGL((dwconv3x3_wgrad_row_k<float, 2>), grid, dim3(256), lds, (hipStream_t)stream, (const float*)dz, (const float*)x, partial, N, H, W, C, SEG, SPR, SPC, cvp, (const float*)zpre, (float*)dz_out);
    } else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

static int attn_geom(int Nkv, int heads, int head_dim) { return (head_dim == 64 && Nkv >= 1 && Nkv <= 64 * AT_MAXK && heads >= 1) ? 0 : -2; }

// persistent attention blocks per launch: the blocks of one (b, head) share its K / V and walk the query tiles between them
static int attn_target_blocks() { return 512; }
static int attn_bwd_gx(int B, int heads, int Nq) {
    const int ntile = (Nq + 63) / 64;
    // backward: one block per CU measured best (the block state - K, V, two 64-query tiles, dS / P and their transposes - fills the LDS anyway)
    int gx = (attn_target_blocks() / 2 + B * heads - 1) / (B * heads);
    return gx > ntile ? ntile : (gx < 1 ? 1 : gx);
}

int pn2_attn_fwd(int dt, const void* q, int ld_q, const void* kv, int ld_kv, void* out, int ld_o, float* lse, int B, int Nq, int Nkv, int heads, int head_dim,
                 float scale, void* stream) {
    if (!q || !kv || !out || !lse) return -1;
    if (int rc = attn_geom(Nkv, heads, head_dim)) return rc;
    const int NK = (Nkv + 63) / 64, NP = NK * 64;
    const size_t lds = (size_t)2 * 64 * (NP + 1) * 4;
    const int qpb = 64;
    const dim3 grid((Nq + qpb - 1) / qpb, heads, B);
    hipStream_t st = (hipStream_t)stream;
#define PN2_ATTN_FWD(NKV) { if (lds > 64 * 1024) { static bool done = false; if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_k<T, NKV>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; } } \
        hipLaunchKernelGGL((attn_fwd_k<T, NKV>), grid, dim3(256), lds, st, (const T*)q, ld_q, (const T*)kv, ld_kv, (T*)out, ld_o, lse, Nq, Nkv, heads, scale, qpb); }
    constexpr bool use_mfma = true;
    if (dt == PN2_BF16 && use_mfma && (ld_q % 8) == 0 && (ld_kv % 8) == 0 && (ld_o % 8) == 0) {
        const int NPK = NK == 3 ? 256 : NP;                        // 129..192 keys run the 256-key instantiation (zero-padded keys are masked)
        const size_t lm = ((size_t)NPK * 72 + (size_t)NPK * ATR + 8 * 16 * (NPK + 8)) * 2;
        const int ntile = (Nq + 127) / 128;
        int gx = (attn_target_blocks() + B * heads - 1) / (B * heads); if (gx > ntile) gx = ntile; if (gx < 1) gx = 1;
        const dim3 gm(gx, heads, B);
#define PN2_ATTN_FWD_M(NKV) { if (lm > 64 * 1024) { static bool done = false; if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_mfma_k<NKV>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm); done = true; } } \
        hipLaunchKernelGGL((attn_fwd_mfma_k<NKV>), gm, dim3(512), lm, st, (const bf16_t*)q, ld_q, (const bf16_t*)kv, ld_kv, (bf16_t*)out, ld_o, lse, Nq, Nkv, heads, scale); }
        if (NK == 1) PN2_ATTN_FWD_M(1) else if (NK == 2) PN2_ATTN_FWD_M(2) else PN2_ATTN_FWD_M(4)
#undef PN2_ATTN_FWD_M
        PN2_CHECK_LAUNCH();
        return 0;
    }
    VIT_DISPATCH(dt, { if (NK == 1) PN2_ATTN_FWD(1) else if (NK == 2) PN2_ATTN_FWD(2) else if (NK == 3) PN2_ATTN_FWD(3) else PN2_ATTN_FWD(4) })
#undef PN2_ATTN_FWD
    PN2_CHECK_LAUNCH();
    return 0;
}

static bool attn_use_mfma(int dt, int ld_a, int ld_b) {
    constexpr bool on = true;
    return on && dt == PN2_BF16 && (ld_a % 8) == 0 && (ld_b % 8) == 0;
}

int pn2_attn_bwd_blocks(int dt, int B, int heads, int Nq) {
    if (Nq < 1 || B < 1 || heads < 1) return -1;
    return dt == PN2_BF16 ? attn_bwd_gx(B, heads, Nq) : (Nq + AT_QCHUNK - 1) / AT_QCHUNK;
}

int pn2_attn_bwd(int dt, const void* q, int ld_q, const void* kv, int ld_kv, const void* out, int ld_o, const void* dout, int ld_do, const float* lse, void* dq, int ld_dq,
                 void* dkv, int ld_dkv, float* partial, float* delta, int B, int Nq, int Nkv, int heads, int head_dim, float scale, void* stream) {
    if (!q || !kv || !out || !dout || !lse || !dq || !dkv || !partial || !delta) return -1;
    if (int rc = attn_geom(Nkv, heads, head_dim)) return rc;
    const int NK = (Nkv + 63) / 64, NP = NK * 64;
    hipStream_t st = (hipStream_t)stream;
    if (attn_use_mfma(dt, ld_q, ld_kv) && (ld_do % 8) == 0 && (ld_dq % 8) == 0) {
        const int nqt = attn_bwd_gx(B, heads, Nq);                   // persistent blocks (= partial slots) per (b, head)
        hipLaunchKernelGGL(attn_delta_k<bf16_t>, dim3((unsigned)(((size_t)B * Nq * heads + 3) / 4)), dim3(256), 0, st, (const bf16_t*)dout, ld_do, (const bf16_t*)out, ld_o, delta, B, Nq, heads);
        const dim3 grid(nqt, heads, B);
        for (int key0 = 0; key0 < NP; key0 += 128) {
            const int nkb = NP - key0 >= 128 ? 2 : 1;
            const int np = nkb * 64;
            const size_t lm = ((size_t)np * ATR + (size_t)np * 72 + 2 * 64 * ATR + 64 * (np + 8) + 2 * (size_t)np * 72) * 2;
            if (nkb == 2) {
                static bool done = false;
                if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_mfma_k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm); done = true; }
                hipLaunchKernelGGL(attn_bwd_mfma_k<2>, grid, dim3(256), lm, st, (const bf16_t*)q, ld_q, (const bf16_t*)kv, ld_kv, (const bf16_t*)dout, ld_do, lse, delta, (bf16_t*)dq, ld_dq,
                                   partial, Nq, Nkv, heads, scale, key0, NP, key0 > 0 ? 1 : 0);
            } else {
                static bool done1 = false;
                if (!done1) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_mfma_k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lm); done1 = true; }
                hipLaunchKernelGGL(attn_bwd_mfma_k<1>, grid, dim3(256), lm, st, (const bf16_t*)q, ld_q, (const bf16_t*)kv, ld_kv, (const bf16_t*)dout, ld_do, lse, delta, (bf16_t*)dq, ld_dq,
                                   partial, Nq, Nkv, heads, scale, key0, NP, key0 > 0 ? 1 : 0);
            }
        }
        hipLaunchKernelGGL(attn_bwd_kv_reduce_k<bf16_t>, dim3(Nkv, heads, B), dim3(64), 0, st, partial, (bf16_t*)dkv, ld_dkv, Nkv, NP, heads, nqt);
        PN2_CHECK_LAUNCH();
        return 0;
    }
    const int nqb = (Nq + AT_QCHUNK - 1) / AT_QCHUNK;
    const int QB = NK == 4 ? 2 : AT_QB;                 // 256 keys: smaller query groups so that K, V and the tiles fit the 160 KiB of LDS
    const size_t lds = ((size_t)2 * 64 * (NP + 1) + 2 * 4 * QB * NP + 2 * 4 * QB * 64) * 4;
    const dim3 grid(nqb, heads, B);
#define PN2_ATTN_BWD(NKV, QBV) { if (lds > 64 * 1024) { static bool done = false; if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_k<T, NKV, QBV>), \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); done = true; } } \
        hipLaunchKernelGGL((attn_bwd_k<T, NKV, QBV>), grid, dim3(256), lds, st, (const T*)q, ld_q, (const T*)kv, ld_kv, (const T*)dout, ld_do, lse, (T*)dq, ld_dq, \
                           partial, Nq, Nkv, heads, scale); }
    VIT_DISPATCH(dt, {
        if (NK == 1) PN2_ATTN_BWD(1, AT_QB) else if (NK == 2) PN2_ATTN_BWD(2, AT_QB) else if (NK == 3) PN2_ATTN_BWD(3, AT_QB) else PN2_ATTN_BWD(4, 2)
        hipLaunchKernelGGL(attn_bwd_kv_reduce_k<T>, dim3(Nkv, heads, B), dim3(64), 0, st, partial, (T*)dkv, ld_dkv, Nkv, NP, heads, nqb); })
#undef PN2_ATTN_BWD
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_scale_samples(int dt, const void* x, void* y, const float* scale, const void* res, int N, long long per_sample, void* stream) {
    if (!x || !y || !scale || N < 1) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (per_sample % V) return -2;
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(scale_samples_k<T>, dim3(grid_for((size_t)N * per_sample / V)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, scale, (const T*)res,
                                          (size_t)per_sample / V, (size_t)N * per_sample / V); })
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
