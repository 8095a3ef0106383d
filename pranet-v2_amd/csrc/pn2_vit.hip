// pn2_vit.hip — the non-GEMM kernels of the PVTv2 encoder (reference: /root/reference/binary_seg/lib/pvtv2.py):
//   LayerNorm forward / backward            nn.LayerNorm at pvtv2.py:71,119,126,169,224-247 (eps 1e-5 / 1e-6)
//   column sums                             bias gradients of nn.Linear / biased nn.Conv2d (:19,22,62-65,70,167)
//   depth-wise 3x3 conv (+bias, +GELU)      DWConv :363-374 followed by nn.GELU in Mlp.forward :42-49
//   spatial-reduction attention             Attention.forward :90-111 (head_dim 64, <= 256 reduced key/value tokens)
// Tokens [B, N, C] of the reference are NHWC pixels here (N = H*W), so no transposes are needed anywhere.
// The Linear layers themselves run on the implicit-GEMM conv kernels (1x1) of pn2_conv.hip.
// All kernels are deterministic (fixed-order reductions, no floating-point atomics).
#include "pn2_common.h"
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
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_k(const T* __restrict__ dy, int ld_dy, const T* __restrict__ x, int ld_x, int M, int C, const float* __restrict__ gamma,
                                                const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dx, int ld_dx, int acc_dx,
                                                float* __restrict__ pg, float* __restrict__ pb, int rows_per_blk, int LPR) {
    constexpr int V = TT<T>::VEC;
    extern __shared__ float sh[];            // [2][slots][C] for the cross-slot reduction
    const int CV = C / V, rpw = 64 / LPR, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane % LPR, slot = wid * rpw + lane / LPR;
    const int nslot = 4 * rpw;
    const int r0 = blockIdx.x * rows_per_blk;
    int r1 = r0 + rows_per_blk; if (r1 > M) r1 = M;
    float ag[LN_NV][V], ab[LN_NV][V], gm[LN_NV][V];
#pragma unroll
    for (int k = 0; k < LN_NV; ++k)
#pragma unroll
        for (int e = 0; e < V; ++e) { ag[k][e] = 0.f; ab[k][e] = 0.f; const int c = (lr + k * LPR) * V + e; gm[k][e] = c < C ? gamma[c] : 0.f; }
    for (int rb = r0; rb < r1; rb += nslot) {           // uniform trip count: the shuffles below need every lane
        const int row = rb + slot;
        const bool live = row < r1;
        float g[LN_NV][V], xh[LN_NV][V];
        float s1 = 0.f, s2 = 0.f;
        const float mu = live ? mean[row] : 0.f, rs = live ? rstd[row] : 0.f;
#pragma unroll
        for (int k = 0; k < LN_NV; ++k) {
            const int cv = lr + k * LPR;
            if (live && cv < CV) {
                float d[V], xv[V];
                ldv<T>(dy + (size_t)row * ld_dy + cv * V, d);
                ldv<T>(x + (size_t)row * ld_x + cv * V, xv);
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
        for (int k = 0; k < LN_NV; ++k) {
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
    for (int k = 0; k < LN_NV; ++k) {
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

// out[c] (+)= sum_b p[b*ld + c]   (fixed order; double accumulation)
__global__ __launch_bounds__(256) void colsum_finalize_k(const float* __restrict__ p, int nblk, int C, int ld, float* __restrict__ out, int accumulate) {
    __shared__ double sh[4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < C) for (int r = rl; r < nblk; r += 4) s += (double)p[(size_t)r * ld + c];
    sh[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < C) {
        const float v = (float)((sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]));
        out[c] = accumulate ? out[c] + v : v;
    }
}

// partial[blk][C] = sum over the block's rows of dy[row][c]
template <typename T>
__global__ __launch_bounds__(256) void colsum_k(const T* __restrict__ dy, int ld, int M, int C, float* __restrict__ partial, int rows_per_blk, int CVP) {
    constexpr int V = TT<T>::VEC;
    extern __shared__ float sh[];            // [R][CVP*V]
    const int CV = C / V, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int r0 = blockIdx.x * rows_per_blk;
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
                partial[(size_t)blockIdx.x * C + cv * V + e] = s;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------ depth-wise 3x3 (+bias, +GELU)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {       // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

// z = sum_taps w[c][tap] * x[pixel + tap][c] (+ b[c]) ; y = gelu(z) (optional).  flip: correlate with the mirrored kernel (data gradient).
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_k(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b, T* __restrict__ z, T* __restrict__ y,
                                                   int N, int H, int W, int C, int flip, int accumulate) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V;
    const size_t total = (size_t)N * H * W * CV;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int cv = (int)(idx % CV); size_t p = idx / CV;
        const int ox = (int)(p % W); p /= W; const int oy = (int)(p % H); const int n = (int)(p / H);
        float a[V];
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] = b ? b[cv * V + e] : 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                float xv[V];
                ldv<T>(x + (((size_t)n * H + iy) * W + ix) * C + cv * V, xv);
                const int tw = flip ? 8 - t : t;
#pragma unroll
                for (int e = 0; e < V; ++e) a[e] += w[(cv * V + e) * 9 + tw] * xv[e];
            }
        }
        const size_t o = (((size_t)n * H + oy) * W + ox) * C + cv * V;
        if (accumulate) { float old[V]; ldv<T>(z + o, old);
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] += old[e]; }
        stv<T>(z + o, a);
        if (y) {
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] = gelu_f(a[e]);
            stv<T>(y + o, a);
        }
    }
}

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
template <typename T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_k(const T* __restrict__ dz, const T* __restrict__ x, float* __restrict__ partial, int N, int H, int W, int C,
                                                         int pix_per_blk, int CVP) {
    constexpr int V = TT<T>::VEC;
    extern __shared__ float sh[];            // [R][CVP*V] reused for each of the 10 sums
    const int CV = C / V, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int M = N * H * W, p0 = blockIdx.x * pix_per_blk;
    int p1 = p0 + pix_per_blk; if (p1 > M) p1 = M;
    for (int cvb = 0; cvb < CV; cvb += CVP) {
        const int cv = cvb + cvl;
        float a[10][V];
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int e = 0; e < V; ++e) a[t][e] = 0.f;
        if (cv < CV) {
            for (int m = p0 + rl; m < p1; m += R) {
                const int ox = m % W, oy = (m / W) % H, n = m / (W * H);
                float d[V];
                ldv<T>(dz + (size_t)m * C + cv * V, d);
#pragma unroll
                for (int e = 0; e < V; ++e) a[9][e] += d[e];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
                    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                        float xv[V];
                        ldv<T>(x + (((size_t)n * H + iy) * W + ix) * C + cv * V, xv);
#pragma unroll
                        for (int e = 0; e < V; ++e) a[t][e] += d[e] * xv[e];
                    }
                }
            }
        }
        for (int t = 0; t < 10; ++t) {
#pragma unroll
            for (int e = 0; e < V; ++e) sh[(rl * CVP + cvl) * V + e] = a[t][e];
            __syncthreads();
            if (rl == 0 && cv < CV) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    float s = 0.f;
                    for (int r = 0; r < R; ++r) s += sh[(r * CVP + cvl) * V + e];
                    const int c = cv * V + e;
                    partial[(size_t)blockIdx.x * C * 10 + (t < 9 ? c * 9 + t : C * 9 + c)] = s;
                }
            }
            __syncthreads();
        }
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

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_k(const T* __restrict__ q, int ld_q, const T* __restrict__ kv, int ld_kv, T* __restrict__ out, int ld_o,
                                                  float* __restrict__ lse, int Nq, int Nkv, int heads, float scale, int q_per_blk) {
    extern __shared__ float lds[];
    const int NP = (Nkv + 63) & ~63, NK = NP >> 6, RS = NP + 1;
    float* Kt = lds; float* Vt = lds + 64 * RS;
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    attn_stage_kv<T>(kv + (size_t)b * Nkv * ld_kv, ld_kv, Nkv, NP, heads, h, Kt, Vt);
    __syncthreads();
    const int q0 = blockIdx.x * q_per_blk;
    int q1 = q0 + q_per_blk; if (q1 > Nq) q1 = Nq;
    for (int qi = q0 + wid; qi < q1; qi += 4) {
        const size_t qo = ((size_t)b * Nq + qi);
        const float qd = TT<T>::ld(q + qo * ld_q + h * 64 + lane) * scale;
        float s[AT_MAXK];
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) s[k] = 0.f;
#pragma unroll 8
        for (int d = 0; d < 64; ++d) {
            const float qq = __shfl(qd, d);
#pragma unroll
            for (int k = 0; k < AT_MAXK; ++k) if (k < NK) s[k] += qq * Kt[d * RS + k * 64 + lane];
        }
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) if (k < NK && k * 64 + lane < Nkv) mx = fmaxf(mx, s[k]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f, p[AT_MAXK];
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) { p[k] = (k < NK && k * 64 + lane < Nkv) ? expf(s[k] - mx) : 0.f; sum += p[k]; }
        sum = wave_sum(sum);
        const float inv = 1.f / sum;
        float o_ = 0.f;
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) {
            if (k < NK) {
#pragma unroll 8
                for (int l = 0; l < 64; ++l) o_ += __shfl(p[k], l) * Vt[lane * RS + k * 64 + l];
            }
        }
        TT<T>::st(out + qo * ld_o + h * 64 + lane, o_ * inv);
        if (lane == 0) lse[((size_t)b * heads + h) * Nq + qi] = mx + logf(sum);
    }
}

// backward, per query: recompute p from the saved log-sum-exp, dP = dO V^T, dS = p (dP - sum p dP), dq = scale dS K; P and dS are stored
// ([B][heads][Nq][NP] fp32) for the key/value pass below
template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_q_k(const T* __restrict__ q, int ld_q, const T* __restrict__ kv, int ld_kv, const T* __restrict__ dout, int ld_do,
                                                    const float* __restrict__ lse, T* __restrict__ dq, int ld_dq, float* __restrict__ Pm, float* __restrict__ dSm,
                                                    int Nq, int Nkv, int heads, float scale, int q_per_blk) {
    extern __shared__ float lds[];
    const int NP = (Nkv + 63) & ~63, NK = NP >> 6, RS = NP + 1;
    float* Kt = lds; float* Vt = lds + 64 * RS;
    const int b = blockIdx.z, h = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    attn_stage_kv<T>(kv + (size_t)b * Nkv * ld_kv, ld_kv, Nkv, NP, heads, h, Kt, Vt);
    __syncthreads();
    const int q0 = blockIdx.x * q_per_blk;
    int q1 = q0 + q_per_blk; if (q1 > Nq) q1 = Nq;
    for (int qi = q0 + wid; qi < q1; qi += 4) {
        const size_t qo = ((size_t)b * Nq + qi);
        const float qd = TT<T>::ld(q + qo * ld_q + h * 64 + lane) * scale;
        const float dod = TT<T>::ld(dout + qo * ld_do + h * 64 + lane);
        const float L = lse[((size_t)b * heads + h) * Nq + qi];
        float s[AT_MAXK], dp[AT_MAXK];
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) { s[k] = 0.f; dp[k] = 0.f; }
#pragma unroll 8
        for (int d = 0; d < 64; ++d) {
            const float qq = __shfl(qd, d), gg = __shfl(dod, d);
#pragma unroll
            for (int k = 0; k < AT_MAXK; ++k) if (k < NK) { s[k] += qq * Kt[d * RS + k * 64 + lane]; dp[k] += gg * Vt[d * RS + k * 64 + lane]; }
        }
        float p[AT_MAXK], delta = 0.f;
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) { p[k] = (k < NK && k * 64 + lane < Nkv) ? expf(s[k] - L) : 0.f; delta += p[k] * dp[k]; }
        delta = wave_sum(delta);
        float ds[AT_MAXK], dqd = 0.f;
        const size_t po = (((size_t)b * heads + h) * Nq + qi) * NP;
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) {
            ds[k] = p[k] * (dp[k] - delta);
            if (k < NK) { Pm[po + k * 64 + lane] = p[k]; dSm[po + k * 64 + lane] = ds[k]; }
        }
#pragma unroll
        for (int k = 0; k < AT_MAXK; ++k) {
            if (k < NK) {
#pragma unroll 8
                for (int l = 0; l < 64; ++l) dqd += __shfl(ds[k], l) * Kt[lane * RS + k * 64 + l];
            }
        }
        TT<T>::st(dq + qo * ld_dq + h * 64 + lane, dqd * scale);
    }
}

// backward, keys/values: dK[l] = scale * sum_q dS[q][l] q[q], dV[l] = sum_q P[q][l] dO[q] for a chunk of AT_KC keys of one (b, head);
// the 4 waves split the queries, lane = dimension; fixed-order cross-wave sum through LDS
constexpr int AT_KC = 8;

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kv_k(const T* __restrict__ q, int ld_q, const T* __restrict__ dout, int ld_do, const float* __restrict__ Pm,
                                                     const float* __restrict__ dSm, T* __restrict__ dkv, int ld_dkv, int Nq, int Nkv, int heads, float scale) {
    __shared__ float red[4][2 * AT_KC][64];
    const int NP = (Nkv + 63) & ~63;
    const int b = blockIdx.z, h = blockIdx.y, l0 = blockIdx.x * AT_KC, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float ak[AT_KC], av[AT_KC];
#pragma unroll
    for (int j = 0; j < AT_KC; ++j) { ak[j] = 0.f; av[j] = 0.f; }
    const size_t pb = ((size_t)b * heads + h) * Nq;
    for (int qi = wid; qi < Nq; qi += 4) {
        const size_t qo = ((size_t)b * Nq + qi);
        const float qd = TT<T>::ld(q + qo * ld_q + h * 64 + lane), dod = TT<T>::ld(dout + qo * ld_do + h * 64 + lane);
        const float* pr = Pm + (pb + qi) * NP + l0; const float* dr = dSm + (pb + qi) * NP + l0;
#pragma unroll
        for (int j = 0; j < AT_KC; ++j) { ak[j] += dr[j] * qd; av[j] += pr[j] * dod; }     // l0 + j < NP always (NP multiple of 64, AT_KC divides 64)
    }
#pragma unroll
    for (int j = 0; j < AT_KC; ++j) { red[wid][j][lane] = ak[j]; red[wid][AT_KC + j][lane] = av[j]; }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * AT_KC * 64; i += 256) {
        const int j = i >> 6, d = i & 63, l = l0 + (j % AT_KC);
        if (l >= Nkv) continue;
        const float s = (red[0][j][d] + red[1][j][d]) + (red[2][j][d] + red[3][j][d]);
        T* dst = dkv + ((size_t)b * Nkv + l) * ld_dkv + (j < AT_KC ? h * 64 : heads * 64 + h * 64) + d;
        TT<T>::st(dst, j < AT_KC ? s * scale : s);
    }
}

// y[n][r][c] = x[n][r][c] * s[n]   (DropPath: per-sample keep mask / keep_prob, timm.models.layers.DropPath used at pvtv2.py:125,148-149)
template <typename T>
__global__ __launch_bounds__(256) void scale_samples_k(const T* __restrict__ x, T* __restrict__ y, const float* __restrict__ s, size_t vec_per_sample, size_t nvec) {
    constexpr int V = TT<T>::VEC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float v[V];
        ldv<T>(x + i * V, v);
        const float f = s[i / vec_per_sample];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= f;
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

static int rows_for(int M, int unit) {          /* rows per block of the column-sum style reductions: ~512 blocks, a multiple of `unit` */
    int rows = (M + 511) / 512;
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
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(ln_bwd_k<T>, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const T*)dy, ld_dy, (const T*)x, ld_x, M, C, gamma, mean, rstd,
                                          (T*)dx, ld_dx, accumulate_dx, pg, pb, rows, lpr); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_ln_slots(int dt, int C) { const int lpr = ln_lpr(dt, C); return lpr < 0 ? -1 : 4 * (64 / lpr); }

int pn2_colsum_finalize(const float* partial, int nblk, int C, int ld, float* out, int accumulate, void* stream) {
    if (!partial || !out || nblk < 1 || C < 1) return -1;
    hipLaunchKernelGGL(colsum_finalize_k, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, partial, nblk, C, ld, out, accumulate);
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

int pn2_colsum_unit(int dt, int C) { const int V = dt == PN2_F32 ? 4 : 8; int cvp = pow2ceil(C / V); if (cvp > 256) cvp = 256; return 256 / cvp; }

int pn2_dwconv3x3(int dt, const void* x, const float* w, const float* b, void* z, void* y_gelu, int N, int H, int W, int C, int flip, int accumulate, void* stream) {
    if (!x || !w || !z) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -2;
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(dwconv3x3_k<T>, dim3(grid_for((size_t)N * H * W * (C / V))), dim3(256), 0, (hipStream_t)stream, (const T*)x, w, b, (T*)z, (T*)y_gelu,
                                          N, H, W, C, flip, accumulate); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_gelu_bwd(int dt, const void* dy, const void* z, void* dz, long long n, void* stream) {
    if (!dy || !z || !dz) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (n % V) return -2;
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(gelu_bwd_k<T>, dim3(grid_for((size_t)n / V)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)z, (T*)dz, (size_t)n / V); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_dwconv3x3_wgrad(int dt, const void* dz, const void* x, float* partial, int nblk, int N, int H, int W, int C, void* stream) {
    if (!dz || !x || !partial || nblk < 1) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -2;
    int cvp = pow2ceil(C / V); if (cvp > 256) cvp = 256;
    const int M = N * H * W, pix = rows_for(M, 256 / cvp);
    if ((M + pix - 1) / pix != nblk) return -2;            // nblk must be pn2_rows_blocks(N*H*W, pn2_colsum_unit(dt, C))
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(dwconv3x3_wgrad_k<T>, dim3(nblk), dim3(256), 256 * TT<T>::VEC * 4, (hipStream_t)stream, (const T*)dz, (const T*)x, partial, N, H, W, C, pix, cvp); })
    PN2_CHECK_LAUNCH();
    return 0;
}

static int attn_geom(int Nkv, int heads, int head_dim) { return (head_dim == 64 && Nkv >= 1 && Nkv <= 64 * AT_MAXK && heads >= 1) ? 0 : -2; }

int pn2_attn_fwd(int dt, const void* q, int ld_q, const void* kv, int ld_kv, void* out, int ld_o, float* lse, int B, int Nq, int Nkv, int heads, int head_dim,
                 float scale, void* stream) {
    if (!q || !kv || !out || !lse) return -1;
    if (int rc = attn_geom(Nkv, heads, head_dim)) return rc;
    const int NP = (Nkv + 63) & ~63;
    const size_t lds = (size_t)2 * 64 * (NP + 1) * 4;
    const int qpb = 64;
    VIT_DISPATCH(dt, {
        static bool done = false;
        if (!done) { hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 257 * 4); done = true; }
        hipLaunchKernelGGL(attn_fwd_k<T>, dim3((Nq + qpb - 1) / qpb, heads, B), dim3(256), lds, (hipStream_t)stream, (const T*)q, ld_q, (const T*)kv, ld_kv, (T*)out, ld_o, lse,
                           Nq, Nkv, heads, scale, qpb); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_attn_bwd(int dt, const void* q, int ld_q, const void* kv, int ld_kv, const void* dout, int ld_do, const float* lse, void* dq, int ld_dq, void* dkv, int ld_dkv,
                 float* P_scratch, float* dS_scratch, int B, int Nq, int Nkv, int heads, int head_dim, float scale, void* stream) {
    if (!q || !kv || !dout || !lse || !dq || !dkv || !P_scratch || !dS_scratch) return -1;
    if (int rc = attn_geom(Nkv, heads, head_dim)) return rc;
    const int NP = (Nkv + 63) & ~63;
    const size_t lds = (size_t)2 * 64 * (NP + 1) * 4;
    const int qpb = 64;
    VIT_DISPATCH(dt, {
        static bool done = false;
        if (!done) { hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_q_k<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 64 * 257 * 4); done = true; }
        hipLaunchKernelGGL(attn_bwd_q_k<T>, dim3((Nq + qpb - 1) / qpb, heads, B), dim3(256), lds, (hipStream_t)stream, (const T*)q, ld_q, (const T*)kv, ld_kv, (const T*)dout, ld_do,
                           lse, (T*)dq, ld_dq, P_scratch, dS_scratch, Nq, Nkv, heads, scale, qpb);
        hipLaunchKernelGGL(attn_bwd_kv_k<T>, dim3((Nkv + AT_KC - 1) / AT_KC, heads, B), dim3(256), 0, (hipStream_t)stream, (const T*)q, ld_q, (const T*)dout, ld_do, P_scratch,
                           dS_scratch, (T*)dkv, ld_dkv, Nq, Nkv, heads, scale); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_scale_samples(int dt, const void* x, void* y, const float* scale, int N, long long per_sample, void* stream) {
    if (!x || !y || !scale || N < 1) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (per_sample % V) return -2;
    VIT_DISPATCH(dt, { hipLaunchKernelGGL(scale_samples_k<T>, dim3(grid_for((size_t)N * per_sample / V)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, scale,
                                          (size_t)per_sample / V, (size_t)N * per_sample / V); })
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
