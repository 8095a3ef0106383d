// pn2_bn.hip — BatchNorm2d (train / eval) fused with the ReLU and residual add that follow it.
// Reference semantics: nn.BatchNorm2d defaults (eps 1e-5, momentum 0.1, biased variance for
// normalisation, unbiased for running_var) at /root/reference/binary_seg/lib/pranet.py:37,41-42 and
// /root/reference/binary_seg/lib/Res2Net_v1b.py:33,45,50,103,106,110,135; ReLU / residual at
// Res2Net_v1b.py:63,72,88-89 and pranet.py:82,358-360.
//
// All of these are HBM-bound streaming kernels: 16-byte vectors along the NHWC channel axis, one pass.
// Batch statistics arrive as per-row-block partials from the conv epilogue (pn2_conv.hip), so the
// forward never re-reads the conv output to compute them.
#include <cstdlib>
#include "pn2_common.h"
#include "../../include/pn2.h"

namespace {

template <typename T, int W> struct VL {   // W == VEC: 16-byte vector ; W == 1: scalar
    __device__ static __forceinline__ void load(const T* p, float* f) {
        if constexpr (W == 1) f[0] = TT<T>::ld(p);
        else TT<T>::unpack(*reinterpret_cast<const uint4*>(p), f);
    }
    __device__ static __forceinline__ void store(T* p, const float* f) {
        if constexpr (W == 1) TT<T>::st(p, f[0]);
        else *reinterpret_cast<uint4*>(p) = TT<T>::pack(f);
    }
};

// ---------------------------------------------------------------------------------------------
// reduce [nblk][Cp] partial rows: CPB channels x (256 / CPB) row-lanes per block, accumulate in double.  CPB = 8 for short partial buffers;
// tall ones (the stem / layer1 convs leave thousands of rows) take fewer channels per block so that every thread walks <= ~8 rows.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_partials(const float* p1, const float* p2, int nblk, int Cn, int Cp, int c, int rl, int RL, double& s1, double& s2) {
    // Cn = channels of this BN, Cp = row stride of the partial buffers, RL = row lanes
    s1 = 0.0; s2 = 0.0;
    if (c < Cn) {
        int r = rl;
        for (; r + 3 * RL < nblk; r += 4 * RL) {        // 4 independent row loads in flight per thread
            const float a0 = p1[(size_t)r * Cp + c], a1 = p1[(size_t)(r + RL) * Cp + c], a2 = p1[(size_t)(r + 2 * RL) * Cp + c], a3 = p1[(size_t)(r + 3 * RL) * Cp + c];
            const float b0 = p2[(size_t)r * Cp + c], b1 = p2[(size_t)(r + RL) * Cp + c], b2 = p2[(size_t)(r + 2 * RL) * Cp + c], b3 = p2[(size_t)(r + 3 * RL) * Cp + c];
            s1 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            s2 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        if (r < nblk) {         // <= 3 rows left: one more batch of loads (absent rows add an exact +0.0), summed in row order
            float a[3], b[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool ok = r + u * RL < nblk;
                const size_t rr = ok ? (size_t)(r + u * RL) : (size_t)r;
                a[u] = p1[rr * Cp + c]; b[u] = p2[rr * Cp + c];
                if (!ok) { a[u] = 0.f; b[u] = 0.f; }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) { s1 += (double)a[u]; s2 += (double)b[u]; }
        }
    }
}

// sum (s1, s2) over the row lanes that share a channel (thread = rl*CPB + cl): butterfly inside each wave, then the 4 wave partials
// through LDS.  Result valid in the threads < CPB.  Fixed order -> deterministic.
__device__ __forceinline__ void block_reduce_rows(double (*sh)[4][8], int CPB, double& s1, double& s2) {
    for (int off = CPB; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane < CPB) { sh[0][wid][lane] = s1; sh[1][wid][lane] = s2; }
    __syncthreads();
    if ((int)threadIdx.x < CPB) {
        s1 = (sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x]) + (sh[0][2][threadIdx.x] + sh[0][3][threadIdx.x]);
        s2 = (sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x]) + (sh[1][2][threadIdx.x] + sh[1][3][threadIdx.x]);
    }
}
inline int finalize_cpb(int nblk) {
    int cpb = 8;
    while (cpb > 1 && nblk / (256 / cpb) > 8) cpb >>= 1;
    return cpb;
}

__device__ __forceinline__ void bn_finalize_body(const float* __restrict__ psum, const float* __restrict__ psq, int nblk, const pn2_bn_desc& d,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var,
                                                 float* scale, float* shift, float* mean_o, float* invstd_o, int CPB, int bid, double (*sh)[4][8]) {
    const int cl = threadIdx.x % CPB, rl = threadIdx.x / CPB, c = bid * CPB + cl;
    // the per-channel parameters are requested BEFORE the partial rows are walked: loaded after the reduction they would add a second,
    // fully exposed memory latency to a kernel that is nothing but latency (156 of these per step)
    const int lc0 = (rl == 0 && c < d.Cp) ? phys2log(c, d.gw, d.gwp, d.C) : -1;
    float pg = 0.f, pb = 0.f, prm = 0.f, prv = 0.f;
    if (lc0 >= 0) { pg = gamma[lc0]; pb = beta[lc0]; if (running_mean) { prm = running_mean[lc0]; prv = running_var[lc0]; } }
    double s1, s2, k0 = 0.0;
    if (d.tile_rows > 0) {
        // Chan merge of per-tile (mean_t, M2_t): with K = the mean of tile 0 as a shift,
        //   mean = K + A / M,  M2 = sum M2_t + B - A^2 / M,   A = sum n_t (mean_t - K),  B = sum n_t (mean_t - K)^2   (all in double)
        // s1 carries A, s2 carries sum M2_t + B; the A^2 / M term is applied after the block reduction.
        const int ldp = d.ldp ? d.ldp : d.Cp, RL = 256 / CPB;
        s1 = 0.0; s2 = 0.0;
        if (c < d.Cp) {
            const float k0f = psum[c];
            k0 = (double)k0f;
            for (int r = rl; r < nblk; r += 4 * RL) {          // the 8 loads of four tiles are in flight together: the walk is latency, not arithmetic
                float pm[4], pq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int rr = min(r + u * RL, nblk - 1);
                    pm[u] = psum[(size_t)rr * ldp + c]; pq[u] = psq[(size_t)rr * ldp + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (r + u * RL < nblk) {
                        const int nt = min(d.tile_rows, d.M - (r + u * RL) * d.tile_rows);
                        const double dm = (double)pm[u] - k0;
                        s1 += nt * dm; s2 += (double)pq[u] + nt * dm * dm;
                    }
                }
            }
        }
    } else reduce_partials(psum, psq, nblk, d.Cp, d.ldp ? d.ldp : d.Cp, c, rl, 256 / CPB, s1, s2);
    block_reduce_rows(sh, CPB, s1, s2);
    if (rl == 0 && c < d.Cp) {
        const int lc = phys2log(c, d.gw, d.gwp, d.C);
        if (lc < 0) { scale[c] = 0.f; shift[c] = 0.f; mean_o[c] = 0.f; invstd_o[c] = 0.f; return; }
        double mean, var;
        if (d.tile_rows > 0) { mean = k0 + s1 / d.M; var = (s2 - s1 * s1 / d.M) / d.M; }
        else { mean = s1 / d.M; var = s2 / d.M - mean * mean; }
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)d.eps));
        const float sc = pg * invstd;
        scale[c] = sc; shift[c] = pb - (float)mean * sc;
        mean_o[c] = (float)mean; invstd_o[c] = invstd;
        if (running_mean) {
            const double unb = d.M > 1 ? var * ((double)d.M / (double)(d.M - 1)) : var;
            running_mean[lc] = (1.f - d.momentum) * prm + d.momentum * (float)mean;
            running_var[lc] = (1.f - d.momentum) * prv + d.momentum * (float)unb;
        }
    }
}

__global__ __launch_bounds__(256) void bn_finalize_k(const float* __restrict__ psum, const float* __restrict__ psq, int nblk, pn2_bn_desc d,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var,
                                                     float* scale, float* shift, float* mean_o, float* invstd_o, int CPB) {
    __shared__ double sh[2][4][8];
    bn_finalize_body(psum, psq, nblk, d, gamma, beta, running_mean, running_var, scale, shift, mean_o, invstd_o, CPB, blockIdx.x, sh);
}
// table-driven launches (pn2_*_multi): the same bodies over a DEVICE job table - the independent chains of a model (the three RFB modules and
// their three branches each, the three parallel 3x3 convs of a Res2Net stage block) advance in lock step, one launch per kernel kind and position
__global__ __launch_bounds__(256) void bn_finalize_tab(const pn2_bnfin_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    __shared__ double sh[2][4][8];
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_bnfin_job j = jobs[jb];
    bn_finalize_body(j.psum, j.psq, j.nblk, j.d, j.gamma, j.beta, j.running_mean, j.running_var, j.scale, j.shift, j.mean, j.invstd, j.cpb, blockIdx.x - bstart[jb], sh);
}

__global__ void bn_eval_prepare_k(pn2_bn_desc d, const float* gamma, const float* beta, const float* rm, const float* rv, float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.Cp) return;
    const int lc = phys2log(c, d.gw, d.gwp, d.C);
    if (lc < 0) { scale[c] = 0.f; shift[c] = 0.f; return; }
    const float sc = gamma[lc] / sqrtf(rv[lc] + d.eps);
    scale[c] = sc; shift[c] = beta[lc] - rm[lc] * sc;
}

// every eval-mode BatchNorm of a model in one launch (inference graphs: one node instead of one per layer)
__global__ __launch_bounds__(256) void bn_eval_prepare_tab(const pn2_bnprep_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_bnprep_job j = jobs[jb];
    const int c = (blockIdx.x - bstart[jb]) * 256 + threadIdx.x;
    if (c >= j.d.Cp) return;
    const int lc = phys2log(c, j.d.gw, j.d.gwp, j.d.C);
    if (lc < 0) { j.scale[c] = 0.f; j.shift[c] = 0.f; return; }
    const float sc = j.gamma[lc] / sqrtf(j.running_var[lc] + j.d.eps);
    j.scale[c] = sc; j.shift[c] = j.beta[lc] - j.running_mean[lc] * sc;
}

// ---------------------------------------------------------------------------------------------
// y = act(x*scale + shift + res)
// ---------------------------------------------------------------------------------------------
template <typename Ti, typename To, int W>
__device__ __forceinline__ void affine_act_body(const Ti* __restrict__ x, int ld_x, To* __restrict__ y, int ld_y, int M, int C,
                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                const Ti* __restrict__ res, int ld_res, int relu, unsigned bid, unsigned nb) {
    const int CV = C / W;
    const size_t total = (size_t)M * CV;
    for (size_t idx = (size_t)bid * 256 + threadIdx.x; idx < total; idx += (size_t)nb * 256) {
        int c; const int m = (int)divmod_idx(idx, CV, c); c *= W;
        float v[W], r[W];
        VL<Ti, W>::load(x + (size_t)m * ld_x + c, v);
        if (res) VL<Ti, W>::load(res + (size_t)m * ld_res + c, r);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float t = v[e];
            if (scale) t = fmaf(t, scale[c + e], shift[c + e]);
            else if (shift) t += shift[c + e];
            if (res) t += r[e];
            v[e] = relu ? (relu == 2 ? fminf(fmaxf(t, 0.f), 6.f) : fmaxf(t, 0.f)) : t;
        }
        if constexpr (sizeof(Ti) == sizeof(To)) VL<To, W>::store(y + (size_t)m * ld_y + c, v);
        else {
#pragma unroll
            for (int e = 0; e < W; ++e) TT<To>::st(y + (size_t)m * ld_y + c + e, v[e]);
        }
    }
}
template <typename Ti, typename To, int W>
__global__ __launch_bounds__(256) void affine_act_k(const Ti* __restrict__ x, int ld_x, To* __restrict__ y, int ld_y, int M, int C,
                                                    const float* __restrict__ scale, const float* __restrict__ shift,
                                                    const Ti* __restrict__ res, int ld_res, int relu) {
    affine_act_body<Ti, To, W>(x, ld_x, y, ld_y, M, C, scale, shift, res, ld_res, relu, blockIdx.x, gridDim.x);
}
// table form of the element-wise kernel for bf16 -> fp32 outputs (the K-channel head maps: W = 1): the head convs of independent chains at one lock-step position
__global__ __launch_bounds__(256) void affine_act_tab_f32out(const pn2_affine_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_affine_job j = jobs[jb];
    affine_act_body<bf16_t, float, 1>((const bf16_t*)j.x, j.ld_x, (float*)j.y, j.ld_y, j.M, j.C, j.scale, j.shift, (const bf16_t*)j.res, j.ld_res, j.relu,
                                      blockIdx.x - bstart[jb], bstart[jb + 1] - bstart[jb]);
}

// ---------------------------------------------------------------------------------------------
// backward pass 1: partial sums of dz and dz*xhat over row chunks
// ---------------------------------------------------------------------------------------------
// V consecutive per-channel fp32 parameters as 16-byte loads (c is a multiple of V; slices start at multiples of 8 channels).  As scalar loads
// they were 56 of the 72 vector-memory instructions a thread of the BatchNorm-backward pass issued.
template <int V>
__device__ __forceinline__ void ldpar(const float* __restrict__ p, float (&v)[V]) {
#pragma unroll
    for (int e = 0; e < V; e += 4) {
        const float4 t = *reinterpret_cast<const float4*>(p + e);
        v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
    }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm + ReLU + MaxPool(3, 2, 1) forward as ONE pass (the stem of Res2Net_v1b.py:137-139: bn1 -> relu -> maxpool): the normalised 176 x 176 x 64 activation is
// consumed by the pool alone, so it is never written - the pass pools T(relu(raw*scale + shift)) on the fly: values and argmax bytes exactly as the normalise pass
// followed by maxpool3x3s2_fwd give them.  (The backward keeps its three passes: gathering the pooled gradient inside the two BatchNorm passes was 4x slower)
// ---------------------------------------------------------------------------------------------
template <typename T, int W>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_k(const T* __restrict__ x, int ld_x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             T* __restrict__ y, int ld_y, unsigned char* __restrict__ arg, int N, int H, int Wd, int C, int OH, int OW) {
    const int CV = C / W;
    const size_t total = (size_t)N * OH * OW * CV;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int cv = (int)(idx % CV); unsigned p = (unsigned)(idx / CV);
        const int ox = (int)(p % OW); p /= OW; const int oy = (int)(p % OH); const int n = (int)(p / OH);
        float sc[W], sh[W], best[W]; int bi[W];
        ldpar<W>(scale + cv * W, sc); ldpar<W>(shift + cv * W, sh);
#pragma unroll
        for (int e = 0; e < W; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * 2 - 1 + r; if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const int ix = ox * 2 - 1 + s_; if ((unsigned)ix >= (unsigned)Wd) continue;
                float v[W];
                VL<T, W>::load(x + ((size_t)(n * H + iy) * Wd + ix) * ld_x + cv * W, v);
#pragma unroll
                for (int e = 0; e < W; ++e) v[e] = fmaxf(fmaf(v[e], sc[e], sh[e]), 0.f);
                if constexpr (sizeof(T) == 2 && W == 8) { const uint4 pk = TT<T>::pack(v); TT<T>::unpack(pk, v); }      // the value the normalise pass would have stored
#pragma unroll
                for (int e = 0; e < W; ++e) if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 3 + s_; }
            }
        }
        const size_t o = (size_t)(n * OH + oy) * OW + ox;
        VL<T, W>::store(y + o * ld_y + cv * W, best);
        if constexpr (W == 8) {
            uint2 a;
            a.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
            a.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
            *reinterpret_cast<uint2*>(arg + o * C + cv * W) = a;
        } else {
#pragma unroll
            for (int e = 0; e < W; ++e) arg[o * C + cv * W + e] = (unsigned char)bi[e];
        }
    }
}

// quotient / remainder of a < 2^24 by d (quotient < 2^20): float reciprocal + one correction step instead of a ~35-instruction integer division
__device__ __forceinline__ unsigned fdivmod(unsigned a, unsigned d, float rcp_d, unsigned& rem) {
    unsigned q = (unsigned)((float)a * rcp_d);
    int r = (int)a - (int)(q * d);
    if (r < 0) { --q; r += (int)d; } else if (r >= (int)d) { ++q; r -= (int)d; }
    rem = (unsigned)r;
    return q;
}

// Backward of that op in QUAD form (even H, W).  MaxPool(3, 2, 1): input pixel (iy, ix) belongs to the windows oy = (iy + 1 - r) / 2, r in {0, 1, 2} where integral - so
// the 2 x 2 input quad rows {2k, 2k+1} x cols {2j, 2j+1} is served by exactly the four windows (k, k+1) x (j, j+1).  A thread takes one quad and one channel vector:
// 4 pooled-gradient vectors + 4 argmax vectors + 4 raw vectors give it the gradient of all four pixels (each the sum maxpool3x3s2_bwd forms, in its tap order, rounded to T),
// without the materialised 176 x 176 gradient tensor: no pool-backward launch, and the two BatchNorm passes read 3/8 of the bytes.  (The same gather per ROW inside the generic
// BatchNorm passes was 4x slower: up to four dependent look-ups per row and a pixel decode each; here the decode is per quad and every load is issued up front.)
template <typename T, int W> struct QuadGrad {
    float dy[4][W], xr[4][W];            // pixel order: (2k,2j) (2k,2j+1) (2k+1,2j) (2k+1,2j+1)
    __device__ __forceinline__ void load(const T* __restrict__ dp, int ld_dp, const unsigned char* __restrict__ arg, const T* __restrict__ x, int ld_x,
                                         int n, int k, int j, int H, int Wd, int OH, int OW, int C, int c) {
        float g[4][W]; unsigned char a[4][W];          // windows (k,j) (k,j+1) (k+1,j) (k+1,j+1)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oy = k + (q >> 1), ox = j + (q & 1);
            const bool ok = oy < OH && ox < OW;
            const size_t o = ((size_t)n * OH + (ok ? oy : k)) * OW + (ok ? ox : j);
            VL<T, W>::load(dp + o * ld_dp + c, g[q]);
            if constexpr (W == 8) { const uint2 t = *reinterpret_cast<const uint2*>(arg + o * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[q][e] = (unsigned char)(t.x >> (8 * e)); a[q][4 + e] = (unsigned char)(t.y >> (8 * e)); } }
            else { const unsigned t = *reinterpret_cast<const unsigned*>(arg + o * C + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[q][e] = (unsigned char)(t >> (8 * e)); }
            if (!ok) {
#pragma unroll
                for (int e = 0; e < W; ++e) a[q][e] = 255;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) VL<T, W>::load(x + ((size_t)(n * H + 2 * k + (p >> 1)) * Wd + 2 * j + (p & 1)) * ld_x + c, xr[p]);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            // taps in maxpool3x3s2_bwd's order (r ascending, then s ascending); tap = r*3 + s of the window that holds the pixel at (r, s)
            dy[0][e] = a[0][e] == 4 ? g[0][e] : 0.f;                                                                  // (2k, 2j): window (k, j), r = 1, s = 1
            dy[1][e] = (a[1][e] == 3 ? g[1][e] : 0.f) + (a[0][e] == 5 ? g[0][e] : 0.f);                               // (2k, 2j+1): s = 0 -> (k, j+1); s = 2 -> (k, j)
            dy[2][e] = (a[2][e] == 1 ? g[2][e] : 0.f) + (a[0][e] == 7 ? g[0][e] : 0.f);                               // (2k+1, 2j): r = 0 -> (k+1, j); r = 2 -> (k, j)
            dy[3][e] = (((a[3][e] == 0 ? g[3][e] : 0.f) + (a[2][e] == 2 ? g[2][e] : 0.f)) + (a[1][e] == 6 ? g[1][e] : 0.f)) + (a[0][e] == 8 ? g[0][e] : 0.f);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int e = 0; e < W; ++e) dy[p][e] = TT<T>::round(dy[p][e]);          // what maxpool3x3s2_bwd stores
    }
};

template <typename T, int W>
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_k(const T* __restrict__ dp, int ld_dp, const unsigned char* __restrict__ arg, const T* __restrict__ x, int ld_x,
                                                            int N, int H, int Wd, int OH, int OW, int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ msc, const float* __restrict__ msh, float* __restrict__ p1, float* __restrict__ p2) {
    // block = CV channel vectors x (256 / CV) quad lanes; quads dealt round-robin to (block, lane): fixed order -> deterministic partial rows [gridDim.x][C]
    __shared__ float sh[2][256][W];
    const int CV = C / W, QL = 256 / CV, cv = threadIdx.x % CV, ql = threadIdx.x / CV, c = cv * W;
    const int QH = H >> 1, QW = Wd >> 1;
    const unsigned nq = (unsigned)N * QH * QW;
    const float rqw = __builtin_amdgcn_rcpf((float)QW), rqh = __builtin_amdgcn_rcpf((float)QH);
    float mu[W], is[W], ks[W], kh[W], a1[W], a2[W];
    ldpar<W>(mean + c, mu); ldpar<W>(invstd + c, is); ldpar<W>(msc + c, ks); ldpar<W>(msh + c, kh);
#pragma unroll
    for (int e = 0; e < W; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (ql < QL) {
        for (unsigned q = blockIdx.x * QL + ql; q < nq; q += gridDim.x * QL) {
            unsigned j, k; const unsigned t = fdivmod(q, (unsigned)QW, rqw, j), n = fdivmod(t, (unsigned)QH, rqh, k);
            QuadGrad<T, W> G;
            G.load(dp, ld_dp, arg, x, ld_x, (int)n, (int)k, (int)j, H, Wd, OH, OW, C, c);
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int e = 0; e < W; ++e) {
                    const float dz = (fmaf(G.xr[p][e], ks[e], kh[e]) > 0.f) ? G.dy[p][e] : 0.f;
                    a1[e] += dz; a2[e] += dz * (G.xr[p][e] - mu[e]) * is[e];
                }
        }
    }
#pragma unroll
    for (int e = 0; e < W; ++e) { sh[0][threadIdx.x][e] = a1[e]; sh[1][threadIdx.x][e] = a2[e]; }
    __syncthreads();
    if (ql == 0) {
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float s1 = 0.f, s2 = 0.f;
            for (int r = 0; r < QL; ++r) { s1 += sh[0][r * CV + cv][e]; s2 += sh[1][r * CV + cv][e]; }
            p1[(size_t)blockIdx.x * C + c + e] = s1; p2[(size_t)blockIdx.x * C + c + e] = s2;
        }
    }
}

template <typename T, int W>
__global__ __launch_bounds__(256) void pool_bn_bwd_apply_k(const T* __restrict__ dp, int ld_dp, const unsigned char* __restrict__ arg, const T* __restrict__ x, int ld_x,
                                                           int N, int H, int Wd, int OH, int OW, int C, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, const float* __restrict__ msc, const float* __restrict__ msh, T* __restrict__ dz_, int ld_dz) {
    const int CV = C / W, QL = 256 / CV, cv = threadIdx.x % CV, ql = threadIdx.x / CV, c = cv * W;
    const int QH = H >> 1, QW = Wd >> 1;
    const unsigned nq = (unsigned)N * QH * QW;
    const float rqw = __builtin_amdgcn_rcpf((float)QW), rqh = __builtin_amdgcn_rcpf((float)QH);
    if (ql >= QL) return;
    float ka[W], kb[W], kd[W], kmu[W], ks[W], kh[W], c1[W], c2[W], is[W];
    ldpar<W>(msc + c, ks); ldpar<W>(msh + c, kh);
    ldpar<W>(coef + c, ka); ldpar<W>(coef + C + c, c1); ldpar<W>(coef + 2 * C + c, c2); ldpar<W>(invstd + c, is); ldpar<W>(mean + c, kmu);
#pragma unroll
    for (int e = 0; e < W; ++e) { kb[e] = __fmul_rn(__fmul_rn(-ka[e], c2[e]), is[e]); kd[e] = __fmul_rn(-ka[e], c1[e]); }        // as bn_bwd_apply_rows_body
    for (unsigned q = blockIdx.x * QL + ql; q < nq; q += gridDim.x * QL) {
        unsigned j, k; const unsigned t = fdivmod(q, (unsigned)QW, rqw, j), n = fdivmod(t, (unsigned)QH, rqh, k);
        QuadGrad<T, W> G;
        G.load(dp, ld_dp, arg, x, ld_x, (int)n, (int)k, (int)j, H, Wd, OH, OW, C, c);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float o[W];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const float dz = (fmaf(G.xr[p][e], ks[e], kh[e]) > 0.f) ? G.dy[p][e] : 0.f;
                o[e] = __fmaf_rn(ka[e], dz, __fmaf_rn(kb[e], G.xr[p][e] - kmu[e], kd[e]));
            }
            VL<T, W>::store(dz_ + ((size_t)((int)n * H + 2 * (int)k + (p >> 1)) * Wd + 2 * (int)j + (p & 1)) * ld_dz + c, o);
        }
    }
}

template <typename T, typename Tdy, int W, bool LEAN = false>       // LEAN: no stored-activation mask (y) - its staging registers disappear
__device__ __forceinline__ void bn_bwd_reduce_body(const Tdy* __restrict__ dy, int ld_dy, int Cdy, const T* __restrict__ y_, int ld_y,
                                                   const T* __restrict__ x, int ld_x, int M, int Cp, const float* __restrict__ mean,
                                                   const float* __restrict__ invstd, float* __restrict__ p1, float* __restrict__ p2,
                                                   int rows_per_blk, int CVP, const float* __restrict__ msc, const float* __restrict__ msh, int r6, int bid) {
    extern __shared__ __attribute__((aligned(16))) float shf[];   // [2][R][CVP*W]
    const T* __restrict__ y = LEAN ? nullptr : y_;
    const int CV = Cp / W;
    const int R = 256 / CVP;
    const int cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int r0 = bid * rows_per_blk;
    int r1 = r0 + rows_per_blk; if (r1 > M) r1 = M;
    for (int cvb = 0; cvb < CV; cvb += CVP) {
        const int cv = cvb + cvl, c = cv * W;
        float a1[W], a2[W], mu[W], is[W], ks[W], kh[W];
#pragma unroll
        for (int e = 0; e < W; ++e) { a1[e] = 0.f; a2[e] = 0.f; mu[e] = 0.f; is[e] = 0.f; ks[e] = 0.f; kh[e] = 1.f; }
        if (cv < CV) {
#pragma unroll
            for (int e = 0; e < W; ++e) { if constexpr (W % 4 != 0) { mu[e] = mean[c + e]; is[e] = invstd[c + e]; if (msc) { ks[e] = msc[c + e]; kh[e] = msh[c + e]; } } }
            if constexpr (W % 4 == 0) { ldpar<W>(mean + c, mu); ldpar<W>(invstd + c, is); if (msc) { ldpar<W>(msc + c, ks); ldpar<W>(msh + c, kh); } }
            for (int m = r0 + rl; m < r1; m += R * 4) {
                float g[4][W], xv[4][W], yv[4][W];
#pragma unroll
                for (int u = 0; u < 4; ++u) {          // issue all loads of 4 rows before using any
                    const int mm = m + u * R;
                    if (mm < r1) {
                        if (W > 1 || c < Cdy) VL<Tdy, W>::load(dy + (size_t)mm * ld_dy + c, g[u]); else g[u][0] = 0.f;
                        VL<T, W>::load(x + (size_t)mm * ld_x + c, xv[u]);
                        if (y) VL<T, W>::load(y + (size_t)mm * ld_y + c, yv[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (m + u * R < r1) {
#pragma unroll
                        for (int e = 0; e < W; ++e) {
                            // ReLU mask: from the stored output y, or recomputed from the raw conv output (same fmaf as the forward)
                            const bool off = y ? (!(yv[u][e] > 0.f) || (r6 && yv[u][e] >= 6.f)) : (msc && !(fmaf(xv[u][e], ks[e], kh[e]) > 0.f));
                            const float dz = off ? 0.f : g[u][e];
                            a1[e] += dz; a2[e] += dz * (xv[u][e] - mu[e]) * is[e];
                        }
                    }
                }
            }
        }
        // combine the row lanes: butterfly over the lanes of a wave that share a channel vector (CVP < 64), then <= 4 partial
        // rows (one per wave, or one per row lane when CVP >= 64) through LDS.  Fixed order -> deterministic.
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        if (CVP < 64) {
            for (int off = CVP; off < 64; off <<= 1) {
#pragma unroll
                for (int e = 0; e < W; ++e) { a1[e] += __shfl_xor(a1[e], off); a2[e] += __shfl_xor(a2[e], off); }
            }
        }
        const int NR = CVP < 64 ? 4 : R;
        const int pr = CVP < 64 ? wid : rl;
        const bool owner = CVP >= 64 || lane < CVP;
        if (owner) {
#pragma unroll
            for (int e = 0; e < W; ++e) { shf[(pr * CVP + cvl) * W + e] = a1[e]; shf[((NR + pr) * CVP + cvl) * W + e] = a2[e]; }
        }
        __syncthreads();
        if (pr == 0 && owner && cv < CV) {
#pragma unroll
            for (int e = 0; e < W; ++e) {
                float s1 = 0.f, s2 = 0.f;
                for (int r = 0; r < NR; ++r) { s1 += shf[(r * CVP + cvl) * W + e]; s2 += shf[((NR + r) * CVP + cvl) * W + e]; }
                p1[(size_t)bid * Cp + c + e] = s1; p2[(size_t)bid * Cp + c + e] = s2;
            }
        }
        __syncthreads();
    }
}

template <typename T, typename Tdy, int W, bool LEAN = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_k(const Tdy* __restrict__ dy, int ld_dy, int Cdy, const T* __restrict__ y, int ld_y,
                                                       const T* __restrict__ x, int ld_x, int M, int Cp, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, float* __restrict__ p1, float* __restrict__ p2,
                                                       int rows_per_blk, int CVP, const float* __restrict__ msc, const float* __restrict__ msh, int r6) {
    bn_bwd_reduce_body<T, Tdy, W, LEAN>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, p1, p2, rows_per_blk, CVP, msc, msh, r6, blockIdx.x);
}
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_tab(const pn2_bnreduce_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_bnreduce_job j = jobs[jb];
    bn_bwd_reduce_body<T, T, TT<T>::VEC>((const T*)j.dy, j.ld_dy, j.Cp, (const T*)j.y, j.ld_y, (const T*)j.x, j.ld_x, j.M, j.Cp, j.mean, j.invstd, j.p1, j.p2,
                                         j.rows_per_blk, j.cvp, j.msc, j.msh, j.r6, blockIdx.x - bstart[jb]);
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_tab_f32dy(const pn2_bnreduce_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_bnreduce_job j = jobs[jb];
    bn_bwd_reduce_body<bf16_t, float, 1>((const float*)j.dy, j.ld_dy, j.pad_, (const bf16_t*)j.y, j.ld_y, (const bf16_t*)j.x, j.ld_x, j.M, j.Cp, j.mean, j.invstd, j.p1, j.p2,
                                         j.rows_per_blk, j.cvp, j.msc, j.msh, j.r6, blockIdx.x - bstart[jb]);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_k(const float* __restrict__ p1, const float* __restrict__ p2, int nblk, pn2_bn_desc d,
                                                         const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                         float* dgamma, float* dbeta, int accumulate, float* coef, int CPB) {
    __shared__ double sh[2][4][8];
    const int cl = threadIdx.x % CPB, rl = threadIdx.x / CPB, c = blockIdx.x * CPB + cl;
    const int lc0 = (rl == 0 && c < d.Cp) ? phys2log(c, d.gw, d.gwp, d.C) : -1;       // parameters first, see bn_finalize_k
    float pg = 0.f, pis = 0.f, pdb = 0.f, pdg = 0.f;
    if (lc0 >= 0) { pg = gamma[lc0]; pis = invstd[c]; if (accumulate) { pdb = dbeta[lc0]; pdg = dgamma[lc0]; } }
    double s1, s2;
    reduce_partials(p1, p2, nblk, d.Cp, d.ldp ? d.ldp : d.Cp, c, rl, 256 / CPB, s1, s2);
    const int cs = d.ldp ? d.ldp : d.Cp;     // plane stride of coef
    block_reduce_rows(sh, CPB, s1, s2);
    if (rl == 0 && c < d.Cp) {
        const int lc = phys2log(c, d.gw, d.gwp, d.C);
        if (lc < 0) { coef[c] = 0.f; coef[cs + c] = 0.f; coef[2 * cs + c] = 0.f; return; }
        dbeta[lc] = pdb + (float)s1; dgamma[lc] = pdg + (float)s2;
        coef[c] = pg * pis;
        coef[cs + c] = (float)(s1 / d.M);
        coef[2 * cs + c] = (float)(s2 / d.M);
    }
}

// pn2_bn_bwd_finalize with the partial rows of the channels coming from up to 4 producers (see pn2_bn_segs)
__device__ __forceinline__ void bn_bwd_finalize_seg_body(const pn2_bn_segs& sg, const pn2_bn_desc& d, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                         float* dgamma, float* dbeta, int accumulate, float* coef, int CPB, int bid, double (*sh)[4][8]) {
    const int cl = threadIdx.x % CPB, rl = threadIdx.x / CPB, c = bid * CPB + cl;
    int si = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) if (k < sg.nseg && c >= sg.c0[k]) si = k;
    const int cc = c - sg.c0[si];
    const int lc0 = (rl == 0 && c < d.Cp) ? phys2log(c, d.gw, d.gwp, d.C) : -1;       // parameters first, see bn_finalize_k
    float pg = 0.f, pis = 0.f, pdb = 0.f, pdg = 0.f;
    if (lc0 >= 0) { pg = gamma[lc0]; pis = invstd[c]; if (accumulate) { pdb = dbeta[lc0]; pdg = dgamma[lc0]; } }
    double s1, s2;
    reduce_partials(sg.p1[si], sg.p2[si], sg.nblk[si], c < d.Cp ? cc + 1 : 0, sg.ldp[si], cc, rl, 256 / CPB, s1, s2);
    const int cs = d.ldp ? d.ldp : d.Cp;     // plane stride of coef
    block_reduce_rows(sh, CPB, s1, s2);
    if (rl == 0 && c < d.Cp) {
        const int lc = phys2log(c, d.gw, d.gwp, d.C);
        if (lc < 0) { coef[c] = 0.f; coef[cs + c] = 0.f; coef[2 * cs + c] = 0.f; return; }
        dbeta[lc] = pdb + (float)s1; dgamma[lc] = pdg + (float)s2;
        coef[c] = pg * pis;
        coef[cs + c] = (float)(s1 / d.M);
        coef[2 * cs + c] = (float)(s2 / d.M);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_seg_k(pn2_bn_segs sg, pn2_bn_desc d, const float* __restrict__ gamma, const float* __restrict__ invstd,
                                                             float* dgamma, float* dbeta, int accumulate, float* coef, int CPB) {
    __shared__ double sh[2][4][8];
    bn_bwd_finalize_seg_body(sg, d, gamma, invstd, dgamma, dbeta, accumulate, coef, CPB, blockIdx.x, sh);
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_tab(const pn2_bnbfin_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    __shared__ double sh[2][4][8];
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_bnbfin_job j = jobs[jb];
    bn_bwd_finalize_seg_body(j.sg, j.d, j.gamma, j.invstd, j.dgamma, j.dbeta, j.accumulate, j.coef, j.cpb, blockIdx.x - bstart[jb], sh);
}

template <typename T, typename Tdy, int W>
__device__ __forceinline__ void bn_bwd_apply_body(const Tdy* __restrict__ dy, int ld_dy, int Cdy, const T* __restrict__ y, int ld_y,
                                                  const T* __restrict__ x, int ld_x, int M, int Cp, const float* __restrict__ mean,
                                                  const float* __restrict__ invstd, const float* __restrict__ coef, T* __restrict__ dx, int ld_dx,
                                                  T* __restrict__ dres, int ld_dres, int dres_accum, int r6, unsigned bid, unsigned nb) {
    const int CV = Cp / W;
    const size_t total = (size_t)M * CV;
    for (size_t idx = (size_t)bid * 256 + threadIdx.x; idx < total; idx += (size_t)nb * 256) {
        int c; const int m = (int)divmod_idx(idx, CV, c); c *= W;
        float g[W], xv[W], yv[W], o[W], rr[W];
        if (W > 1 || c < Cdy) VL<Tdy, W>::load(dy + (size_t)m * ld_dy + c, g); else g[0] = 0.f;
        if (coef) VL<T, W>::load(x + (size_t)m * ld_x + c, xv);
        if (y) VL<T, W>::load(y + (size_t)m * ld_y + c, yv);
        if (dres && dres_accum) VL<T, W>::load(dres + (size_t)m * ld_dres + c, rr);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const float dz = (y && (!(yv[e] > 0.f) || (r6 && yv[e] >= 6.f))) ? 0.f : g[e];
            if (coef) {
                const float xh = (xv[e] - mean[c + e]) * invstd[c + e];
                o[e] = coef[c + e] * (dz - coef[Cp + c + e] - xh * coef[2 * Cp + c + e]);
            } else o[e] = dz;
            rr[e] = (dres && dres_accum) ? rr[e] + dz : dz;
        }
        VL<T, W>::store(dx + (size_t)m * ld_dx + c, o);
        if (dres) VL<T, W>::store(dres + (size_t)m * ld_dres + c, rr);
    }
}
template <typename T, typename Tdy, int W>
__global__ __launch_bounds__(256) void bn_bwd_apply_k(const Tdy* __restrict__ dy, int ld_dy, int Cdy, const T* __restrict__ y, int ld_y,
                                                      const T* __restrict__ x, int ld_x, int M, int Cp, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ coef, T* __restrict__ dx, int ld_dx,
                                                      T* __restrict__ dres, int ld_dres, int dres_accum, int r6) {
    bn_bwd_apply_body<T, Tdy, W>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, coef, dx, ld_dx, dres, ld_dres, dres_accum, r6, blockIdx.x, gridDim.x);
}
// table forms for fp32 gradients of bf16 layers (PN2_MULTI_F32DY: the K-channel head maps; job.pad_ = Cdy, the gradient's channel count): element-wise apply, scalar reduce
__global__ __launch_bounds__(256) void bn_bwd_apply_tab_f32dy(const pn2_bnapply_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_bnapply_job j = jobs[jb];
    bn_bwd_apply_body<bf16_t, float, 1>((const float*)j.dy, j.ld_dy, j.pad_, (const bf16_t*)j.y, j.ld_y, (const bf16_t*)j.x, j.ld_x, j.M, j.Cp, j.mean, j.invstd, j.coef,
                                        (bf16_t*)j.dx, j.ld_dx, (bf16_t*)j.dres, j.ld_dres, j.dres_accum, j.r6, blockIdx.x - bstart[jb], bstart[jb + 1] - bstart[jb]);
}

// ---------------------------------------------------------------------------------------------
// row-streaming variants (16-byte vectors): a thread owns ONE channel vector and walks down the rows,
// so the per-channel parameters live in registers instead of being re-fetched for every element, and
// U independent row loads are in flight per thread.  256 threads = CVP channel-vectors x R row lanes.
// ---------------------------------------------------------------------------------------------
constexpr int RU = 4;   // rows in flight per thread

template <typename T>
__device__ __forceinline__ void affine_rows_body(const T* __restrict__ x, int ld_x, T* __restrict__ y, int ld_y, int M, int C,
                                                 const float* __restrict__ scale, const float* __restrict__ shift,
                                                 const T* __restrict__ res, int ld_res, int relu, int rows_per_blk, int CVP,
                                                 const T* __restrict__ add, int ld_add, T* __restrict__ y2, int ld_y2, int bid,
                                                 T* __restrict__ y3 = nullptr, int ld_y3 = 0, int c_lo = 0) {
    // add / y2 (optional): second output y2 = y + add - the "sp + spx[i+1]" of Bottle2neck.forward (Res2Net_v1b.py:68) written by the pass
    // that produces sp instead of by a separate element-wise launch
    // y3 (optional): the channels >= c_lo are ALSO written to y3[m][c - c_lo] - Bottle2neck's pass-through slice spx[3] lands in the concat buffer (Res2Net_v1b.py:78-79) without a copy launch
    constexpr int V = TT<T>::VEC;
    const int CV = C / V, R = 256 / CVP;
    const int cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int r0 = bid * rows_per_blk;
    int r1 = r0 + rows_per_blk; if (r1 > M) r1 = M;
    for (int cv = cvl; cv < CV; cv += CVP) {
        const int c = cv * V;
        float sc[V], sh[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (scale) ldpar<V>(scale + c, sc);
        if (shift) ldpar<V>(shift + c, sh);
        for (int m = r0 + rl; m < r1; m += R * RU) {
            uint4 vx[RU], vr[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int mm = m + u * R;
                if (mm < r1) {
                    vx[u] = *reinterpret_cast<const uint4*>(x + (size_t)mm * ld_x + c);
                    if (res) vr[u] = *reinterpret_cast<const uint4*>(res + (size_t)mm * ld_res + c);
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int mm = m + u * R;
                if (mm < r1) {
                    float v[V], r[V];
                    TT<T>::unpack(vx[u], v);
                    if (res) TT<T>::unpack(vr[u], r);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float t = fmaf(v[e], sc[e], sh[e]);
                        if (res) t += r[e];
                        v[e] = relu ? (relu == 2 ? fminf(fmaxf(t, 0.f), 6.f) : fmaxf(t, 0.f)) : t;
                    }
                    const uint4 pk = TT<T>::pack(v);
                    *reinterpret_cast<uint4*>(y + (size_t)mm * ld_y + c) = pk;
                    if (y3 && c >= c_lo) *reinterpret_cast<uint4*>(y3 + (size_t)mm * ld_y3 + (c - c_lo)) = pk;
                    if (y2) {                     // sum of the STORED (rounded) y and the other operand, as a separate add of the two tensors gives
                        float a[V];
                        TT<T>::unpack(pk, v);
                        TT<T>::unpack(*reinterpret_cast<const uint4*>(add + (size_t)mm * ld_add + c), a);
#pragma unroll
                        for (int e = 0; e < V; ++e) v[e] += a[e];
                        *reinterpret_cast<uint4*>(y2 + (size_t)mm * ld_y2 + c) = TT<T>::pack(v);
                    }
                }
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void affine_rows_k(const T* __restrict__ x, int ld_x, T* __restrict__ y, int ld_y, int M, int C,
                                                     const float* __restrict__ scale, const float* __restrict__ shift,
                                                     const T* __restrict__ res, int ld_res, int relu, int rows_per_blk, int CVP,
                                                     const T* __restrict__ add = nullptr, int ld_add = 0, T* __restrict__ y2 = nullptr, int ld_y2 = 0,
                                                     T* __restrict__ y3 = nullptr, int ld_y3 = 0, int c_lo = 0) {
    affine_rows_body<T>(x, ld_x, y, ld_y, M, C, scale, shift, res, ld_res, relu, rows_per_blk, CVP, add, ld_add, y2, ld_y2, blockIdx.x, y3, ld_y3, c_lo);
}
template <typename T>
__global__ __launch_bounds__(256) void affine_rows_tab(const pn2_affine_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_affine_job j = jobs[jb];
    affine_rows_body<T>((const T*)j.x, j.ld_x, (T*)j.y, j.ld_y, j.M, j.C, j.scale, j.shift, (const T*)j.res, j.ld_res, j.relu, j.rows_per_blk, j.cvp,
                        (const T*)j.add, j.ld_add, (T*)j.y2, j.ld_y2, blockIdx.x - bstart[jb]);
}

// LEAN: the common form inside a training step - ReLU mask recomputed from the raw conv output (or none), no stored y, no residual gradient: the
// y / dres staging registers disappear (183 -> ~100 VGPRs, twice the resident waves of a kernel that lives on memory-level parallelism)
template <typename T, bool LEAN>
__device__ __forceinline__ void bn_bwd_apply_rows_body(const T* __restrict__ dy, int ld_dy, const T* __restrict__ y_, int ld_y,
                                                       const T* __restrict__ x, int ld_x, int M, int Cp, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ coef, T* __restrict__ dx, int ld_dx,
                                                       T* __restrict__ dres_, int ld_dres, int dres_accum, int rows_per_blk, int CVP,
                                                       const float* __restrict__ msc, const float* __restrict__ msh, int r6, int bid) {
    constexpr int V = TT<T>::VEC;
    const T* __restrict__ y = LEAN ? nullptr : y_;
    T* __restrict__ dres = LEAN ? nullptr : dres_;
    const int CV = Cp / V, R = 256 / CVP;
    const int cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int r0 = bid * rows_per_blk;
    int r1 = r0 + rows_per_blk; if (r1 > M) r1 = M;
    for (int cv = cvl; cv < CV; cv += CVP) {
        const int c = cv * V;
        // dx = a*dz + b*(x - mu) + d  with  a = g, b = -g*c2*invstd, d = -g*c1   (g = gamma*invstd)
        float ka[V], kb[V], kd[V], kmu[V], ks[V], kh[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { ks[e] = 0.f; kh[e] = 1.f; ka[e] = 1.f; kb[e] = 0.f; kd[e] = 0.f; kmu[e] = 0.f; }
        if (msc) { ldpar<V>(msc + c, ks); ldpar<V>(msh + c, kh); }
        if (coef) {
            float c1[V], c2[V], is[V];
            ldpar<V>(coef + c, ka); ldpar<V>(coef + Cp + c, c1); ldpar<V>(coef + 2 * Cp + c, c2); ldpar<V>(invstd + c, is); ldpar<V>(mean + c, kmu);
#pragma unroll
            for (int e = 0; e < V; ++e) { kb[e] = __fmul_rn(__fmul_rn(-ka[e], c2[e]), is[e]); kd[e] = __fmul_rn(-ka[e], c1[e]); }
        }
        for (int m = r0 + rl; m < r1; m += R * RU) {
            uint4 vg[RU], vx[RU], vy[RU], vr[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int mm = m + u * R;
                if (mm < r1) {
                    vg[u] = *reinterpret_cast<const uint4*>(dy + (size_t)mm * ld_dy + c);
                    if (coef || msc) vx[u] = *reinterpret_cast<const uint4*>(x + (size_t)mm * ld_x + c);
                    if (y) vy[u] = *reinterpret_cast<const uint4*>(y + (size_t)mm * ld_y + c);
                    if (dres && dres_accum) vr[u] = *reinterpret_cast<const uint4*>(dres + (size_t)mm * ld_dres + c);
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int mm = m + u * R;
                if (mm < r1) {
                    float g[V], xv[V], yv[V], rr[V], o[V];
                    TT<T>::unpack(vg[u], g);
                    if (coef || msc) TT<T>::unpack(vx[u], xv);
                    if (y) TT<T>::unpack(vy[u], yv);
                    if (dres && dres_accum) TT<T>::unpack(vr[u], rr);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const bool off = y ? (!(yv[e] > 0.f) || (r6 && yv[e] >= 6.f)) : (msc && !(fmaf(xv[e], ks[e], kh[e]) > 0.f));
                        const float dz = off ? 0.f : g[e];
                        // explicit fused ops: every instantiation (lean / general / table-driven) rounds the same way, bit for bit
                        o[e] = coef ? __fmaf_rn(ka[e], dz, __fmaf_rn(kb[e], xv[e] - kmu[e], kd[e])) : dz;
                        rr[e] = (dres && dres_accum) ? rr[e] + dz : dz;
                    }
                    *reinterpret_cast<uint4*>(dx + (size_t)mm * ld_dx + c) = TT<T>::pack(o);
                    if (dres) *reinterpret_cast<uint4*>(dres + (size_t)mm * ld_dres + c) = TT<T>::pack(rr);
                }
            }
        }
    }
}

template <typename T, bool LEAN>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_k(const T* __restrict__ dy, int ld_dy, const T* __restrict__ y, int ld_y,
                                                           const T* __restrict__ x, int ld_x, int M, int Cp, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ coef, T* __restrict__ dx, int ld_dx,
                                                           T* __restrict__ dres, int ld_dres, int dres_accum, int rows_per_blk, int CVP,
                                                           const float* __restrict__ msc, const float* __restrict__ msh, int r6) {
    bn_bwd_apply_rows_body<T, LEAN>(dy, ld_dy, y, ld_y, x, ld_x, M, Cp, mean, invstd, coef, dx, ld_dx, dres, ld_dres, dres_accum, rows_per_blk, CVP, msc, msh, r6, blockIdx.x);
}
template <typename T, bool LEAN>
__global__ __launch_bounds__(256) void bn_bwd_apply_rows_tab(const pn2_bnapply_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_bnapply_job j = jobs[jb];
    bn_bwd_apply_rows_body<T, LEAN>((const T*)j.dy, j.ld_dy, (const T*)j.y, j.ld_y, (const T*)j.x, j.ld_x, j.M, j.Cp, j.mean, j.invstd, j.coef, (T*)j.dx, j.ld_dx,
                              (T*)j.dres, j.ld_dres, j.dres_accum, j.rows_per_blk, j.cvp, j.msc, j.msh, j.r6, blockIdx.x - bstart[jb]);
}


inline void rows_geometry(int M, int CV, int& cvp, int& rows_per_blk, int& nblk) {
    cvp = 1; while (cvp < CV && cvp < 256) cvp <<= 1;
    const int R = 256 / cvp;
    // aim at ~1024 workgroups (4 per CU).  With scalar parameter loads every block paid a ~60-instruction prologue and 512 was the optimum; with the
    // 16-byte parameter loads the step time is flat from 768 up (15.59 / 15.57 / 15.51 / 15.51 / 15.49 ms at 384 / 512 / 768 / 1024 / 2048)
    constexpr int target = 1024;         // workgroups of a streaming pass (swept 256 .. 2048: flat up to 512 per pass kind)
    int want = (M + target - 1) / target;
    rows_per_blk = ((want + R - 1) / R) * R;
    if (rows_per_blk < R) rows_per_blk = R;
    if (rows_per_blk > R * RU * 4) rows_per_blk = R * RU * 4;
    nblk = (M + rows_per_blk - 1) / rows_per_blk;
}

inline int grid_for(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }
inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

template <typename Ti, typename To>
int affine_dispatch(const void* x, int ld_x, void* y, int ld_y, int M, int C, const float* scale, const float* shift, const void* res, int ld_res, int relu, hipStream_t st) {
    constexpr int V = TT<Ti>::VEC;
    const bool vec = sizeof(Ti) == sizeof(To) && C % V == 0 && ld_x % V == 0 && ld_y % V == 0 && (!res || ld_res % V == 0);
    if constexpr (sizeof(Ti) == sizeof(To)) {
        if (vec) {
            int cvp, rpb, nblk;
            rows_geometry(M, C / V, cvp, rpb, nblk);
            hipLaunchKernelGGL((affine_rows_k<Ti>), dim3(nblk), dim3(256), 0, st, (const Ti*)x, ld_x, (Ti*)y, ld_y, M, C, scale, shift, (const Ti*)res, ld_res, relu, rpb, cvp);
            PN2_CHECK_LAUNCH();
            return 0;
        }
    }
    if (vec) hipLaunchKernelGGL((affine_act_k<Ti, To, V>), dim3(grid_for((size_t)M * (C / V))), dim3(256), 0, st, (const Ti*)x, ld_x, (To*)y, ld_y, M, C, scale, shift, (const Ti*)res, ld_res, relu);
    else hipLaunchKernelGGL((affine_act_k<Ti, To, 1>), dim3(grid_for((size_t)M * C)), dim3(256), 0, st, (const Ti*)x, ld_x, (To*)y, ld_y, M, C, scale, shift, (const Ti*)res, ld_res, relu);
    PN2_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename Tdy>
int bwd_reduce_dispatch(const void* dy, int ld_dy, int Cdy, const void* y, int ld_y, const void* x, int ld_x, int M, int Cp,
                        const float* mean, const float* invstd, float* p1, float* p2, int nblk, const float* msc, const float* msh, int r6, hipStream_t st) {
    constexpr int V = TT<T>::VEC;
    const bool vec = sizeof(T) == sizeof(Tdy) && Cdy == Cp && Cp % V == 0 && ld_dy % V == 0 && ld_x % V == 0 && (!y || ld_y % V == 0);
    const int rows = (M + nblk - 1) / nblk;
    if (vec) {
        int cvp = pow2ceil(Cp / V); if (cvp > 256) cvp = 256;
        if (!y) hipLaunchKernelGGL((bn_bwd_reduce_k<T, Tdy, V, true>), dim3(nblk), dim3(256), 2 * 256 * V * 4, st, (const Tdy*)dy, ld_dy, Cdy, (const T*)y, ld_y, (const T*)x, ld_x, M, Cp, mean, invstd, p1, p2, rows, cvp, msc, msh, r6);
        else hipLaunchKernelGGL((bn_bwd_reduce_k<T, Tdy, V>), dim3(nblk), dim3(256), 2 * 256 * V * 4, st, (const Tdy*)dy, ld_dy, Cdy, (const T*)y, ld_y, (const T*)x, ld_x, M, Cp, mean, invstd, p1, p2, rows, cvp, msc, msh, r6);
    } else {
        int cvp = pow2ceil(Cp); if (cvp > 256) cvp = 256;
        hipLaunchKernelGGL((bn_bwd_reduce_k<T, Tdy, 1>), dim3(nblk), dim3(256), 2 * 256 * 4, st, (const Tdy*)dy, ld_dy, Cdy, (const T*)y, ld_y, (const T*)x, ld_x, M, Cp, mean, invstd, p1, p2, rows, cvp, msc, msh, r6);
    }
    PN2_CHECK_LAUNCH();
    return 0;
}

template <typename T, typename Tdy>
int bwd_apply_dispatch(const void* dy, int ld_dy, int Cdy, const void* y, int ld_y, const void* x, int ld_x, int M, int Cp, const float* mean,
                       const float* invstd, const float* coef, void* dx, int ld_dx, void* dres, int ld_dres, int dres_accum, const float* msc, const float* msh, int r6, hipStream_t st) {
    constexpr int V = TT<T>::VEC;
    const bool vec = sizeof(T) == sizeof(Tdy) && Cdy == Cp && Cp % V == 0 && ld_dy % V == 0 && ld_dx % V == 0 && (!coef || ld_x % V == 0) &&
                     (!y || ld_y % V == 0) && (!dres || ld_dres % V == 0);
    if constexpr (sizeof(T) == sizeof(Tdy)) {
        if (vec) {
            int cvp, rpb, nblk;
            rows_geometry(M, Cp / V, cvp, rpb, nblk);
            if (!y && !dres) hipLaunchKernelGGL((bn_bwd_apply_rows_k<T, true>), dim3(nblk), dim3(256), 0, st, (const T*)dy, ld_dy, (const T*)y, ld_y, (const T*)x, ld_x, M, Cp, mean, invstd, coef, (T*)dx, ld_dx, (T*)dres, ld_dres, dres_accum, rpb, cvp, msc, msh, r6);
            else hipLaunchKernelGGL((bn_bwd_apply_rows_k<T, false>), dim3(nblk), dim3(256), 0, st, (const T*)dy, ld_dy, (const T*)y, ld_y, (const T*)x, ld_x, M, Cp, mean, invstd, coef, (T*)dx, ld_dx, (T*)dres, ld_dres, dres_accum, rpb, cvp, msc, msh, r6);
            PN2_CHECK_LAUNCH();
            return 0;
        }
    }
    if (vec) hipLaunchKernelGGL((bn_bwd_apply_k<T, Tdy, V>), dim3(grid_for((size_t)M * (Cp / V))), dim3(256), 0, st, (const Tdy*)dy, ld_dy, Cdy, (const T*)y, ld_y, (const T*)x, ld_x, M, Cp, mean, invstd, coef, (T*)dx, ld_dx, (T*)dres, ld_dres, dres_accum, r6);
    else hipLaunchKernelGGL((bn_bwd_apply_k<T, Tdy, 1>), dim3(grid_for((size_t)M * Cp)), dim3(256), 0, st, (const Tdy*)dy, ld_dy, Cdy, (const T*)y, ld_y, (const T*)x, ld_x, M, Cp, mean, invstd, coef, (T*)dx, ld_dx, (T*)dres, ld_dres, dres_accum, r6);
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // namespace

extern "C" {

int pn2_bn_finalize(const float* psum, const float* psq, int nblk, const pn2_bn_desc* d, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd, void* stream) {
    if (!psum || !psq || !d || !gamma || !beta || !scale || !shift || !mean || !invstd) return -1;
    const int cpb = finalize_cpb(nblk);
    hipLaunchKernelGGL(bn_finalize_k, dim3((d->Cp + cpb - 1) / cpb), dim3(256), 0, (hipStream_t)stream, psum, psq, nblk, *d, gamma, beta, running_mean, running_var, scale, shift, mean, invstd, cpb);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bn_eval_prepare(const pn2_bn_desc* d, const float* gamma, const float* beta, const float* rm, const float* rv, float* scale, float* shift, void* stream) {
    if (!d || !gamma || !beta || !rm || !rv || !scale || !shift) return -1;
    hipLaunchKernelGGL(bn_eval_prepare_k, dim3((d->Cp + 255) / 256), dim3(256), 0, (hipStream_t)stream, *d, gamma, beta, rm, rv, scale, shift);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bn_eval_prepare_multi(const pn2_bnprep_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    hipLaunchKernelGGL(bn_eval_prepare_tab, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_affine_act(int dt_in, const void* x, int ld_x, int dt_out, void* y, int ld_y, int M, int Cout, const float* scale, const float* shift,
                   const void* res, int ld_res, int relu, void* stream) {
    if (!x || !y) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dt_in == PN2_BF16 && dt_out == PN2_BF16) return affine_dispatch<bf16_t, bf16_t>(x, ld_x, y, ld_y, M, Cout, scale, shift, res, ld_res, relu, st);
    if (dt_in == PN2_BF16 && dt_out == PN2_F32) return affine_dispatch<bf16_t, float>(x, ld_x, y, ld_y, M, Cout, scale, shift, res, ld_res, relu, st);
    if (dt_in == PN2_F32 && dt_out == PN2_F32) return affine_dispatch<float, float>(x, ld_x, y, ld_y, M, Cout, scale, shift, res, ld_res, relu, st);
    return -3;
}

int pn2_bn_bwd_blocks(int M, int Cp, int dt) {
    // >= 8 rows per thread: 256 threads = CVP channel vectors x R row lanes
    const int V = dt == PN2_F32 ? 4 : 8;
    const int cv = Cp % V == 0 ? Cp / V : Cp;
    int cvp = 1; while (cvp < cv && cvp < 256) cvp <<= 1;
    const int rows = (256 / cvp) * 2;     // >= 2 rows per thread, at most 512 partial rows for the finalize pass (measured optimum)
    int b = (M + rows - 1) / rows;
    constexpr int cap = 512;
    return b > cap ? cap : (b < 1 ? 1 : b);
}

int pn2_bn_bwd_reduce(int dt, int dt_dy, const void* dy, int ld_dy, int Cdy, const void* y, int ld_y, int dt_y, const void* x, int ld_x,
                      int M, int Cp, const float* mean, const float* invstd, float* p1, float* p2, int nblk,
                      const float* mask_scale, const float* mask_shift, int relu6, void* stream) {
    if (!dy || !x || !mean || !invstd || !p1 || !p2) return -1;
    if (y && dt_y != dt) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16 && dt_dy == PN2_BF16) return bwd_reduce_dispatch<bf16_t, bf16_t>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, p1, p2, nblk, mask_scale, mask_shift, relu6, st);
    if (dt == PN2_BF16 && dt_dy == PN2_F32) return bwd_reduce_dispatch<bf16_t, float>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, p1, p2, nblk, mask_scale, mask_shift, relu6, st);
    if (dt == PN2_F32 && dt_dy == PN2_F32) return bwd_reduce_dispatch<float, float>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, p1, p2, nblk, mask_scale, mask_shift, relu6, st);
    return -3;
}

int pn2_bn_bwd_finalize(const float* p1, const float* p2, int nblk, const pn2_bn_desc* d, const float* gamma, const float* invstd,
                        float* dgamma, float* dbeta, int accumulate, float* coef, void* stream) {
    if (!p1 || !p2 || !d || !gamma || !invstd || !dgamma || !dbeta || !coef) return -1;
    const int cpb = finalize_cpb(nblk);
    hipLaunchKernelGGL(bn_bwd_finalize_k, dim3((d->Cp + cpb - 1) / cpb), dim3(256), 0, (hipStream_t)stream, p1, p2, nblk, *d, gamma, invstd, dgamma, dbeta, accumulate, coef, cpb);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bn_bwd_finalize_seg(const pn2_bn_segs* segs, const pn2_bn_desc* d, const float* gamma, const float* invstd,
                            float* dgamma, float* dbeta, int accumulate, float* coef, void* stream) {
    if (!segs || !d || !gamma || !invstd || !dgamma || !dbeta || !coef) return -1;
    if (segs->nseg < 1 || segs->nseg > 4 || segs->c0[0] != 0) return -2;
    int nmax = 1;
    for (int k = 0; k < segs->nseg; ++k) {
        if (!segs->p1[k] || !segs->p2[k] || segs->nblk[k] < 1 || segs->ldp[k] < 1) return -1;
        if (k && segs->c0[k] <= segs->c0[k - 1]) return -2;
        if (segs->nblk[k] > nmax) nmax = segs->nblk[k];
    }
    const int cpb = finalize_cpb(nmax);
    hipLaunchKernelGGL(bn_bwd_finalize_seg_k, dim3((d->Cp + cpb - 1) / cpb), dim3(256), 0, (hipStream_t)stream, *segs, *d, gamma, invstd, dgamma, dbeta, accumulate, coef, cpb);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bn_bwd_apply(int dt, int dt_dy, const void* dy, int ld_dy, int Cdy, const void* y, int ld_y, int dt_y, const void* x, int ld_x,
                     int M, int Cp, const float* mean, const float* invstd, const float* coef, void* dx, int ld_dx,
                     void* dres, int ld_dres, int dres_accum, const float* mask_scale, const float* mask_shift, int relu6, void* stream) {
    if (!dy || !dx) return -1;
    if (coef && (!x || !mean || !invstd)) return -1;
    if (y && dt_y != dt) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16 && dt_dy == PN2_BF16) return bwd_apply_dispatch<bf16_t, bf16_t>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, coef, dx, ld_dx, dres, ld_dres, dres_accum, mask_scale, mask_shift, relu6, st);
    if (dt == PN2_BF16 && dt_dy == PN2_F32) return bwd_apply_dispatch<bf16_t, float>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, coef, dx, ld_dx, dres, ld_dres, dres_accum, mask_scale, mask_shift, relu6, st);
    if (dt == PN2_F32 && dt_dy == PN2_F32) return bwd_apply_dispatch<float, float>(dy, ld_dy, Cdy, y, ld_y, x, ld_x, M, Cp, mean, invstd, coef, dx, ld_dx, dres, ld_dres, dres_accum, mask_scale, mask_shift, relu6, st);
    return -3;
}


/* BatchNorm + ReLU + MaxPool(3, 2, 1) forward as one pass (stem of Res2Net_v1b.py:137-139).  Vector rows only (C, ld multiples of the 16-byte vector). */
int pn2_bn_relu_maxpool_fwd(int dt, const void* raw, int ld_raw, const float* scale, const float* shift, void* y, int ld_y, unsigned char* idx,
                            int N, int H, int W, int C, int OH, int OW, void* stream) {
    if (!raw || !scale || !shift || !y || !idx) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V || ld_raw % V || ld_y % V || OH != (H - 1) / 2 + 1 || OW != (W - 1) / 2 + 1) return -2;
    if ((size_t)N * OH * OW * (C / V) >= ((size_t)1 << 32) - ((size_t)8192 << 8)) return -2;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for((size_t)N * OH * OW * (C / V));
    if (dt == PN2_BF16) hipLaunchKernelGGL((bn_relu_maxpool_fwd_k<bf16_t, 8>), dim3(grid), dim3(256), 0, st, (const bf16_t*)raw, ld_raw, scale, shift, (bf16_t*)y, ld_y, idx, N, H, W, C, OH, OW);
    else if (dt == PN2_F32) hipLaunchKernelGGL((bn_relu_maxpool_fwd_k<float, 4>), dim3(grid), dim3(256), 0, st, (const float*)raw, ld_raw, scale, shift, (float*)y, ld_y, idx, N, H, W, C, OH, OW);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

/* Backward of pn2_bn_relu_maxpool_fwd without the full-resolution gradient tensor (even H, W; C / V a power of two <= 256): pn2_bn_bwd_reduce / pn2_bn_bwd_apply with the
 * incoming gradient formed per 2 x 2 input quad from the pooled gradient and the argmax bytes.  p1 / p2: nblk partial rows (any nblk >= 1; pn2_bn_bwd_finalize follows). */
int pn2_pool_bn_bwd_reduce(int dt, const void* dpool, int ld_dp, const unsigned char* idx, const void* raw, int ld_raw, int N, int H, int W, int C, int OH, int OW,
                           const float* mean, const float* invstd, const float* mask_scale, const float* mask_shift, float* p1, float* p2, int nblk, void* stream) {
    if (!dpool || !idx || !raw || !mean || !invstd || !mask_scale || !mask_shift || !p1 || !p2 || nblk < 1) return -1;
    const int V = dt == PN2_F32 ? 4 : 8, CV = C / V;
    if (C % V || ld_dp % V || ld_raw % V || (H & 1) || (W & 1) || OH != H / 2 || OW != W / 2 || CV < 1 || CV > 256 || (CV & (CV - 1))) return -2;
    if ((size_t)N * (H / 2) * (W / 2) >= ((size_t)1 << 24) - 65536 * 32) return -2;          // (quad indices decoded through a float reciprocal)
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL((pool_bn_bwd_reduce_k<bf16_t, 8>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)dpool, ld_dp, idx, (const bf16_t*)raw, ld_raw, N, H, W, OH, OW, C, mean, invstd, mask_scale, mask_shift, p1, p2);
    else if (dt == PN2_F32) hipLaunchKernelGGL((pool_bn_bwd_reduce_k<float, 4>), dim3(nblk), dim3(256), 0, st, (const float*)dpool, ld_dp, idx, (const float*)raw, ld_raw, N, H, W, OH, OW, C, mean, invstd, mask_scale, mask_shift, p1, p2);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}
int pn2_pool_bn_bwd_apply(int dt, const void* dpool, int ld_dp, const unsigned char* idx, const void* raw, int ld_raw, int N, int H, int W, int C, int OH, int OW,
                          const float* mean, const float* invstd, const float* coef, const float* mask_scale, const float* mask_shift, void* dz, int ld_dz, void* stream) {
    if (!dpool || !idx || !raw || !mean || !invstd || !coef || !mask_scale || !mask_shift || !dz) return -1;
    const int V = dt == PN2_F32 ? 4 : 8, CV = C / V;
    if (C % V || ld_dp % V || ld_raw % V || ld_dz % V || (H & 1) || (W & 1) || OH != H / 2 || OW != W / 2 || CV < 1 || CV > 256 || (CV & (CV - 1))) return -2;
    const size_t nq = (size_t)N * (H / 2) * (W / 2);
    if (nq >= ((size_t)1 << 24) - 65536 * 32) return -2;
    const int QL = 256 / CV;
    int grid = (int)((nq + QL - 1) / QL); if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL((pool_bn_bwd_apply_k<bf16_t, 8>), dim3(grid), dim3(256), 0, st, (const bf16_t*)dpool, ld_dp, idx, (const bf16_t*)raw, ld_raw, N, H, W, OH, OW, C, mean, invstd, coef, mask_scale, mask_shift, (bf16_t*)dz, ld_dz);
    else if (dt == PN2_F32) hipLaunchKernelGGL((pool_bn_bwd_apply_k<float, 4>), dim3(grid), dim3(256), 0, st, (const float*)dpool, ld_dp, idx, (const float*)raw, ld_raw, N, H, W, OH, OW, C, mean, invstd, coef, mask_scale, mask_shift, (float*)dz, ld_dz);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

/* pn2_affine_act (same dtype in / out, no residual) with a second output y2 = y + add: the branch sum of Bottle2neck.forward
 * (Res2Net_v1b.py:66-68, sp = sp + spx[i]) produced by the pass that writes sp.  16-byte aligned rows only (-2 otherwise). */
int pn2_affine_act_sum(int dt, const void* x, int ld_x, void* y, int ld_y, int M, int C, const float* scale, const float* shift, int relu,
                       const void* add, int ld_add, void* y2, int ld_y2, void* stream) {
    if (!x || !y || !add || !y2) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V || ld_x % V || ld_y % V || ld_add % V || ld_y2 % V) return -2;
    int cvp, rpb, nblk;
    rows_geometry(M, C / V, cvp, rpb, nblk);
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL((affine_rows_k<bf16_t>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)x, ld_x, (bf16_t*)y, ld_y, M, C, scale, shift, (const bf16_t*)nullptr, 0, relu, rpb, cvp,
                                           (const bf16_t*)add, ld_add, (bf16_t*)y2, ld_y2);
    else if (dt == PN2_F32) hipLaunchKernelGGL((affine_rows_k<float>), dim3(nblk), dim3(256), 0, st, (const float*)x, ld_x, (float*)y, ld_y, M, C, scale, shift, (const float*)nullptr, 0, relu, rpb, cvp,
                                               (const float*)add, ld_add, (float*)y2, ld_y2);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_affine_act_tee(int dt, const void* x, int ld_x, void* y, int ld_y, int M, int C, const float* scale, const float* shift, int relu,
                       void* y3, int ld_y3, int c_lo, void* stream) {
    if (!x || !y || !y3) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V || ld_x % V || ld_y % V || ld_y3 % V || c_lo % V || c_lo < 0 || c_lo >= C) return -2;
    int cvp, rpb, nblk;
    rows_geometry(M, C / V, cvp, rpb, nblk);
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL((affine_rows_k<bf16_t>), dim3(nblk), dim3(256), 0, st, (const bf16_t*)x, ld_x, (bf16_t*)y, ld_y, M, C, scale, shift, (const bf16_t*)nullptr, 0, relu, rpb, cvp,
                                           (const bf16_t*)nullptr, 0, (bf16_t*)nullptr, 0, (bf16_t*)y3, ld_y3, c_lo);
    else if (dt == PN2_F32) hipLaunchKernelGGL((affine_rows_k<float>), dim3(nblk), dim3(256), 0, st, (const float*)x, ld_x, (float*)y, ld_y, M, C, scale, shift, (const float*)nullptr, 0, relu, rpb, cvp,
                                               (const float*)nullptr, 0, (float*)nullptr, 0, (float*)y3, ld_y3, c_lo);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

/* ------------------------------------------------------------------------------------------------ table-driven launches
 * pn2_*_job_blocks fill the derived geometry of a job (host side) and return its workgroup count (< 0: this call cannot be batched - launch it on
 * its own); pn2_*_multi run many jobs of one kind in ONE launch from a DEVICE job table (block_start_dev: njobs + 1 prefix sums).  Same arithmetic,
 * bit for bit, as the single launches. */
int pn2_bn_finalize_job_blocks(pn2_bnfin_job* j) {
    if (!j || !j->psum || !j->psq || !j->gamma || !j->beta || !j->scale || !j->shift || !j->mean || !j->invstd || j->nblk < 1) return -1;
    j->cpb = finalize_cpb(j->nblk);
    return (j->d.Cp + j->cpb - 1) / j->cpb;
}
int pn2_bn_finalize_multi(const pn2_bnfin_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    hipLaunchKernelGGL(bn_finalize_tab, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_affine_job_blocks(int dt, pn2_affine_job* j) {
    if (!j || !j->x || !j->y) return -1;
    if (dt == (PN2_BF16 | PN2_MULTI_F32OUT)) {          // bf16 in, fp32 out, element-wise (pn2_affine_act's own grid)
        if (j->y2 || j->add || j->M < 1 || j->C < 1) return -2;
        return grid_for((size_t)j->M * j->C);
    }
    const int V = dt == PN2_F32 ? 4 : 8;
    if (j->C % V || j->ld_x % V || j->ld_y % V || (j->res && j->ld_res % V) || (j->y2 && (!j->add || j->ld_add % V || j->ld_y2 % V))) return -2;
    int nblk;
    rows_geometry(j->M, j->C / V, j->cvp, j->rows_per_blk, nblk);
    return nblk;
}
int pn2_affine_multi(int dt, const pn2_affine_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    if (dt == (PN2_BF16 | PN2_MULTI_F32OUT)) hipLaunchKernelGGL(affine_act_tab_f32out, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_BF16) hipLaunchKernelGGL((affine_rows_tab<bf16_t>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_F32) hipLaunchKernelGGL((affine_rows_tab<float>), dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bn_bwd_finalize_job_blocks(pn2_bnbfin_job* j) {
    if (!j || !j->gamma || !j->invstd || !j->dgamma || !j->dbeta || !j->coef) return -1;
    if (j->sg.nseg < 1 || j->sg.nseg > 4 || j->sg.c0[0] != 0) return -2;
    int nmax = 1;
    for (int k = 0; k < j->sg.nseg; ++k) {
        if (!j->sg.p1[k] || !j->sg.p2[k] || j->sg.nblk[k] < 1 || j->sg.ldp[k] < 1) return -1;
        if (j->sg.nblk[k] > nmax) nmax = j->sg.nblk[k];
    }
    j->cpb = finalize_cpb(nmax);
    return (j->d.Cp + j->cpb - 1) / j->cpb;
}
int pn2_bn_bwd_finalize_multi(const pn2_bnbfin_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    hipLaunchKernelGGL(bn_bwd_finalize_tab, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bn_bwd_apply_job_blocks(int dt, pn2_bnapply_job* j) {
    if (!j || !j->dy || !j->dx) return -1;
    if (j->coef && (!j->x || !j->mean || !j->invstd)) return -1;
    if (dt == (PN2_BF16 | PN2_MULTI_F32DY)) return (j->M < 1 || j->Cp < 1 || j->pad_ < 1) ? -2 : grid_for((size_t)j->M * j->Cp);
    const int V = dt == PN2_F32 ? 4 : 8;
    if (j->Cp % V || j->ld_dy % V || j->ld_dx % V || (j->coef && j->ld_x % V) || (j->y && j->ld_y % V) || (j->dres && j->ld_dres % V)) return -2;
    int nblk;
    rows_geometry(j->M, j->Cp / V, j->cvp, j->rows_per_blk, nblk);
    return nblk;
}
int pn2_bn_bwd_apply_multi(int dt, const pn2_bnapply_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (dt == (PN2_BF16 | PN2_MULTI_F32DY)) {
        hipLaunchKernelGGL(bn_bwd_apply_tab_f32dy, dim3(total_blocks), dim3(256), 0, st, jobs_dev, block_start_dev, njobs);
        PN2_CHECK_LAUNCH();
        return 0;
    }
    const bool lean = dt & PN2_MULTI_LEAN;
    dt &= ~PN2_MULTI_LEAN;
    if (dt == PN2_BF16 && lean) hipLaunchKernelGGL((bn_bwd_apply_rows_tab<bf16_t, true>), dim3(total_blocks), dim3(256), 0, st, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_BF16) hipLaunchKernelGGL((bn_bwd_apply_rows_tab<bf16_t, false>), dim3(total_blocks), dim3(256), 0, st, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_F32 && lean) hipLaunchKernelGGL((bn_bwd_apply_rows_tab<float, true>), dim3(total_blocks), dim3(256), 0, st, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_F32) hipLaunchKernelGGL((bn_bwd_apply_rows_tab<float, false>), dim3(total_blocks), dim3(256), 0, st, jobs_dev, block_start_dev, njobs);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bn_bwd_reduce_job_blocks(int dt, pn2_bnreduce_job* j) {
    if (!j || !j->dy || !j->x || !j->mean || !j->invstd || !j->p1 || !j->p2 || j->nblk < 1) return -1;
    if (dt == (PN2_BF16 | PN2_MULTI_F32DY)) {          // the scalar form of bwd_reduce_dispatch<bf16_t, float>
        if (j->pad_ < 1) return -2;
        int cvp = pow2ceil(j->Cp); if (cvp > 256) cvp = 256;
        j->cvp = cvp; j->rows_per_blk = (j->M + j->nblk - 1) / j->nblk;
        return j->nblk;
    }
    const int V = dt == PN2_F32 ? 4 : 8;
    if (j->Cp % V || j->ld_dy % V || j->ld_x % V || (j->y && j->ld_y % V)) return -2;
    int cvp = pow2ceil(j->Cp / V); if (cvp > 256) cvp = 256;
    j->cvp = cvp; j->rows_per_blk = (j->M + j->nblk - 1) / j->nblk;
    return j->nblk;
}
int pn2_bn_bwd_reduce_multi(int dt, const pn2_bnreduce_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    if (dt == (PN2_BF16 | PN2_MULTI_F32DY)) hipLaunchKernelGGL(bn_bwd_reduce_tab_f32dy, dim3(total_blocks), dim3(256), 2 * 256 * 4, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_BF16) hipLaunchKernelGGL((bn_bwd_reduce_tab<bf16_t>), dim3(total_blocks), dim3(256), 2 * 256 * 8 * 4, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_F32) hipLaunchKernelGGL((bn_bwd_reduce_tab<float>), dim3(total_blocks), dim3(256), 2 * 256 * 4 * 4, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
