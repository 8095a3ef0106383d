// pn2_tail.hip — the memory-bound tail of the PraNet-V2 step: DSRA fusion, V1 reverse-attention gate,
// the dual structure loss (forward + backward in one pass over all four supervision pairs), the fused
// clamp+Adam update, and the MyTest_med.py eval tail.
//
// Reference code restated:
//   DSRA fusion      /root/reference/binary_seg/lib/pranet.py:365-368,385-389,407-411
//   RA gate (V1)     /root/reference/binary_seg/lib/PraNet_Res2Net.py:153-154,166-167,177-178
//   structure_loss   /root/reference/binary_seg/MyTrain_med.py:19-38 (called 4x at :78-81)
//   clip + Adam      /root/reference/binary_seg/utils/utils.py:7-17 ; MyTrain_med.py:85-86,149
//   eval tail        /root/reference/binary_seg/MyTest_med.py:104-111
#include "pn2_common.h"
#include "../../include/pn2.h"

namespace {

constexpr int MAXK = 32;
inline int grid_for(size_t total, int cap = 16384) { size_t g = (total + 255) / 256; return (int)(g > (size_t)cap ? cap : (g < 1 ? 1 : g)); }

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// ------------------------------------------------------------------------------------------ DSRA
__global__ __launch_bounds__(256) void dsra_fwd_k(const float* __restrict__ fg, const float* __restrict__ cf, const float* __restrict__ cb,
                                                  float* __restrict__ out, int M, int K, int sm) {
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const size_t o = (size_t)m * K;
        if (sm) {
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, cf[o + k] - cb[o + k]);
            float e[MAXK], s = 0.f;
            for (int k = 0; k < K; ++k) { e[k] = expf(cf[o + k] - cb[o + k] - mx); s += e[k]; }
            for (int k = 0; k < K; ++k) { const float f = fg[o + k]; out[o + k] = f + f * (e[k] / s); }
        } else
            for (int k = 0; k < K; ++k) { const float f = fg[o + k]; out[o + k] = f + f * (cf[o + k] - cb[o + k]); }
    }
}

__global__ __launch_bounds__(256) void dsra_bwd_k(const float* __restrict__ fg, const float* __restrict__ cf, const float* __restrict__ cb,
                                                  const float* __restrict__ dout, float* __restrict__ dfg, float* __restrict__ dcf, float* __restrict__ dcb,
                                                  int M, int K, int sm) {
    // no FMA contraction here: every product is rounded, so K == 1 yields d/dcrop == 0 exactly (as torch's softmax backward does)
#pragma clang fp contract(off)
    for (int m = blockIdx.x * 256 + threadIdx.x; m < M; m += gridDim.x * 256) {
        const size_t o = (size_t)m * K;
        if (sm) {
            float mx = -INFINITY;
            for (int k = 0; k < K; ++k) mx = fmaxf(mx, cf[o + k] - cb[o + k]);
            float p[MAXK], s = 0.f;
            for (int k = 0; k < K; ++k) { p[k] = expf(cf[o + k] - cb[o + k] - mx); s += p[k]; }
            float dot = 0.f;
            for (int k = 0; k < K; ++k) { p[k] /= s; dot += dout[o + k] * fg[o + k] * p[k]; }
            for (int k = 0; k < K; ++k) {
                const float g = dout[o + k];
                dfg[o + k] = g + g * p[k];
                const float dd = p[k] * (g * fg[o + k] - dot);
                dcf[o + k] = dd; dcb[o + k] = -dd;
            }
        } else
            for (int k = 0; k < K; ++k) {
                const float g = dout[o + k], d = cf[o + k] - cb[o + k];
                dfg[o + k] = g + g * d;
                dcf[o + k] = g * fg[o + k]; dcb[o + k] = -g * fg[o + k];
            }
    }
}

// ------------------------------------------------------------------------------------------ RA gate (V1)
template <typename T>
__global__ __launch_bounds__(256) void ra_gate_fwd_k(const T* __restrict__ x, int ld_x, const float* __restrict__ crop, T* __restrict__ out, int ld_o, int M, int C) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V;
    const size_t total = (size_t)M * CV;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int c; const int m = (int)divmod_idx(idx, CV, c); c *= V;
        const float gte = 1.f - sigmoidf_(crop[m]);
        float v[V];
        TT<T>::unpack(*reinterpret_cast<const uint4*>(x + (size_t)m * ld_x + c), v);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] *= gte;
        *reinterpret_cast<uint4*>(out + (size_t)m * ld_o + c) = TT<T>::pack(v);
    }
}

// POST: the gate was applied BEHIND the 1x1 conv (pn2_conv_gemm_gated): x is the gated conv output raw = gte * u and dout its gradient dz, so
// dx = gte * dz is the gradient of the un-gated GEMM result u and dcrop = -s * gte * sum(dz * u) = -s * sum(dz * raw): no division by the gate
template <typename T, bool POST>
__global__ __launch_bounds__(256) void ra_gate_bwd_k(const T* __restrict__ x, int ld_x, const float* __restrict__ crop, const T* __restrict__ dout, int ld_do,
                                                     T* __restrict__ dx, int ld_dx, int dx_accum, float* __restrict__ dcrop, int M, int C) {
    // one wave per pixel row: lanes sweep channel vectors, wave-reduce the dot product for dcrop
    constexpr int V = TT<T>::VEC;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int m = blockIdx.x * 4 + wv; m < M; m += gridDim.x * 4) {
        const float s = sigmoidf_(crop[m]), gte = 1.f - s;
        float dot = 0.f;
        for (int c = lane * V; c < C; c += 64 * V) {
            float xv[V], g[V], o[V];
            TT<T>::unpack(*reinterpret_cast<const uint4*>(x + (size_t)m * ld_x + c), xv);
            TT<T>::unpack(*reinterpret_cast<const uint4*>(dout + (size_t)m * ld_do + c), g);
            if (dx_accum) TT<T>::unpack(*reinterpret_cast<const uint4*>(dx + (size_t)m * ld_dx + c), o);
#pragma unroll
            for (int e = 0; e < V; ++e) { dot += g[e] * xv[e]; o[e] = dx_accum ? o[e] + g[e] * gte : g[e] * gte; }
            *reinterpret_cast<uint4*>(dx + (size_t)m * ld_dx + c) = TT<T>::pack(o);
        }
        dot = wave_sum(dot);
        if (lane == 0) dcrop[m] = POST ? -s * dot : -s * gte * dot;
    }
}

// ------------------------------------------------------------------------------------------ structure loss
// weit = 1 + 5*|avgpool31x31(mask, stride 1, pad 15, count_include_pad) - mask|  (MyTrain_med.py:21)
// Separable box filter with sliding-window sums: a (row, 16-column segment) lane sums its first window once and then adds the entering /
// subtracts the leaving sample (2 LDS reads per output instead of ks); same for (column, 8-row segment) lanes on the row sums.  {0,1} masks
// sum exactly.  Odd LDS row strides keep both walks bank-conflict free.  (The 32x32-tile / full-window version took 55 us at 32x352x352.)
constexpr int LTY = 32, LTX = 64, LSX = 16, LSY = 8;
__global__ __launch_bounds__(256) void loss_weights_k(const float* __restrict__ mask, float* __restrict__ weit, int H, int W, int ks, long long* __restrict__ clr = nullptr, int nclr = 0) {
    extern __shared__ float sh[];                  // tile TSY x TSX, then row sums TSY x HSX
    if (clr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)          // pn2_loss_weights_clear: the image-sum accumulators of the one-pass tail, zeroed by the launch in front of it
        for (int i = threadIdx.x; i < nclr; i += 256) clr[i] = 0;
    const int R = ks / 2, TSY = LTY + ks - 1, TSXr = LTX + ks - 1, TSX = TSXr | 1, HSX = LTX + 1;
    float* tile = sh; float* hs = sh + TSY * TSX;
    const int n = blockIdx.z, ty0 = blockIdx.y * LTY, tx0 = blockIdx.x * LTX;
    const float* mk = mask + (size_t)n * H * W;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < TSY; r += 16) {          // 8 loads in flight per lane (TSXr <= 126: two 64-lane column steps cover a row)
        float v[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int yy = ty0 - R + r + 4 * u, xx = tx0 - R + tx + 64 * hh;
                v[u][hh] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? mk[(size_t)yy * W + xx] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
                if (r + 4 * u < TSY && tx + 64 * hh < TSXr) tile[(r + 4 * u) * TSX + tx + 64 * hh] = v[u][hh];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TSY * (LTX / LSX); i += 256) {
        const int seg = i / TSY, r = i - seg * TSY;
        const float* base = tile + r * TSX + seg * LSX;
        float* out = hs + r * HSX + seg * LSX;
        float s = 0.f;
        for (int k = 0; k < ks; ++k) s += base[k];
        out[0] = s;
#pragma unroll
        for (int j = 1; j < LSX; ++j) { s += base[j + ks - 1] - base[j - 1]; out[j] = s; }
    }
    __syncthreads();
    const float inv = 1.f / (float)(ks * ks);
    for (int i = threadIdx.x; i < LTX * (LTY / LSY); i += 256) {
        const int seg = i / LTX, c = i - seg * LTX, r0 = seg * LSY, xx = tx0 + c;
        const float* base = hs + r0 * HSX + c;
        float s = 0.f;
        for (int k = 0; k < ks; ++k) s += base[k * HSX];
#pragma unroll
        for (int j = 0; j < LSY; ++j) {
            if (j) s += base[(j + ks - 1) * HSX] - base[(j - 1) * HSX];
            const int yy = ty0 + r0 + j;
            if (yy < H && xx < W) weit[(size_t)n * H * W + (size_t)yy * W + xx] = 1.f + 5.f * fabsf(s * inv - tile[(r0 + j + R) * TSX + c + R]);
        }
    }
}

__device__ __forceinline__ float bce_logits(float x, float z) { return fmaxf(x, 0.f) - x * z + log1pf(__expf(-fabsf(x))); }

// sigmoid(x) and softplus(x) = log(1 + e^x) from ONE hardware exp + one hardware log (v_exp_f32 / v_log_f32): e = e^-|x| <= 1, so
// nothing overflows; log(1 + e) loses at most one ulp of 1.0 (6e-8 absolute) against log1p, far below the 1e-6 loss tolerance.
// BCE-with-logits(x, z) = softplus(x) - x*z, the same stable form torch uses.
__device__ __forceinline__ void sig_softplus(float x, float& sig, float& sp) {
    const float e = __expf(-fabsf(x)), r = __frcp_rn(1.f + e);
    sig = x >= 0.f ? r : e * r;
    sp = fmaxf(x, 0.f) + __logf(1.f + e);
}

// partial[p][n][blk][5] = { sum w*bce_fg, sum w*bce_bg, sum p*m*w, sum (p+m)*w, sum w }
__global__ __launch_bounds__(256) void loss_fwd_k(const float* __restrict__ preds, long long map_stride, int P, const float* __restrict__ mask,
                                                  const float* __restrict__ weit, float* __restrict__ partial, int N, int HW) {
    __shared__ float sh[5][4];
    const int p = blockIdx.z, n = blockIdx.y, nb = gridDim.x;
    const float* xf = preds + (size_t)p * map_stride + (size_t)n * HW;
    const float* xb = preds + (size_t)(P + p) * map_stride + (size_t)n * HW;
    const float* mk = mask + (size_t)n * HW; const float* wt = weit + (size_t)n * HW;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int chunk = (HW + nb - 1) / nb, i0 = blockIdx.x * chunk;
    int i1 = i0 + chunk; if (i1 > HW) i1 = HW;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) {
        const float m = mk[i], w = wt[i], f = xf[i], b = xb[i];
        const float pr = 1.f / (1.f + expf(-f));
        a[0] += w * bce_logits(f, m); a[1] += w * bce_logits(b, 1.f - m);
        a[2] += pr * m * w; a[3] += (pr + m) * w; a[4] += w;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 5; ++j) { a[j] = wave_sum(a[j]); if (lane == 0) sh[j][wv] = a[j]; }
    __syncthreads();
    if (threadIdx.x < 5) {
        const int j = threadIdx.x;
        partial[(((size_t)p * N + n) * nb + blockIdx.x) * 5 + j] = sh[j][0] + sh[j][1] + sh[j][2] + sh[j][3];
    }
}

__global__ void loss_finalize_k(const float* __restrict__ partial, int P, int N, int nb, float* __restrict__ sums, float* __restrict__ wsum, float* __restrict__ loss) {
    __shared__ float per[64 * 8];                  // per (p,n) loss term
    __shared__ double acc[64 * 8 * 5];             // per (p,n,j) band sums: one thread each (the band loop of one thread per (p,n) took 17 us)
    const int t = threadIdx.x;
    // 4 adjacent lanes share one (p,n,j) band sum (bands b = q, q+4, ...; 8 loads in flight per lane), combined in a fixed order:
    // one lane per sum walked the bands in 26 us of dependent-latency loads
    for (int it = t; it < ((P * N * 5 * 4 + 63) & ~63); it += blockDim.x) {
        const int idx = it >> 2, q = it & 3;
        double a = 0.0;
        if (idx < P * N * 5) {
            const int pn = idx / 5, j = idx - pn * 5;
            const float* src = partial + (size_t)pn * nb * 5 + j;
            int b = q;
            for (; b + 28 < nb; b += 32) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(b + 4 * u) * 5];
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)v[u];
            }
            for (; b < nb; b += 4) a += (double)src[(size_t)b * 5];
        }
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
        if (q == 0 && idx < P * N * 5) acc[idx] = a;
    }
    __syncthreads();
    if (t < P * N) {
        const int p = t / N, n = t % N;
        const double* a = acc + (size_t)t * 5;
        float* s = sums + ((size_t)p * N + n) * 4;
        s[0] = (float)a[0]; s[1] = (float)a[1]; s[2] = (float)a[2]; s[3] = (float)a[3];
        if (p == 0) wsum[n] = (float)a[4];
        const float wbce = (float)(a[0] / a[4]), wbce2 = (float)(a[1] / a[4]);
        const float wiou = 1.f - ((float)a[2] + 1.f) / ((float)a[3] - (float)a[2] + 1.f);
        per[t] = wbce + wiou + 0.8f * wbce2;
    }
    __syncthreads();
    if (t == 0) {
        float tot = 0.f;
        for (int p = 0; p < P; ++p) {
            float s = 0.f;
            for (int n = 0; n < N; ++n) s += per[p * N + n];
            loss[p] = s / (float)N; tot += loss[p];
        }
        loss[P] = tot;
    }
}

__global__ __launch_bounds__(256) void loss_bwd_k(const float* __restrict__ preds, float* __restrict__ dpreds, long long map_stride, int P,
                                                  const float* __restrict__ mask, const float* __restrict__ weit, const float* __restrict__ wsum,
                                                  const float* __restrict__ sums, float gscale, int N, int HW, long long dmap_stride = -1, const float* __restrict__ gdev = nullptr) {
    const int p = blockIdx.z, n = blockIdx.y;
    if (dmap_stride < 0) dmap_stride = map_stride;
    const size_t of = (size_t)p * map_stride + (size_t)n * HW, ob = (size_t)(P + p) * map_stride + (size_t)n * HW;
    const size_t df = (size_t)p * dmap_stride + (size_t)n * HW, db = (size_t)(P + p) * dmap_stride + (size_t)n * HW;
    const float* s = sums + ((size_t)p * N + n) * 4;
    const float Wn = wsum[n], I = s[2], U = s[3], D = U - I + 1.f;
    const float gs = (gdev ? gscale * gdev[p] : gscale) / (float)N, invW = 1.f / Wn, invD2 = 1.f / (D * D);          // gdev: the pair's upstream gradient on the DEVICE (autograd's)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const float m = mask[(size_t)n * HW + i], w = weit[(size_t)n * HW + i];
        const float f = preds[of + i], b = preds[ob + i];
        const float pr = 1.f / (1.f + expf(-f)), pb = 1.f / (1.f + expf(-b));
        const float dwiou = -(m * w * D - (I + 1.f) * (w - m * w)) * invD2;
        dpreds[df + i] = gs * (w * (pr - m) * invW + dwiou * pr * (1.f - pr));
        dpreds[db + i] = gs * 0.8f * w * (pb - (1.f - m)) * invW;
    }
}

// ------------------------------------------------------------------------------------------ fused DSRA tail (K = 1)
constexpr int TRB = 8;          // output rows per forward block
__device__ __forceinline__ int pn2_dsra_tail_blocks_dev(int OH) { return (OH + TRB - 1) / TRB; }

// 4 consecutive output pixels (oy, ox..ox+3) of one lateral map from its low-res rows staged in LDS (rows ylo.. of width mp.w);
// same expression as bilinear_fwd_k
__device__ __forceinline__ void tail_up4(const float* rows, int ylo, const pn2_tail_map& mp, int ac, int oy, int ox, float* z) {
    int y0, y1; float ly0, ly1;
    bl_src(oy, mp.rh, ac, mp.h, y0, y1, ly0, ly1);
    const float* r0 = rows + (y0 - ylo) * mp.w; const float* r1 = rows + (y1 - ylo) * mp.w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int x0, x1; float lx0, lx1;
        bl_src(ox + e, mp.rw, ac, mp.w, x0, x1, lx0, lx1);
        z[e] = ly0 * (lx0 * r0[x0] + lx1 * r0[x1]) + ly1 * (lx0 * r1[x0] + lx1 * r1[x1]);
    }
}

template <int P>
__global__ __launch_bounds__(256) void tail_fwd_k(pn2_tail_desc d, float* __restrict__ lat, const float* __restrict__ mask, const float* __restrict__ weit,
                                                  float* __restrict__ partial) {
    extern __shared__ float lds[];
    __shared__ float red[4 * P + 1][4];
    __shared__ int s_off[2 * P], s_ylo[2 * P];
    const int n = blockIdx.y, band = blockIdx.x, nb = gridDim.x, ac = d.align_corners;
    const int oyA = band * TRB;
    int oyB = oyA + TRB - 1; if (oyB > d.OH - 1) oyB = d.OH - 1;
    int off = 0;
    for (int j = 0; j < 2 * P; ++j) {
        const pn2_tail_map& mp = d.maps[j];
        int ya0, ya1, yb0, yb1; float t0, t1;
        bl_src(oyA, mp.rh, ac, mp.h, ya0, ya1, t0, t1);
        bl_src(oyB, mp.rh, ac, mp.h, yb0, yb1, t0, t1);
        const int nr = yb1 - ya0 + 1;
        const float* src = mp.src + ((size_t)n * mp.h + ya0) * mp.w;
        for (int i = threadIdx.x; i < nr * mp.w; i += 256) lds[off + i] = src[i];
        if (threadIdx.x == 0) { s_off[j] = off; s_ylo[j] = ya0; }
        off += nr * mp.w;
    }
    __syncthreads();
    float acc[P][4], wacc = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p) { acc[p][0] = 0.f; acc[p][1] = 0.f; acc[p][2] = 0.f; acc[p][3] = 0.f; }
    const int OW4 = d.OW >> 2, rows = oyB - oyA + 1;
    const size_t img = (size_t)d.OH * d.OW;
    for (int it = threadIdx.x; it < rows * OW4; it += 256) {
        const int r = it / OW4, ox = (it - r * OW4) * 4, oy = oyA + r;
        const size_t pix = (size_t)n * img + (size_t)oy * d.OW + ox;
        const float4 m4 = *reinterpret_cast<const float4*>(mask + pix), w4 = *reinterpret_cast<const float4*>(weit + pix);
        const float m[4] = {m4.x, m4.y, m4.z, m4.w}, w[4] = {w4.x, w4.y, w4.z, w4.w};
        wacc += (w[0] + w[1]) + (w[2] + w[3]);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            float f[4], b[4];
            tail_up4(lds + s_off[p], s_ylo[p], d.maps[p], ac, oy, ox, f);
            tail_up4(lds + s_off[P + p], s_ylo[P + p], d.maps[P + p], ac, oy, ox, b);
            *reinterpret_cast<float4*>(lat + (size_t)p * d.N * img + pix) = make_float4(f[0], f[1], f[2], f[3]);
            *reinterpret_cast<float4*>(lat + (size_t)(P + p) * d.N * img + pix) = make_float4(b[0], b[1], b[2], b[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pr, spf, pb, spb;
                sig_softplus(f[e], pr, spf); sig_softplus(b[e], pb, spb);
                acc[p][0] += w[e] * (spf - f[e] * m[e]); acc[p][1] += w[e] * (spb - b[e] * (1.f - m[e]));
                acc[p][2] += pr * m[e] * w[e]; acc[p][3] += (pr + m[e]) * w[e];
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float v = wave_sum(acc[p][k]); if (lane == 0) red[p * 4 + k][wv] = v; }
    { const float v = wave_sum(wacc); if (lane == 0) red[4 * P][wv] = v; }
    __syncthreads();
    if (threadIdx.x < 5 * P) {          // partial[p][n][band][5] = {w*bce_fg, w*bce_bg, p*m*w, (p+m)*w, w}
        const int p = threadIdx.x / 5, k = threadIdx.x - p * 5, row = k < 4 ? p * 4 + k : 4 * P;
        partial[(((size_t)p * d.N + n) * nb + band) * 5 + k] = (red[row][0] + red[row][1]) + (red[row][2] + red[row][3]);
    }
}

// backward work list: maps that share a low-res geometry are processed together (<= 4 per block, they share the mask / weit reads)
struct tail_groups { int ng; int start[PN2_TAIL_MAX_MAPS + 1]; int nm[PN2_TAIL_MAX_MAPS]; int idx[PN2_TAIL_MAX_MAPS][4]; };

__global__ __launch_bounds__(256) void tail_bwd_k(pn2_tail_desc d, tail_groups G, const float* __restrict__ mask, const float* __restrict__ weit,
                                                  const float* __restrict__ wsum, const float* __restrict__ sums, float gscale) {
    extern __shared__ float lds[];          // [nm][nr][w] low-res rows, then [nm][OW] column sums
    int g = 0;
    while (g + 1 < G.ng && (int)blockIdx.x >= G.start[g + 1]) ++g;
    const int nm = G.nm[g], ac = d.align_corners, P = d.P;
    const pn2_tail_map& m0 = d.maps[G.idx[g][0]];
    const int h = m0.h, w = m0.w;
    const int local = blockIdx.x - G.start[g], n = local / h, iy = local - n * h;
    int oy0, oy1;
    bl_range(iy, m0.rh, ac, d.OH, oy0, oy1);
    int ya0, ya1, yb0, yb1; float t0, t1;
    bl_src(oy0, m0.rh, ac, h, ya0, ya1, t0, t1);
    bl_src(oy1, m0.rh, ac, h, yb0, yb1, t0, t1);
    const int ylo = ya0, nr = yb1 - ya0 + 1;
    for (int q = 0; q < nm; ++q) {
        const float* src = d.maps[G.idx[g][q]].src + ((size_t)n * h + ylo) * w;
        for (int i = threadIdx.x; i < nr * w; i += 256) lds[q * nr * w + i] = src[i];
    }
    float* col = lds + ((nm * nr * w + 3) & ~3);       // 16-byte aligned for the float4 column sums
    __syncthreads();
    const int OW = d.OW, LV = OW >> 2;
    int lvp = 64; while (lvp < LV) lvp <<= 1;
    const int R = 256 / lvp, jv = threadIdx.x % lvp, rl = threadIdx.x / lvp;
    const size_t img = (size_t)d.OH * OW;
    const float gs = gscale / (float)d.N, invW = 1.f / wsum[n];
    // per-map constants of this image
    float cI[4], cD[4], cinvD2[4]; int isbg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = G.idx[g][q < nm ? q : 0], p = j >= P ? j - P : j;
        const float* s = sums + ((size_t)p * d.N + n) * 4;
        cI[q] = s[2]; cD[q] = s[3] - s[2] + 1.f; cinvD2[q] = 1.f / (cD[q] * cD[q]); isbg[q] = j >= P;
    }
    float a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q][0] = 0.f; a[q][1] = 0.f; a[q][2] = 0.f; a[q][3] = 0.f; }
    if (jv < LV) {
        for (int oy = oy0 + rl; oy <= oy1; oy += R) {
            int y0, y1; float ly0, ly1;
            bl_src(oy, m0.rh, ac, h, y0, y1, ly0, ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            const size_t pix = (size_t)n * img + (size_t)oy * OW + jv * 4;
            const float4 m4 = *reinterpret_cast<const float4*>(mask + pix), w4 = *reinterpret_cast<const float4*>(weit + pix);
            const float m[4] = {m4.x, m4.y, m4.z, m4.w}, wt[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nm) {
                    float z[4];
                    tail_up4(lds + q * nr * w, ylo, d.maps[G.idx[g][q]], ac, oy, jv * 4, z);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float ex = __expf(-fabsf(z[e])), rr = __frcp_rn(1.f + ex), pr = z[e] >= 0.f ? rr : ex * rr;
                        float dl;
                        if (isbg[q]) dl = gs * 0.8f * wt[e] * (pr - (1.f - m[e])) * invW;
                        else {
                            const float dwiou = -(m[e] * wt[e] * cD[q] - (cI[q] + 1.f) * (wt[e] - m[e] * wt[e])) * cinvD2[q];
                            dl = gs * (wt[e] * (pr - m[e]) * invW + dwiou * pr * (1.f - pr));
                        }
                        a[q][e] += wy * dl;
                    }
                }
            }
        }
    }
    for (int r = 0; r < R; ++r) {              // combine the row lanes in a fixed order
        if (rl == r && jv < LV) {
            for (int q = 0; q < nm; ++q) {
                float4* c4 = reinterpret_cast<float4*>(col + q * OW) + jv;
                if (r == 0) *c4 = make_float4(a[q][0], a[q][1], a[q][2], a[q][3]);
                else { float4 o = *c4; o.x += a[q][0]; o.y += a[q][1]; o.z += a[q][2]; o.w += a[q][3]; *c4 = o; }
            }
        }
        __syncthreads();
    }
    for (int o = threadIdx.x; o < nm * w; o += 256) {
        const int q = o / w, ix = o - q * w;
        const pn2_tail_map& mp = d.maps[G.idx[g][q]];
        int ox0, ox1;
        bl_range(ix, mp.rw, ac, OW, ox0, ox1);
        float sacc = 0.f;
        for (int ox = ox0; ox <= ox1; ++ox) {
            int x0, x1; float lx0, lx1;
            bl_src(ox, mp.rw, ac, w, x0, x1, lx0, lx1);
            const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
            sacc += wx * col[q * OW + ox];
        }
        float* dst = mp.dsrc + ((size_t)n * h + iy) * w + ix;
        *dst = mp.accumulate ? *dst + sacc : sacc;
    }
}

// ------------------------------------------------------------------------------------------ fused DSRA tail, band kernels
// Used when every lateral map is magnified by more than TRB-1 (x8 / x16 / x32 in pranet.py:354,371,393,415): a band of TRB output rows then
// touches at most 3 low-res rows of a map.  One block = (geometry group, row band, image).  The band's TRB vertically interpolated low-res
// rows are staged in LDS once ([map][row][w+1], last column duplicated so the right tap is always x0+1); a thread owns 4 output columns, keeps
// their horizontal taps in registers and walks the band's rows: 2 LDS reads + 2 flops per logit instead of the full 4-tap index math.
// Backward: the same walk forms dL/dlogit once per pixel (the row-per-block kernel above visits every pixel twice per geometry), folds the
// vertical adjoint into <= 3 slot accumulators, meets the other row lanes in LDS in a fixed order, applies the horizontal adjoint with the
// tap weight shared by the group's maps, and leaves one partial row set per band; tail_band_fin_k sums the <= 3 bands that touch a low-res
// row, again in a fixed order (deterministic, no atomics).
// single-instruction transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, <= 1 ulp each): __expf / __logf / __frcp_rn expand to 10-instruction
// denormal-safe / correctly rounded sequences, which made the walk ALU-bound.  Arguments here are e^-|z| in (0, 1] and 1 + e in [1, 2].
__device__ __forceinline__ float hw_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504f); }
__device__ __forceinline__ float hw_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718f; }
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

struct tail_band_aux { int poff[PN2_TAIL_MAX_MAPS]; int ptot; int R; int nbn; };
constexpr int TBM = 2;          // maps per band block: 24 slot accumulators in the backward, every block the same weight

template <int CTRL> __device__ __forceinline__ float tdpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row (row_ror:8/4/2/1 fold into v_add_f32_dpp, no LDS permutes); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += tdpp<0x128>(v); v += tdpp<0x124>(v); v += tdpp<0x122>(v); v += tdpp<0x121>(v);
    return v;
}

// XCD-aware work order: workgroup i runs on XCD i % 8, so the groups of one (band, image) take consecutive slots of ONE XCD and share
// its L2 for the mask / weit rows.  Returns false for the padding blocks of the last 8-wide stripe.
__device__ __forceinline__ bool tail_band_ids(int ng, int nbn, int nb, int& g, int& band, int& n) {
    const int lin = blockIdx.x, xcd = lin & 7, k = lin >> 3;
    g = k % ng;
    const int bn = (k / ng) * 8 + xcd;
    if (bn >= nbn) return false;
    n = bn / nb; band = bn - n * nb;
    return true;
}

// stage v[q][r][0..w] = ly0*src[y0][x] + ly1*src[y1][x] for the rows of the band ([TBM][TRB][w+1]); slot weights of row r towards low-res
// rows ylo..ylo+2.  A (map, row) pair takes a power-of-two lane segment >= w+1, so the index math is shifts and masks.
__device__ __forceinline__ void tail_band_stage(const pn2_tail_desc& d, const tail_groups& G, int g, int n, int oyA, int rows, float* v, float* wslot) {
    const pn2_tail_map& m0 = d.maps[G.idx[g][0]];
    const int h = m0.h, w = m0.w, wp = w + 1, ac = d.align_corners, nm = G.nm[g];
    int ylo, t1; float t2, t3;
    bl_src(oyA, m0.rh, ac, h, ylo, t1, t2, t3);
    int sh = 2; while ((1 << sh) < wp && sh < 6) ++sh;
    const int seg = 1 << sh, x0 = threadIdx.x & (seg - 1), item0 = threadIdx.x >> sh, items = blockDim.x >> sh;
    const float* s0 = d.maps[G.idx[g][0]].src + (size_t)n * h * w;
    const float* s1 = d.maps[G.idx[g][1]].src + (size_t)n * h * w;
    for (int it = item0; it < nm * TRB; it += items) {
        const int q = it >> 3, r = it & (TRB - 1);
        if (r >= rows) continue;
        int y0, y1; float l0, l1;
        bl_src(oyA + r, m0.rh, ac, h, y0, y1, l0, l1);
        const float* src = q ? s1 : s0;
        for (int x = x0; x < wp; x += seg) {
            const int xs = x < w ? x : w - 1;
            v[it * wp + x] = l0 * src[y0 * w + xs] + l1 * src[y1 * w + xs];
        }
    }
    if (wslot && (int)threadIdx.x < rows) {
        int y0, y1; float l0, l1;
        bl_src(oyA + threadIdx.x, m0.rh, ac, h, y0, y1, l0, l1);
#pragma unroll
        for (int s = 0; s < 3; ++s) wslot[threadIdx.x * 3 + s] = (y0 - ylo == s ? l0 : 0.f) + (y1 - ylo == s ? l1 : 0.f);
    }
}

__global__ __launch_bounds__(256) void tail_band_fwd_k(pn2_tail_desc d, tail_groups G, tail_band_aux A, float* __restrict__ lat, const float* __restrict__ mask,
                                                       const float* __restrict__ weit, float* __restrict__ partial) {
    extern __shared__ float lds[];
    __shared__ float red[3 * TBM + 2][16];
    const int nb = pn2_dsra_tail_blocks_dev(d.OH), ac = d.align_corners, P = d.P, R = A.R;
    int g, band, n;
    if (!tail_band_ids(G.ng, A.nbn, nb, g, band, n)) return;
    const int nm = G.nm[g];
    const pn2_tail_map& m0 = d.maps[G.idx[g][0]];
    const int w = m0.w, wp = w + 1, OW = d.OW, LV = OW >> 2;
    const int oyA = band * TRB, rows = min(TRB, d.OH - oyA);
    tail_band_stage(d, G, g, n, oyA, rows, lds, nullptr);
    __syncthreads();
    const int rl = threadIdx.x / LV, jv = threadIdx.x - rl * LV;
    float acc[TBM][3], wacc = 0.f, mwacc = 0.f;
#pragma unroll
    for (int q = 0; q < TBM; ++q) { acc[q][0] = 0.f; acc[q][1] = 0.f; acc[q][2] = 0.f; }
    int isbg[TBM]; size_t mapoff[TBM];
    const size_t img = (size_t)d.OH * OW;
#pragma unroll
    for (int q = 0; q < TBM; ++q) { const int j = G.idx[g][q]; isbg[q] = j >= P; mapoff[q] = (size_t)j * d.N * img; }
    if (rl < R) {
        int xo[4]; float lx0[4], lx1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { int x1; bl_src(jv * 4 + e, m0.rw, ac, w, xo[e], x1, lx0[e], lx1[e]); }
        const bool same = xo[0] == xo[1] && xo[0] == xo[2] && xo[0] == xo[3];
#define PN2_TAIL_FWD_ROW(r, m4, w4)                                                                                                                   \
        do {                                                                                                                                          \
            const size_t pix = (size_t)n * img + (size_t)(oyA + (r)) * OW + jv * 4;                                                                   \
            const float wt[4] = {w4.x, w4.y, w4.z, w4.w};                                                                                             \
            const float mw[4] = {m4.x * w4.x, m4.y * w4.y, m4.z * w4.z, m4.w * w4.w};                                                                 \
            wacc += (wt[0] + wt[1]) + (wt[2] + wt[3]); mwacc += (mw[0] + mw[1]) + (mw[2] + mw[3]);                                                    \
_Pragma("unroll")                                                                                                                                     \
            for (int q = 0; q < TBM; ++q) {                                                                                                           \
                if (q < nm) {                                                                                                                         \
                    const float* vr = lds + (q * TRB + (r)) * wp;                                                                                     \
                    float z[4];                                                                                                                       \
                    if (same) {                                                                                                                       \
                        const float ta = vr[xo[0]], tb = vr[xo[0] + 1];                                                                               \
_Pragma("unroll")                                                                                                                                     \
                        for (int e = 0; e < 4; ++e) z[e] = lx0[e] * ta + lx1[e] * tb;                                                                 \
                    } else {                                                                                                                          \
_Pragma("unroll")                                                                                                                                     \
                        for (int e = 0; e < 4; ++e) z[e] = lx0[e] * vr[xo[e]] + lx1[e] * vr[xo[e] + 1];                                               \
                    }                                                                                                                                 \
                    *reinterpret_cast<float4*>(lat + mapoff[q] + pix) = make_float4(z[0], z[1], z[2], z[3]);                                          \
                    if (isbg[q]) {                                                                                                                    \
_Pragma("unroll")                                                                                                                                     \
                        for (int e = 0; e < 4; ++e) {                                                                                                 \
                            const float sp = fmaxf(z[e], 0.f) + hw_log(1.f + hw_exp(-fabsf(z[e])));                                                   \
                            acc[q][0] += wt[e] * sp - z[e] * (wt[e] - mw[e]);                                                                         \
                        }                                                                                                                             \
                    } else {                                                                                                                          \
_Pragma("unroll")                                                                                                                                     \
                        for (int e = 0; e < 4; ++e) {                                                                                                 \
                            const float ex = hw_exp(-fabsf(z[e])), rr = hw_rcp(1.f + ex), pr = z[e] >= 0.f ? rr : ex * rr;                            \
                            const float sp = fmaxf(z[e], 0.f) - hw_log(rr);                                                                           \
                            acc[q][0] += wt[e] * sp - z[e] * mw[e]; acc[q][1] += pr * mw[e]; acc[q][2] += pr * wt[e];                                 \
                        }                                                                                                                             \
                    }                                                                                                                                 \
                }                                                                                                                                     \
            }                                                                                                                                         \
        } while (0)
        if (R * 4 >= TRB) {
            // <= 4 rows per thread: every row's mask / weit vector is requested BEFORE the first one is used - a load -> use walk pays a full
            // memory latency per row (4 per block, on top of the staging round trip)
            float4 mq[4], wq[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = rl + t * R, rc = r < rows ? r : rl;
                const size_t px_ = (size_t)n * img + (size_t)(oyA + rc) * OW + jv * 4;
                mq[t] = *reinterpret_cast<const float4*>(mask + px_); wq[t] = *reinterpret_cast<const float4*>(weit + px_);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r = rl + t * R;
                if (r < rows) { const float4 m4 = mq[t], w4 = wq[t]; PN2_TAIL_FWD_ROW(r, m4, w4); }
            }
        } else {
            for (int r = rl; r < rows; r += R) {
                const size_t px_ = (size_t)n * img + (size_t)(oyA + r) * OW + jv * 4;
                const float4 m4 = *reinterpret_cast<const float4*>(mask + px_), w4 = *reinterpret_cast<const float4*>(weit + px_);
                PN2_TAIL_FWD_ROW(r, m4, w4);
            }
        }
#undef PN2_TAIL_FWD_ROW
    }
    // 16-lane DPP sums, then the <= 16 row partials of the block meet in LDS (fixed order)
    const int rrow = threadIdx.x >> 4, nrow = blockDim.x >> 4;
    const bool rlead = (threadIdx.x & 15) == 0;
#pragma unroll
    for (int q = 0; q < TBM; ++q)
#pragma unroll
        for (int k = 0; k < 3; ++k) { const float s = row16_sum(acc[q][k]); if (rlead) red[q * 3 + k][rrow] = s; }
    { const float s = row16_sum(wacc); if (rlead) red[3 * TBM][rrow] = s; }
    { const float s = row16_sum(mwacc); if (rlead) red[3 * TBM + 1][rrow] = s; }
    __syncthreads();
    if (threadIdx.x < 4 * nm) {          // partial[p][n][band][5] = {w*bce_fg, w*bce_bg, p*m*w, (p+m)*w, w}
        const int q = threadIdx.x >> 2, k = threadIdx.x & 3, j = G.idx[g][q];
        auto tot = [&](int row) { float s = red[row][0]; for (int i = 1; i < nrow; ++i) s += red[row][i]; return s; };
        if (j >= P) { if (k == 0) partial[(((size_t)(j - P) * d.N + n) * nb + band) * 5 + 1] = tot(q * 3); }
        else {
            float* dst = partial + (((size_t)j * d.N + n) * nb + band) * 5;
            if (k == 0) dst[0] = tot(q * 3);
            else if (k == 1) dst[2] = tot(q * 3 + 1);
            else if (k == 2) dst[3] = tot(q * 3 + 2) + tot(3 * TBM + 1);
            else dst[4] = tot(3 * TBM);
        }
    }
}

__global__ __launch_bounds__(256) void tail_band_bwd_k(pn2_tail_desc d, tail_groups G, tail_band_aux A, const float* __restrict__ mask,
                                                       const float* __restrict__ weit, const float* __restrict__ wsum, const float* __restrict__ sums,
                                                       float gscale, float* __restrict__ pbuf) {
    extern __shared__ float lds[];          // v[TBM][TRB][w+1] | wslot[TRB][3] | col[nm][3][OW]
    const int nb = pn2_dsra_tail_blocks_dev(d.OH), ac = d.align_corners, P = d.P, R = A.R;
    int g, band, n;
    if (!tail_band_ids(G.ng, A.nbn, nb, g, band, n)) return;
    const int nm = G.nm[g];
    const pn2_tail_map& m0 = d.maps[G.idx[g][0]];
    const int w = m0.w, wp = w + 1, OW = d.OW, LV = OW >> 2;
    const int oyA = band * TRB, rows = min(TRB, d.OH - oyA);
    float* wslot = lds + ((TBM * TRB * wp + 3) & ~3);
    float* col = wslot + TRB * 3;
    tail_band_stage(d, G, g, n, oyA, rows, lds, wslot);
    __syncthreads();
    const int rl = threadIdx.x / LV, jv = threadIdx.x - rl * LV;
    const size_t img = (size_t)d.OH * OW;
    const float gs = gscale / (float)d.N, gw = gs / wsum[n];
    // per-map constants of this image: fg  dl = gw*(wt*pr - mw) + (cA*(wt-mw) - cB*mw) * pr*(1-pr) ;  bg  dl = 0.8*gw*(wt*pb - (wt-mw))
    float cA[TBM], cB[TBM]; int isbg[TBM];
#pragma unroll
    for (int q = 0; q < TBM; ++q) {
        const int j = G.idx[g][q], p = j >= P ? j - P : j;
        const float* s = sums + ((size_t)p * d.N + n) * 4;
        const float I1 = s[2] + 1.f, D = s[3] - s[2] + 1.f, invD2 = 1.f / (D * D);
        cA[q] = gs * I1 * invD2; cB[q] = gs * D * invD2; isbg[q] = j >= P;
    }
    float a[TBM][3][4];
#pragma unroll
    for (int q = 0; q < TBM; ++q)
#pragma unroll
        for (int s = 0; s < 3; ++s) { a[q][s][0] = 0.f; a[q][s][1] = 0.f; a[q][s][2] = 0.f; a[q][s][3] = 0.f; }
    if (rl < R) {
        int xo[4]; float lx0[4], lx1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { int x1; bl_src(jv * 4 + e, m0.rw, ac, w, xo[e], x1, lx0[e], lx1[e]); }
        const bool same = xo[0] == xo[1] && xo[0] == xo[2] && xo[0] == xo[3];
        for (int r = rl; r < rows; r += R) {
            const size_t pix = (size_t)n * img + (size_t)(oyA + r) * OW + jv * 4;
            const float4 m4 = *reinterpret_cast<const float4*>(mask + pix), w4 = *reinterpret_cast<const float4*>(weit + pix);
            const float wt[4] = {w4.x, w4.y, w4.z, w4.w};
            const float mw[4] = {m4.x * w4.x, m4.y * w4.y, m4.z * w4.z, m4.w * w4.w};
            const float ws0 = wslot[r * 3], ws1 = wslot[r * 3 + 1], ws2 = wslot[r * 3 + 2];
#pragma unroll
            for (int q = 0; q < TBM; ++q) {
                if (q < nm) {
                    const float* vr = lds + (q * TRB + r) * wp;
                    const float ta = vr[xo[0]], tb = vr[xo[0] + 1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float z = same ? lx0[e] * ta + lx1[e] * tb : lx0[e] * vr[xo[e]] + lx1[e] * vr[xo[e] + 1];
                        const float ex = hw_exp(-fabsf(z)), rr = hw_rcp(1.f + ex), pr = z >= 0.f ? rr : ex * rr;
                        const float u = wt[e] - mw[e];
                        float dl;
                        if (isbg[q]) dl = 0.8f * gw * (wt[e] * pr - u);
                        else dl = gw * (wt[e] * pr - mw[e]) + (cA[q] * u - cB[q] * mw[e]) * (pr - pr * pr);
                        a[q][0][e] += ws0 * dl; a[q][1][e] += ws1 * dl; a[q][2][e] += ws2 * dl;
                    }
                }
            }
        }
    }
    for (int r = 0; r < R; ++r) {              // the row lanes meet in a fixed order
        if (rl == r) {
#pragma unroll
            for (int q = 0; q < TBM; ++q) {
                if (q < nm) {
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        float4* c4 = reinterpret_cast<float4*>(col + (q * 3 + s) * OW) + jv;
                        if (r == 0) *c4 = make_float4(a[q][s][0], a[q][s][1], a[q][s][2], a[q][s][3]);
                        else { float4 o = *c4; o.x += a[q][s][0]; o.y += a[q][s][1]; o.z += a[q][s][2]; o.w += a[q][s][3]; *c4 = o; }
                    }
                }
            }
        }
        __syncthreads();
    }
    // horizontal adjoint: PW lanes share one low-res column, the tap weight of an output column is shared by the block's maps and slots
    int PW = 1; while (PW * 2 * w <= (int)blockDim.x && PW < 64) PW <<= 1;
    const int ix = threadIdx.x / PW, part = threadIdx.x - ix * PW;
    float sacc[TBM][3];
#pragma unroll
    for (int q = 0; q < TBM; ++q) { sacc[q][0] = 0.f; sacc[q][1] = 0.f; sacc[q][2] = 0.f; }
    if (ix < w) {
        int ox0, ox1;
        bl_range(ix, m0.rw, ac, OW, ox0, ox1);
        for (int ox = ox0 + part; ox <= ox1; ox += PW) {
            int x0, x1; float l0, l1;
            bl_src(ox, m0.rw, ac, w, x0, x1, l0, l1);
            const float wx = (x0 == ix ? l0 : 0.f) + (x1 == ix ? l1 : 0.f);
#pragma unroll
            for (int q = 0; q < TBM; ++q)
                if (q < nm) { sacc[q][0] += wx * col[(q * 3) * OW + ox]; sacc[q][1] += wx * col[(q * 3 + 1) * OW + ox]; sacc[q][2] += wx * col[(q * 3 + 2) * OW + ox]; }
        }
    }
    for (int o = PW >> 1; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < TBM; ++q) { sacc[q][0] += __shfl_xor(sacc[q][0], o); sacc[q][1] += __shfl_xor(sacc[q][1], o); sacc[q][2] += __shfl_xor(sacc[q][2], o); }
    if (ix < w && part == 0) {
        float* base = pbuf + ((size_t)n * nb + band) * A.ptot;
#pragma unroll
        for (int q = 0; q < TBM; ++q)
            if (q < nm) { float* dst = base + A.poff[G.idx[g][q]] + ix; dst[0] = sacc[q][0]; dst[w] = sacc[q][1]; dst[2 * w] = sacc[q][2]; }
    }
}

// dsrc[j][n][y][x] = sum over the bands whose rows touch low-res row y (ascending band order)
__global__ __launch_bounds__(256) void tail_band_fin_k(pn2_tail_desc d, tail_band_aux A, const float* __restrict__ pbuf, int nb) {
    const int j = blockIdx.y, local = blockIdx.x * 256 + threadIdx.x;        // the map is uniform per block: its descriptor stays in SGPRs
    const pn2_tail_map& mp = d.maps[j];
    const int h = mp.h, w = mp.w, ac = d.align_corners;
    if (local >= d.N * h * w) return;
    const int n = local / (h * w), rem = local - n * h * w, y = rem / w, x = rem - y * w;
    int oy0, oy1;
    bl_range(y, mp.rh, ac, d.OH, oy0, oy1);
    float s = 0.f;
    for (int b = oy0 / TRB; b <= oy1 / TRB; ++b) {
        const int oyA = b * TRB, oyB = min(oyA + TRB - 1, d.OH - 1);
        int ylo, yhi, t; float t0, t1;
        bl_src(oyA, mp.rh, ac, h, ylo, t, t0, t1);
        bl_src(oyB, mp.rh, ac, h, t, yhi, t0, t1);
        if (y >= ylo && y <= yhi) s += pbuf[((size_t)n * nb + b) * A.ptot + A.poff[j] + (y - ylo) * w + x];
    }
    float* dst = mp.dsrc + local;
    *dst = mp.accumulate ? *dst + s : s;
}

// ------------------------------------------------------------------------------------------ fused DSRA tail in ONE pass (round 5)
// PMC on the band kernels above (profiles/r05_tail_*): they issue 25-30 vector instructions per logit, forward and backward each, and both walk every
// pixel - stage the low-res rows, fetch mask / weit, interpolate, take the sigmoid.  Two rewrites that kept the two-pass structure (one block per (band,
// image) over all 2P maps; per-logit arithmetic on packed pairs, ~11 instructions) measured 105-123 us against 108: with the walk halved, what is left is
// the per-block fixed cost and the second pass itself.  This kernel removes the second pass.  The gradient of the structure loss is LINEAR in three
// per-logit quantities whose coefficients are the only thing that needs the image-wide sums:
//     dL/dz_fg = gw*(w*p - m*w) + cA*(w*p(1-p)) - (cA + cB)*(m*w*p(1-p)),     dL/dz_bg = 0.8*gw*(w*p - (w - m*w)),
//     gw = g/(N*sum w), cA = g*(I + 1)/(N*D^2), cB = g/(N*D), D = U - I + 1      (I = sum p*m*w, U = sum (p + m)*w per image; MyTrain_med.py:19-38)
// so the forward walk also pushes those quantities through the bilinear adjoint (vertical: two slot accumulators per thread; horizontal: a thread-local
// dot with its four tap weights, then a short gather over the threads of a low-res column) and leaves band partials; tail_one_fin_k sums the <= 3 bands of
// a low-res row and applies gw / cA / cB, which it forms from the forward's loss partials.  With p = 1/2 + gs, hw = w/2, c1 = w/2 - m*w (per pixel, shared
// by the 2P maps) the walk accumulates  X1 = hw*gs, X2 = hw*gs^2, X3 = c1*gs^2 (fg), Y1 = hw*gs (bg)  and, once per pair geometry, Shw = hw, Sc1 = c1:
//     fg: gw*(2 X1 + Sc1) + cA*(Shw/2 - 2 X2) - (cA + cB)*((Shw - Sc1)/4 - X2 + X3)          bg: 0.8*gw*(2 Y1 - Sc1)
// Served geometry: align_corners = 0 and every map magnified by a power of two >= 8 in both directions (all training scales of MyTrain_med.py:55,70-73:
// 256 / 352 / 448 px -> x8, x16, x32): then a pixel quad has one left tap column, a row lane's rows one upper tap row, and the run of quads behind a
// low-res column is (i*mag + mag/2)/4 exactly.  Everything else takes pn2_dsra_tail_fwd + _bwd.  No atomics, fixed summation orders.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 f2s(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 f2fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
constexpr float T_LOG2E = 1.44269504f, T_2LN2 = 1.38629436f;
// u = 1 + e^-|z|
__device__ __forceinline__ f2 tail_u(f2 z) {
    return f2{__builtin_amdgcn_exp2f(-fabsf(z.x) * T_LOG2E), __builtin_amdgcn_exp2f(-fabsf(z.y) * T_LOG2E)} + f2s(1.f);
}
// |z| + 2 ln2 log2(u)  (= 2 softplus(z) - z):   w*bce(z, m) = hw*t + z*c1,   w*bce(z, 1 - m) = hw*t - z*c1
__device__ __forceinline__ f2 tail_t(f2 z, f2 u) {
    return f2{fmaf(__builtin_amdgcn_logf(u.x), T_2LN2, fabsf(z.x)), fmaf(__builtin_amdgcn_logf(u.y), T_2LN2, fabsf(z.y))};
}
// sigmoid(z) - 1/2 = copysign(1/u - 1/2, z)      (e/(1 + e) = 1 - 1/(1 + e))
__device__ __forceinline__ f2 tail_gs(f2 z, f2 u) {
    const f2 g = f2{__builtin_amdgcn_rcpf(u.x), __builtin_amdgcn_rcpf(u.y)} - f2s(0.5f);
    return f2{copysignf(g.x, z.x), copysignf(g.y, z.y)};
}

constexpr int TNK = 12;          // gradient rows of a pair in cc: (X1, X2, X3, Y1, Shw, Sc1) x 2 slots
struct tail_one_aux {
    int voff[PN2_TAIL_MAX_MAPS];            // float offset of map j's staged source rows ([3][w + 1]) in LDS
    int poff[PN2_TAIL_MAX_MAPS];            // pair p's block of the band partials: [6][3][w] floats
    int mag[PN2_TAIL_MAX_MAPS];             // pair p's magnification
    int shp[PN2_TAIL_MAX_MAPS];             // first pair of pair p's geometry: its Shw / Sc1 rows serve pair p too
    int ptot, vtot, R, nb;
};

// the thread's 4 logits on output row r of one map: z[h] = columns (2h, 2h + 1); raw: the map's staged source rows, rt = {dy0*(w+1), dy1*(w+1), ly0, ly1}
__device__ __forceinline__ void tail_z4(const float* raw, const float4 rt, int x0, const f2* la, const f2* lb, f2* z) {
    const float* r0 = raw + __float_as_int(rt.x) + x0; const float* r1 = raw + __float_as_int(rt.y) + x0;
    const f2 tv = f2fma(f2s(rt.w), f2{r1[0], r1[1]}, f2s(rt.z) * f2{r0[0], r0[1]});          // (v[x0], v[x0 + 1]), v = ly0*s[y0] + ly1*s[y1] as bilinear_fwd_k
    z[0] = f2fma(lb[0], f2s(tv.y), la[0] * f2s(tv.x)); z[1] = f2fma(lb[1], f2s(tv.y), la[1] * f2s(tv.x));
}
// thread-local horizontal adjoint of a slot accumulator (4 columns): (sum lx0*a, sum lx1*a) -> the quad's left / right tap column
__device__ __forceinline__ void tail_cc(float* cc, int k, const f2* acc, const f2* la, const f2* lb) {
    const f2 s0 = f2fma(la[1], acc[1], la[0] * acc[0]), s1 = f2fma(lb[1], acc[1], lb[0] * acc[0]);
    *reinterpret_cast<float2*>(cc + (size_t)k * 2) = make_float2(s0.x + s0.y, s1.x + s1.y);
}

// horizontal gather of one pair (see tail_one_k): NQ = mag/4 quads per low-res column (3 NQ / 2 at column 0, which also owns the clamped left border), every
// LDS read of a (row, row lane) requested before the first add - with run-time trip counts the loads and adds alternated, one LDS latency each
template <int NQ>
__device__ __forceinline__ void tail_gather(const float* cc, const int* sbt_p, int R, int RPT, int LV, int w, int mag, int nrows_, float* pbp) {
    constexpr int NL = (3 * NQ) / 2;
    int sh = 4; while ((1 << sh) < w) ++sh;          // lanes per row: power of two >= w (w <= 63)
    const int ix = threadIdx.x & ((1 << sh) - 1), g0 = threadIdx.x >> sh, ng = blockDim.x >> sh;
    if (ix >= w) return;
    auto J = [&](int i) { return i <= 0 ? 0 : min(LV, (i * mag + (mag >> 1)) >> 2); };
    const int jB = J(ix), jC = J(ix + 1), jA = ix >= 1 ? J(ix - 1) : jB;
    const float rsel = ix == w - 1 ? 1.f : 0.f;          // column w - 1 also takes the right taps of its own quads (their x1 is clamped)
    for (int cS = g0; cS < nrows_; cS += ng) {
        const int c = (cS * 11) >> 5, S = cS - c * 3;
        float acc = 0.f;
        for (int rr = 0; rr < R; ++rr) {
            const int s_ = S - sbt_p[rr * RPT];
            if (s_ == 0 || s_ == 1) {
                const float2* row = reinterpret_cast<const float2*>(cc + ((size_t)(rr * TNK + c * 2 + s_) * LV) * 2);
                float2 eL[NL]; float eR[NL];
#pragma unroll
                for (int k = 0; k < NL; ++k) { const int j = jB + k; eL[k] = row[j < jC ? j : jB]; }
#pragma unroll
                for (int k = 0; k < NL; ++k) { const int j = jA + k; eR[k] = row[j < jB ? j : jA].y; }
#pragma unroll
                for (int k = 0; k < NL; ++k) acc += jB + k < jC ? fmaf(rsel, eL[k].y, eL[k].x) : 0.f;
#pragma unroll
                for (int k = 0; k < NL; ++k) acc += jA + k < jB ? eR[k] : 0.f;
            }
        }
        pbp[cS * w + ix] = acc;
    }
}

// Order-independent image sums: v * 2^30 split into its integer part and 50 fractional bits (|v| < 2^31; everything below 2^-80 is dropped - per VALUE, so the
// result still does not depend on the order), added with 64-bit integer atomics (no return value); tail_isum_get puts the two words together again.
__device__ __forceinline__ void tail_isum_add(long long* a, float v) {
    double s_ = (double)v * 0x1p30;
    s_ = fmin(fmax(s_, -0x1p61), 0x1p61);
    const double f = floor(s_);
    const long long hi = (long long)f, lo = (long long)((s_ - f) * 0x1p50);
    atomicAdd(reinterpret_cast<unsigned long long*>(a), (unsigned long long)hi);
    if (lo) atomicAdd(reinterpret_cast<unsigned long long*>(a) + 1, (unsigned long long)lo);
}
__device__ __forceinline__ double tail_isum_get(const long long* a) { return (double)a[0] * 0x1p-30 + (double)a[1] * 0x1p-80; }

template <int P>
__global__ __launch_bounds__(256) void tail_one_k(pn2_tail_desc d, tail_one_aux A, float* __restrict__ lat, const float* __restrict__ mask,
                                                                                            const float* __restrict__ weit, float* __restrict__ partial, float* __restrict__ pbuf, long long* __restrict__ isum) {
    extern __shared__ float lds[];          // raw[2P][3][w + 1] | cc[R][TNK][LV][2]
    __shared__ __attribute__((aligned(16))) float tab[P * TRB * 4];
    __shared__ float wst[P * TRB * 2];          // slot weights of a row: towards its upper / lower tap row
    __shared__ int sbt[P * TRB];                // upper tap row of a row, relative to the band's first
    __shared__ float red[4 * P + 2][16];
    const int nb = A.nb, R = A.R;
    const int n = blockIdx.x / nb, band = blockIdx.x - n * nb;
    const int OW = d.OW, LV = OW >> 2;
    const int oyA = band * TRB, rows = min(TRB, d.OH - oyA), RPT = TRB / R;          // RPT <= 4 (host)
    const int rl0 = threadIdx.x / LV, jv0 = threadIdx.x - rl0 * LV;
    const bool live = rl0 < R;
    const int rl = live ? rl0 : 0, jv = live ? jv0 : 0;          // padding lanes of the last wave shadow lane 0's pixels and contribute nothing
    const int rA = rl * RPT;                                     // the thread's rows rA .. rA + RPT - 1 (contiguous: one upper tap row)
    const size_t img = (size_t)d.OH * OW, base = (size_t)n * img + (size_t)oyA * OW;
    float* cc = lds + A.vtot;
    // ---- every global load of the block is requested before the first one is used: mask / weit rows, then the low-res source rows
    f2 qh[4][2], qc[4][2];          // hw = w/2, c1 = w/2 - m*w of the thread's pixels (rows x column pairs)
    {
        float4 mq[4], wq[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r = rA + t, rc = (t < RPT && r < rows) ? r : 0;
            const size_t px = base + (size_t)rc * OW + jv * 4;
            mq[t] = *reinterpret_cast<const float4*>(mask + px); wq[t] = *reinterpret_cast<const float4*>(weit + px);
        }
        float a0[2 * P];
#pragma unroll
        for (int j = 0; j < 2 * P; ++j) {
            const pn2_tail_map& mp = d.maps[j];
            const int w = mp.w, wp = w + 1, cnt = 3 * wp, h = mp.h;
            const float* src = mp.src + (size_t)n * h * w;
            int ylo, t1; float t2, t3;
            bl_src(oyA, mp.rh, 0, h, ylo, t1, t2, t3);
            auto fetch = [&](int e) {
                const int rr = (e >= wp ? 1 : 0) + (e >= 2 * wp ? 1 : 0), x = e - rr * wp, xs = x < w ? x : w - 1;
                const int y = ylo + rr < h ? ylo + rr : h - 1;
                return src[y * w + xs];
            };
            a0[j] = (int)threadIdx.x < cnt ? fetch(threadIdx.x) : 0.f;
            if (cnt > (int)blockDim.x) for (int e = threadIdx.x + blockDim.x; e < cnt; e += blockDim.x) lds[A.voff[j] + e] = fetch(e);          // (w > 63 only)
        }
        if ((int)threadIdx.x < P * TRB) {
            const int p = threadIdx.x / TRB, r = threadIdx.x - p * TRB;
            const pn2_tail_map& mp = d.maps[p];
            int y0, y1, ylo, t1; float l0, l1, t2, t3;
            bl_src(oyA, mp.rh, 0, mp.h, ylo, t1, t2, t3);
            bl_src(oyA + (r < rows ? r : 0), mp.rh, 0, mp.h, y0, y1, l0, l1);
            float* t = tab + (p * TRB + r) * 4;
            t[0] = __int_as_float((y0 - ylo) * (mp.w + 1)); t[1] = __int_as_float((y1 - ylo) * (mp.w + 1)); t[2] = l0; t[3] = l1;
            wst[(p * TRB + r) * 2] = y1 == y0 ? l0 + l1 : l0; wst[(p * TRB + r) * 2 + 1] = y1 == y0 ? 0.f : l1;          // (bottom border: both taps are row h - 1)
            sbt[p * TRB + r] = y0 - ylo;
        }
#pragma unroll
        for (int j = 0; j < 2 * P; ++j)
            if ((int)threadIdx.x < 3 * (d.maps[j].w + 1)) lds[A.voff[j] + threadIdx.x] = a0[j];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f2 w0 = f2{wq[t].x, wq[t].y}, w1 = f2{wq[t].z, wq[t].w};
            qh[t][0] = w0 * f2s(0.5f); qh[t][1] = w1 * f2s(0.5f);
            qc[t][0] = f2fma(-f2{mq[t].x, mq[t].y}, w0, qh[t][0]); qc[t][1] = f2fma(-f2{mq[t].z, mq[t].w}, w1, qh[t][1]);
        }
    }
    __syncthreads();
    bool rowok[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) rowok[t] = live && t < RPT && rA + t < rows;
    const int rrow = threadIdx.x >> 4, nrow = blockDim.x >> 4;
    const bool rlead = (threadIdx.x & 15) == 0;
    {       // sum w/2, sum (w/2 - m*w) over the block's pixels
        f2 hs = f2s(0.f), cs = f2s(0.f);
#pragma unroll
        for (int t = 0; t < 4; ++t) if (rowok[t]) { hs += qh[t][0] + qh[t][1]; cs += qc[t][0] + qc[t][1]; }
        const float s0 = row16_sum(hs.x + hs.y), s1 = row16_sum(cs.x + cs.y);
        if (rlead) { red[4 * P][rrow] = s0; red[4 * P + 1][rrow] = s1; }
    }
    float* ccme = cc + ((size_t)rl * TNK * LV + jv) * 2;          // cc[rl][k][jv][2]
    float* pb = pbuf + ((size_t)n * nb + band) * A.ptot;
#pragma unroll 1
    for (int p = 0; p < P; ++p) {
        const pn2_tail_map& mf = d.maps[p];
        const int w = mf.w;
        int x0; f2 la[2], lb[2];
        {
            int xo[4]; float l0[4], l1[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { int x1; bl_src(jv * 4 + e, mf.rw, 0, w, xo[e], x1, l0[e], l1[e]); }
            x0 = xo[0];          // (= xo[1..3]: power-of-two magnification)
            la[0] = f2{l0[0], l0[1]}; la[1] = f2{l0[2], l0[3]}; lb[0] = f2{l1[0], l1[1]}; lb[1] = f2{l1[2], l1[3]};
        }
        const float* vf = lds + A.voff[p]; const float* vb = lds + A.voff[P + p];
        float* latf = lat + (size_t)p * d.N * img + base + jv * 4; float* latb = lat + (size_t)(P + p) * d.N * img + base + jv * 4;
        const float* tp = tab + p * TRB * 4; const float* wp_ = wst + p * TRB * 2;
        if (live && A.shp[p] == p) {       // ---- the geometry's map-independent rows: Shw, Sc1 (once per geometry)
            f2 sh[2][2], sc[2][2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) { sh[s_][0] = f2s(0.f); sh[s_][1] = f2s(0.f); sc[s_][0] = f2s(0.f); sc[s_][1] = f2s(0.f); }
#pragma unroll
            for (int t = 0; t < 4; ++t)
                if (rowok[t]) {
                    const f2 ws0 = f2s(wp_[(rA + t) * 2]), ws1 = f2s(wp_[(rA + t) * 2 + 1]);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        sh[0][h] = f2fma(ws0, qh[t][h], sh[0][h]); sh[1][h] = f2fma(ws1, qh[t][h], sh[1][h]);
                        sc[0][h] = f2fma(ws0, qc[t][h], sc[0][h]); sc[1][h] = f2fma(ws1, qc[t][h], sc[1][h]);
                    }
                }
            tail_cc(ccme, 8 * LV, sh[0], la, lb); tail_cc(ccme, 9 * LV, sh[1], la, lb);
            tail_cc(ccme, 10 * LV, sc[0], la, lb); tail_cc(ccme, 11 * LV, sc[1], la, lb);
        }
        {       // ---- fg map: logits, loss sums, X1 / X2 / X3
            f2 A0 = f2s(0.f), B1 = f2s(0.f), B2 = f2s(0.f);
            f2 x1[2][2], x2[2][2], x3[2][2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int h = 0; h < 2; ++h) { x1[s_][h] = f2s(0.f); x2[s_][h] = f2s(0.f); x3[s_][h] = f2s(0.f); }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (rowok[t]) {
                    const int r = rA + t;
                    const f2 ws0 = f2s(wp_[r * 2]), ws1 = f2s(wp_[r * 2 + 1]);
                    f2 z[2];
                    tail_z4(vf, *reinterpret_cast<const float4*>(tp + r * 4), x0, la, lb, z);
                    *reinterpret_cast<float4*>(latf + (size_t)r * OW) = make_float4(z[0].x, z[0].y, z[1].x, z[1].y);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f2 hw = qh[t][h], c1 = qc[t][h];
                        const f2 u = tail_u(z[h]), gs = tail_gs(z[h], u);
                        A0 = f2fma(z[h], c1, f2fma(hw, tail_t(z[h], u), A0));
                        const f2 a = hw * gs, v = c1 * gs, b = a * gs, g = v * gs;
                        B1 += a; B2 += v;
                        x1[0][h] = f2fma(ws0, a, x1[0][h]); x1[1][h] = f2fma(ws1, a, x1[1][h]);
                        x2[0][h] = f2fma(ws0, b, x2[0][h]); x2[1][h] = f2fma(ws1, b, x2[1][h]);
                        x3[0][h] = f2fma(ws0, g, x3[0][h]); x3[1][h] = f2fma(ws1, g, x3[1][h]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);          // one row body at a time
            }
            if (live) {
                tail_cc(ccme, 0, x1[0], la, lb); tail_cc(ccme, LV, x1[1], la, lb);
                tail_cc(ccme, 2 * LV, x2[0], la, lb); tail_cc(ccme, 3 * LV, x2[1], la, lb);
                tail_cc(ccme, 4 * LV, x3[0], la, lb); tail_cc(ccme, 5 * LV, x3[1], la, lb);
            }
            const float a0 = row16_sum(A0.x + A0.y), b1 = row16_sum(B1.x + B1.y), b2 = row16_sum(B2.x + B2.y);
            if (rlead) { red[p * 4][rrow] = a0; red[p * 4 + 1][rrow] = b1; red[p * 4 + 2][rrow] = b2; }
        }
        {       // ---- bg map: logits, loss sum, Y1
            f2 A3 = f2s(0.f);
            f2 y1[2][2];
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_) { y1[s_][0] = f2s(0.f); y1[s_][1] = f2s(0.f); }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (rowok[t]) {
                    const int r = rA + t;
                    const f2 ws0 = f2s(wp_[r * 2]), ws1 = f2s(wp_[r * 2 + 1]);
                    f2 z[2];
                    tail_z4(vb, *reinterpret_cast<const float4*>(tp + r * 4), x0, la, lb, z);
                    *reinterpret_cast<float4*>(latb + (size_t)r * OW) = make_float4(z[0].x, z[0].y, z[1].x, z[1].y);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f2 hw = qh[t][h];
                        const f2 u = tail_u(z[h]);
                        A3 = f2fma(-z[h], qc[t][h], f2fma(hw, tail_t(z[h], u), A3));
                        const f2 a = hw * tail_gs(z[h], u);
                        y1[0][h] = f2fma(ws0, a, y1[0][h]); y1[1][h] = f2fma(ws1, a, y1[1][h]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (live) { tail_cc(ccme, 6 * LV, y1[0], la, lb); tail_cc(ccme, 7 * LV, y1[1], la, lb); }
            const float a3 = row16_sum(A3.x + A3.y);
            if (rlead) red[p * 4 + 3][rrow] = a3;
        }
        __syncthreads();
        // ---- the quads behind low-res column ix: left taps from quads [J(ix), J(ix+1)), right taps from [J(ix-1), J(ix)) (column w - 1 also takes the right
        //      taps of its own quads: their x1 is clamped), J(i) = (i*mag + mag/2)/4; row lane rl's slot s belongs to the band's low-res row
        //      sbt[first row of rl] + s.  A thread keeps its column and walks rows (comp, band row) of the 18 (12: Shw / Sc1 come from an earlier pair); fixed order.
        {
            const int mag = A.mag[p], nrows_ = A.shp[p] == p ? 18 : 12;
            if (mag == 8) tail_gather<2>(cc, sbt + p * TRB, R, RPT, LV, w, mag, nrows_, pb + A.poff[p]);
            else if (mag == 16) tail_gather<4>(cc, sbt + p * TRB, R, RPT, LV, w, mag, nrows_, pb + A.poff[p]);
            else tail_gather<8>(cc, sbt + p * TRB, R, RPT, LV, w, mag, nrows_, pb + A.poff[p]);
        }
        __syncthreads();          // cc is rewritten by the next pair
    }
    if (threadIdx.x < 5 * P) {          // partial[p][n][band][5] = {w*bce_fg, w*bce_bg, p*m*w, (p+m)*w, w}
        const int p = threadIdx.x / 5, k = threadIdx.x - p * 5;
        auto tot = [&](int row) { float s_ = red[row][0]; for (int i = 1; i < nrow; ++i) s_ += red[row][i]; return s_; };
        const float HW_ = tot(4 * P), C1_ = tot(4 * P + 1), MW_ = HW_ - C1_;          // sum w/2, sum (w/2 - m*w), sum m*w
        float val;
        if (k == 0) val = tot(p * 4);
        else if (k == 1) val = tot(p * 4 + 3);
        else if (k == 2) val = 0.5f * MW_ + (tot(p * 4 + 1) - tot(p * 4 + 2));                 // sum p*m*w = sum (1/2 + gs)*(hw - c1)
        else if (k == 3) val = (HW_ + 2.f * tot(p * 4 + 1)) + MW_;                              // sum p*w + sum m*w
        else val = 2.f * HW_;
        partial[(((size_t)p * d.N + n) * nb + band) * 5 + k] = val;
        // the image's sums, formed on the way (isum[pair][5][2], zero on entry): two-word FIXED-POINT integer atomics (tail_isum_add) - integer addition is
        // associative, so the order in which the bands of an image arrive cannot change a bit of the loss or of the gradient coefficients, whatever the
        // spread of the band sums (round 5 used fp64 atomics, exact only while the sums span < 2^(29 - log2 nb)).  tail_one_fin_k reads five sums per image
        // instead of reducing nb partial rows per workgroup
        if (isum) tail_isum_add(isum + (((size_t)p * d.N + n) * 5 + k) * 2, val);
    }
}

// The loss itself, by ONE extra workgroup of tail_one_fin_k (row 0 of a grid with 2P + 1 rows, dispatched first) instead of a third launch: per[p][n] from the
// image sums isum[pair][5] that tail_one_k accumulated, then loss[p] = mean over the images in the order of loss_total_k.  (Two versions that reduced the forward's
// partial rows inside this workgroup - 110 KB through one CU - took 16-25 us, longer than the rest of the launch.)
constexpr int TLB_MAXPN = 1024;          // P * N the loss block serves (host: more -> loss_total_k)
__device__ __forceinline__ void tail_loss_block(const pn2_tail_desc& d, const long long* __restrict__ isum, float* __restrict__ loss) {
    __shared__ float s_per[TLB_MAXPN], s_lp[8];
    const int P = d.P, N = d.N, PN = P * N;
    for (int pair = threadIdx.x; pair < PN; pair += 256) {
        const long long* a = isum + (size_t)pair * 10;
        const double a0 = tail_isum_get(a), a1 = tail_isum_get(a + 2), a2 = tail_isum_get(a + 4), a3 = tail_isum_get(a + 6), a4 = tail_isum_get(a + 8);
        const float wbce = (float)(a0 / a4), wbce2 = (float)(a1 / a4);
        const float wiou = 1.f - ((float)a2 + 1.f) / ((float)a3 - (float)a2 + 1.f);
        s_per[pair] = wbce + wiou + 0.8f * wbce2;
    }
    __syncthreads();
    if ((int)threadIdx.x < P) {
        const int p = threadIdx.x;
        float s_ = 0.f;
        for (int i = 0; i < N; ++i) s_ += s_per[p * N + i];
        s_lp[p] = s_ / (float)N; loss[p] = s_lp[p];
    }
    __syncthreads();
    if (threadIdx.x == 0) { float tot = 0.f; for (int k = 0; k < P; ++k) tot += s_lp[k]; loss[P] = tot; }
}

// dsrc[j][n][y][x] from the band partials of tail_one_k: sum over the <= 3 bands that touch low-res row y (ascending), combined with gw / cA / cB of the
// pixel's image, which the block forms from the forward's loss partials (16 lanes per sum, double, fixed order - what loss_finalize_k does in one workgroup
// for the whole batch).  The block of fg map p that holds an image's first pixel also leaves sums / wsum / the image's loss term per[p][n].
constexpr int TFI = 64;         // images a 256-pixel block can span (h*w >= 5: host)
__global__ __launch_bounds__(256) void tail_one_fin_k(pn2_tail_desc d, tail_one_aux A, const float* __restrict__ pbuf, const float* __restrict__ partial,
                                                      float gscale, float* __restrict__ sums_out, float* __restrict__ wsum_out, float* __restrict__ per, float* __restrict__ loss,
                                                      const long long* __restrict__ isum) {
    const int lossrow = (int)gridDim.y == 2 * d.P + 1 ? 1 : 0;          // the host asked for the loss (isum given): row 0 of the grid (dispatched first), one workgroup
    if (lossrow && blockIdx.y == 0) {
        if (blockIdx.x == 0) tail_loss_block(d, isum, loss);
        return;
    }
    __shared__ double s_d[TFI * 5];
    const int j = (int)blockIdx.y - lossrow, P = d.P, p = j >= P ? j - P : j, nb = A.nb;
    const pn2_tail_map& mp = d.maps[j];
    const int h = mp.h, w = mp.w, hw_ = h * w, total = d.N * hw_;
    const int first = blockIdx.x * 256;
    if (first >= total) return;
    const int last = min(first + 255, total - 1), n0 = first / hw_, n1 = last / hw_, ni = n1 - n0 + 1;
    // the pixel's band partials first: their loads are in flight while the image sums below are formed (one memory latency instead of two in a row)
    const int local = min(first + (int)threadIdx.x, total - 1);
    const int n = local / hw_, rem = local - n * hw_, y = rem / w, x = rem - y * w;
    float t1 = 0.f, t2 = 0.f, t3 = 0.f, th = 0.f, tc = 0.f;
    {
        int oy0, oy1;
        bl_range(y, mp.rh, 0, d.OH, oy0, oy1);
        for (int b = oy0 / TRB; b <= oy1 / TRB; ++b) {
            const int oyA = b * TRB, oyB = min(oyA + TRB - 1, d.OH - 1);
            int ylo, yhi, t; float u0, u1;
            bl_src(oyA, mp.rh, 0, h, ylo, t, u0, u1);
            bl_src(oyB, mp.rh, 0, h, t, yhi, u0, u1);
            if (y >= ylo && y <= yhi) {
                const float* bq = pbuf + ((size_t)n * nb + b) * A.ptot + (y - ylo) * w + x;
                const float* q = bq + A.poff[p]; const float* qs = bq + A.poff[A.shp[p]];          // [comp][3][w]; Shw / Sc1 of the geometry's first pair
                const int cs = 3 * w;
                if (j < P) { t1 += q[0]; t2 += q[cs]; t3 += q[2 * cs]; th += qs[4 * cs]; } else t1 += q[3 * cs];
                tc += qs[5 * cs];
            }
        }
    }
    if (isum) {          // the image sums as tail_one_k accumulated them
        for (int c = threadIdx.x; c < 5 * ni; c += 256) s_d[c] = tail_isum_get(isum + (((size_t)p * d.N + n0) * 5 + c) * 2);
    } else {
        const int grp = threadIdx.x >> 4, l = threadIdx.x & 15;
        for (int c = grp; c < 5 * ni; c += 16) {
            const int i = c / 5, k = c - i * 5;
            const float* src = partial + ((size_t)p * d.N + (n0 + i)) * nb * 5 + k;
            double a = 0.0;
            for (int b0 = l; b0 < nb; b0 += 128) {
                float vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int b = b0 + 16 * u; vv[u] = b < nb ? src[(size_t)b * 5] : 0.f; }
#pragma unroll
                for (int u = 0; u < 8; ++u) a += (double)vv[u];
            }
            a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
            if (l == 0) s_d[c] = a;
        }
    }
    __syncthreads();
    if (j < P && (int)threadIdx.x < ni) {          // the image's first pixel lies in this block: its sums, wsum and loss term (as loss_finalize_k)
        const int ni_ = n0 + threadIdx.x;
        if (ni_ * hw_ >= first) {
            const double* a = s_d + threadIdx.x * 5;
            float* so = sums_out + ((size_t)p * d.N + ni_) * 4;
            so[0] = (float)a[0]; so[1] = (float)a[1]; so[2] = (float)a[2]; so[3] = (float)a[3];
            if (p == 0) wsum_out[ni_] = (float)a[4];
            const float wbce = (float)(a[0] / a[4]), wbce2 = (float)(a[1] / a[4]);
            const float wiou = 1.f - ((float)a[2] + 1.f) / ((float)a[3] - (float)a[2] + 1.f);
            per[(size_t)p * d.N + ni_] = wbce + wiou + 0.8f * wbce2;
        }
    }
    if (first + (int)threadIdx.x >= total) return;
    const double* a = s_d + (n - n0) * 5;
    const float gs_ = gscale / (float)d.N, I = (float)a[2], U = (float)a[3], D = U - I + 1.f;
    const float gw = gs_ / (float)a[4], cA = gs_ * (I + 1.f) / (D * D), cB = gs_ / D;
    const float g = j < P ? gw * (2.f * t1 + tc) + cA * (0.5f * th - 2.f * t2) - (cA + cB) * (0.25f * (th - tc) - t2 + t3)
                          : 0.8f * gw * (2.f * t1 - tc);
    float* dst = mp.dsrc + local;
    *dst = mp.accumulate ? *dst + g : g;
}

// loss[p] = mean over the images of per[p][n], loss[P] = their sum (the order of loss_finalize_k)
__global__ void loss_total_k(const float* __restrict__ per, int P, int N, float* __restrict__ loss) {
    __shared__ float lp[8];
    const int p = threadIdx.x;
    if (p < P) {
        float s = 0.f;
        int i = 0;
        for (; i + 8 <= N; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = per[(size_t)p * N + i + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; i < N; ++i) s += per[(size_t)p * N + i];
        lp[p] = s / (float)N; loss[p] = lp[p];
    }
    __syncthreads();
    if (p == 0) { float tot = 0.f; for (int k = 0; k < P; ++k) tot += lp[k]; loss[P] = tot; }
}

// ------------------------------------------------------------------------------------------ clamp + Adam
__global__ void adam_tick_k(float* bc, float b1, float b2) {
    // bc = {1-b1^t, 1-b2^t, b1^t, b2^t}; host initialises {0,0,1,1}
    bc[2] *= b1; bc[3] *= b2; bc[0] = 1.f - bc[2]; bc[1] = 1.f - bc[3];
}

__global__ __launch_bounds__(256) void clamp_adam_k(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                                                    float lr, float b1, float b2, float eps, float clip, float gsc, const float* __restrict__ bc, float wd) {
    // bc[7] != 0: lr / clip / weight decay live on the DEVICE (bc[4..6]) so that a captured hipGraph follows adjust_lr() without a re-capture
    if (bc[7] != 0.f) { lr = bc[4]; clip = bc[5]; wd = bc[6]; }
    const float step = lr / bc[0], rs2 = 1.f / sqrtf(bc[1]), keep = 1.f - lr * wd;       // wd: decoupled weight decay (torch.optim.AdamW), 0 = Adam
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 P = reinterpret_cast<float4*>(p)[i], G = reinterpret_cast<float4*>(g)[i], Mv = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
        float* pp = &P.x; float* gg = &G.x; float* mm = &Mv.x; float* vv = &V.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ge = fminf(fmaxf(gg[e] * gsc, -clip), clip);
            gg[e] = ge;
            mm[e] = b1 * mm[e] + (1.f - b1) * ge;
            vv[e] = b2 * vv[e] + (1.f - b2) * ge * ge;
            pp[e] = pp[e] * keep - step * mm[e] / (sqrtf(vv[e]) * rs2 + eps);
        }
        reinterpret_cast<float4*>(p)[i] = P; reinterpret_cast<float4*>(g)[i] = G; reinterpret_cast<float4*>(m)[i] = Mv; reinterpret_cast<float4*>(v)[i] = V;
    }
    for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float ge = fminf(fmaxf(g[i] * gsc, -clip), clip);
        g[i] = ge;
        m[i] = b1 * m[i] + (1.f - b1) * ge;
        v[i] = b2 * v[i] + (1.f - b2) * ge * ge;
        p[i] = p[i] * keep - step * m[i] / (sqrtf(v[i]) * rs2 + eps);
    }
}

// ------------------------------------------------------------------------------------------ eval metrics
// Integer histograms of the uint8 prediction map (all pixels / pixels where gt > 0.5).  Every quantity of the reference's 256-threshold
// sweep (eval.py:22-50, Fmeasure_calu eval_functions.py:131-166: NumRec, NumAnd, num_obj -> precision, recall, specificity, Dice, F, IoU)
// and the MAE is a function of these 512 counts, so the host finishes in float64 with the reference's own expressions (bit-identical).
__global__ __launch_bounds__(256) void eval_hist_k(const unsigned char* __restrict__ pred, const float* __restrict__ gt, long long n, unsigned* __restrict__ hist) {
    __shared__ unsigned h[512];
    h[threadIdx.x] = 0; h[threadIdx.x + 256] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const unsigned v = pred[i];
        atomicAdd(&h[v], 1u);
        if (gt[i] > 0.5f) atomicAdd(&h[256 + v], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
    if (h[threadIdx.x + 256]) atomicAdd(&hist[threadIdx.x + 256], h[threadIdx.x + 256]);
}

// ------------------------------------------------------------------------------------------ eval tail
__global__ __launch_bounds__(256) void minmax_k(const float* __restrict__ x, long long n, float* __restrict__ part) {
    __shared__ float smn[4], smx[4];
    float mn = INFINITY, mx = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float s = 1.f / (1.f + expf(-x[i])); mn = fminf(mn, s); mx = fmaxf(mx, s);
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 + blockIdx.x * 2] = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        part[3 + blockIdx.x * 2] = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    }
}
__global__ void minmax_final_k(float* part, int nb) {
    if (threadIdx.x == 0) {
        float mn = INFINITY, mx = -INFINITY;
        for (int b = 0; b < nb; ++b) { mn = fminf(mn, part[2 + 2 * b]); mx = fmaxf(mx, part[3 + 2 * b]); }
        part[0] = mn; part[1] = mx;
    }
}
__global__ __launch_bounds__(256) void eval_u8_k(const float* __restrict__ x, unsigned char* __restrict__ out, const float* __restrict__ mm, long long n) {
    const float mn = mm[0], den = mm[1] - mm[0] + 1e-8f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float s = 1.f / (1.f + expf(-x[i]));
        out[i] = (unsigned char)(((s - mn) / den) * 255.f);
    }
}

}  // namespace

extern "C" {

int pn2_dsra_fuse_fwd(const float* fg, const float* cf, const float* cb, float* out, int M, int K, int sm, void* stream) {
    if (!fg || !cf || !cb || !out) return -1;
    if (K > MAXK) return -2;
    hipLaunchKernelGGL(dsra_fwd_k, dim3(grid_for(M)), dim3(256), 0, (hipStream_t)stream, fg, cf, cb, out, M, K, sm);
    PN2_CHECK_LAUNCH();
    return 0;
}
int pn2_dsra_fuse_bwd(const float* fg, const float* cf, const float* cb, const float* dout, float* dfg, float* dcf, float* dcb, int M, int K, int sm, void* stream) {
    if (!fg || !cf || !cb || !dout || !dfg || !dcf || !dcb) return -1;
    if (K > MAXK) return -2;
    hipLaunchKernelGGL(dsra_bwd_k, dim3(grid_for(M)), dim3(256), 0, (hipStream_t)stream, fg, cf, cb, dout, dfg, dcf, dcb, M, K, sm);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_ra_gate_fwd(int dt, const void* x, int ld_x, const float* crop, void* out, int ld_out, int M, int C, void* stream) {
    if (!x || !crop || !out) return -1;
    if (C % 8 || ld_x % 8 || ld_out % 8) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL(ra_gate_fwd_k<bf16_t>, dim3(grid_for((size_t)M * C / 8)), dim3(256), 0, st, (const bf16_t*)x, ld_x, crop, (bf16_t*)out, ld_out, M, C);
    else if (dt == PN2_F32) hipLaunchKernelGGL(ra_gate_fwd_k<float>, dim3(grid_for((size_t)M * C / 4)), dim3(256), 0, st, (const float*)x, ld_x, crop, (float*)out, ld_out, M, C);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}
int pn2_ra_gate_bwd(int dt, const void* x, int ld_x, const float* crop, const void* dout, int ld_dout, void* dx, int ld_dx, int dx_accum, float* dcrop, int M, int C, void* stream) {
    if (!x || !crop || !dout || !dx || !dcrop) return -1;
    if (C % 8 || ld_x % 8 || ld_dout % 8 || ld_dx % 8) return -2;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (M + 3) / 4 > 8192 ? 8192 : (M + 3) / 4;
    if (dt == PN2_BF16) hipLaunchKernelGGL((ra_gate_bwd_k<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)x, ld_x, crop, (const bf16_t*)dout, ld_dout, (bf16_t*)dx, ld_dx, dx_accum, dcrop, M, C);
    else if (dt == PN2_F32) hipLaunchKernelGGL((ra_gate_bwd_k<float, false>), dim3(grid), dim3(256), 0, st, (const float*)x, ld_x, crop, (const float*)dout, ld_dout, (float*)dx, ld_dx, dx_accum, dcrop, M, C);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}
int pn2_ra_gate_post_bwd(int dt, const void* raw, int ld_raw, const float* crop, const void* dz, int ld_dz, void* dzg, int ld_dzg, float* dcrop, int M, int C, void* stream) {
    if (!raw || !crop || !dz || !dzg || !dcrop) return -1;
    if (C % 8 || ld_raw % 8 || ld_dz % 8 || ld_dzg % 8) return -2;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (M + 3) / 4 > 8192 ? 8192 : (M + 3) / 4;
    if (dt == PN2_BF16) hipLaunchKernelGGL((ra_gate_bwd_k<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)raw, ld_raw, crop, (const bf16_t*)dz, ld_dz, (bf16_t*)dzg, ld_dzg, 0, dcrop, M, C);
    else if (dt == PN2_F32) hipLaunchKernelGGL((ra_gate_bwd_k<float, true>), dim3(grid), dim3(256), 0, st, (const float*)raw, ld_raw, crop, (const float*)dz, ld_dz, (float*)dzg, ld_dzg, 0, dcrop, M, C);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_loss_weights(const float* mask, float* weit, int N, int H, int W, int ksize, void* stream) {
    if (!mask || !weit) return -1;
    if (ksize < 1 || !(ksize & 1) || ksize > 63) return -2;
    const int TSY = LTY + ksize - 1, TSX = (LTX + ksize - 1) | 1;
    const size_t lds = (size_t)(TSY * TSX + TSY * (LTX + 1)) * 4;
    hipLaunchKernelGGL(loss_weights_k, dim3((W + LTX - 1) / LTX, (H + LTY - 1) / LTY, N), dim3(256), lds, (hipStream_t)stream, mask, weit, H, W, ksize);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_loss_weights_clear(const float* mask, float* weit, int N, int H, int W, int ksize, long long* clear, int nclear, void* stream) {
    if (!mask || !weit || (nclear > 0 && !clear) || nclear < 0) return -1;
    if (ksize < 1 || !(ksize & 1) || ksize > 63) return -2;
    const int TSY = LTY + ksize - 1, TSX = (LTX + ksize - 1) | 1;
    const size_t lds = (size_t)(TSY * TSX + TSY * (LTX + 1)) * 4;
    hipLaunchKernelGGL(loss_weights_k, dim3((W + LTX - 1) / LTX, (H + LTY - 1) / LTY, N), dim3(256), lds, (hipStream_t)stream, mask, weit, H, W, ksize, clear, nclear);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_loss_blocks(int HW) { int b = (HW + 4095) / 4096; return b > 64 ? 64 : (b < 1 ? 1 : b); }

int pn2_structure_loss_fwd(const float* preds, long long map_stride, int P, const float* mask, const float* weit, float* partial,
                           float* sums, float* wsum, float* loss, int N, int HW, void* stream) {
    if (!preds || !mask || !weit || !partial || !sums || !wsum || !loss) return -1;
    if (P * N > 512 || P > 8) return -2;
    const int nb = pn2_loss_blocks(HW);
    hipLaunchKernelGGL(loss_fwd_k, dim3(nb, N, P), dim3(256), 0, (hipStream_t)stream, preds, map_stride, P, mask, weit, partial, N, HW);
    hipLaunchKernelGGL(loss_finalize_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, partial, P, N, nb, sums, wsum, loss);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_structure_loss_bwd(const float* preds, float* dpreds, long long map_stride, int P, const float* mask, const float* weit, const float* wsum,
                           const float* sums, float gscale, int N, int HW, void* stream) {
    if (!preds || !dpreds || !mask || !weit || !wsum || !sums) return -1;
    int nb = (HW + 1023) / 1024; if (nb > 256) nb = 256;
    hipLaunchKernelGGL(loss_bwd_k, dim3(nb, N, P), dim3(256), 0, (hipStream_t)stream, preds, dpreds, map_stride, P, mask, weit, wsum, sums, gscale, N, HW);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_structure_loss_bwd_dev(const float* preds, float* dpreds, long long map_stride, long long dmap_stride, int P, const float* mask, const float* weit, const float* wsum,
                               const float* sums, const float* gscale_dev, float gscale, int N, int HW, void* stream) {
    if (!preds || !dpreds || !mask || !weit || !wsum || !sums) return -1;
    if (map_stride < 0 || dmap_stride < 0) return -2;
    int nb = (HW + 1023) / 1024; if (nb > 256) nb = 256;
    hipLaunchKernelGGL(loss_bwd_k, dim3(nb, N, P), dim3(256), 0, (hipStream_t)stream, preds, dpreds, map_stride, P, mask, weit, wsum, sums, gscale, N, HW, dmap_stride, gscale_dev);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_adam_tick(float* bc, float beta1, float beta2, void* stream) {
    if (!bc) return -1;
    hipLaunchKernelGGL(adam_tick_k, dim3(1), dim3(1), 0, (hipStream_t)stream, bc, beta1, beta2);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_clamp_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                   float eps, float clip, float grad_scale, const float* bias_corr, float weight_decay, void* stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || !bias_corr) return -1;
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return -2;
    hipLaunchKernelGGL(clamp_adam_k, dim3(grid_for((size_t)(n / 4 + 1), 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, clip, grad_scale, bias_corr, weight_decay);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_eval_tail(const float* res, unsigned char* out, float* minmax, long long n, void* stream) {
    if (!res || !out || !minmax) return -1;
    const int nb = grid_for((size_t)n, 512);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(minmax_k, dim3(nb), dim3(256), 0, st, res, n, minmax);
    hipLaunchKernelGGL(minmax_final_k, dim3(1), dim3(64), 0, st, minmax, nb);
    hipLaunchKernelGGL(eval_u8_k, dim3(grid_for((size_t)n)), dim3(256), 0, st, res, out, minmax, n);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_dsra_tail_blocks(int OH) { return (OH + TRB - 1) / TRB; }

static int tail_check(const pn2_tail_desc* d) {
    if (!d || d->P < 1 || d->P > 4 || (d->OW & 3) || d->OW > 1024 || d->N < 1 || d->P * d->N > 512) return -2;
    for (int j = 0; j < 2 * d->P; ++j) {
        const pn2_tail_map& m = d->maps[j];
        if (!m.src || m.h < 1 || m.w < 1 || m.w > d->OW || m.h > d->OH) return -2;
    }
    return 0;
}

// geometry groups (<= 4 maps of equal low-res size and scale); returns the LDS bytes the row-per-block backward needs
static size_t tail_make_groups(const pn2_tail_desc* d, tail_groups& G, int cap = 4) {
    G.ng = 0; G.start[0] = 0;
    bool used[PN2_TAIL_MAX_MAPS] = {false};
    size_t lds = 0;
    for (int j = 0; j < 2 * d->P; ++j) {
        if (used[j]) continue;
        const pn2_tail_map& a = d->maps[j];
        int g = G.ng++;
        G.nm[g] = 0;
        for (int k = j; k < 2 * d->P && G.nm[g] < cap; ++k) {
            const pn2_tail_map& b = d->maps[k];
            if (!used[k] && b.h == a.h && b.w == a.w && b.rh == a.rh && b.rw == a.rw) { used[k] = true; G.idx[g][G.nm[g]++] = k; }
        }
        for (int q = G.nm[g]; q < 4; ++q) G.idx[g][q] = G.idx[g][0];
        G.start[g + 1] = G.start[g] + d->N * a.h;
        // low-res rows a window can touch: window ~ 2/rh + 4 output rows -> (window * rh) + 3 source rows
        const int win = (int)(2.f / a.rh) + 6, nr = (int)(win * a.rh) + 4;
        const size_t need = ((size_t)G.nm[g] * nr * a.w + 4 + (size_t)G.nm[g] * d->OW) * 4;
        if (need > lds) lds = need;
    }
    return lds;
}

// 0 = row kernels, 1 = band kernels (one block per geometry group of <= 2 maps), 2 = all maps per block (default where the geometry allows)
static int tail_mode() {
    const char* e = getenv("PN2_TAIL_BAND");
    return (e && (e[0] == '0' || e[0] == '1')) ? e[0] - '0' : 2;
}

// band kernels apply when a band of TRB output rows touches <= 3 low-res rows of every map; PN2_TAIL_BAND=0 keeps the row kernels
static bool tail_band_ok(const pn2_tail_desc* d) {
    if (tail_mode() == 0 || d->N > 65535) return false;
    for (int j = 0; j < 2 * d->P; ++j) {
        const pn2_tail_map& m = d->maps[j];
        if (m.rh * (float)(TRB - 1) >= 0.999f || m.rw * 4.f >= 0.999f) return false;
    }
    return true;
}

static void tail_band_geometry(const pn2_tail_desc* d, const tail_groups& G, tail_band_aux& A, int& threads, int& blocks, size_t& lds_f, size_t& lds_b) {
    const int LV = d->OW >> 2;
    int R = 1; while (R * 2 * LV <= 256 && R * 2 <= TRB) R <<= 1;
    A.R = R; threads = (R * LV + 63) & ~63;
    A.nbn = pn2_dsra_tail_blocks(d->OH) * d->N;
    blocks = G.ng * ((A.nbn + 7) & ~7);
    A.ptot = 0;
    for (int j = 0; j < 2 * d->P; ++j) { A.poff[j] = A.ptot; A.ptot += 3 * d->maps[j].w; }
    lds_f = lds_b = 0;
    for (int g = 0; g < G.ng; ++g) {
        const size_t v = (((size_t)TBM * TRB * (d->maps[G.idx[g][0]].w + 1) + 3) & ~(size_t)3) * 4;
        const size_t b = v + (TRB * 3 + (size_t)G.nm[g] * 3 * d->OW) * 4;
        if (v > lds_f) lds_f = v;
        if (b > lds_b) lds_b = b;
    }
}

// the one-pass kernel: align_corners = 0, every map magnified by the same power of two >= 8 in both directions, pair p = (maps[p], maps[P + p]) of one
// geometry, <= 4 rows per thread, maps of >= 5 pixels (tail_one_fin_k's image window)
static bool tail_one_geometry(const pn2_tail_desc* d, tail_one_aux& A, int& threads, size_t& lds) {
    if (tail_mode() != 2 || d->align_corners || d->OW > 512 || (long long)d->N * pn2_dsra_tail_blocks(d->OH) > 0x7fffffffLL) return false;
    const int LV = d->OW >> 2;
    int R = 1; while (R * 2 * LV <= 256 && R * 2 <= TRB) R <<= 1;
    if (TRB / R > 4) return false;
    A.R = R; threads = (R * LV + 63) & ~63;
    A.nb = pn2_dsra_tail_blocks(d->OH);
    A.ptot = 0; A.vtot = 0;
    for (int p = 0; p < d->P; ++p) {
        const pn2_tail_map& a = d->maps[p]; const pn2_tail_map& b = d->maps[d->P + p];
        if (a.h != b.h || a.w != b.w || a.rh != b.rh || a.rw != b.rw || a.h * a.w < 5 || a.w > 63) return false;
        const int mag = d->OW / a.w;
        if (mag < 8 || mag > 32 || (mag & (mag - 1)) || mag * a.w != d->OW || mag * a.h != d->OH || a.rw != 1.f / (float)mag || a.rh != 1.f / (float)mag) return false;
        A.mag[p] = mag;
        A.shp[p] = p;
        for (int q = p - 1; q >= 0; --q) if (d->maps[q].w == a.w && d->maps[q].h == a.h) A.shp[p] = q;
        A.poff[p] = A.ptot; A.ptot += 18 * a.w;
    }
    for (int j = 0; j < 2 * d->P; ++j) { A.voff[j] = A.vtot; A.vtot += 3 * (d->maps[j].w + 1); }
    A.vtot = (A.vtot + 3) & ~3;
    lds = ((size_t)A.vtot + (size_t)R * TNK * LV * 2) * 4;
    return lds <= 60 * 1024;
}

int pn2_dsra_tail_scratch(const pn2_tail_desc* d) {
    if (tail_check(d) || !tail_band_ok(d)) return 0;
    long long ptot = 0;
    for (int j = 0; j < 2 * d->P; ++j) ptot += 3 * d->maps[j].w;
    const long long need = (long long)d->N * pn2_dsra_tail_blocks(d->OH) * ptot;
    return need > 0x7fffffffLL ? 0 : (int)need;          // (never reached with tail_check's limits: N*P <= 512, OW <= 1024)
}

int pn2_dsra_tail_fwd(const pn2_tail_desc* d, float* lat, const float* mask, const float* weit, float* partial,
                      float* sums, float* wsum, float* loss, void* stream) {
    if (!lat || !mask || !weit || !partial || !sums || !wsum || !loss) return -1;
    if (int rc = tail_check(d)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int nb = pn2_dsra_tail_blocks(d->OH);
    bool band = tail_mode() >= 1 && tail_band_ok(d);
    if (band) {
        tail_groups G; tail_band_aux A; int threads, blocks; size_t lds_f, lds_b;
        tail_make_groups(d, G, TBM);
        tail_band_geometry(d, G, A, threads, blocks, lds_f, lds_b);
        if (lds_f <= 60 * 1024) hipLaunchKernelGGL(tail_band_fwd_k, dim3(blocks), dim3(threads), lds_f, st, *d, G, A, lat, mask, weit, partial);
        else band = false;
    }
    if (!band) {
        size_t lds = 0;
        for (int j = 0; j < 2 * d->P; ++j) lds += (size_t)((int)(TRB * d->maps[j].rh) + 3) * d->maps[j].w * 4;
        if (lds > 60 * 1024) return -2;
        dim3 grid(nb, d->N);
        switch (d->P) {
            case 1: hipLaunchKernelGGL(tail_fwd_k<1>, grid, dim3(256), lds, st, *d, lat, mask, weit, partial); break;
            case 2: hipLaunchKernelGGL(tail_fwd_k<2>, grid, dim3(256), lds, st, *d, lat, mask, weit, partial); break;
            case 3: hipLaunchKernelGGL(tail_fwd_k<3>, grid, dim3(256), lds, st, *d, lat, mask, weit, partial); break;
            default: hipLaunchKernelGGL(tail_fwd_k<4>, grid, dim3(256), lds, st, *d, lat, mask, weit, partial); break;
        }
    }
    hipLaunchKernelGGL(loss_finalize_k, dim3(1), dim3(1024), 0, st, partial, d->P, d->N, nb, sums, wsum, loss);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_dsra_tail_bwd(const pn2_tail_desc* d, const float* mask, const float* weit, const float* wsum, const float* sums,
                      float gscale, float* scratch, long long scratch_floats, void* stream) {
    if (!mask || !weit || !wsum || !sums) return -1;
    if (int rc = tail_check(d)) return rc;
    for (int j = 0; j < 2 * d->P; ++j) if (!d->maps[j].dsrc) return -1;
    hipStream_t st = (hipStream_t)stream;
    tail_groups G;
    const long long need = pn2_dsra_tail_scratch(d);
    if (need > 0 && scratch && scratch_floats >= need) {
        tail_band_aux A; int threads, blocks; size_t lds_f, lds_b;
        tail_make_groups(d, G, TBM);
        tail_band_geometry(d, G, A, threads, blocks, lds_f, lds_b);
        if (lds_b <= 60 * 1024) {
            const int nb = pn2_dsra_tail_blocks(d->OH);
            hipLaunchKernelGGL(tail_band_bwd_k, dim3(blocks), dim3(threads), lds_b, st, *d, G, A, mask, weit, wsum, sums, gscale, scratch);
            int emax = 0;
            for (int j = 0; j < 2 * d->P; ++j) emax = std::max(emax, d->N * d->maps[j].h * d->maps[j].w);
            hipLaunchKernelGGL(tail_band_fin_k, dim3((emax + 255) / 256, 2 * d->P), dim3(256), 0, st, *d, A, scratch, nb);
            PN2_CHECK_LAUNCH();
            return 0;
        }
    }
    const size_t lds = tail_make_groups(d, G);
    if (lds > 60 * 1024) return -2;
    hipLaunchKernelGGL(tail_bwd_k, dim3(G.start[G.ng]), dim3(256), lds, st, *d, G, mask, weit, wsum, sums, gscale);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_dsra_tail_fused_ok(const pn2_tail_desc* d) {
    if (tail_check(d)) return 0;
    tail_one_aux A; int threads; size_t lds;
    return tail_one_geometry(d, A, threads, lds) && (long long)d->N * A.nb * A.ptot <= 0x7fffffffLL ? 1 : 0;
}

int pn2_dsra_tail_fused_scratch(const pn2_tail_desc* d) {
    if (tail_check(d)) return 0;
    tail_one_aux A; int threads; size_t lds;
    if (!tail_one_geometry(d, A, threads, lds)) return 0;
    const long long need = (long long)d->N * A.nb * A.ptot;
    return need > 0x7fffffffLL ? 0 : (int)need;
}

int pn2_dsra_tail_fwd_bwd(const pn2_tail_desc* d, float* lat, const float* mask, const float* weit, float* partial, float* sums, float* wsum,
                          float* per, float* loss, float gscale, float* scratch, long long scratch_floats, long long* isum, void* stream) {
    if (!lat || !mask || !weit || !partial || !sums || !wsum || !per || !loss || !scratch) return -1;
    if (int rc = tail_check(d)) return rc;
    for (int j = 0; j < 2 * d->P; ++j) if (!d->maps[j].dsrc) return -1;
    tail_one_aux A; int threads; size_t lds;
    if (!tail_one_geometry(d, A, threads, lds)) return -2;
    if (scratch_floats < (long long)d->N * A.nb * A.ptot) return -2;
    if (isum && (d->P * d->N > TLB_MAXPN || d->P > 8)) return -2;          // (P * N * 5 two-word fixed-point sums, ZERO on entry: pn2_loss_weights_clear; NULL: two more passes, see pn2.h)
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(A.nb * d->N), blk(threads);
    switch (d->P) {
        case 1: hipLaunchKernelGGL(tail_one_k<1>, grid, blk, lds, st, *d, A, lat, mask, weit, partial, scratch, isum); break;
        case 2: hipLaunchKernelGGL(tail_one_k<2>, grid, blk, lds, st, *d, A, lat, mask, weit, partial, scratch, isum); break;
        case 3: hipLaunchKernelGGL(tail_one_k<3>, grid, blk, lds, st, *d, A, lat, mask, weit, partial, scratch, isum); break;
        default: hipLaunchKernelGGL(tail_one_k<4>, grid, blk, lds, st, *d, A, lat, mask, weit, partial, scratch, isum); break;
    }
    int emax = 0;
    for (int j = 0; j < 2 * d->P; ++j) emax = std::max(emax, d->N * d->maps[j].h * d->maps[j].w);
    hipLaunchKernelGGL(tail_one_fin_k, dim3((emax + 255) / 256, 2 * d->P + (isum ? 1 : 0)), dim3(256), 0, st, *d, A, scratch, partial, gscale, sums, wsum, per, loss, (const long long*)isum);
    if (!isum) hipLaunchKernelGGL(loss_total_k, dim3(1), dim3(64), 0, st, per, d->P, d->N, loss);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_eval_hist(const unsigned char* pred_u8, const float* gt, long long n, unsigned* hist, void* stream) {
    if (!pred_u8 || !gt || !hist || n < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(hist, 0, 512 * sizeof(unsigned), st) != hipSuccess) return -4;
    long long g = (n + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(eval_hist_k, dim3((unsigned)(g > 1024 ? 1024 : g)), dim3(256), 0, st, pred_u8, gt, n, hist);
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
