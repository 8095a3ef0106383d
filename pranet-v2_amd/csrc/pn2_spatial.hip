// pn2_spatial.hip — pooling, bilinear resampling, element-wise glue and layout conversion.
// HBM-bound NHWC streaming kernels (16-byte channel vectors, grid-stride, no atomics: every backward
// is written in gather form so results are deterministic).
//
// Reference ops replaced:
//   nn.MaxPool2d(3,2,1)                          /root/reference/binary_seg/lib/Res2Net_v1b.py:112
//   nn.AvgPool2d(3,stride,1)                     Res2Net_v1b.py:40,80
//   nn.AvgPool2d(s,s,ceil_mode,count_include_pad=False)   Res2Net_v1b.py:131-132
//   nn.Upsample(x2,bilinear,align_corners=True)  /root/reference/binary_seg/lib/pranet.py:93,111-118
//   F.interpolate(scale_factor,bilinear)         pranet.py:349-354,370-376,392-398,414-415
//   torch.split/cat, `sp + spx[i]`, `a * b`      Res2Net_v1b.py:65-80 ; pranet.py:111-119
#include "pn2_common.h"
#include "../../include/pn2.h"

namespace {

struct __attribute__((packed, aligned(4))) f32x3_t { float a, b, c; };      // 12-byte vector of the fp32 K-channel maps (K = 9 = 3 x 3)
template <typename T, int W> struct VL {
    __device__ static __forceinline__ void load(const T* p, float* f) {
        if constexpr (W == 1) f[0] = TT<T>::ld(p);
        else if constexpr (W == 3) { const f32x3_t v = *reinterpret_cast<const f32x3_t*>(p); f[0] = v.a; f[1] = v.b; f[2] = v.c; }
        else TT<T>::unpack(*reinterpret_cast<const uint4*>(p), f);
    }
    __device__ static __forceinline__ void store(T* p, const float* f) {
        if constexpr (W == 1) TT<T>::st(p, f[0]);
        else if constexpr (W == 3) { f32x3_t v; v.a = f[0]; v.b = f[1]; v.c = f[2]; *reinterpret_cast<f32x3_t*>(p) = v; }
        else *reinterpret_cast<uint4*>(p) = TT<T>::pack(f);
    }
};

// The kernels index with 32 bits and step by gridDim.x * 256 <= 16384 * 256: the C entry points refuse (status -2) element counts within one step
// of 2^32, where `idx += step` would wrap and the loop never end (PN2_TOO_MANY); grid_for keeps the impossible-grid backstop.
constexpr size_t PIX_MAX = 0xFFFFFFFFull - 16384ull * 256ull;
#define PN2_TOO_MANY(total) do { if ((size_t)(total) > PIX_MAX) return -2; } while (0)
inline int grid_for(size_t total) { if (total > PIX_MAX) return -1; size_t g = (total + 255) / 256; return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g)); }

// 32-bit element index (the C entry points refuse tensors of 2^32 or more vectors): the 64-bit `idx % CV`, `p % W`, `p / H` chains of a pixel decode were
// ~100 instructions each - most of what these streaming kernels executed
#define PIX_LOOP(total) for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < (total); idx += gridDim.x * 256u)

// ------------------------------------------------------------------------------------------ max pool
template <typename T, int W>
__global__ __launch_bounds__(256) void maxpool_fwd_k(const T* __restrict__ x, int ld_x, T* __restrict__ y, int ld_y, unsigned char* __restrict__ arg,
                                                     int N, int H, int Wd, int C, int OH, int OW) {
    const int CV = C / W;
    const size_t total = (size_t)N * OH * OW * CV;
    PIX_LOOP(total) {
        const int cv = (int)(idx % CV); unsigned p = idx / CV;
        const int ox = (int)(p % OW); p /= OW; const int oy = (int)(p % OH); const int n = (int)(p / OH);
        float best[W]; int bi[W];
#pragma unroll
        for (int e = 0; e < W; ++e) { best[e] = -INFINITY; bi[e] = 0; }
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * 2 - 1 + r; if ((unsigned)iy >= (unsigned)H) continue;
            for (int s = 0; s < 3; ++s) {
                const int ix = ox * 2 - 1 + s; if ((unsigned)ix >= (unsigned)Wd) continue;
                float v[W];
                VL<T, W>::load(x + ((size_t)(n * H + iy) * Wd + ix) * ld_x + cv * W, v);
#pragma unroll
                for (int e = 0; e < W; ++e) if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = r * 3 + s; }
            }
        }
        const size_t o = (size_t)(n * OH + oy) * OW + ox;
        VL<T, W>::store(y + o * ld_y + cv * W, best);
#pragma unroll
        for (int e = 0; e < W; ++e) arg[o * C + cv * W + e] = (unsigned char)bi[e];
    }
}

template <typename T, int W>
__global__ __launch_bounds__(256) void maxpool_bwd_k(const T* __restrict__ dy, int ld_dy, const unsigned char* __restrict__ arg, T* __restrict__ dx, int ld_dx,
                                                     int N, int H, int Wd, int C, int OH, int OW) {
    const int CV = C / W;
    const size_t total = (size_t)N * H * Wd * CV;
    PIX_LOOP(total) {
        const int cv = (int)(idx % CV); unsigned p = idx / CV;
        const int ix = (int)(p % Wd); p /= Wd; const int iy = (int)(p % H); const int n = (int)(p / H);
        float acc[W];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] = 0.f;
        for (int r = 0; r < 3; ++r) {
            const int ty = iy + 1 - r; if (ty < 0 || (ty & 1)) continue; const int oy = ty >> 1; if (oy >= OH) continue;
            for (int s = 0; s < 3; ++s) {
                const int tx = ix + 1 - s; if (tx < 0 || (tx & 1)) continue; const int ox = tx >> 1; if (ox >= OW) continue;
                const size_t o = (size_t)(n * OH + oy) * OW + ox;
                float g[W];
                VL<T, W>::load(dy + o * ld_dy + cv * W, g);
#pragma unroll
                for (int e = 0; e < W; ++e) if (arg[o * C + cv * W + e] == r * 3 + s) acc[e] += g[e];
            }
        }
        VL<T, W>::store(dx + ((size_t)(n * H + iy) * Wd + ix) * ld_dx + cv * W, acc);
    }
}

// ------------------------------------------------------------------------------------------ avg pool
__device__ __forceinline__ float avg_div(int o, int k, int stride, int pad, int In, int include_pad, int& lo, int& hi) {
    int s = o * stride - pad, e = s + k; if (e > In + pad) e = In + pad;
    const int full = e - s;
    lo = s < 0 ? 0 : s; hi = e > In ? In : e;
    return (float)(include_pad ? full : (hi - lo));
}

template <typename T, int W>
__global__ __launch_bounds__(256) void avgpool_fwd_k(const T* __restrict__ x, int ld_x, T* __restrict__ y, int ld_y, int N, int H, int Wd, int C, int OH, int OW,
                                                     int k, int stride, int pad, int inc) {
    const int CV = C / W;
    const size_t total = (size_t)N * OH * OW * CV;
    PIX_LOOP(total) {
        const int cv = (int)(idx % CV); unsigned p = idx / CV;
        const int ox = (int)(p % OW); p /= OW; const int oy = (int)(p % OH); const int n = (int)(p / OH);
        int y0, y1, x0, x1;
        const float dh = avg_div(oy, k, stride, pad, H, inc, y0, y1), dw = avg_div(ox, k, stride, pad, Wd, inc, x0, x1);
        float acc[W];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] = 0.f;
        for (int iy = y0; iy < y1; ++iy)
            for (int ix = x0; ix < x1; ++ix) {
                float v[W];
                VL<T, W>::load(x + ((size_t)(n * H + iy) * Wd + ix) * ld_x + cv * W, v);
#pragma unroll
                for (int e = 0; e < W; ++e) acc[e] += v[e];
            }
        const float inv = 1.f / (dh * dw);
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] *= inv;
        VL<T, W>::store(y + ((size_t)(n * OH + oy) * OW + ox) * ld_y + cv * W, acc);
    }
}

template <typename T, int W>
__global__ __launch_bounds__(256) void avgpool_bwd_k(const T* __restrict__ dy, int ld_dy, T* __restrict__ dx, int ld_dx, int N, int H, int Wd, int C, int OH, int OW,
                                                     int k, int stride, int pad, int inc, int accumulate) {
    const int CV = C / W;
    const size_t total = (size_t)N * H * Wd * CV;
    PIX_LOOP(total) {
        const int cv = (int)(idx % CV); unsigned p = idx / CV;
        const int ix = (int)(p % Wd); p /= Wd; const int iy = (int)(p % H); const int n = (int)(p / H);
        float acc[W];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] = 0.f;
        int oy0 = (iy + pad - k + stride) / stride; if (iy + pad - k + 1 < 0) oy0 = 0;
        int ox0 = (ix + pad - k + stride) / stride; if (ix + pad - k + 1 < 0) ox0 = 0;
        int oy1 = (iy + pad) / stride; if (oy1 >= OH) oy1 = OH - 1;
        int ox1 = (ix + pad) / stride; if (ox1 >= OW) ox1 = OW - 1;
        for (int oy = oy0; oy <= oy1; ++oy)
            for (int ox = ox0; ox <= ox1; ++ox) {
                int a, b;
                const float dh = avg_div(oy, k, stride, pad, H, inc, a, b);
                if (iy < a || iy >= b) continue;
                const float dw = avg_div(ox, k, stride, pad, Wd, inc, a, b);
                if (ix < a || ix >= b) continue;
                float g[W];
                VL<T, W>::load(dy + ((size_t)(n * OH + oy) * OW + ox) * ld_dy + cv * W, g);
                const float inv = 1.f / (dh * dw);
#pragma unroll
                for (int e = 0; e < W; ++e) acc[e] += g[e] * inv;
            }
        T* d = dx + ((size_t)(n * H + iy) * Wd + ix) * ld_dx + cv * W;
        if (accumulate) { float o[W]; VL<T, W>::load(d, o);
#pragma unroll
            for (int e = 0; e < W; ++e) acc[e] += o[e]; }
        VL<T, W>::store(d, acc);
    }
}

// ------------------------------------------------------------------------------------------ bilinear
// PyTorch upsample_bilinear2d index math (area_pixel_compute_source_index):
//   align_corners: src = r*dst ; else src = max(r*(dst+0.5)-0.5, 0) ; i0 = (int)src ; i1 = i0 + (i0 < In-1) ; l1 = src - i0.
template <typename T, int W>
__global__ __launch_bounds__(256) void bilinear_fwd_k(const T* __restrict__ x, int ld_x, T* __restrict__ y, int ld_y, int N, int H, int Wd, int C, int OH, int OW,
                                                      int ac, float rh, float rw) {
    const int CV = C / W;
    const size_t total = (size_t)N * OH * OW * CV;
    PIX_LOOP(total) {
        const int cv = (int)(idx % CV); unsigned p = idx / CV;
        const int ox = (int)(p % OW); p /= OW; const int oy = (int)(p % OH); const int n = (int)(p / OH);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        bl_src(oy, rh, ac, H, y0, y1, ly0, ly1); bl_src(ox, rw, ac, Wd, x0, x1, lx0, lx1);
        const T* b = x + (size_t)n * H * Wd * ld_x + cv * W;
        float a[W], bq[W], c[W], d[W], o[W];
        VL<T, W>::load(b + (size_t)(y0 * Wd + x0) * ld_x, a); VL<T, W>::load(b + (size_t)(y0 * Wd + x1) * ld_x, bq);
        VL<T, W>::load(b + (size_t)(y1 * Wd + x0) * ld_x, c); VL<T, W>::load(b + (size_t)(y1 * Wd + x1) * ld_x, d);
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = ly0 * (lx0 * a[e] + lx1 * bq[e]) + ly1 * (lx0 * c[e] + lx1 * d[e]);
        VL<T, W>::store(y + ((size_t)(n * OH + oy) * OW + ox) * ld_y + cv * W, o);
    }
}

template <typename T, int W>
__global__ __launch_bounds__(256) void bilinear_bwd_k(const T* __restrict__ dy, int ld_dy, T* __restrict__ dx, int ld_dx, int N, int H, int Wd, int C, int OH, int OW,
                                                      int ac, float rh, float rw, int accumulate) {
    const int CV = C / W;
    const size_t total = (size_t)N * H * Wd * CV;
    PIX_LOOP(total) {
        const int cv = (int)(idx % CV); unsigned p = idx / CV;
        const int ix = (int)(p % Wd); p /= Wd; const int iy = (int)(p % H); const int n = (int)(p / H);
        int oy0, oy1, ox0, ox1;
        bl_range(iy, rh, ac, OH, oy0, oy1); bl_range(ix, rw, ac, OW, ox0, ox1);
        float acc[W];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] = 0.f;
        for (int oy = oy0; oy <= oy1; ++oy) {
            int y0, y1; float ly0, ly1;
            bl_src(oy, rh, ac, H, y0, y1, ly0, ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox0; ox <= ox1; ++ox) {
                int x0, x1; float lx0, lx1;
                bl_src(ox, rw, ac, Wd, x0, x1, lx0, lx1);
                const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
                if (wx == 0.f) continue;
                float g[W];
                VL<T, W>::load(dy + ((size_t)(n * OH + oy) * OW + ox) * ld_dy + cv * W, g);
                const float w = wy * wx;
#pragma unroll
                for (int e = 0; e < W; ++e) acc[e] += w * g[e];
            }
        }
        T* d = dx + ((size_t)(n * H + iy) * Wd + ix) * ld_dx + cv * W;
        if (accumulate) { float o[W]; VL<T, W>::load(d, o);
#pragma unroll
            for (int e = 0; e < W; ++e) acc[e] += o[e]; }
        VL<T, W>::store(d, acc);
    }
}

// large magnification (the x8 / x16 / x32 lateral maps, K channels): one input pixel gathers hundreds of output pixels, and there
// are few input pixels -> one WAVE per (pixel, channel), the lanes share the window (consecutive lanes = consecutive ox), then a
// fixed-order butterfly.  Deterministic.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_wave_k(const T* __restrict__ dy, int ld_dy, T* __restrict__ dx, int ld_dx, int N, int H, int Wd, int C, int OH, int OW,
                                                           int ac, float rh, float rw, int accumulate) {
    const size_t item = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (item >= (size_t)N * H * Wd * C) return;
    const int c = (int)(item % C); size_t p = item / C;
    const int ix = (int)(p % Wd); p /= Wd; const int iy = (int)(p % H); const int n = (int)(p / H);
    int oy0, oy1, ox0, ox1;
    bl_range(iy, rh, ac, OH, oy0, oy1); bl_range(ix, rw, ac, OW, ox0, ox1);
    const int ncol = ox1 - ox0 + 1, ntap = (oy1 - oy0 + 1) * ncol;
    float acc = 0.f;
    for (int t = lane; t < ntap; t += 64) {
        const int r = t / ncol, oy = oy0 + r, ox = ox0 + (t - r * ncol);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        bl_src(oy, rh, ac, H, y0, y1, ly0, ly1); bl_src(ox, rw, ac, Wd, x0, x1, lx0, lx1);
        const float w = ((y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f)) * ((x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f));
        if (w != 0.f) acc += w * TT<T>::ld(dy + ((size_t)(n * OH + oy) * OW + ox) * ld_dy + c);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        T* d = dx + ((size_t)(n * H + iy) * Wd + ix) * ld_dx + c;
        TT<T>::st(d, accumulate ? TT<T>::ld(d) + acc : acc);
    }
}

// same case, contiguous K-channel rows (ld_dy == C): one block per input row (n, iy).  The block first reduces its window of
// output rows along y into one row of OW*C column sums (every dy row is read as one coalesced line), then the Wd*C results are
// short dot products over that row in LDS.  dy is read ~2x in total (each output row feeds two input rows).  Deterministic.
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_rows_k(const T* __restrict__ dy, T* __restrict__ dx, int ld_dx, int N, int H, int Wd, int C, int OH, int OW,
                                                           int ac, float rh, float rw, int accumulate) {
    extern __shared__ float col[];     // [OW * C]
    const int n = blockIdx.x / H, iy = blockIdx.x - n * H;
    int oy0, oy1;
    bl_range(iy, rh, ac, OH, oy0, oy1);
    const int L = OW * C;
    const T* base = dy + (size_t)n * OH * L;
    if constexpr (sizeof(T) == 4) {
        if ((L & 3) == 0 && L <= 1024) {
            // 16-byte loads, the window rows split over R row lanes (more bytes in flight: the x16 / x32 maps have few blocks)
            const int LV = L >> 2;
            int lvp = 64; while (lvp < LV) lvp <<= 1;
            const int R = 256 / lvp, jv = threadIdx.x % lvp, rl = threadIdx.x / lvp;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (jv < LV) {
#pragma unroll 8
                for (int oy = oy0 + rl; oy <= oy1; oy += R) {
                    int y0, y1; float ly0, ly1;
                    bl_src(oy, rh, ac, H, y0, y1, ly0, ly1);
                    const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
                    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)oy * L + jv * 4);
                    a.x += wy * v.x; a.y += wy * v.y; a.z += wy * v.z; a.w += wy * v.w;
                }
            }
            for (int r = 0; r < R; ++r) {          // combine the row lanes in a fixed order
                if (rl == r && jv < LV) {
                    float4* c4 = reinterpret_cast<float4*>(col) + jv;
                    if (r == 0) *c4 = a;
                    else { float4 o = *c4; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; *c4 = o; }
                }
                __syncthreads();
            }
            goto reduce_x;
        }
    }
    for (int j = threadIdx.x; j < L; j += 256) {
        float a = 0.f;
#pragma unroll 4
        for (int oy = oy0; oy <= oy1; ++oy) {
            int y0, y1; float ly0, ly1;
            bl_src(oy, rh, ac, H, y0, y1, ly0, ly1);
            const float wy = (y0 == iy ? ly0 : 0.f) + (y1 == iy ? ly1 : 0.f);
            a += wy * TT<T>::ld(base + (size_t)oy * L + j);
        }
        col[j] = a;
    }
reduce_x:
    __syncthreads();
    for (int o = threadIdx.x; o < Wd * C; o += 256) {
        const int ix = o / C, c = o - ix * C;
        int ox0, ox1;
        bl_range(ix, rw, ac, OW, ox0, ox1);
        float a = 0.f;
        for (int ox = ox0; ox <= ox1; ++ox) {
            int x0, x1; float lx0, lx1;
            bl_src(ox, rw, ac, Wd, x0, x1, lx0, lx1);
            const float wx = (x0 == ix ? lx0 : 0.f) + (x1 == ix ? lx1 : 0.f);
            a += wx * col[ox * C + c];
        }
        T* d = dx + ((size_t)(n * H + iy) * Wd + ix) * ld_dx + c;
        TT<T>::st(d, accumulate ? TT<T>::ld(d) + a : a);
    }
}

// ------------------------------------------------------------------------------------------ element-wise
template <typename T, int W>
__global__ __launch_bounds__(256) void binary_k(int op, const T* __restrict__ a, int ld_a, const T* __restrict__ b, int ld_b, T* __restrict__ out, int ld_o,
                                                int M, int C, int accumulate) {
    const int CV = C / W;
    const size_t total = (size_t)M * CV;
    PIX_LOOP(total) {
        const int m = (int)(idx / CV), c = (int)(idx % CV) * W;
        float x[W], y[W], o[W];
        VL<T, W>::load(a + (size_t)m * ld_a + c, x); VL<T, W>::load(b + (size_t)m * ld_b + c, y);
        if (accumulate) VL<T, W>::load(out + (size_t)m * ld_o + c, o);
#pragma unroll
        for (int e = 0; e < W; ++e) { const float r = op == 0 ? x[e] + y[e] : x[e] * y[e]; o[e] = accumulate ? o[e] + r : r; }
        VL<T, W>::store(out + (size_t)m * ld_o + c, o);
    }
}

// backward of out = a * b in ONE pass over the gradient: ga (+)= g * b, gb (+)= g * a (two pn2_binary launches otherwise; the aggregation's products, pranet.py:111-119)
template <typename T, int W>
__global__ __launch_bounds__(256) void mul_bwd_k(const T* __restrict__ g, int ld_g, const T* __restrict__ a, int ld_a, const T* __restrict__ b, int ld_b,
                                                 T* __restrict__ ga, int ld_ga, int acc_a, T* __restrict__ gb, int ld_gb, int acc_b, int M, int C) {
    const int CV = C / W;
    const size_t total = (size_t)M * CV;
    PIX_LOOP(total) {
        const int m = (int)(idx / CV), c = (int)(idx % CV) * W;
        float gv[W], x[W], y[W], oa[W], ob[W];
        VL<T, W>::load(g + (size_t)m * ld_g + c, gv); VL<T, W>::load(a + (size_t)m * ld_a + c, x); VL<T, W>::load(b + (size_t)m * ld_b + c, y);
        if (acc_a) VL<T, W>::load(ga + (size_t)m * ld_ga + c, oa);
        if (acc_b) VL<T, W>::load(gb + (size_t)m * ld_gb + c, ob);
#pragma unroll
        for (int e = 0; e < W; ++e) {
            const float ra = gv[e] * y[e], rb = gv[e] * x[e];
            oa[e] = acc_a ? oa[e] + ra : ra; ob[e] = acc_b ? ob[e] + rb : rb;
        }
        VL<T, W>::store(ga + (size_t)m * ld_ga + c, oa); VL<T, W>::store(gb + (size_t)m * ld_gb + c, ob);
    }
}

template <typename Ti, typename To, int W>
__global__ __launch_bounds__(256) void copy_k(const Ti* __restrict__ s, int ld_s, To* __restrict__ d, int ld_d, int M, int C, int accumulate) {
    const int CV = C / W;
    const size_t total = (size_t)M * CV;
    PIX_LOOP(total) {
        const int m = (int)(idx / CV), c = (int)(idx % CV) * W;
        float x[W];
        VL<Ti, W>::load(s + (size_t)m * ld_s + c, x);
        if constexpr (sizeof(Ti) == sizeof(To)) {
            if (accumulate) { float o[W]; VL<To, W>::load(d + (size_t)m * ld_d + c, o);
#pragma unroll
                for (int e = 0; e < W; ++e) x[e] += o[e]; }
            VL<To, W>::store(d + (size_t)m * ld_d + c, x);
        } else {
#pragma unroll
            for (int e = 0; e < W; ++e) { To* q = d + (size_t)m * ld_d + c + e; TT<To>::st(q, accumulate ? x[e] + TT<To>::ld(q) : x[e]); }
        }
    }
}

// table-driven copies of one dtype (16-byte vectors): the pass-through slices of independent chains at one lock-step position (the branch0 outputs of the three RFB
// modules into their concat buffers, pranet.py:77-79, and the same slices of the gradient on the way back) as ONE launch
template <typename T>
__global__ __launch_bounds__(256) void copy_tab_k(const pn2_copy_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    constexpr int W = TT<T>::VEC;
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_copy_job j = jobs[jb];
    const unsigned b = blockIdx.x - bstart[jb], nb = bstart[jb + 1] - bstart[jb];
    const T* __restrict__ s = (const T*)j.src; T* __restrict__ d = (T*)j.dst;
    const int CV = j.C / W;
    const size_t total = (size_t)j.M * CV;
    for (size_t idx = (size_t)b * 256u + threadIdx.x; idx < total; idx += (size_t)nb * 256u) {
        const int m = (int)(idx / CV), c = (int)(idx % CV) * W;
        float x[W];
        VL<T, W>::load(s + (size_t)m * j.ld_s + c, x);
        if (j.accumulate) { float o[W]; VL<T, W>::load(d + (size_t)m * j.ld_d + c, o);
#pragma unroll
            for (int e = 0; e < W; ++e) x[e] += o[e]; }
        VL<T, W>::store(d + (size_t)m * j.ld_d + c, x);
    }
}

template <typename To>
__global__ __launch_bounds__(256) void nchw_to_nhwc_k(const float* __restrict__ x, To* __restrict__ y, int ld_y, int N, int C, int HW, int Cp) {
    const size_t total = (size_t)N * HW;
    PIX_LOOP(total) {
        const int n = (int)(idx / HW), p = (int)(idx % HW);
        To* d = y + (size_t)idx * ld_y;
        constexpr int V = TT<To>::VEC;
        if (Cp == V && (ld_y % V) == 0 && (reinterpret_cast<size_t>(y) & 15) == 0) {      // the usual case (3 channels in an 8 / 4 slot pixel): one 16-byte store
            float f[V];
#pragma unroll
            for (int c = 0; c < V; ++c) f[c] = c < C ? x[((size_t)n * C + c) * HW + p] : 0.f;
            *reinterpret_cast<uint4*>(d) = TT<To>::pack(f);
        } else {
            for (int c = 0; c < Cp; ++c) TT<To>::st(d + c, c < C ? x[((size_t)n * C + c) * HW + p] : 0.f);
        }
    }
}

__global__ __launch_bounds__(1024) void bias_grad_k(const float* __restrict__ dy, int M, int K, float* db, int accumulate) {
    // one block per channel k, 1024 lanes with 8 independent loads in flight each (a 256-lane walk of dependent-latency loads took 56 us on
    // the 61952-row K = 1 head maps); fixed-order tree reduce -> deterministic
    __shared__ float sh[1024];
    const int k = blockIdx.x, t = threadIdx.x;
    float s = 0.f;
    if (K == 1 && (M & 3) == 0 && (reinterpret_cast<size_t>(dy) & 15) == 0) {
        const float4* d4 = reinterpret_cast<const float4*>(dy);
        const int M4 = M >> 2;
        int m = t;
        for (; m + 3 * 1024 < M4; m += 4 * 1024) {
            const float4 a = d4[m], b = d4[m + 1024], c = d4[m + 2048], d = d4[m + 3072];
            s += ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w)) + ((c.x + c.y) + (c.z + c.w)) + ((d.x + d.y) + (d.z + d.w));
        }
        for (; m < M4; m += 1024) { const float4 a = d4[m]; s += (a.x + a.y) + (a.z + a.w); }
    } else {
        int m = t;
        for (; m + 7 * 1024 < M; m += 8 * 1024) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = dy[(size_t)(m + u * 1024) * K + k];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; m < M; m += 1024) s += dy[(size_t)m * K + k];
    }
    sh[t] = s; __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if (t < o) sh[t] += sh[t + o]; __syncthreads(); }
    if (t == 0) db[k] = accumulate ? db[k] + sh[0] : sh[0];
}

template <typename T> bool vec_ok(int C, int a, int b = 0, int c = 0, int d = 0) {
    constexpr int V = TT<T>::VEC;
    return C % V == 0 && a % V == 0 && b % V == 0 && c % V == 0 && d % V == 0;
}

#define DISPATCH_T(dt, CALL)                                     \
    if ((dt) == PN2_BF16) { using T = bf16_t; CALL }             \
    else if ((dt) == PN2_F32) { using T = float; CALL }          \
    else return -3;

}  // namespace

extern "C" {

int pn2_maxpool3x3s2_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, unsigned char* idx, int N, int H, int W, int C, int OH, int OW, void* stream) {
    if (!x || !y || !idx) return -1;
    PN2_TOO_MANY((size_t)N * OH * OW * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if (vec_ok<T>(C, ld_x, ld_y)) hipLaunchKernelGGL((maxpool_fwd_k<T, TT<T>::VEC>), dim3(grid_for((size_t)N * OH * OW * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)x, ld_x, (T*)y, ld_y, idx, N, H, W, C, OH, OW);
        else hipLaunchKernelGGL((maxpool_fwd_k<T, 1>), dim3(grid_for((size_t)N * OH * OW * C)), dim3(256), 0, st, (const T*)x, ld_x, (T*)y, ld_y, idx, N, H, W, C, OH, OW);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_maxpool3x3s2_bwd(int dt, const void* dy, int ld_dy, const unsigned char* idx, void* dx, int ld_dx, int N, int H, int W, int C, int OH, int OW, void* stream) {
    if (!dy || !dx || !idx) return -1;
    PN2_TOO_MANY((size_t)N * H * W * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if (vec_ok<T>(C, ld_dy, ld_dx)) hipLaunchKernelGGL((maxpool_bwd_k<T, TT<T>::VEC>), dim3(grid_for((size_t)N * H * W * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)dy, ld_dy, idx, (T*)dx, ld_dx, N, H, W, C, OH, OW);
        else hipLaunchKernelGGL((maxpool_bwd_k<T, 1>), dim3(grid_for((size_t)N * H * W * C)), dim3(256), 0, st, (const T*)dy, ld_dy, idx, (T*)dx, ld_dx, N, H, W, C, OH, OW);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_avgpool_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, int OH, int OW, int k, int stride, int pad, int inc, void* stream) {
    if (!x || !y) return -1;
    PN2_TOO_MANY((size_t)N * OH * OW * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if (vec_ok<T>(C, ld_x, ld_y)) hipLaunchKernelGGL((avgpool_fwd_k<T, TT<T>::VEC>), dim3(grid_for((size_t)N * OH * OW * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)x, ld_x, (T*)y, ld_y, N, H, W, C, OH, OW, k, stride, pad, inc);
        else hipLaunchKernelGGL((avgpool_fwd_k<T, 1>), dim3(grid_for((size_t)N * OH * OW * C)), dim3(256), 0, st, (const T*)x, ld_x, (T*)y, ld_y, N, H, W, C, OH, OW, k, stride, pad, inc);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_avgpool_bwd(int dt, const void* dy, int ld_dy, void* dx, int ld_dx, int N, int H, int W, int C, int OH, int OW, int k, int stride, int pad, int inc, int accumulate, void* stream) {
    if (!dy || !dx) return -1;
    PN2_TOO_MANY((size_t)N * H * W * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if (vec_ok<T>(C, ld_dy, ld_dx)) hipLaunchKernelGGL((avgpool_bwd_k<T, TT<T>::VEC>), dim3(grid_for((size_t)N * H * W * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)dy, ld_dy, (T*)dx, ld_dx, N, H, W, C, OH, OW, k, stride, pad, inc, accumulate);
        else hipLaunchKernelGGL((avgpool_bwd_k<T, 1>), dim3(grid_for((size_t)N * H * W * C)), dim3(256), 0, st, (const T*)dy, ld_dy, (T*)dx, ld_dx, N, H, W, C, OH, OW, k, stride, pad, inc, accumulate);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bilinear_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, int OH, int OW, int ac, float rh, float rw, void* stream) {
    if (!x || !y) return -1;
    PN2_TOO_MANY((size_t)N * OH * OW * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if (vec_ok<T>(C, ld_x, ld_y)) hipLaunchKernelGGL((bilinear_fwd_k<T, TT<T>::VEC>), dim3(grid_for((size_t)N * OH * OW * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)x, ld_x, (T*)y, ld_y, N, H, W, C, OH, OW, ac, rh, rw);
        else if (sizeof(T) == 4 && C % 3 == 0 && ld_x % 3 == 0 && ld_y % 3 == 0)      // fp32 K = 9 class maps: 12-byte vectors
            hipLaunchKernelGGL((bilinear_fwd_k<float, 3>), dim3(grid_for((size_t)N * OH * OW * C / 3)), dim3(256), 0, st, (const float*)x, ld_x, (float*)y, ld_y, N, H, W, C, OH, OW, ac, rh, rw);
        else hipLaunchKernelGGL((bilinear_fwd_k<T, 1>), dim3(grid_for((size_t)N * OH * OW * C)), dim3(256), 0, st, (const T*)x, ld_x, (T*)y, ld_y, N, H, W, C, OH, OW, ac, rh, rw);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bilinear_bwd(int dt, const void* dy, int ld_dy, void* dx, int ld_dx, int N, int H, int W, int C, int OH, int OW, int ac, float rh, float rw, int accumulate, void* stream) {
    if (!dy || !dx) return -1;
    PN2_TOO_MANY((size_t)N * H * W * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if ((C < TT<T>::VEC || (sizeof(T) == 4 && C <= 16)) && OH >= 4 * H && OW >= 4 * W && ld_dy == C && OW * C <= 8192)      // K-channel fp32 maps (K = 1 .. 9)
            hipLaunchKernelGGL((bilinear_bwd_rows_k<T>), dim3(N * H), dim3(256), OW * C * 4, st, (const T*)dy, (T*)dx, ld_dx, N, H, W, C, OH, OW, ac, rh, rw, accumulate);
        else if (C < TT<T>::VEC && OH >= 4 * H && OW >= 4 * W)
            hipLaunchKernelGGL((bilinear_bwd_wave_k<T>), dim3((unsigned)(((size_t)N * H * W * C + 3) / 4)), dim3(256), 0, st, (const T*)dy, ld_dy, (T*)dx, ld_dx, N, H, W, C, OH, OW, ac, rh, rw, accumulate);
        else if (vec_ok<T>(C, ld_dy, ld_dx)) hipLaunchKernelGGL((bilinear_bwd_k<T, TT<T>::VEC>), dim3(grid_for((size_t)N * H * W * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)dy, ld_dy, (T*)dx, ld_dx, N, H, W, C, OH, OW, ac, rh, rw, accumulate);
        else if (sizeof(T) == 4 && C % 3 == 0 && ld_dy % 3 == 0 && ld_dx % 3 == 0)      // fp32 K = 9 class maps: 12-byte vectors
            hipLaunchKernelGGL((bilinear_bwd_k<float, 3>), dim3(grid_for((size_t)N * H * W * C / 3)), dim3(256), 0, st, (const float*)dy, ld_dy, (float*)dx, ld_dx, N, H, W, C, OH, OW, ac, rh, rw, accumulate);
        else hipLaunchKernelGGL((bilinear_bwd_k<T, 1>), dim3(grid_for((size_t)N * H * W * C)), dim3(256), 0, st, (const T*)dy, ld_dy, (T*)dx, ld_dx, N, H, W, C, OH, OW, ac, rh, rw, accumulate);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_binary(int dt, int op, const void* a, int ld_a, const void* b, int ld_b, void* out, int ld_out, int M, int C, int accumulate, void* stream) {
    if (!a || !b || !out) return -1;
    PN2_TOO_MANY((size_t)M * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if (vec_ok<T>(C, ld_a, ld_b, ld_out)) hipLaunchKernelGGL((binary_k<T, TT<T>::VEC>), dim3(grid_for((size_t)M * C / TT<T>::VEC)), dim3(256), 0, st, op, (const T*)a, ld_a, (const T*)b, ld_b, (T*)out, ld_out, M, C, accumulate);
        else hipLaunchKernelGGL((binary_k<T, 1>), dim3(grid_for((size_t)M * C)), dim3(256), 0, st, op, (const T*)a, ld_a, (const T*)b, ld_b, (T*)out, ld_out, M, C, accumulate);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_copy(int dt_in, const void* src, int ld_s, int dt_out, void* dst, int ld_d, int M, int C, int accumulate, void* stream) {
    if (!src || !dst) return -1;
    PN2_TOO_MANY((size_t)M * C);
    hipStream_t st = (hipStream_t)stream;
    if (dt_in == dt_out) {
        DISPATCH_T(dt_in, {
            if (vec_ok<T>(C, ld_s, ld_d)) hipLaunchKernelGGL((copy_k<T, T, TT<T>::VEC>), dim3(grid_for((size_t)M * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)src, ld_s, (T*)dst, ld_d, M, C, accumulate);
            else hipLaunchKernelGGL((copy_k<T, T, 1>), dim3(grid_for((size_t)M * C)), dim3(256), 0, st, (const T*)src, ld_s, (T*)dst, ld_d, M, C, accumulate);
        })
    } else if (dt_in == PN2_F32 && dt_out == PN2_BF16)
        hipLaunchKernelGGL((copy_k<float, bf16_t, 1>), dim3(grid_for((size_t)M * C)), dim3(256), 0, st, (const float*)src, ld_s, (bf16_t*)dst, ld_d, M, C, accumulate);
    else if (dt_in == PN2_BF16 && dt_out == PN2_F32)
        hipLaunchKernelGGL((copy_k<bf16_t, float, 1>), dim3(grid_for((size_t)M * C)), dim3(256), 0, st, (const bf16_t*)src, ld_s, (float*)dst, ld_d, M, C, accumulate);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_mul_bwd(int dt, const void* g, int ld_g, const void* a, int ld_a, const void* b, int ld_b, void* ga, int ld_ga, int acc_a, void* gb, int ld_gb, int acc_b,
                int M, int C, void* stream) {
    if (!g || !a || !b || !ga || !gb || ga == gb) return -1;
    PN2_TOO_MANY((size_t)M * C);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_T(dt, {
        if (vec_ok<T>(C, ld_g, ld_a, ld_b) && vec_ok<T>(C, ld_ga, ld_gb))
            hipLaunchKernelGGL((mul_bwd_k<T, TT<T>::VEC>), dim3(grid_for((size_t)M * C / TT<T>::VEC)), dim3(256), 0, st, (const T*)g, ld_g, (const T*)a, ld_a, (const T*)b, ld_b, (T*)ga, ld_ga, acc_a, (T*)gb, ld_gb, acc_b, M, C);
        else hipLaunchKernelGGL((mul_bwd_k<T, 1>), dim3(grid_for((size_t)M * C)), dim3(256), 0, st, (const T*)g, ld_g, (const T*)a, ld_a, (const T*)b, ld_b, (T*)ga, ld_ga, acc_a, (T*)gb, ld_gb, acc_b, M, C);
    })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_copy_job_blocks(int dt, const pn2_copy_job* j) {
    if (!j || !j->src || !j->dst || j->M < 1 || j->C < 1) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if ((dt != PN2_F32 && dt != PN2_BF16) || j->C % V || j->ld_s % V || j->ld_d % V) return -2;
    const size_t vecs = (size_t)j->M * (j->C / V);
    const size_t nb = (vecs + 1023) / 1024;          // four 16-byte vectors per thread
    return (int)(nb > 4096 ? 4096 : nb);
}
int pn2_copy_multi(int dt, const pn2_copy_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    if (dt == PN2_BF16) hipLaunchKernelGGL(copy_tab_k<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else if (dt == PN2_F32) hipLaunchKernelGGL(copy_tab_k<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_nchw_to_nhwc(int dt_out, const float* x, void* y, int ld_y, int N, int C, int HW, int Cp, void* stream) {
    if (!x || !y) return -1;
    PN2_TOO_MANY((size_t)N * HW);
    hipStream_t st = (hipStream_t)stream;
    if (dt_out == PN2_BF16) hipLaunchKernelGGL(nchw_to_nhwc_k<bf16_t>, dim3(grid_for((size_t)N * HW)), dim3(256), 0, st, x, (bf16_t*)y, ld_y, N, C, HW, Cp);
    else if (dt_out == PN2_F32) hipLaunchKernelGGL(nchw_to_nhwc_k<float>, dim3(grid_for((size_t)N * HW)), dim3(256), 0, st, x, (float*)y, ld_y, N, C, HW, Cp);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_bias_grad(const float* dy, int M, int K, float* db, int accumulate, void* stream) {
    if (!dy || !db) return -1;
    hipLaunchKernelGGL(bias_grad_k, dim3(K), dim3(1024), 0, (hipStream_t)stream, dy, M, K, db, accumulate);
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
