// pn2_conv.hip — implicit-GEMM convolutions on gfx950 matrix cores.
//
// Replaces what the reference dispatches for nn.Conv2d forward/backward on the hot path
// (call sites: /root/reference/binary_seg/lib/Res2Net_v1b.py:32,44,49,102-108,133 and
// /root/reference/binary_seg/lib/pranet.py:34-36,52-73,94-104,303-325).
//
//   conv_gather_gemm : out[m][co] = sum_{tap,ci} gather(in, m, tap, ci) * Wp[co][tap*Cin_p+ci]
//       - forward mode   : gather reads x at (oy*stride - pad + r*dil, ...)
//       - transposed mode: gather reads dy at ((iy + pad - r*dil)/stride, ...)  == dgrad
//       - epilogue: optional per-channel sum / sum-of-squares partials (fused BN batch stats),
//         optional accumulate into the destination (gradient accumulation), LDS-staged 16-byte stores.
//   conv_wgrad       : dWp[co][k] = sum_m dy[m][co] * gather(x, m, k)   (pixels are the contraction
//       index; both operands are staged pixel-major and read with ds_read_b64_tr_b16 for bf16).
//
// Tiling is for 64-wide wavefronts: 256 threads = 4 waves, each wave owns a (16*MT)x(16*NT) block of
// v_mfma_f32_16x16x32_bf16 (or v_mfma_f32_16x16x4_f32 for exact-fp32 parity runs) accumulators.
#include "pn2_common.h"
#include "../../include/pn2.h"

namespace {

constexpr int BM = 128;        // output pixels per workgroup (forward / dgrad)
constexpr int ROWB = 64;       // bytes of K per LDS row per step (32 bf16 or 16 f32)
constexpr int RS = ROWB + 16;  // padded LDS row stride (bytes)

template <typename T> struct MMA;
template <> struct MMA<bf16_t> {
    static constexpr int BK = 32;
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};
template <> struct MMA<float> {
    static constexpr int BK = 16;
    // lane (g = lane>>4) holds k = 4g..4g+3 of this 16-deep step; MFMA j contracts {j, 4+j, 8+j, 12+j}
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
};

// bijective XCD-aware remap: hardware places block b on XCD b%8; give every XCD a contiguous
// range of logical tiles so neighbouring tiles (which share A rows / weight panels) share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

struct GatherGeom {
    int H, W, OH, OW, KH, KW, stride, sshift, pad_h, pad_w, dil_h, dil_w, transposed;
};

// resolve tap (r,s) of output pixel (oy0, ox0 precomputed) to an input pixel; returns false if padding
__device__ __forceinline__ bool tap_pixel(const GatherGeom& g, int iy0, int ix0, int r, int s, int& iy, int& ix) {
    if (!g.transposed) {
        iy = iy0 + r * g.dil_h; ix = ix0 + s * g.dil_w;
        return (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
    }
    const int ty = iy0 - r * g.dil_h, tx = ix0 - s * g.dil_w;
    const int msk = g.stride - 1;
    iy = ty >> g.sshift; ix = tx >> g.sshift;
    return ty >= 0 && tx >= 0 && !(ty & msk) && !(tx & msk) && iy < g.H && ix < g.W;
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad gather-GEMM
// ------------------------------------------------------------------------------------------------
template <typename T, int BN, int WM, int WN, bool PW>
__global__ __launch_bounds__(256) void conv_gather_gemm(const T* __restrict__ in, const T* __restrict__ wp, T* __restrict__ out,
                                                        float* __restrict__ psum, float* __restrict__ psq, pn2_conv_desc d) {
    constexpr int VEC = TT<T>::VEC, BK = MMA<T>::BK;
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int STAGE = (BM + BN) * RS;
    constexpr int CRS = BN * (int)sizeof(T) + 16;
    constexpr int NA = BM / 64, NB = (BN + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN, l15 = lane & 15, g = lane >> 4;
    const int M = d.N * d.OH * d.OW;
    const int nbn = (d.Cout + BN - 1) / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int bn = bid % nbn, bm = bid / nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int taps = d.KH * d.KW;
    const int ktot = taps * d.Cin_p;
    const int ksteps = (ktot + BK - 1) / BK;

    GatherGeom gg;
    gg.H = d.H; gg.W = d.W; gg.OH = d.OH; gg.OW = d.OW; gg.KH = d.KH; gg.KW = d.KW; gg.stride = d.stride;
    gg.sshift = d.stride == 2 ? 1 : 0; gg.pad_h = d.pad_h; gg.pad_w = d.pad_w; gg.dil_h = d.dil_h; gg.dil_w = d.dil_w;
    gg.transposed = d.transposed;

    // ---- per-thread A rows
    const int kv = tid & 3;
    int rbase[NA], riy0[NA], rix0[NA]; bool rok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + (tid >> 2) + 64 * i;
        rok[i] = m < M;
        const int mm = rok[i] ? m : 0;
        if (PW) { rbase[i] = mm; riy0[i] = 0; rix0[i] = 0; }
        else {
            const int hw = d.OH * d.OW;
            const int n = mm / hw, rem = mm - n * hw;
            const int oy = rem / d.OW, ox = rem - oy * d.OW;
            rbase[i] = n * d.H * d.W;
            if (!d.transposed) { riy0[i] = oy * d.stride - d.pad_h; rix0[i] = ox * d.stride - d.pad_w; }
            else { riy0[i] = oy + d.pad_h; rix0[i] = ox + d.pad_w; }
        }
    }
    int ci = kv * VEC, tap = 0;
    if (!PW) { while (ci >= d.Cin_p) { ci -= d.Cin_p; ++tap; } }
    const T* bptr = wp + (size_t)(n0 + (tid >> 2)) * d.Kp + kv * VEC;

    uint4 ra[NA], rb[NB];
    auto gload = [&](int step) {
        if (PW) {
            const int k = step * BK + kv * VEC;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ra[i] = make_uint4(0, 0, 0, 0);
                if (rok[i] && k < d.Cin_p) ra[i] = *reinterpret_cast<const uint4*>(in + (size_t)rbase[i] * d.ld_in + k);
            }
        } else {
            const int r = tap / d.KW, s = tap - r * d.KW;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                ra[i] = make_uint4(0, 0, 0, 0);
                int iy, ix;
                if (rok[i] && tap < taps && tap_pixel(gg, riy0[i], rix0[i], r, s, iy, ix))
                    ra[i] = *reinterpret_cast<const uint4*>(in + (size_t)(rbase[i] + iy * d.W + ix) * d.ld_in + ci);
            }
            ci += BK;
            while (ci >= d.Cin_p) { ci -= d.Cin_p; ++tap; }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (BN >= 64 || tid < 128) rb[i] = *reinterpret_cast<const uint4*>(bptr + (size_t)(64 * i) * d.Kp + (size_t)step * BK);
        }
    };
    auto lstore = [&](int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + BM * RS;
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<uint4*>(As + ((tid >> 2) + 64 * i) * RS + kv * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i)
            if (BN >= 64 || tid < 128) *reinterpret_cast<uint4*>(Bs + ((tid >> 2) + 64 * i) * RS + kv * 16) = rb[i];
    };

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    gload(0);
    lstore(0);
    __syncthreads();
    for (int step = 0; step < ksteps; ++step) {
        const int cur = step & 1;
        if (step + 1 < ksteps) gload(step + 1);
        const char* As = smem + cur * STAGE;
        const char* Bs = As + BM * RS;
        uint4 a[MT], b[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const uint4*>(As + (wm * WTM + i * 16 + l15) * RS + g * 16);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const uint4*>(Bs + (wn * WTN + j * 16 + l15) * RS + g * 16);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) MMA<T>::run(acc[i][j], a[i], b[j]);
        if (step + 1 < ksteps) lstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: stats partials + LDS-staged coalesced store
    char* Cs = smem;
    float* red = reinterpret_cast<float*>(smem + BM * CRS);   // [2][WM][BN]
    if (d.flags & PN2_CONV_STATS) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; s += v; q += v * v; }
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            if (g == 0) {
                red[wm * BN + wn * WTN + j * 16 + l15] = s;
                red[(WM + wm) * BN + wn * WTN + j * 16 + l15] = q;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * WTM + i * 16 + g * 4 + r, col = wn * WTN + j * 16 + l15;
                TT<T>::st(reinterpret_cast<T*>(Cs + row * CRS) + col, acc[i][j][r]);
            }
    __syncthreads();
    if ((d.flags & PN2_CONV_STATS) && tid < BN) {
        const int col = n0 + tid;
        if (col < d.Cout) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { s += red[w * BN + tid]; q += red[(WM + w) * BN + tid]; }
            psum[(size_t)bm * d.Cout + col] = s;
            psq[(size_t)bm * d.Cout + col] = q;
        }
    }
    constexpr int VPR = BN / VEC;
    const bool vec_ok = (d.Cout % VEC == 0) && (d.ld_out % VEC == 0);
    const bool accum = d.flags & PN2_CONV_ACCUM;
    for (int idx = tid; idx < BM * VPR; idx += 256) {
        const int row = idx / VPR, cv = idx - row * VPR;
        const int m = m0 + row, col = n0 + cv * VEC;
        if (m >= M || col >= d.Cout) continue;
        uint4 v = *reinterpret_cast<const uint4*>(Cs + row * CRS + cv * 16);
        T* dst = out + (size_t)m * d.ld_out + col;
        if (vec_ok) {
            if (accum) {
                float x[VEC], y[VEC];
                TT<T>::unpack(v, x);
                TT<T>::unpack(*reinterpret_cast<const uint4*>(dst), y);
#pragma unroll
                for (int e = 0; e < VEC; ++e) x[e] += y[e];
                v = TT<T>::pack(x);
            }
            *reinterpret_cast<uint4*>(dst) = v;
        } else {
            float x[VEC];
            TT<T>::unpack(v, x);
            for (int e = 0; e < VEC && col + e < d.Cout; ++e) TT<T>::st(dst + e, accum ? x[e] + TT<T>::ld(dst + e) : x[e]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: slab[s][co][k] = sum over this split's pixels of dy[m][co] * gather(x, m, k)
// ------------------------------------------------------------------------------------------------
constexpr int WGP = 32;   // pixels (contraction) per step

template <typename T> struct WG;
template <> struct WG<bf16_t> {
    static constexpr int PAD = 32;   // row stride == 32 B (mod 256 B): conflict-free ds_read_b64_tr_b16
    // A/B fragment of one 16-wide channel block: k-slot (g, j) <-> pixel (j>>2)*16 + g*4 + (j&3)
    __device__ static __forceinline__ uint4 frag(const char* tile, int rs, int chan0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const char* p = tile + (g * 4 + (i >> 2)) * rs + (chan0 + (i & 3) * 4) * 2;
        s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p));
        s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)(p + 16 * rs));
        uint2 lo = __builtin_bit_cast(uint2, v0), hi = __builtin_bit_cast(uint2, v1);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) { MMA<bf16_t>::run(acc, a, b); }
    static constexpr int NFRAG = 1;
};
template <> struct WG<float> {
    static constexpr int PAD = 64;
    static constexpr int NFRAG = 2;  // two uint4 = 8 pixels-slots per lane per 32-pixel step
};

template <typename T, int BMC, int BNK, int WM, int WN, bool PW>
__global__ __launch_bounds__(256) void conv_wgrad(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ slab, pn2_wgrad_desc d) {
    constexpr int VEC = TT<T>::VEC;
    constexpr int WTM = BMC / WM, WTN = BNK / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int RSY = BMC * (int)sizeof(T) + WG<T>::PAD, RSX = BNK * (int)sizeof(T) + WG<T>::PAD;
    constexpr int STAGE = WGP * (RSY + RSX);
    constexpr int VPRY = BMC / VEC, VPRX = BNK / VEC;
    constexpr int NY = (WGP * VPRY + 255) / 256, NX = (WGP * VPRX + 255) / 256;
    constexpr int RSTEPY = 256 / VPRY > WGP ? WGP : 256 / VPRY;   // rows covered per pass
    constexpr int RSTEPX = 256 / VPRX;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN, l15 = lane & 15, g = lane >> 4;
    const int M = d.N * d.OH * d.OW;
    const int tk = d.Kp / BNK;
    const int bco = blockIdx.x / tk, bk = blockIdx.x % tk;
    const int co0 = bco * BMC, k0 = bk * BNK;
    const int total_steps = (M + WGP - 1) / WGP;
    const int spb = (total_steps + gridDim.y - 1) / gridDim.y;
    const int s_begin = blockIdx.y * spb;
    int s_end = s_begin + spb; if (s_end > total_steps) s_end = total_steps;
    const int taps = d.KH * d.KW;

    // dy loader: rows yrow + i*RSTEPY, channel vector ycv
    const int ycv = tid % VPRY, yrow = tid / VPRY;
    const bool y_active = (WGP * VPRY >= 256) || (tid < WGP * VPRY);
    const bool yc_ok = (co0 + ycv * VEC) < d.Cout_p;
    // x loader: fixed k-vector per thread
    const int xkv = tid % VPRX, xrow = tid / VPRX;
    const int kk = k0 + xkv * VEC;
    int xtap = 0, xci = kk;
    if (!PW) { xtap = kk / d.Cin_p; xci = kk - xtap * d.Cin_p; }
    const bool xk_ok = PW ? (kk < d.Cin_p) : (xtap < taps);
    const int xr = PW ? 0 : xtap / d.KW, xs = PW ? 0 : xtap - (xtap / d.KW) * d.KW;
    int pn[NX], poy[NX], pox[NX];     // pixel state per owned row (generic path)
    if (!PW) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            int m = s_begin * WGP + xrow + i * RSTEPX;
            if (m >= M) m = M - 1;
            const int hw = d.OH * d.OW;
            const int n = m / hw, rem = m - n * hw;
            pn[i] = n; poy[i] = rem / d.OW; pox[i] = rem - poy[i] * d.OW;
        }
    }

    uint4 ry[NY], rx[NX];
    auto gload = [&](int step) {
        const int mb = step * WGP;
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int m = mb + yrow + i * RSTEPY;
            ry[i] = make_uint4(0, 0, 0, 0);
            if (y_active && yc_ok && m < M) ry[i] = *reinterpret_cast<const uint4*>(dy + (size_t)m * d.ld_dy + co0 + ycv * VEC);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int m = mb + xrow + i * RSTEPX;
            rx[i] = make_uint4(0, 0, 0, 0);
            if (PW) {
                if (xk_ok && m < M) rx[i] = *reinterpret_cast<const uint4*>(x + (size_t)m * d.ld_x + kk);
            } else {
                const int iy = poy[i] * d.stride - d.pad_h + xr * d.dil_h, ix = pox[i] * d.stride - d.pad_w + xs * d.dil_w;
                if (xk_ok && m < M && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W)
                    rx[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(pn[i] * d.H + iy) * d.W + ix) * d.ld_x + xci);
                pox[i] += WGP;
                while (pox[i] >= d.OW) { pox[i] -= d.OW; ++poy[i]; }
                while (poy[i] >= d.OH) { poy[i] -= d.OH; ++pn[i]; }
            }
        }
    };
    auto lstore = [&](int stage) {
        char* Ys = smem + stage * STAGE;
        char* Xs = Ys + WGP * RSY;
#pragma unroll
        for (int i = 0; i < NY; ++i)
            if (y_active) *reinterpret_cast<uint4*>(Ys + (yrow + i * RSTEPY) * RSY + ycv * 16) = ry[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) *reinterpret_cast<uint4*>(Xs + (xrow + i * RSTEPX) * RSX + xkv * 16) = rx[i];
    };

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (s_begin < s_end) {
        gload(s_begin);
        lstore(0);
        __syncthreads();
        for (int step = s_begin; step < s_end; ++step) {
            const int cur = (step - s_begin) & 1;
            if (step + 1 < s_end) gload(step + 1);
            const char* Ys = smem + cur * STAGE;
            const char* Xs = Ys + WGP * RSY;
            if constexpr (sizeof(T) == 2) {
                uint4 a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = WG<bf16_t>::frag(Ys, RSY, wm * WTM + i * 16, lane);
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = WG<bf16_t>::frag(Xs, RSX, wn * WTN + j * 16, lane);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) MMA<bf16_t>::run(acc[i][j], a[i], b[j]);
            } else {
#pragma unroll
                for (int q = 0; q < WGP / 4; ++q) {       // 4 pixels per v_mfma_f32_16x16x4_f32
                    float a[MT], b[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float*>(Ys + (q * 4 + g) * RSY + (wm * WTM + i * 16 + l15) * 4);
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float*>(Xs + (q * 4 + g) * RSX + (wn * WTN + j * 16 + l15) * 4);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
                }
            }
            if (step + 1 < s_end) lstore(cur ^ 1);
            __syncthreads();
        }
    }
    float* dst = slab + ((size_t)blockIdx.y * d.Rp + co0) * d.Kp + k0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[(size_t)(wm * WTM + i * 16 + g * 4 + r) * d.Kp + wn * WTN + j * 16 + l15] = acc[i][j][r];
}

// ------------------------------------------------------------------------------------------------
// weight packing (OIHW fp32 master -> K-contiguous panels in the compute dtype) and grad unpacking
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_weight(const float* __restrict__ w, T* __restrict__ wp, pn2_pack_desc p) {
    // forward : wp[phys_out(co)][tap*Cin_p + phys_in(ci)] ; transposed: wp[phys_in(ci)][tap*Cout_p + phys_out(co)]
    const size_t total = (size_t)p.Rp * p.Kp;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(idx / p.Kp), k = (int)(idx - (size_t)row * p.Kp);
        const int taps = p.KH * p.KW;
        const int cg = p.transposed ? p.Cout_p : p.Cin_p;
        const int tap = k / cg, c = k - tap * cg;
        float v = 0.f;
        if (tap < taps) {
            int co, ci;
            if (!p.transposed) {
                co = row < p.Cout_p ? phys2log(row, p.gw_out, p.gwp_out, p.Cout) : -1;
                ci = phys2log(c, p.gw_in, p.gwp_in, p.Cin);
            } else {
                ci = row < p.Cin_p ? phys2log(row, p.gw_in, p.gwp_in, p.Cin) : -1;
                co = phys2log(c, p.gw_out, p.gwp_out, p.Cout);
            }
            if (co >= 0 && ci >= 0) v = w[((size_t)co * p.Cin + ci) * taps + tap];
        }
        TT<T>::st(wp + idx, v);
    }
}

__global__ void wgrad_reduce_unpack(const float* __restrict__ slab, float* __restrict__ gw, pn2_pack_desc p, int nsplit, int accumulate) {
    const int taps = p.KH * p.KW;
    const size_t total = (size_t)p.Cout * p.Cin * taps;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int tap = (int)(idx % taps);
        const int ci = (int)((idx / taps) % p.Cin), co = (int)(idx / ((size_t)taps * p.Cin));
        const int prow = (co / p.gw_out) * p.gwp_out + co % p.gw_out;
        const int pcol = tap * p.Cin_p + (ci / p.gw_in) * p.gwp_in + ci % p.gw_in;
        const float* s = slab + (size_t)prow * p.Kp + pcol;
        float v = 0.f;
        for (int i = 0; i < nsplit; ++i) v += s[(size_t)i * p.Rp * p.Kp];
        gw[idx] = accumulate ? gw[idx] + v : v;
    }
}

template <typename T, int BN, int WM, int WN>
int launch_gemm(const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc& d, hipStream_t st) {
    const int M = d.N * d.OH * d.OW;
    const int grid = ((M + BM - 1) / BM) * ((d.Cout + BN - 1) / BN);
    constexpr int main_b = 2 * (BM + BN) * RS, epi_b = BM * (BN * (int)sizeof(T) + 16) + 2 * WM * BN * 4;
    constexpr int lds = main_b > epi_b ? main_b : epi_b;
    const bool pw = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0;
    if (pw) hipLaunchKernelGGL((conv_gather_gemm<T, BN, WM, WN, true>), dim3(grid), dim3(256), lds, st, (const T*)in, (const T*)wp, (T*)out, psum, psq, d);
    else hipLaunchKernelGGL((conv_gather_gemm<T, BN, WM, WN, false>), dim3(grid), dim3(256), lds, st, (const T*)in, (const T*)wp, (T*)out, psum, psq, d);
    PN2_CHECK_LAUNCH();
    return 0;
}

template <typename T, int BMC, int WM, int WN>
int launch_wgrad(const void* dy, const void* x, float* slab, const pn2_wgrad_desc& d, int nsplit, hipStream_t st) {
    constexpr int BNK = 128;
    constexpr int lds = 2 * WGP * (BMC * (int)sizeof(T) + WG<T>::PAD + BNK * (int)sizeof(T) + WG<T>::PAD);
    dim3 grid((d.Rp / BMC) * (d.Kp / BNK), nsplit);
    const bool pw = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0;
    if (pw) hipLaunchKernelGGL((conv_wgrad<T, BMC, BNK, WM, WN, true>), grid, dim3(256), lds, st, (const T*)dy, (const T*)x, slab, d);
    else hipLaunchKernelGGL((conv_wgrad<T, BMC, BNK, WM, WN, false>), grid, dim3(256), lds, st, (const T*)dy, (const T*)x, slab, d);
    PN2_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int gemm_dispatch(const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc& d, hipStream_t st) {
    int bn = pn2_conv_tile_n(d.Cout);
    if (sizeof(T) == 4 && bn == 128) bn = 64;   // fp32 128x128 epilogue tile would exceed 64 KiB of LDS
    if (bn == 128) {
        if constexpr (sizeof(T) == 2) return launch_gemm<T, 128, 2, 2>(in, wp, out, psum, psq, d, st);
    }
    if (bn == 64) return launch_gemm<T, 64, 2, 2>(in, wp, out, psum, psq, d, st);
    return launch_gemm<T, 32, 4, 1>(in, wp, out, psum, psq, d, st);
}

template <typename T>
int wgrad_dispatch(const void* dy, const void* x, float* slab, const pn2_wgrad_desc& d, int nsplit, hipStream_t st) {
    const int bmc = pn2_wgrad_tile_co(d.Cout_p);
    if (d.Rp % bmc || d.Kp % 128) return -2;
    if (bmc == 128) return launch_wgrad<T, 128, 2, 2>(dy, x, slab, d, nsplit, st);
    if (bmc == 64) return launch_wgrad<T, 64, 2, 2>(dy, x, slab, d, nsplit, st);
    return launch_wgrad<T, 32, 1, 4>(dy, x, slab, d, nsplit, st);
}

}  // namespace

extern "C" {

int pn2_conv_tile_n(int cout) {
    // smallest padding waste among the built N tiles; ties go to the wider tile
    int best = 32, waste = ((cout + 31) / 32) * 32 - cout;
    const int w64 = ((cout + 63) / 64) * 64 - cout, w128 = ((cout + 127) / 128) * 128 - cout;
    if (w64 <= waste) { best = 64; waste = w64; }
    if (w128 <= waste) { best = 128; }
    return best;
}

int pn2_wgrad_tile_co(int cout_p) { return cout_p > 64 ? 128 : (cout_p > 32 ? 64 : 32); }

int pn2_conv_stat_blocks(int m) { return (m + BM - 1) / BM; }

int pn2_conv_gemm(int dtype, const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc* d, void* stream) {
    if (!in || !wp || !out || !d) return -1;
    if (d->Cin_p % 8 || d->ld_in % 8 || d->Kp % 128 || (d->stride != 1 && d->stride != 2)) return -2;
    if ((d->flags & PN2_CONV_STATS) && (!psum || !psq)) return -1;
    if (dtype == PN2_BF16) return gemm_dispatch<bf16_t>(in, wp, out, psum, psq, *d, (hipStream_t)stream);
    if (dtype == PN2_F32) return gemm_dispatch<float>(in, wp, out, psum, psq, *d, (hipStream_t)stream);
    return -3;
}

int pn2_conv_wgrad(int dtype, const void* dy, const void* x, float* slab, const pn2_wgrad_desc* d, int nsplit, void* stream) {
    if (!dy || !x || !slab || !d || nsplit < 1) return -1;
    if (d->Cin_p % 8 || d->ld_x % 8 || d->ld_dy % 8 || d->Cout_p % 8) return -2;
    if (dtype == PN2_BF16) return wgrad_dispatch<bf16_t>(dy, x, slab, *d, nsplit, (hipStream_t)stream);
    if (dtype == PN2_F32) return wgrad_dispatch<float>(dy, x, slab, *d, nsplit, (hipStream_t)stream);
    return -3;
}

int pn2_pack_weight(int dtype, const float* w, void* wp, const pn2_pack_desc* p, void* stream) {
    if (!w || !wp || !p) return -1;
    const size_t total = (size_t)p->Rp * p->Kp;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == PN2_BF16) hipLaunchKernelGGL(pack_weight<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wp, *p);
    else if (dtype == PN2_F32) hipLaunchKernelGGL(pack_weight<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, (float*)wp, *p);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_wgrad_reduce(const float* slab, float* gw, const pn2_pack_desc* p, int nsplit, int accumulate, void* stream) {
    if (!slab || !gw || !p) return -1;
    const size_t total = (size_t)p->Cout * p->Cin * p->KH * p->KW;
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    hipLaunchKernelGGL(wgrad_reduce_unpack, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, gw, *p, nsplit, accumulate);
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
