// pn2_common.h — shared device helpers for the PraNet-V2 gfx950 kernels.
// All activations are NHWC "views": base pointer (already offset to the first channel of the
// view), row = one pixel, `ld` = channel stride between consecutive pixels (elements).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;   // storage type; arithmetic always in f32
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) double f64x4_t;

#define PN2_F32 0
#define PN2_BF16 1
#define PN2_F32F 2          // fp32 storage like PN2_F32, contractions on the f32 matrix pipe (v_mfma_f32_16x16x4_f32, 157 TF/s) instead of the f64 one - conv GEMM / wgrad entry points only

// storage type of PN2_F32F: a float under another name, so that the conv kernels (templated on the storage type) pick another MFMA form for it
struct f32f_t { float v; };

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {           // round-to-nearest-even, gfx950 hardware convert
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %1" : "=v"(r) : "v"(f));
    return (bf16_t)r;
}

template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int VEC = 4;                       // elements per 16-byte vector
    static constexpr int DT = PN2_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float round(float v) { return v; }          // the value as the storage type holds it
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct TT<f32f_t> : TT<float> {
    static constexpr int DT = PN2_F32F;
    __device__ static __forceinline__ float ld(const f32f_t* p) { return p->v; }
    __device__ static __forceinline__ void st(f32f_t* p, float v) { p->v = v; }
};
template <> struct TT<bf16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = PN2_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ float round(float v) { return bf2f(f2bf(v)); }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    // gfx950 hardware convert: two f32 -> packed bf16x2, round-to-nearest-even (one VALU op per pair instead of ~12)
    __device__ static __forceinline__ unsigned cvt2(float lo, float hi) {
        unsigned r;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
        return r;
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(cvt2(f[0], f[1]), cvt2(f[2], f[3]), cvt2(f[4], f[5]), cvt2(f[6], f[7]));
    }
};

// physical channel p of a group-padded layout (groups of `gw` logical channels stored in `gwp` slots)
// -> logical channel, or -1 for a pad slot.  Identity layout: gw == gwp.
__device__ __forceinline__ int phys2log(int p, int gw, int gwp, int C) {
    int g = p / gwp, o = p - g * gwp;
    int c = g * gw + o;
    return (o < gw && c < C) ? c : -1;
}

// bijective XCD-aware remap: hardware places block b on XCD b%8; give every XCD a contiguous range of logical tiles / pixel chunks so that
// neighbours (shared A rows and weight panels of the GEMMs, halo rows of the depth-wise convs) meet in one L2 instead of being fetched by several.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// flat 64-bit element index -> quotient / remainder by an int divisor, with 32-bit arithmetic whenever the index fits (a 64-bit division is ~100
// instructions on this part, and pixel decodes chain three or four of them)
__device__ __forceinline__ size_t divmod_idx(size_t idx, int d, int& rem) {
    if (idx >> 32) { const size_t q = idx / (size_t)d; rem = (int)(idx - q * (size_t)d); return q; }
    const unsigned i = (unsigned)idx, q = i / (unsigned)d;
    rem = (int)(i - q * (unsigned)d);
    return q;
}

// job lookup for the table-driven launches: largest j with bstart[j] <= b   (bstart has njobs + 1 entries)
__device__ __forceinline__ int find_job(const int* __restrict__ bstart, int njobs, int b) {
    int lo = 0, hi = njobs;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (bstart[mid] <= b) lo = mid; else hi = mid; }
    return lo;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// PyTorch upsample_bilinear2d index math (area_pixel_compute_source_index), shared by pn2_spatial.hip and the fused DSRA tail:
//   align_corners: src = r*dst ; else src = max(r*(dst+0.5)-0.5, 0) ; i0 = (int)src ; i1 = i0 + (i0 < In-1) ; l1 = src - i0.
__device__ __forceinline__ void bl_src(int o, float r, int ac, int In, int& i0, int& i1, float& l0, float& l1) {
    float s = ac ? r * (float)o : fmaxf(r * ((float)o + 0.5f) - 0.5f, 0.f);
    i0 = (int)s; if (i0 > In - 1) i0 = In - 1;
    i1 = i0 + (i0 < In - 1 ? 1 : 0);
    l1 = s - (float)i0; l0 = 1.f - l1;
}

// candidate output range whose 2-tap footprint can touch input index i (monotone src): conservative +-1, exact test in loop
__device__ __forceinline__ void bl_range(int i, float r, int ac, int On, int& lo, int& hi) {
    float a, b;
    if (ac) { if (r <= 0.f) { lo = 0; hi = On - 1; return; } a = ((float)i - 1.f) / r; b = ((float)i + 1.f) / r; }
    else { a = ((float)i - 0.5f) / r - 0.5f; b = ((float)i + 1.5f) / r - 0.5f; }
    lo = (int)floorf(a) - 1; hi = (int)ceilf(b) + 1;
    if (lo < 0) lo = 0; if (hi > On - 1) hi = On - 1;
}

#define PN2_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
