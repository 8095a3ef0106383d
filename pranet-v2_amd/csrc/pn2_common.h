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

#define PN2_F32 0
#define PN2_BF16 1

__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {           // round-to-nearest-even (NaN kept quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct TT;
template <> struct TT<float> {
    static constexpr int VEC = 4;                       // elements per 16-byte vector
    static constexpr int DT = PN2_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x); f[1] = __uint_as_float(v.y); f[2] = __uint_as_float(v.z); f[3] = __uint_as_float(v.w);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    }
};
template <> struct TT<bf16_t> {
    static constexpr int VEC = 8;
    static constexpr int DT = PN2_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
    __device__ static __forceinline__ void unpack(const uint4& v, float* f) {
        f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
        f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
        f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
        f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    __device__ static __forceinline__ uint4 pack(const float* f) {
        uint4 v;
        v.x = (unsigned)f2bf(f[0]) | ((unsigned)f2bf(f[1]) << 16);
        v.y = (unsigned)f2bf(f[2]) | ((unsigned)f2bf(f[3]) << 16);
        v.z = (unsigned)f2bf(f[4]) | ((unsigned)f2bf(f[5]) << 16);
        v.w = (unsigned)f2bf(f[6]) | ((unsigned)f2bf(f[7]) << 16);
        return v;
    }
};

// physical channel p of a group-padded layout (groups of `gw` logical channels stored in `gwp` slots)
// -> logical channel, or -1 for a pad slot.  Identity layout: gw == gwp.
__device__ __forceinline__ int phys2log(int p, int gw, int gwp, int C) {
    int g = p / gwp, o = p - g * gwp;
    int c = g * gw + o;
    return (o < gw && c < C) ? c : -1;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

#define PN2_CHECK_LAUNCH() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
