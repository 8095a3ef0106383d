// pn2_dw.h - packed channel vectors of the depth-wise "sliding window" walks (pn2_vit.hip: 3x3 + GELU of the PVTv2 Mlp; pn2_emcad.hip: K x K of MSDC).
#pragma once
#include "pn2_common.h"

namespace {

// VT channels of one pixel as raw 32-bit words (bf16: VT/2 words, fp32: VT words); loads / stores are 16, 8 or 4 bytes wide
template <typename T, int VT> struct DwVec {
    static constexpr int NW = VT * (int)sizeof(T) / 4;          // 32-bit words per packed vector
    unsigned w[NW];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < NW; ++i) w[i] = 0u;
    }
    __device__ __forceinline__ void load(const T* p) {
        if constexpr (NW == 4) { const uint4 v = *reinterpret_cast<const uint4*>(p); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
        else if constexpr (NW == 2) { const uint2 v = *reinterpret_cast<const uint2*>(p); w[0] = v.x; w[1] = v.y; }
        else { w[0] = *reinterpret_cast<const unsigned*>(p); }
    }
    __device__ __forceinline__ void store(T* p) const {
        if constexpr (NW == 4) *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
        else if constexpr (NW == 2) *reinterpret_cast<uint2*>(p) = make_uint2(w[0], w[1]);
        else *reinterpret_cast<unsigned*>(p) = w[0];
    }
    __device__ __forceinline__ void unpack(float* f) const {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int i = 0; i < NW; ++i) f[i] = __uint_as_float(w[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
        }
    }
    __device__ __forceinline__ void pack(const float* f) {
        if constexpr (sizeof(T) == 4) {
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = __float_as_uint(f[i]);
        } else {
#pragma unroll
            for (int i = 0; i < NW; ++i) w[i] = TT<bf16_t>::cvt2(f[2 * i], f[2 * i + 1]);
        }
    }
};


}  // namespace
