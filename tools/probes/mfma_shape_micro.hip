// Micro: the LDS-fed inner loop of a 64 x 64 wave tile with v_mfma_f32_16x16x32_bf16 (what every bf16 kernel of csrc/ issues) against v_mfma_f32_32x32x16_bf16
// (VERDICT r5 item 5), operands resident in LDS (K-contiguous 144-byte rows, the register-staged conv kernel's image; fragments by ds_read_b128), four waves per
// workgroup on a 128 x 128 tile, 1 / 2 / 3 workgroups per CU.  No global traffic inside the timed loop: what is measured is issue + LDS + matrix pipe.
// Per 32-deep K step and wave: 16x16x32 - 8 ds_read_b128 + 16 MFMAs (8 passes each); 32x32x16 - 8 ds_read_b128 + 8 MFMAs (16 passes each): the fragment reads per
// flop are set by the WAVE tile (64 + 64 rows per K), not by the MFMA shape - the two differ in instruction count and accumulator layout only.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_micro tools/probes/mfma_shape_micro.hip && /tmp/mfma_shape_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int RS = 144;          // padded row stride (bytes) of a 64-deep (128-byte) K slab

static unsigned short f2bf_h(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

// A [128][64] bf16, B [128][64] bf16 (K-contiguous), C [128][128] fp32 = A * B^T summed `iters` times over the same slab (x iters in the check)
template <int SHAPE>
__global__ __launch_bounds__(256) void k_loop(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B, float* __restrict__ C, int iters, int store) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem; char* Bs = smem + 128 * RS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
    for (int i = tid; i < 128 * 8; i += 256) {
        const int r = i >> 3, c = i & 7;
        *reinterpret_cast<uint4*>(As + r * RS + c * 16) = *reinterpret_cast<const uint4*>(A + r * 64 + c * 8);
        *reinterpret_cast<uint4*>(Bs + r * RS + c * 16) = *reinterpret_cast<const uint4*>(B + r * 64 + c * 8);
    }
    __syncthreads();
    if constexpr (SHAPE == 16) {
        const int l15 = lane & 15, g = lane >> 4;
        f32x4 acc[4][4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + i * 16 + l15) * RS + ks * 64 + g * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + j * 16 + l15) * RS + ks * 64 + g * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
        if (store)
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r)
                C[(size_t)(wm * 64 + i * 16 + g * 4 + r) * 128 + wn * 64 + j * 16 + l15] = acc[i][j][r];
    } else {
        const int l31 = lane & 31, h = lane >> 5;
        f32x16 acc[2][2];
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {          // 16-deep sub-steps: lane (h, l31) holds k = 8h .. 8h + 7 of row l31
                bf16x8 a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const bf16x8*>(As + (wm * 64 + i * 32 + l31) * RS + ks * 32 + h * 16);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(Bs + (wn * 64 + j * 32 + l31) * RS + ks * 32 + h * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            asm volatile("" ::: "memory");
        }
        if (store)
            for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r)
                C[(size_t)(wm * 64 + i * 32 + (r >> 2) * 8 + h * 4 + (r & 3)) * 128 + wn * 64 + j * 32 + l31] = acc[i][j][r];
    }
}

template <int SHAPE> static void run(const unsigned short* dA, const unsigned short* dB, float* dC, const std::vector<float>& ref, int wg_per_cu) {
    const int lds = wg_per_cu == 1 ? 150 * 1024 : (wg_per_cu == 2 ? 76 * 1024 : 50 * 1024);          // the LDS request sets how many workgroups share a CU
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_loop<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    std::vector<float> C(128 * 128);
    k_loop<SHAPE><<<1, 256, lds>>>(dA, dB, dC, 3, 1);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double err = 0; for (size_t i = 0; i < C.size(); ++i) err = fmax(err, fabs(C[i] - 3.0 * ref[i]));
    const int iters = 4000, grid = 256 * wg_per_cu * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_loop<SHAPE><<<grid, 256, lds>>>(dA, dB, dC, 200, 0);
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0); k_loop<SHAPE><<<grid, 256, lds>>>(dA, dB, dC, iters, 0); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = fminf(best, ms);
    }
    const double fl = 2.0 * 128 * 128 * 64 * (double)iters * grid;
    printf("mfma %s  %d workgroup(s) per CU: %8.3f ms  %7.1f TFLOP/s  (max |C - ref| %.3g)\n", SHAPE == 16 ? "16x16x32" : "32x32x16", wg_per_cu, best, fl / best / 1e9, err);
}

int main() {
    std::vector<unsigned short> A(128 * 64), B(128 * 64);
    std::vector<float> Af(128 * 64), Bf(128 * 64), ref(128 * 128);
    srand(1);
    for (size_t i = 0; i < A.size(); ++i) { Af[i] = (float)(rand() % 17 - 8) * 0.125f; Bf[i] = (float)(rand() % 13 - 6) * 0.25f; A[i] = f2bf_h(Af[i]); B[i] = f2bf_h(Bf[i]); }
    for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) { double s = 0; for (int k = 0; k < 64; ++k) s += (double)Af[i * 64 + k] * Bf[j * 64 + k]; ref[i * 128 + j] = (float)s; }
    unsigned short *dA, *dB; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 128 * 128 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    for (int w = 1; w <= 3; ++w) { run<16>(dA, dB, dC, ref, w); run<32>(dA, dB, dC, ref, w); }
    return 0;
}
