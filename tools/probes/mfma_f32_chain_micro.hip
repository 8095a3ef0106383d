// v_mfma_f32_16x16x4_f32 issue-rate micro: (A) plain accumulation, (B) the product's form - chains of 4 MFMAs from C = 0 met by a round-to-nearest add (MMA<f32f_t>::run_block).
// Register operands only (no LDS, no global traffic in the loop); 8 independent 16 x 16 blocks per wave; 4 waves per workgroup; 1..4 workgroups per CU.
// hipcc --offload-arch=gfx950 -O3 -o mfma_f32_chain_micro mfma_f32_chain_micro.hip && ./mfma_f32_chain_micro
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE> __global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    f32x4 acc[8], prev[8];
    for (int i = 0; i < 8; ++i) prev[i] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a[4], b[8];
    for (int i = 0; i < 4; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int i = 0; i < 8; ++i) b[i] = seed - threadIdx.x * 1e-3f - i;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[i], acc[i], 0, 0, 0);
        } else if (MODE == 2) {
            f32x4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[i], f32x4{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk & 3], b[i], t[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += t[i];
        } else if (MODE == 3) {
            // chains of 4; the adds of a group are issued between the MFMAs of the NEXT group (one VALU quad per MFMA)
            f32x4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[i], f32x4{0, 0, 0, 0}, 0, 0, 0);
                acc[i] += prev[i];
            }
#pragma unroll
            for (int kk = 1; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[i], t[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) prev[i] = t[i];
        } else {
            f32x4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[i], f32x4{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
            for (int kk = 1; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[i], t[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += t[i];
        }
        a[0] += 1e-9f;          // (keeps the loop body from being hoisted)
    }
    f32x4 s = acc[0];
    for (int i = 1; i < 8; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += prev[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wg = 1; wg <= 4; ++wg)
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 1.0f);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256 * wg), dim3(256), 0, 0, out, iters / 2, 1.0f);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 1.0f);
                else hipLaunchKernelGGL(k<1>, dim3(256 * wg), dim3(256), 0, 0, out, iters, 1.0f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2048.0 * 32 * iters * 4.0 * 256 * wg;
            printf("%s  %d workgroup(s)/CU: %8.3f ms  %6.1f TFLOP/s\n", mode == 0 ? "plain accumulate        " : mode == 1 ? "chains of 4 + add       " : mode == 2 ? "chains of 8 + add       " : "chains of 4, add 1 late ", wg, ms, fl / ms * 1e-9);
        }
    return 0;
}
