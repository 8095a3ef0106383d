// pn2_input.hip - the input transform of binary_seg/utils/dataloader.py:104-111 / :176-181 on the device:
//   transforms.Resize((S, S))  = PIL.Image.resize(BILINEAR): separable antialiased triangle filter, uint8 after each pass, 22-bit fixed-point taps
//   transforms.ToTensor()      = uint8 HWC -> float32 CHW / 255
//   transforms.Normalize(m, s) = (t - m) / s
// Bit-exact with Pillow (the taps are computed on the host in double exactly as Pillow's precompute_coeffs / normalize_coeffs_8bpc do).
#include <cmath>
#include <cstdint>
#include "pn2_common.h"
#include "../../include/pn2.h"

namespace {

constexpr int RS_PREC = 32 - 8 - 2;

// one separable pass: dst[o][i][c] = clip8((2^21 + sum_x src[..][xmin[o] + x][..] * kk[o][x]) >> 22) along `axis` (1: width, 0: height)
__global__ __launch_bounds__(256) void resize_pass_k(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int H, int W, int C, int out_size, int axis,
                                                     const int* __restrict__ xmin, const int* __restrict__ cnt, const int* __restrict__ kk, int ksize) {
    const int OH = axis == 0 ? out_size : H, OW = axis == 1 ? out_size : W;
    const size_t total = (size_t)OH * OW * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C), ox = (int)((i / C) % OW), oy = (int)(i / ((size_t)C * OW));
        const int o = axis == 1 ? ox : oy;
        const int x0 = xmin[o], n = cnt[o];
        const int* k = kk + (size_t)o * ksize;
        int acc = 1 << (RS_PREC - 1);
        if (axis == 1) { const unsigned char* s = src + ((size_t)oy * W + x0) * C + c; for (int x = 0; x < n; ++x) acc += (int)s[(size_t)x * C] * k[x]; }
        else { const unsigned char* s = src + ((size_t)x0 * W + ox) * C + c; for (int x = 0; x < n; ++x) acc += (int)s[(size_t)x * W * C] * k[x]; }
        acc >>= RS_PREC;
        dst[i] = (unsigned char)(acc < 0 ? 0 : (acc > 255 ? 255 : acc));
    }
}

// ToTensor (+ Normalize when mean/std are given): dst[c][y][x] = (src[y][x][c] / 255 - mean[c]) / std[c]   (true fp32 divisions, as torch)
__global__ __launch_bounds__(256) void to_tensor_k(const unsigned char* __restrict__ src, float* __restrict__ dst, int H, int W, int C, const float* __restrict__ mean,
                                                   const float* __restrict__ std) {
    const size_t total = (size_t)H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((size_t)W * H));
        float t = (float)src[((size_t)y * W + x) * C + c] / 255.0f;
        if (mean) t = (t - mean[c]) / std[c];
        dst[i] = t;
    }
}

inline unsigned grid_of(size_t total) { const size_t g = (total + 255) / 256; return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g)); }

}  // namespace

extern "C" {

int pn2_resize_ksize(int in_size, int out_size) {
    if (in_size < 1 || out_size < 1) return -1;
    double fs = (double)in_size / out_size; if (fs < 1.0) fs = 1.0;
    return (int)std::ceil(1.0 * fs) * 2 + 1;
}

/* HOST: the bilinear taps of one axis exactly as Pillow computes them (Resample.c precompute_coeffs + normalize_coeffs_8bpc):
 * xmin[out_size], count[out_size], kk[out_size][pn2_resize_ksize(in_size, out_size)] int32 fixed point (22 fractional bits) */
int pn2_resize_coeffs(int in_size, int out_size, int* xmin, int* count, int* kk) {
    if (!xmin || !count || !kk) return -1;
    const int ksize = pn2_resize_ksize(in_size, out_size);
    if (ksize < 0) return -1;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale, support = 1.0 * filterscale, ss = 1.0 / filterscale;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int lo = (int)(center - support + 0.5); if (lo < 0) lo = 0;
        int hi = (int)(center + support + 0.5); if (hi > in_size) hi = in_size;
        const int n = hi - lo;
        double w[64 * 1024 / 8]; double ww = 0.0;
        if (n > (int)(sizeof(w) / sizeof(w[0]))) return -2;
        for (int x = 0; x < n; ++x) {
            double a = (x + lo - center + 0.5) * ss; if (a < 0.0) a = -a;
            w[x] = a < 1.0 ? 1.0 - a : 0.0;
            ww += w[x];
        }
        int* k = kk + (size_t)xx * ksize;
        for (int x = 0; x < ksize; ++x) k[x] = 0;
        for (int x = 0; x < n; ++x) {
            const double v = (ww != 0.0 ? w[x] / ww : w[x]);
            k[x] = v < 0 ? (int)(-0.5 + v * (1 << RS_PREC)) : (int)(0.5 + v * (1 << RS_PREC));
        }
        xmin[xx] = lo; count[xx] = n;
    }
    return 0;
}

/* one pass of PIL.Image.resize(BILINEAR) on a device uint8 [H][W][C] image: axis 1 -> [H][out_size][C], axis 0 -> [out_size][W][C];
 * xmin / count / kk: DEVICE copies of pn2_resize_coeffs(axis length, out_size).  Pillow's order is width first, then height. */
int pn2_resize_u8_pass(const unsigned char* src, unsigned char* dst, int H, int W, int C, int out_size, int axis, const int* xmin_dev, const int* count_dev,
                       const int* kk_dev, int ksize, void* stream) {
    if (!src || !dst || !xmin_dev || !count_dev || !kk_dev || H < 1 || W < 1 || C < 1 || out_size < 1 || (axis != 0 && axis != 1)) return -1;
    const size_t total = (size_t)(axis == 0 ? out_size : H) * (axis == 1 ? out_size : W) * C;
    hipLaunchKernelGGL(resize_pass_k, dim3(grid_of(total)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, C, out_size, axis, xmin_dev, count_dev, kk_dev, ksize);
    PN2_CHECK_LAUNCH();
    return 0;
}

/* transforms.ToTensor() [+ transforms.Normalize(mean, std) when mean != NULL]: uint8 [H][W][C] -> fp32 [C][H][W] (one sample of the NCHW batch) */
int pn2_u8_to_tensor(const unsigned char* src, float* dst, int H, int W, int C, const float* mean_dev, const float* std_dev, void* stream) {
    if (!src || !dst || H < 1 || W < 1 || C < 1 || ((mean_dev == nullptr) != (std_dev == nullptr))) return -1;
    hipLaunchKernelGGL(to_tensor_k, dim3(grid_of((size_t)H * W * C)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, C, mean_dev, std_dev);
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
