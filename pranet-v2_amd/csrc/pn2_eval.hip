// pn2_eval.hip — the remaining metrics of the reference's eval_for_testAllInOne on the GPU (binary_seg/eval.py:18-66):
//   Sm   StructureMeasure   utils/eval_functions.py:5-94     -> pn2_eval_region_sums (integer moments of the four centroid quadrants; S_Object comes from the
//                                                                histograms of pn2_eval_hist)
//   wFm  original_WFb       utils/eval_functions.py:96-129   -> pn2_eval_wfm (exact Euclidean feature transform with scipy's tie-breaking, 7x7 Gaussian of the
//                                                                propagated error, weighted TP / FP sums)
//   meanEm EnhancedMeasure  utils/eval_functions.py:168-192  -> needs no kernel: for a binarised map the alignment matrix takes four values, so every
//                                                                threshold's score is a function of the two histograms (pn2/evaltail.py)
// All sums that decide a metric are integers (atomics on integers are order-independent) or fixed-order double reductions: deterministic.  The host finishes
// in float64 with the reference's expressions (pn2/evaltail.py), as it does for the threshold sweep.
#include "pn2_common.h"
#include "../../include/pn2.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ void block_add3(u64 a, u64 b, u64 c, u64* dst) {
    __shared__ u64 sh[3][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); c += __shfl_xor(c, o); }
    if (lane == 0) { sh[0][wv] = a; sh[1][wv] = b; sh[2][wv] = c; }
    __syncthreads();
    if (threadIdx.x < 3) { const u64 s = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3]; if (s) atomicAdd(dst + threadIdx.x, s); }
    __syncthreads();
}

// out[0..2] = sum of row indices, sum of column indices, count of the pixels with gt > 0.5   (centroid of S_Region, eval_functions.py:28-35)
__global__ __launch_bounds__(256) void eval_centroid_k(const float* __restrict__ gt, int H, int W, u64* __restrict__ out) {
    u64 sr = 0, sc = 0, cnt = 0;
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        if (gt[i] > 0.5f) { const int r = (int)(i / W); sr += r; sc += (int)(i - (long long)r * W); ++cnt; }
    block_add3(sr, sc, cnt, out);
}

// out[3 + q*5 + {0..4}] = {pixels, sum k, sum k^2, sum g, sum k*g} of quadrant q = (row >= X) + 2*(col >= Y): LT, RT, LB, RB of divide() (eval_functions.py:37-48),
// k = prediction byte, g = gt > 0.5; X, Y = int(mean.round()) of the foreground rows / columns (numpy rounds half to even: rint), (H//2, W//2) without foreground
__global__ __launch_bounds__(256) void eval_quadrants_k(const unsigned char* __restrict__ pred, const float* __restrict__ gt, int H, int W, u64* __restrict__ out) {
    const u64 cnt = out[2];
    const int X = cnt ? (int)rint((double)out[0] / (double)cnt) : H / 2, Y = cnt ? (int)rint((double)out[1] / (double)cnt) : W / 2;
    u64 a[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { a[q][0] = 0; a[q][1] = 0; a[q][2] = 0; a[q][3] = 0; }
    const long long n = (long long)H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / W), c = (int)(i - (long long)r * W), q = (r >= X ? 1 : 0) + (c >= Y ? 2 : 0);
        const u64 k = pred[i], g = gt[i] > 0.5f ? 1 : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) if (t == q) { a[t][0] += k; a[t][1] += k * k; a[t][2] += g; a[t][3] += k * g; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        block_add3(a[q][0], a[q][1], a[q][2], out + 3 + q * 5 + 1);
        block_add3(a[q][3], 0, 0, out + 3 + q * 5 + 4);
    }
    if (blockIdx.x == 0 && threadIdx.x < 4) {
        const int q = threadIdx.x;
        const long long rows = (q & 1) ? H - X : X, cols = (q & 2) ? W - Y : Y;
        out[3 + q * 5] = (u64)(rows * cols);
        if (q == 0) { out[23] = (u64)X; out[24] = (u64)Y; }
    }
}

// ---- exact Euclidean feature transform with scipy's order of preference (ndimage.distance_transform_edt(return_indices=True): Maurer's sweep along axis 0,
// then along axis 1, keeping the earlier site unless the next one is STRICTLY closer):
// (1) per column the nearest foreground row of every pixel, the smaller row on a tie (-1: no foreground in the column)
__global__ __launch_bounds__(256) void edt_cols_k(const float* __restrict__ gt, int H, int W, int* __restrict__ rcol) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= W) return;
    int last = -1;
    for (int i = 0; i < H; ++i) { if (gt[(size_t)i * W + j] > 0.5f) last = i; rcol[(size_t)i * W + j] = last; }
    int nxt = -1;
    for (int i = H - 1; i >= 0; --i) {
        if (gt[(size_t)i * W + j] > 0.5f) nxt = i;
        const int a = rcol[(size_t)i * W + j];
        rcol[(size_t)i * W + j] = a < 0 ? nxt : (nxt < 0 || i - a <= nxt - i ? a : nxt);
    }
}
// (2) over the columns' candidates (rcol[i][j'], j') the smallest squared distance, the smaller column on a tie; one block per row, the row's candidates in LDS.
// et[i][j] = E at the nearest foreground pixel (E = |pred/255 - g|, eval_functions.py:97,102-103), dst[i][j] = the distance (foreground pixels: themselves, 0)
__global__ __launch_bounds__(256) void edt_rows_k(const unsigned char* __restrict__ pred, const float* __restrict__ gt, int H, int W, const int* __restrict__ rcol,
                                                  double* __restrict__ et, double* __restrict__ dst) {
    extern __shared__ int cand[];
    const int i = blockIdx.x;
    for (int j = threadIdx.x; j < W; j += 256) cand[j] = rcol[(size_t)i * W + j];
    __syncthreads();
    for (int j = threadIdx.x; j < W; j += 256) {
        long long best = 0x7fffffffffffffffLL; int bj = j;
        for (int c = 0; c < W; ++c) {
            const int r = cand[c];
            if (r >= 0) { const long long dr = r - i, dc = c - j, dd = dr * dr + dc * dc; if (dd < best) { best = dd; bj = c; } }
        }
        const int br = cand[bj];
        const size_t src = (size_t)(br < 0 ? i : br) * W + bj;          // (no foreground at all: the host never uses the result)
        et[(size_t)i * W + j] = fabs((double)pred[src] / 255.0 - (gt[src] > 0.5f ? 1.0 : 0.0));
        dst[(size_t)i * W + j] = sqrt((double)best);
    }
}
// (3) EA = convolve(Et, K, mode='nearest') with the 7x7 Gaussian (accumulated in scipy's order: the flipped kernel row-major), MIN_E_EA, B, Ew and the block's
// sums {Ew over foreground, Ew over background} in a fixed order -> part[block][2]
__global__ __launch_bounds__(256) void wfm_sums_k(const unsigned char* __restrict__ pred, const float* __restrict__ gt, int H, int W, const double* __restrict__ et,
                                                  const double* __restrict__ dst, const double* __restrict__ K, double c5, double* __restrict__ part) {
    __shared__ double sh[2][256];
    double sf = 0.0, sb = 0.0;
    const long long n = (long long)H * W;
    for (long long p = (long long)blockIdx.x * 256 + threadIdx.x; p < n; p += (long long)gridDim.x * 256) {
        const int i = (int)(p / W), j = (int)(p - (long long)i * W);
        const bool fg = gt[p] > 0.5f;
        const double E = fabs((double)pred[p] / 255.0 - (fg ? 1.0 : 0.0));
        double ew;
        if (fg) {
            double ea = 0.0;
            for (int a = 0; a < 7; ++a) {
                const int ii = min(max(i + a - 3, 0), H - 1);
                for (int b = 0; b < 7; ++b) { const int jj = min(max(j + b - 3, 0), W - 1); ea += K[(6 - a) * 7 + (6 - b)] * et[(size_t)ii * W + jj]; }
            }
            ew = ea < E ? ea : E;          // B = 1 on the foreground
            sf += ew;
        } else {
            ew = E * (2.0 - exp(c5 * dst[p]));
            sb += ew;
        }
    }
    sh[0][threadIdx.x] = sf; sh[1][threadIdx.x] = sb;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = sh[0][0]; part[blockIdx.x * 2 + 1] = sh[1][0]; }
}

}  // namespace

extern "C" {

int pn2_eval_region_sums(const unsigned char* pred_u8, const float* gt, int H, int W, unsigned long long* out25, void* stream) {
    if (!pred_u8 || !gt || !out25 || H < 1 || W < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out25, 0, 25 * sizeof(unsigned long long), st) != hipSuccess) return -4;
    const long long n = (long long)H * W;
    const int g = (int)std::min<long long>((n + 256 * 8 - 1) / (256 * 8), 1024);
    hipLaunchKernelGGL(eval_centroid_k, dim3(g), dim3(256), 0, st, gt, H, W, out25);
    hipLaunchKernelGGL(eval_quadrants_k, dim3(g), dim3(256), 0, st, pred_u8, gt, H, W, out25);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_eval_wfm_blocks(int H, int W) { const long long n = (long long)H * W; return (int)std::min<long long>((n + 255) / 256, 2048); }

int pn2_eval_wfm(const unsigned char* pred_u8, const float* gt, int H, int W, const double* K49, double c5, int* work_i, double* work_d, double* part, void* stream) {
    if (!pred_u8 || !gt || !K49 || !work_i || !work_d || !part || H < 1 || W < 1) return -1;
    if ((size_t)W * sizeof(int) > 60 * 1024) return -2;
    hipStream_t st = (hipStream_t)stream;
    double* et = work_d; double* dst = work_d + (size_t)H * W;
    hipLaunchKernelGGL(edt_cols_k, dim3((W + 255) / 256), dim3(256), 0, st, gt, H, W, work_i);
    hipLaunchKernelGGL(edt_rows_k, dim3(H), dim3(256), (size_t)W * sizeof(int), st, pred_u8, gt, H, W, work_i, et, dst);
    hipLaunchKernelGGL(wfm_sums_k, dim3(pn2_eval_wfm_blocks(H, W)), dim3(256), 0, st, pred_u8, gt, H, W, et, dst, K49, c5, part);
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
