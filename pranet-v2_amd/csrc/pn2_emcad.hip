// pn2_emcad.hip — kernels of the EMCAD decoder (reference: /root/reference/multiclass_seg/EMCAD/lib/decoders.py, BASELINE config 5):
//   depth-wise k x k convs (k = 1, 3, 5) feeding BatchNorm      MSDC :83-99, EUCB.up_dwc :170-175
//   grouped 3x3 conv with 2 input channels per group            LGAG.W_g / W_x :193-200 (groups = F_int)
//   channel / pixel gates x * g                                  CAB :233-241 and SAB :252-258 used at EMCAD_dual.forward :442-443 ; LGAG :214
//   global average + max pooling, channel mean + max             CAB :234-237, SAB :253-255
//   nearest x2 up-sampling                                       EUCB :171
//   sum of the three MSDC branches written through channel_shuffle   MSCB.forward :147-154 with channel_shuffle :69-77
//   sigmoid on small fp32 maps                                   CAB / SAB / LGAG.psi
// All element-wise / memory-bound; BatchNorm statistics come out of the producing kernel as per-block partial rows, like the conv
// epilogue of pn2_conv.hip, so pn2_bn_finalize / pn2_affine_act / pn2_bn_bwd_* are reused unchanged.  Deterministic (no atomics).
#include "pn2_common.h"
#include "pn2_dw.h"
#include "../../include/pn2.h"

namespace {

template <typename T> __device__ __forceinline__ void ldv(const T* p, float* f) { TT<T>::unpack(*reinterpret_cast<const uint4*>(p), f); }
template <typename T> __device__ __forceinline__ void stv(T* p, const float* f) { *reinterpret_cast<uint4*>(p) = TT<T>::pack(f); }
inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
inline int grid_for(size_t total) { size_t g = (total + 255) / 256; return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g)); }

// rows (pixels) per block of the "thread owns a channel vector and walks pixels" kernels: ~1024 blocks, a multiple of R
inline void walk_geometry(int M, int CV, int& cvp, int& pix, int& nblk) {
    cvp = pow2ceil(CV); if (cvp > 256) cvp = 256;
    const int R = 256 / cvp;
    pix = (M + 1023) / 1024;
    pix = ((pix + R - 1) / R) * R;
    if (pix < 4 * R) pix = 4 * R;
    nblk = (M + pix - 1) / pix;
}

// cross-row-lane sum of per-thread channel-vector accumulators a[V] -> dst[c] (block partial row), through LDS, fixed order
template <int V>
__device__ __forceinline__ void block_colsum(float* sh, const float* a, int CVP, int R, int cvl, int rl, bool active, float* dst) {
#pragma unroll
    for (int e = 0; e < V; ++e) sh[(rl * CVP + cvl) * V + e] = a[e];
    __syncthreads();
    if (rl == 0 && active) {
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float s = 0.f;
            for (int r = 0; r < R; ++r) s += sh[(r * CVP + cvl) * V + e];
            dst[e] = s;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------ depth-wise K x K
// z[p][c] = sum_taps w[c][tap] x[p + tap][c]; optional per-block partial sums of z and z^2 (BatchNorm batch statistics).
// flip: mirrored kernel = data gradient.  Sliding window (as the 3x3 + GELU kernels of pn2_vit.hip): a thread owns VT channels, keeps their
// K*K weights and the K x K input window (packed) in registers and walks row segments of SEG output pixels - K new loads per output pixel
// instead of K*K.  A block = CVP channel groups x R lanes; a lane walks the segments bid*SPB + rl, + R, ... of the block's range.
template <typename T, int K, int VT>
__device__ __forceinline__ void dwk_load_col(const T* base, const bool (&vy)[K], int ix, int W, int C, DwVec<T, VT> (&col)[K]) {
#pragma unroll
    for (int r = 0; r < K; ++r) {
        col[r].zero();
        if (vy[r] && (unsigned)ix < (unsigned)W) col[r].load(base + ((ptrdiff_t)(r - K / 2) * W + ix) * C);
    }
}

template <typename T, int K, int VT>
__global__ __launch_bounds__(256) void dwconv_row_k(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ z, int N, int H, int W, int C, int flip, int accumulate,
                                                    float* __restrict__ psum, float* __restrict__ psq, int SEG, int SPR, int SPB, int CVP) {
    typedef DwVec<T, VT> Vec;
    constexpr int KK = K * K, PD = K / 2;
    extern __shared__ float sh[];
    const int CV = C / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x), cv = blockIdx.y * CVP + cvl;
    const bool act = cv < CV;
    const int nseg = N * H * SPR;
    float wr[KK][VT], s1[VT], s2[VT];
#pragma unroll
    for (int e = 0; e < VT; ++e) {
        s1[e] = 0.f; s2[e] = 0.f;
#pragma unroll
        for (int t = 0; t < KK; ++t) wr[t][e] = act ? w[(size_t)(cv * VT + e) * KK + (flip ? KK - 1 - t : t)] : 0.f;
    }
    if (act) {
        const int send = min(nseg, (bid + 1) * SPB);
        for (int s = bid * SPB + rl; s < send; s += R) {
            const int sx = s % SPR, row = s / SPR, oy = row % H;
            const int x0 = sx * SEG, x1 = min(W, x0 + SEG);
            bool vy[K];
#pragma unroll
            for (int r = 0; r < K; ++r) vy[r] = (unsigned)(oy + r - PD) < (unsigned)H;
            const T* base = x + ((size_t)row * W) * C + cv * VT;           // pixel (row, 0) of this channel group
            Vec win[K][K], nx[K];                                         // win[j] = column x0 - PD + j
#pragma unroll
            for (int j = 0; j < K; ++j) dwk_load_col<T, K, VT>(base, vy, x0 - PD + j, W, C, win[j]);
            for (int ox = x0; ox < x1; ++ox) {
                dwk_load_col<T, K, VT>(base, vy, ox + 1 < x1 ? ox + PD + 1 : -1, W, C, nx);     // the column the next pixel adds
                float a[VT], xv[VT];
#pragma unroll
                for (int e = 0; e < VT; ++e) a[e] = 0.f;
#pragma unroll
                for (int r = 0; r < K; ++r)
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        win[j][r].unpack(xv);
#pragma unroll
                        for (int e = 0; e < VT; ++e) a[e] += wr[r * K + j][e] * xv[e];
                    }
                const size_t o = ((size_t)row * W + ox) * C + cv * VT;
                Vec ov;
                if (accumulate) {
                    ov.load(z + o); ov.unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) a[e] += xv[e];
                }
                ov.pack(a); ov.store(z + o);
                if (psum) {       // statistics of the values as stored (rounded to T), like the conv epilogue
                    ov.unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) { s1[e] += xv[e]; s2[e] += xv[e] * xv[e]; }
                }
#pragma unroll
                for (int j = 0; j + 1 < K; ++j)
#pragma unroll
                    for (int r = 0; r < K; ++r) win[j][r] = win[j + 1][r];
#pragma unroll
                for (int r = 0; r < K; ++r) win[K - 1][r] = nx[r];
            }
        }
    }
    if (psum) {
        block_colsum<VT>(sh, s1, CVP, R, cvl, rl, act, psum + (size_t)bid * C + cv * VT);
        block_colsum<VT>(sh, s2, CVP, R, cvl, rl, act, psq + (size_t)bid * C + cv * VT);
    }
}

// partial[blk][C*K*K]: [c*KK + tap] = sum_pixels dz[p][c] * x[p + tap][c], same walk
template <typename T, int K, int VT>
__global__ __launch_bounds__(256) void dwconv_wgrad_row_k(const T* __restrict__ dz, const T* __restrict__ x, float* __restrict__ partial, int N, int H, int W, int C,
                                                          int SEG, int SPR, int SPB, int CVP) {
    typedef DwVec<T, VT> Vec;
    constexpr int KK = K * K, PD = K / 2;
    extern __shared__ float sh[];
    const int CV = C / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x), cv = blockIdx.y * CVP + cvl;
    const bool act = cv < CV;
    const int nseg = N * H * SPR;
    float a[KK][VT];
#pragma unroll
    for (int t = 0; t < KK; ++t)
#pragma unroll
        for (int e = 0; e < VT; ++e) a[t][e] = 0.f;
    if (act) {
        const int send = min(nseg, (bid + 1) * SPB);
        for (int s = bid * SPB + rl; s < send; s += R) {
            const int sx = s % SPR, row = s / SPR, oy = row % H;
            const int x0 = sx * SEG, x1 = min(W, x0 + SEG);
            bool vy[K];
#pragma unroll
            for (int r = 0; r < K; ++r) vy[r] = (unsigned)(oy + r - PD) < (unsigned)H;
            const size_t o0 = ((size_t)row * W) * C + cv * VT;
            const T* base = x + o0;
            Vec win[K][K], nx[K], dn;
#pragma unroll
            for (int j = 0; j < K; ++j) dwk_load_col<T, K, VT>(base, vy, x0 - PD + j, W, C, win[j]);
            dn.load(dz + o0 + (size_t)x0 * C);
            for (int ox = x0; ox < x1; ++ox) {
                const Vec dc = dn;
                dwk_load_col<T, K, VT>(base, vy, ox + 1 < x1 ? ox + PD + 1 : -1, W, C, nx);
                if (ox + 1 < x1) dn.load(dz + o0 + (size_t)(ox + 1) * C);
                float d[VT], xv[VT];
                dc.unpack(d);
#pragma unroll
                for (int r = 0; r < K; ++r)
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        win[j][r].unpack(xv);
#pragma unroll
                        for (int e = 0; e < VT; ++e) a[r * K + j][e] += d[e] * xv[e];
                    }
#pragma unroll
                for (int j = 0; j + 1 < K; ++j)
#pragma unroll
                    for (int r = 0; r < K; ++r) win[j][r] = win[j + 1][r];
#pragma unroll
                for (int r = 0; r < K; ++r) win[K - 1][r] = nx[r];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < KK; ++t) {
        float tmp[VT];
        block_colsum<VT>(sh, a[t], CVP, R, cvl, rl, act, tmp);
        if (rl == 0 && act) {
#pragma unroll
            for (int e = 0; e < VT; ++e) partial[(size_t)bid * C * KK + (size_t)(cv * VT + e) * KK + t] = tmp[e];
        }
    }
}

// ------------------------------------------------------------------------------------------ grouped 3x3, two input channels per group
// z[p][o] = sum_{j<2, taps} w[o][j][tap] x[p + tap][2o + j]   (LGAG.W_g / W_x, decoders.py:193-200; the bias is folded by the caller).
// Sliding-window walks as the depth-wise kernels: a thread owns VT outputs (= one 16-byte vector of 2*VT inputs), keeps their 18*VT weights and
// the packed 3x3 input window in registers and walks a row segment.
template <typename T, int VT>
__device__ __forceinline__ void pc_load_col(const T* base, const bool (&vy)[3], int ix, int W, int ld, DwVec<T, 2 * VT> (&col)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        col[r].zero();
        if (vy[r] && (unsigned)ix < (unsigned)W) col[r].load(base + ((ptrdiff_t)(r - 1) * W + ix) * ld);
    }
}
template <typename T, int VT>
__device__ __forceinline__ void pc_load_col1(const T* base, const bool (&vy)[3], int ix, int W, int ld, DwVec<T, VT> (&col)[3]) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        col[r].zero();
        if (vy[r] && (unsigned)ix < (unsigned)W) col[r].load(base + ((ptrdiff_t)(r - 1) * W + ix) * ld);
    }
}

template <typename T, int VT>
__global__ __launch_bounds__(256) void pairconv_fwd_k(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ z, int N, int H, int W, int F,
                                                      float* __restrict__ psum, float* __restrict__ psq, int SEG, int SPR, int SPB, int CVP) {
    typedef DwVec<T, 2 * VT> VI; typedef DwVec<T, VT> VO;
    extern __shared__ float sh[];
    const int FV = F / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x), fv = blockIdx.y * CVP + cvl;
    const bool act = fv < FV;
    const int nseg = N * H * SPR;
    float wr[VT][18], s1[VT], s2[VT];
#pragma unroll
    for (int e = 0; e < VT; ++e) {
        s1[e] = 0.f; s2[e] = 0.f;
#pragma unroll
        for (int t = 0; t < 18; ++t) wr[e][t] = act ? w[(size_t)(fv * VT + e) * 18 + t] : 0.f;
    }
    if (act) {
        const int send = min(nseg, (bid + 1) * SPB);
        for (int s = bid * SPB + rl; s < send; s += R) {
            const int sx = s % SPR, row = s / SPR, oy = row % H;
            const int x0 = sx * SEG, x1 = min(W, x0 + SEG);
            bool vy[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) vy[r] = (unsigned)(oy + r - 1) < (unsigned)H;
            const T* base = x + ((size_t)row * W) * (2 * F) + fv * 2 * VT;
            VI c0[3], c1[3], c2[3], nx[3];
            pc_load_col<T, VT>(base, vy, x0 - 1, W, 2 * F, c0); pc_load_col<T, VT>(base, vy, x0, W, 2 * F, c1); pc_load_col<T, VT>(base, vy, x0 + 1, W, 2 * F, c2);
            for (int ox = x0; ox < x1; ++ox) {
                pc_load_col<T, VT>(base, vy, ox + 1 < x1 ? ox + 2 : -1, W, 2 * F, nx);
                float a[VT], xv[2 * VT];
#pragma unroll
                for (int e = 0; e < VT; ++e) a[e] = 0.f;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    c0[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) a[e] += wr[e][r * 3] * xv[2 * e] + wr[e][9 + r * 3] * xv[2 * e + 1];
                    c1[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) a[e] += wr[e][r * 3 + 1] * xv[2 * e] + wr[e][9 + r * 3 + 1] * xv[2 * e + 1];
                    c2[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) a[e] += wr[e][r * 3 + 2] * xv[2 * e] + wr[e][9 + r * 3 + 2] * xv[2 * e + 1];
                }
                VO ov; ov.pack(a); ov.store(z + ((size_t)row * W + ox) * F + fv * VT);
                ov.unpack(a);                     // statistics of the stored (rounded) values, like the conv epilogue
#pragma unroll
                for (int e = 0; e < VT; ++e) { s1[e] += a[e]; s2[e] += a[e] * a[e]; }
#pragma unroll
                for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; c2[r] = nx[r]; }
            }
        }
    }
    block_colsum<VT>(sh, s1, CVP, R, cvl, rl, act, psum + (size_t)bid * F + fv * VT);
    block_colsum<VT>(sh, s2, CVP, R, cvl, rl, act, psq + (size_t)bid * F + fv * VT);
}

// dx[p][2o + j] (+)= sum_taps w[o][j][8 - tap] dz[p + tap][o]
template <typename T, int VT>
__global__ __launch_bounds__(256) void pairconv_dgrad_k(const T* __restrict__ dz, const float* __restrict__ w, T* __restrict__ dx, int N, int H, int W, int F, int accumulate,
                                                        int SEG, int SPR, int CVP) {
    typedef DwVec<T, 2 * VT> VI; typedef DwVec<T, VT> VO;
    const int FV = F / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x), fv = blockIdx.y * CVP + cvl;
    const int s = bid * R + rl;
    if (s >= N * H * SPR || fv >= FV) return;
    float wr[VT][18];
#pragma unroll
    for (int e = 0; e < VT; ++e)
#pragma unroll
        for (int t = 0; t < 18; ++t) wr[e][t] = w[(size_t)(fv * VT + e) * 18 + t];
    const int sx = s % SPR, row = s / SPR, oy = row % H;
    const int x0 = sx * SEG, x1 = min(W, x0 + SEG);
    bool vy[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) vy[r] = (unsigned)(oy + r - 1) < (unsigned)H;
    const T* base = dz + ((size_t)row * W) * F + fv * VT;
    VO c0[3], c1[3], c2[3], nx[3];
    pc_load_col1<T, VT>(base, vy, x0 - 1, W, F, c0); pc_load_col1<T, VT>(base, vy, x0, W, F, c1); pc_load_col1<T, VT>(base, vy, x0 + 1, W, F, c2);
    for (int ox = x0; ox < x1; ++ox) {
        pc_load_col1<T, VT>(base, vy, ox + 1 < x1 ? ox + 2 : -1, W, F, nx);
        float a[2 * VT], d[VT];
#pragma unroll
        for (int e = 0; e < 2 * VT; ++e) a[e] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {          // tap t = r*3 + j uses the mirrored weight 8 - t
            c0[r].unpack(d);
#pragma unroll
            for (int e = 0; e < VT; ++e) { a[2 * e] += wr[e][8 - r * 3] * d[e]; a[2 * e + 1] += wr[e][17 - r * 3] * d[e]; }
            c1[r].unpack(d);
#pragma unroll
            for (int e = 0; e < VT; ++e) { a[2 * e] += wr[e][7 - r * 3] * d[e]; a[2 * e + 1] += wr[e][16 - r * 3] * d[e]; }
            c2[r].unpack(d);
#pragma unroll
            for (int e = 0; e < VT; ++e) { a[2 * e] += wr[e][6 - r * 3] * d[e]; a[2 * e + 1] += wr[e][15 - r * 3] * d[e]; }
        }
        T* dp = dx + ((size_t)row * W + ox) * (2 * F) + fv * 2 * VT;
        VI ov;
        if (accumulate) {
            float o[2 * VT];
            ov.load(dp); ov.unpack(o);
#pragma unroll
            for (int e = 0; e < 2 * VT; ++e) a[e] += o[e];
        }
        ov.pack(a); ov.store(dp);
#pragma unroll
        for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; c2[r] = nx[r]; }
    }
}

// partial[blk][F*18]: [o*18 + j*9 + tap] = sum_pixels dz[p][o] * x[p + tap][2o + j]
template <typename T, int VT>
__global__ __launch_bounds__(256) void pairconv_wgrad_k(const T* __restrict__ dz, const T* __restrict__ x, float* __restrict__ partial, int N, int H, int W, int F,
                                                        int SEG, int SPR, int SPB, int CVP) {
    typedef DwVec<T, 2 * VT> VI; typedef DwVec<T, VT> VO;
    extern __shared__ float sh[];
    const int FV = F / VT, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP;
    const int bid = xcd_remap(blockIdx.x, gridDim.x), fv = blockIdx.y * CVP + cvl;
    const bool act = fv < FV;
    const int nseg = N * H * SPR;
    float a[18][VT];
#pragma unroll
    for (int t = 0; t < 18; ++t)
#pragma unroll
        for (int e = 0; e < VT; ++e) a[t][e] = 0.f;
    if (act) {
        const int send = min(nseg, (bid + 1) * SPB);
        for (int s = bid * SPB + rl; s < send; s += R) {
            const int sx = s % SPR, row = s / SPR, oy = row % H;
            const int x0 = sx * SEG, x1 = min(W, x0 + SEG);
            bool vy[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) vy[r] = (unsigned)(oy + r - 1) < (unsigned)H;
            const T* base = x + ((size_t)row * W) * (2 * F) + fv * 2 * VT;
            const T* dbase = dz + ((size_t)row * W) * F + fv * VT;
            VI c0[3], c1[3], c2[3], nx[3]; VO dn;
            pc_load_col<T, VT>(base, vy, x0 - 1, W, 2 * F, c0); pc_load_col<T, VT>(base, vy, x0, W, 2 * F, c1); pc_load_col<T, VT>(base, vy, x0 + 1, W, 2 * F, c2);
            dn.load(dbase + (size_t)x0 * F);
            for (int ox = x0; ox < x1; ++ox) {
                const VO dc = dn;
                pc_load_col<T, VT>(base, vy, ox + 1 < x1 ? ox + 2 : -1, W, 2 * F, nx);
                if (ox + 1 < x1) dn.load(dbase + (size_t)(ox + 1) * F);
                float d[VT], xv[2 * VT];
                dc.unpack(d);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    c0[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) { a[r * 3][e] += d[e] * xv[2 * e]; a[9 + r * 3][e] += d[e] * xv[2 * e + 1]; }
                    c1[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) { a[r * 3 + 1][e] += d[e] * xv[2 * e]; a[9 + r * 3 + 1][e] += d[e] * xv[2 * e + 1]; }
                    c2[r].unpack(xv);
#pragma unroll
                    for (int e = 0; e < VT; ++e) { a[r * 3 + 2][e] += d[e] * xv[2 * e]; a[9 + r * 3 + 2][e] += d[e] * xv[2 * e + 1]; }
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { c0[r] = c1[r]; c1[r] = c2[r]; c2[r] = nx[r]; }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 18; ++t) {
        float tmp[VT];
        block_colsum<VT>(sh, a[t], CVP, R, cvl, rl, act, tmp);
        if (rl == 0 && act) {
#pragma unroll
            for (int e = 0; e < VT; ++e) partial[(size_t)bid * F * 18 + (size_t)(fv * VT + e) * 18 + t] = tmp[e];
        }
    }
}

// ------------------------------------------------------------------------------------------ gates
// mode 0: y[n][p][c] = x * g[n][c]   (channel gate, CAB) ; mode 1: y = x * g[n][p]   (pixel gate, SAB / LGAG.psi)
template <typename T>
__global__ __launch_bounds__(256) void gate_mul_k(const T* __restrict__ x, const float* __restrict__ g, T* __restrict__ y, int N, int HW, int C, int mode, int accumulate) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V;
    const size_t total = (size_t)N * HW * CV;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int cv, pp_; const size_t p = divmod_idx(idx, CV, cv); const int n = (int)divmod_idx(p, HW, pp_);
        float v[V], o[V];
        ldv<T>(x + p * C + cv * V, v);
        if (accumulate) ldv<T>(y + p * C + cv * V, o);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float t = v[e] * (mode ? g[p] : g[(size_t)n * C + cv * V + e]);
            o[e] = accumulate ? o[e] + t : t;
        }
        stv<T>(y + p * C + cv * V, o);
    }
}

// pixel gate gradient: dg[n][p] = sum_c dy * x   (one 8/16-lane group per pixel)
template <typename T>
__global__ __launch_bounds__(256) void gate_dpix_k(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ dg, size_t NP, int C) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V;
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    for (size_t p = wave; p < NP; p += nw) {
        float s = 0.f;
        for (int cv = lane; cv < CV; cv += 64) {
            float a[V], b[V];
            ldv<T>(dy + p * C + cv * V, a); ldv<T>(x + p * C + cv * V, b);
#pragma unroll
            for (int e = 0; e < V; ++e) s += a[e] * b[e];
        }
        s = wave_sum(s);
        if (lane == 0) dg[p] = s;
    }
}

// channel gate gradient partials: part[blk][n*C + c] = sum over the block's pixels of sample n of dy * x   (grid: blocks x N)
template <typename T>
__global__ __launch_bounds__(256) void gate_dchan_k(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ part, int N, int HW, int C, int pix_per_blk, int CVP) {
    constexpr int V = TT<T>::VEC;
    extern __shared__ float sh[];
    const int CV = C / V, R = 256 / CVP, cvl = threadIdx.x % CVP, rl = threadIdx.x / CVP, n = blockIdx.y;
    const int p0 = blockIdx.x * pix_per_blk;
    int p1 = p0 + pix_per_blk; if (p1 > HW) p1 = HW;
    for (int cvb = 0; cvb < CV; cvb += CVP) {
        const int cv = cvb + cvl;
        const bool act = cv < CV;
        float a[V];
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] = 0.f;
        if (act) {
            for (int m = p0 + rl; m < p1; m += R) {
                float d[V], xv[V];
                const size_t o = ((size_t)n * HW + m) * C + cv * V;
                ldv<T>(dy + o, d); ldv<T>(x + o, xv);
#pragma unroll
                for (int e = 0; e < V; ++e) a[e] += d[e] * xv[e];
            }
        }
        block_colsum<V>(sh, a, CVP, R, cvl, rl, act, part + (size_t)blockIdx.x * N * C + (size_t)n * C + cv * V);
    }
}

// ------------------------------------------------------------------------------------------ global avg + max pooling (CAB)
// one block per (n, channel-vector chunk): avg[n][c], mx[n][c] (compute dtype, rows of an [N][1][1][C] map) and the argmax pixel
template <typename T>
__global__ __launch_bounds__(256) void global_pool_k(const T* __restrict__ x, T* __restrict__ avg, T* __restrict__ mx, int* __restrict__ arg, int HW, int C) {
    constexpr int V = TT<T>::VEC;
    __shared__ float ssum[32][8]; __shared__ float smax[32][8]; __shared__ int sarg[32][8];
    const int CV = C / V, n = blockIdx.y;
    const int cvl = threadIdx.x % 8, rl = threadIdx.x / 8, cv = blockIdx.x * 8 + cvl;      // 8 channel vectors x 32 row lanes
    float s[V], m[V]; int am[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { s[e] = 0.f; m[e] = -INFINITY; am[e] = 0; }
    if (cv < CV) {
        for (int p = rl; p < HW; p += 32) {
            float v[V];
            ldv<T>(x + ((size_t)n * HW + p) * C + cv * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) { s[e] += v[e]; if (v[e] > m[e]) { m[e] = v[e]; am[e] = p; } }
        }
    }
    // reduce the 32 row lanes per channel (sequential in lane order: first maximum wins, as torch's max pooling does)
#pragma unroll
    for (int e = 0; e < V; ++e) {
        ssum[rl][cvl] = s[e]; smax[rl][cvl] = m[e]; sarg[rl][cvl] = am[e];
        __syncthreads();
        if (rl == 0 && cv < CV) {
            float ts = 0.f, tm = -INFINITY; int ta = 0;
            for (int r = 0; r < 32; ++r) {
                ts += ssum[r][cvl];
                if (smax[r][cvl] > tm || (smax[r][cvl] == tm && sarg[r][cvl] < ta)) { tm = smax[r][cvl]; ta = sarg[r][cvl]; }
            }
            const int c = cv * V + e;
            TT<T>::st(avg + (size_t)n * C + c, ts / (float)HW);
            TT<T>::st(mx + (size_t)n * C + c, tm);
            arg[(size_t)n * C + c] = ta;
        }
        __syncthreads();
    }
}

// dx[n][p][c] (+)= davg[n][c] / HW + (p == arg[n][c]) * dmax[n][c]
template <typename T>
__global__ __launch_bounds__(256) void global_pool_bwd_k(const T* __restrict__ davg, const T* __restrict__ dmax, const int* __restrict__ arg, T* __restrict__ dx,
                                                         int N, int HW, int C, int accumulate) {
    const size_t total = (size_t)N * HW * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int c, pp; const size_t p = divmod_idx(idx, C, c); const int n = (int)divmod_idx(p, HW, pp);
        float v = TT<T>::ld(davg + (size_t)n * C + c) / (float)HW;
        if (arg[(size_t)n * C + c] == pp) v += TT<T>::ld(dmax + (size_t)n * C + c);
        if (accumulate) v += TT<T>::ld(dx + idx);
        TT<T>::st(dx + idx, v);
    }
}

// ------------------------------------------------------------------------------------------ channel mean + max per pixel (SAB)
// out[p][0] = mean_c x, out[p][1] = max_c x, out[p][2..7] = 0 ; arg[p] = first channel attaining the max
template <typename T>
__global__ __launch_bounds__(256) void chan_stats_k(const T* __restrict__ x, T* __restrict__ out, int* __restrict__ arg, size_t NP, int C) {
    const int lane = threadIdx.x & 63;
    const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256) >> 6;
    for (size_t p = wave; p < NP; p += nw) {
        float s = 0.f, m = -INFINITY; int am = 0x7fffffff;
        for (int c = lane; c < C; c += 64) {
            const float v = TT<T>::ld(x + p * C + c);
            s += v;
            if (v > m) { m = v; am = c; }
        }
        s = wave_sum(s);
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(m, o); const int oa = __shfl_xor(am, o);
            if (om > m || (om == m && oa < am)) { m = om; am = oa; }
        }
        if (lane < 8) TT<T>::st(out + p * 8 + lane, lane == 0 ? s / (float)C : (lane == 1 ? m : 0.f));
        if (lane == 0) arg[p] = am;
    }
}

// dx[p][c] (+)= dout[p][0] / C + (c == arg[p]) * dout[p][1]
template <typename T>
__global__ __launch_bounds__(256) void chan_stats_bwd_k(const T* __restrict__ dout, const int* __restrict__ arg, T* __restrict__ dx, size_t NP, int C, int accumulate) {
    const size_t total = NP * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int c; const size_t p = divmod_idx(idx, C, c);
        float v = TT<T>::ld(dout + p * 8) / (float)C;
        if (arg[p] == c) v += TT<T>::ld(dout + p * 8 + 1);
        if (accumulate) v += TT<T>::ld(dx + idx);
        TT<T>::st(dx + idx, v);
    }
}

// ------------------------------------------------------------------------------------------ nearest x2
template <typename T>
__global__ __launch_bounds__(256) void up2_k(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V, OH = 2 * H, OW = 2 * W;
    const size_t total = (size_t)N * OH * OW * CV;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int cv, ox, oy; size_t p = divmod_idx(idx, CV, cv);
        p = divmod_idx(p, OW, ox); const int n = (int)divmod_idx(p, OH, oy);
        *reinterpret_cast<uint4*>(y + (((size_t)n * OH + oy) * OW + ox) * C + cv * V) =
            *reinterpret_cast<const uint4*>(x + (((size_t)n * H + (oy >> 1)) * W + (ox >> 1)) * C + cv * V);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void up2_bwd_k(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int accumulate) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V, OW = 2 * W;
    const size_t total = (size_t)N * H * W * CV;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int cv, ix, iy; size_t p = divmod_idx(idx, CV, cv);
        p = divmod_idx(p, W, ix); const int n = (int)divmod_idx(p, H, iy);
        float a[V], b[V];
        const T* base = dy + (((size_t)n * 2 * H + 2 * iy) * OW + 2 * ix) * C + cv * V;
        ldv<T>(base, a);
        ldv<T>(base + C, b);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += b[e];
        ldv<T>(base + (size_t)OW * C, b);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += b[e];
        ldv<T>(base + (size_t)OW * C + C, b);
#pragma unroll
        for (int e = 0; e < V; ++e) a[e] += b[e];
        T* d = dx + (((size_t)n * H + iy) * W + ix) * C + cv * V;
        if (accumulate) { ldv<T>(d, b);
#pragma unroll
            for (int e = 0; e < V; ++e) a[e] += b[e]; }
        stv<T>(d, a);
    }
}

// ------------------------------------------------------------------------------------------ sum of up to 3 tensors through a channel permutation
// y[p][c] = a[p][perm[c]] (+ b[p][perm[c]] + c3[p][perm[c]])     (MSCB: dout = channel_shuffle(sum of the MSDC branches); backward: one gather)
template <typename T>
__global__ __launch_bounds__(256) void gather_sum_k(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ c3, const int* __restrict__ perm,
                                                    T* __restrict__ y, size_t M, int C) {
    const size_t total = M * C;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        int c; const size_t p = divmod_idx(idx, C, c);
        const size_t s = p * C + perm[c];
        float v = TT<T>::ld(a + s);
        if (b) v += TT<T>::ld(b + s);
        if (c3) v += TT<T>::ld(c3 + s);
        TT<T>::st(y + idx, v);
    }
}

// ------------------------------------------------------------------------------------------ sigmoid on small maps
template <typename Ti>
__global__ __launch_bounds__(256) void sigmoid_k(const Ti* __restrict__ x, float* __restrict__ y, size_t n, int ld, int C) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = TT<Ti>::ld(x + (i / C) * ld + i % C);
        y[i] = 1.f / (1.f + expf(-v));
    }
}
template <typename To>
__global__ __launch_bounds__(256) void sigmoid_bwd_k(const float* __restrict__ dy, const float* __restrict__ y, To* __restrict__ dx, size_t n, int ld, int C, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        To* d = dx + (i / C) * ld + i % C;
        const float g = dy[i] * y[i] * (1.f - y[i]);
        TT<To>::st(d, accumulate ? TT<To>::ld(d) + g : g);
    }
}

// ------------------------------------------------------------------------------------------ multi-class "mutation" loss (config 5)
// EMCAD/trainer.py:106-140 (dual, supervision='mutation') with utils/utils.py:102-138 (DiceLoss, softmax=True):
//   loss = sum over the 15 non-empty subsets s of the 4 scales of
//          lc1 * CE(sum_{i in s} fg_i, label) + lc2 * Dice(softmax(sum fg_i), onehot(label)) + lc3 * BCEWithLogits(sum_{i in s} bg_i, bg_mask)
// (hardware exp2 / log2 / rcp: the kernels are ALU-bound - ~300 transcendentals per pixel - and the sums are means over >= 10^6 terms)
// One pass over the 8 K-channel maps (NHWC fp32): a thread keeps its pixel's 8*K logits in registers and walks the 15 subsets; the per-subset
// sums (CE, BCE, per-class intersect and sum p^2) are reduced per wave and written as partial rows.  The backward recomputes the softmaxes.
constexpr int ML_NS = 15;
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row by rotations (row_ror:8/4/2/1 fold into v_add_f32_dpp - no LDS permutes); every lane gets the total
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0x128>(v); v += dpp_f<0x124>(v); v += dpp_f<0x122>(v); v += dpp_f<0x121>(v);
    return v;
}
template <int K> struct MLW { static constexpr int W = 2 + 2 * K; };      // values per subset: CE, BCE, I[K], Z[K]

struct ml_maps { const float* fg[4]; const float* bg[4]; float* dfg[4]; float* dbg[4]; };

template <int K>
__global__ __launch_bounds__(256) void mloss_fwd_k(ml_maps m, const long long* __restrict__ label, const float* __restrict__ bgm, size_t NP, size_t HW,
                                                   float* __restrict__ partial) {
    constexpr int W = MLW<K>::W;
    // per-value sums stop at the 16-lane rows (DPP adds only); the 16 row partials of the block meet in LDS.  A full 64-lane butterfly per value
    // needs two LDS permutes each, and with ~300 values per pixel their latency was most of this kernel.
    __shared__ float red[16][ML_NS * W + K];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, rrow = wid * 4 + (lane >> 4);
    const bool rlead = (lane & 15) == 0;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool ok = p < NP;
    // Round 6: foreground (softmax-coupled: CE, Dice sums) and background (BCE: every class on its own) parts apart, as in the backward - the background part runs
    // class by class with four logits and fifteen running sums live (205 -> ~110 registers, two -> four waves per SIMD); same sums, same order, bit for bit
    float f[4][K];
    int lab = -1;
    const size_t n = ok ? p / HW : 0, hw = ok ? p % HW : 0;
    if (ok) {
        lab = (int)label[p];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < K; ++k) f[i][k] = m.fg[i][p * K + k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {          // label histogram (sum of target^2 per class), once
        float t = (ok && lab == k) ? 1.f : 0.f;
        t = row16_sum(t);
        if (rlead) red[rrow][ML_NS * W + k] = t;
    }
#pragma unroll
    for (int s = 1; s <= ML_NS; ++s) {
        float z[K], v[W];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            z[k] = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (s >> i & 1) z[k] += f[i][k];
        }
        float mx = z[0];
#pragma unroll
        for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
        float se = 0.f, e[K];
#pragma unroll
        for (int k = 0; k < K; ++k) { e[k] = __expf(z[k] - mx); se += e[k]; }
        const float inv = __builtin_amdgcn_rcpf(se), lse = mx + __logf(se);
        float ce = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const float pk = e[k] * inv;
            if (lab == k) ce = lse - z[k];
            v[2 + k] = (lab == k) ? pk : 0.f;
            v[2 + K + k] = pk * pk;
        }
        v[0] = ce; v[1] = 0.f;
#pragma unroll
        for (int j = 0; j < W; ++j) if (j != 1) v[j] = row16_sum(ok ? v[j] : 0.f);
        if (rlead) {
#pragma unroll
            for (int j = 0; j < W; ++j) if (j != 1) red[rrow][(s - 1) * W + j] = v[j];
        }
    }
    {
        float bce[ML_NS];
#pragma unroll
        for (int s = 0; s < ML_NS; ++s) bce[s] = 0.f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            float b[4] = {0.f, 0.f, 0.f, 0.f}, mk = 0.f;
            if (ok) {
#pragma unroll
                for (int i = 0; i < 4; ++i) b[i] = m.bg[i][p * K + k];
                mk = bgm[(n * K + k) * HW + hw];
            }
#pragma unroll
            for (int s = 1; s <= ML_NS; ++s) {
                float zb = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (s >> i & 1) zb += b[i];
                bce[s - 1] += fmaxf(zb, 0.f) - zb * mk + __logf(1.f + __expf(-fabsf(zb)));
            }
        }
#pragma unroll
        for (int s = 0; s < ML_NS; ++s) {
            const float t = row16_sum(ok ? bce[s] : 0.f);
            if (rlead) red[rrow][s * W + 1] = t;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < ML_NS * W + K; j += 256) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][j];
        partial[(size_t)blockIdx.x * (ML_NS * W + K) + j] = t;
    }
}

// partial rows -> ML_RG group rows in double (every group sums a contiguous range of block rows, 4 independent chains per thread)
constexpr int ML_RG = 128;
template <int K>
__global__ __launch_bounds__(256) void mloss_reduce_k(const float* __restrict__ partial, int nblk, double* __restrict__ grp) {
    constexpr int NV = ML_NS * MLW<K>::W + K;
    const int rb = (nblk + gridDim.x - 1) / gridDim.x, r0 = blockIdx.x * rb, r1 = min(nblk, r0 + rb);
    for (int j = threadIdx.x; j < NV; j += 256) {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = r0;
        for (; r + 3 < r1; r += 4) {
            a0 += (double)partial[(size_t)r * NV + j]; a1 += (double)partial[(size_t)(r + 1) * NV + j];
            a2 += (double)partial[(size_t)(r + 2) * NV + j]; a3 += (double)partial[(size_t)(r + 3) * NV + j];
        }
        for (; r < r1; ++r) a0 += (double)partial[(size_t)r * NV + j];
        grp[(size_t)blockIdx.x * NV + j] = (a0 + a1) + (a2 + a3);
    }
}

// sums[ML_NS*W + K] (double-accumulated over the group rows) and the scalar loss
template <int K>
__global__ void mloss_finalize_k(const double* __restrict__ grp, int ngrp, float* __restrict__ sums, float* __restrict__ loss, double npix, float lc1, float lc2, float lc3) {
    constexpr int W = MLW<K>::W, NV = ML_NS * W + K;
    __shared__ double sh[NV];
    for (int j = threadIdx.x; j < NV; j += blockDim.x) {
        double a = 0.0;
        for (int r = 0; r < ngrp; ++r) a += grp[(size_t)r * NV + j];
        sh[j] = a; sums[j] = (float)a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int s = 0; s < ML_NS; ++s) {
            const double ce = sh[s * W] / npix, bce = sh[s * W + 1] / (npix * K);
            double dice = 0.0;
            for (int k = 0; k < K; ++k) dice += 1.0 - (2.0 * sh[s * W + 2 + k] + 1e-5) / (sh[s * W + 2 + K + k] + sh[ML_NS * W + k] + 1e-5);
            tot += lc1 * ce + lc2 * dice / K + lc3 * bce;
        }
        loss[0] = (float)tot;
    }
}

// Round 6: the kernel was register-bound, not memory-bound (256 VGPRs + 256 AGPRs + 708 bytes of scratch, ONE wave per SIMD: nothing hid its transcendental chains;
// staging the 36-byte pixel rows through LDS as coalesced float4s made it slower still - profiles/r06_emcad_notes.txt).  The foreground part (softmax over the K classes
// couples them: CE + Dice) and the background part (BCE: every class on its own) share nothing but the subset loop, so they run one after the other, the background
// part class by class with four logits and four gradient sums live: ~130 registers, three to four waves per SIMD.
template <int K>
__global__ __launch_bounds__(256) void mloss_bwd_k(ml_maps m, const long long* __restrict__ label, const float* __restrict__ bgm, size_t NP, size_t HW,
                                                   const float* __restrict__ sums, float gscale, float lc1, float lc2, float lc3) {
    constexpr int W = MLW<K>::W;
    // the Dice coefficients of every (subset, class) - uniform over the pixels - once per block in LDS: as 270 global loads per thread the compiler hoisted them
    // all to the top of the fully unrolled subset loop (256 + 256 registers and scratch)
    __shared__ float tA[ML_NS * K], tB[ML_NS * K];
    const float wdice = gscale * lc2 / K;
    for (int j = threadIdx.x; j < ML_NS * K; j += 256) {
        const int s0 = j / K, k = j - s0 * K;
        const float* S = sums + s0 * W;
        // d dice_k / d p_k = -(2 t / D - (2 I + eps) 2 p / D^2),  D = Z + T + eps
        const float D = S[2 + K + k] + sums[ML_NS * W + k] + 1e-5f;
        tA[j] = wdice * 2.f / D; tB[j] = wdice * (2.f * S[2 + k] + 1e-5f) * 2.f / (D * D);
    }
    __syncthreads();
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= NP) return;
    const int lab = (int)label[p];
    const size_t n = p / HW, hw = p % HW;
    const float wce = gscale * lc1 / (float)NP, wbce = gscale * lc3 / ((float)NP * K);
    {       // ---- foreground maps: d(CE + Dice) / d logits
        float f[4][K], gf[4][K];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < K; ++k) { f[i][k] = m.fg[i][p * K + k]; gf[i][k] = 0.f; }
#pragma unroll 1
        for (int s = 1; s <= ML_NS; ++s) {
            const float w0 = s & 1 ? 1.f : 0.f, w1 = s & 2 ? 1.f : 0.f, w2 = s & 4 ? 1.f : 0.f, w3 = s & 8 ? 1.f : 0.f;          // (x * 1 and + 0 are exact: the subset sums keep their order)
            float z[K];
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = ((w0 * f[0][k] + w1 * f[1][k]) + w2 * f[2][k]) + w3 * f[3][k];
            float mx = z[0];
#pragma unroll
            for (int k = 1; k < K; ++k) mx = fmaxf(mx, z[k]);
            float se = 0.f, pr[K], g[K], dot = 0.f;
#pragma unroll
            for (int k = 0; k < K; ++k) { pr[k] = __expf(z[k] - mx); se += pr[k]; }
            const float inv = __builtin_amdgcn_rcpf(se);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                pr[k] *= inv;
                g[k] = -((lab == k ? tA[(s - 1) * K + k] : 0.f) - tB[(s - 1) * K + k] * pr[k]);
                dot += g[k] * pr[k];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float dz = wce * (pr[k] - (lab == k ? 1.f : 0.f)) + pr[k] * (g[k] - dot);
                gf[0][k] += w0 * dz; gf[1][k] += w1 * dz; gf[2][k] += w2 * dz; gf[3][k] += w3 * dz;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < K; ++k) m.dfg[i][p * K + k] = gf[i][k];
    }
    // ---- background maps: d BCE / d logits, class by class (same subset order, same sums per (map, class) as before)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        float b[4], gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = m.bg[i][p * K + k];
        const float mk = bgm[(n * K + k) * HW + hw];
#pragma unroll
        for (int s = 1; s <= ML_NS; ++s) {
            float zb = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) if (s >> i & 1) zb += b[i];
            const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-zb)), dzb = wbce * (sg - mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) if (s >> i & 1) gb[i] += dzb;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) m.dbg[i][p * K + k] = gb[i];
    }
}

}  // namespace

#define EM_DISPATCH(dt, BODY) \
    if ((dt) == PN2_BF16) { typedef bf16_t T; BODY } else if ((dt) == PN2_F32) { typedef float T; BODY } else return -3;
#define EM_K(K_, BODY) \
    if ((K_) == 1) { constexpr int KK_ = 1; BODY } else if ((K_) == 3) { constexpr int KK_ = 3; BODY } else if ((K_) == 5) { constexpr int KK_ = 5; BODY } else return -2;

extern "C" {

// geometry of the K x K depth-wise walks.  kind 0: forward / data gradient, 1: weight gradient.  Channels per thread shrink with the window
// (registers: K*K weights or accumulators per channel); blocks.x is capped near 1024 because it is also the number of partial rows.
static int dwk_geometry(int dt, int kind, int N, int H, int W, int C, int K, int& VT, int& SEG, int& SPR, int& SPB, int& cvp, int& gx, int& gy) {
    const int vmax = dt == PN2_F32 ? 4 : 8;
    VT = K == 1 ? vmax : (K == 3 ? (kind ? 2 : 4) : 2);
    if (VT > vmax) VT = vmax;
    while (VT > 1 && C % VT) VT >>= 1;
    if (VT * (dt == PN2_F32 ? 4 : 2) < 4) return -2;          // at least one 32-bit word per thread (bf16: even channel counts)
    const int CV = C / VT;
    int target = 16;
    if ((long long)N * H * ((W + 15) / 16) * CV < 200000) target = 8;
    SPR = (W + target - 1) / target; SEG = (W + SPR - 1) / SPR;
    const int lanes = (dt == PN2_F32 ? 64 : 128) / VT;        // 256 contiguous bytes of one pixel per block
    cvp = CV >= lanes ? lanes : pow2ceil(CV);
    const int R = 256 / cvp, nseg = N * H * SPR;
    gy = (CV + cvp - 1) / cvp;
    int want = 2048 / gy; if (want < 1) want = 1; if (want > 1024) want = 1024;
    SPB = (nseg + want - 1) / want;
    SPB = ((SPB + R - 1) / R) * R;
    gx = (nseg + SPB - 1) / SPB;
    return 0;
}

/* rows of the BatchNorm partial buffers of pn2_dwconv (wgrad = 0) / of the partial buffer of pn2_dwconv_wgrad (wgrad = 1) */
int pn2_dwconv_blocks(int dt, int N, int H, int W, int C, int K, int wgrad) {
    if (N < 1 || H < 1 || W < 1 || (K != 1 && K != 3 && K != 5)) return -1;
    int VT, SEG, SPR, SPB, cvp, gx, gy;
    if (dwk_geometry(dt, wgrad ? 1 : 0, N, H, W, C, K, VT, SEG, SPR, SPB, cvp, gx, gy)) return -1;
    return gx;
}

#define EM_KV(K_, VT_, BODY) \
    if ((K_) == 1 && (VT_) == 8) { constexpr int KK_ = 1, VV_ = 8; BODY } else if ((K_) == 1 && (VT_) == 4) { constexpr int KK_ = 1, VV_ = 4; BODY } \
    else if ((K_) == 1 && (VT_) == 2) { constexpr int KK_ = 1, VV_ = 2; BODY } else if ((K_) == 3 && (VT_) == 4) { constexpr int KK_ = 3, VV_ = 4; BODY } \
    else if ((K_) == 3 && (VT_) == 2) { constexpr int KK_ = 3, VV_ = 2; BODY } else if ((K_) == 5 && (VT_) == 2) { constexpr int KK_ = 5, VV_ = 2; BODY } \
    else return -2;
#define EM_KV32(K_, VT_, BODY) \
    if ((K_) == 1 && (VT_) == 4) { constexpr int KK_ = 1, VV_ = 4; BODY } else if ((K_) == 1 && (VT_) == 2) { constexpr int KK_ = 1, VV_ = 2; BODY } \
    else if ((K_) == 1 && (VT_) == 1) { constexpr int KK_ = 1, VV_ = 1; BODY } else if ((K_) == 3 && (VT_) == 4) { constexpr int KK_ = 3, VV_ = 4; BODY } \
    else if ((K_) == 3 && (VT_) == 2) { constexpr int KK_ = 3, VV_ = 2; BODY } else if ((K_) == 3 && (VT_) == 1) { constexpr int KK_ = 3, VV_ = 1; BODY } \
    else if ((K_) == 5 && (VT_) == 2) { constexpr int KK_ = 5, VV_ = 2; BODY } else if ((K_) == 5 && (VT_) == 1) { constexpr int KK_ = 5, VV_ = 1; BODY } \
    else return -2;

/* depth-wise K x K conv (K = 1, 3, 5), pad K/2, stride 1, no bias; flip = data gradient; psum/psq: BatchNorm partial rows
 * [pn2_dwconv_blocks(dt, N, H, W, C, K, 0)][C] */
int pn2_dwconv(int dt, const void* x, const float* w, void* z, int N, int H, int W, int C, int K, int flip, int accumulate, float* psum, float* psq, void* stream) {
    if (!x || !w || !z || (psum && !psq)) return -1;
    int VT, SEG, SPR, SPB, cvp, gx, gy;
    if (dwk_geometry(dt, 0, N, H, W, C, K, VT, SEG, SPR, SPB, cvp, gx, gy)) return -2;
    const dim3 grid(gx, gy);
    const size_t lds = (size_t)256 * VT * 4;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) { EM_KV(K, VT, { hipLaunchKernelGGL((dwconv_row_k<bf16_t, KK_, VV_>), grid, dim3(256), lds, st, (const bf16_t*)x, w, (bf16_t*)z, N, H, W, C, flip, accumulate,
                                                            psum, psq, SEG, SPR, SPB, cvp); }) }
    else if (dt == PN2_F32) { EM_KV32(K, VT, { hipLaunchKernelGGL((dwconv_row_k<float, KK_, VV_>), grid, dim3(256), lds, st, (const float*)x, w, (float*)z, N, H, W, C, flip, accumulate,
                                                                  psum, psq, SEG, SPR, SPB, cvp); }) }
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

/* partial[pn2_dwconv_blocks(dt, N, H, W, C, K, 1)][C*K*K] of the depth-wise weight gradient; finish with pn2_colsum_finalize(partial, nblk, C*K*K, C*K*K, dW, acc) */
int pn2_dwconv_wgrad(int dt, const void* dz, const void* x, float* partial, int N, int H, int W, int C, int K, void* stream) {
    if (!dz || !x || !partial) return -1;
    int VT, SEG, SPR, SPB, cvp, gx, gy;
    if (dwk_geometry(dt, 1, N, H, W, C, K, VT, SEG, SPR, SPB, cvp, gx, gy)) return -2;
    const dim3 grid(gx, gy);
    const size_t lds = (size_t)256 * VT * 4;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) { EM_KV(K, VT, { hipLaunchKernelGGL((dwconv_wgrad_row_k<bf16_t, KK_, VV_>), grid, dim3(256), lds, st, (const bf16_t*)dz, (const bf16_t*)x, partial,
                                                            N, H, W, C, SEG, SPR, SPB, cvp); }) }
    else if (dt == PN2_F32) { EM_KV32(K, VT, { hipLaunchKernelGGL((dwconv_wgrad_row_k<float, KK_, VV_>), grid, dim3(256), lds, st, (const float*)dz, (const float*)x, partial,
                                                                  N, H, W, C, SEG, SPR, SPB, cvp); }) }
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

// geometry of the pair-conv walks: VT outputs per thread (one 16-byte vector of 2*VT inputs), segments as the depth-wise kernels
static int pc_geometry(int dt, int N, int H, int W, int F, int& VT, int& SEG, int& SPR, int& SPB, int& cvp, int& gx, int& gy) {
    VT = dt == PN2_F32 ? 2 : 4;
    if (F % VT || N < 1 || H < 1 || W < 1) return -2;
    const int FV = F / VT;
    int target = 16;
    if ((long long)N * H * ((W + 15) / 16) * FV < 200000) target = 8;
    SPR = (W + target - 1) / target; SEG = (W + SPR - 1) / SPR;
    cvp = FV >= 32 ? 32 : pow2ceil(FV);
    const int R = 256 / cvp, nseg = N * H * SPR;
    gy = (FV + cvp - 1) / cvp;
    int want = 2048 / gy; if (want < 1) want = 1; if (want > 1024) want = 1024;
    SPB = (nseg + want - 1) / want;
    SPB = ((SPB + R - 1) / R) * R;
    gx = (nseg + SPB - 1) / SPB;
    return 0;
}

/* rows of the BatchNorm partial buffers of pn2_pairconv3x3_fwd and of the partial buffer of pn2_pairconv3x3_wgrad */
int pn2_pairconv_blocks(int dt, int N, int H, int W, int F) {
    int VT, SEG, SPR, SPB, cvp, gx, gy;
    if (pc_geometry(dt, N, H, W, F, VT, SEG, SPR, SPB, cvp, gx, gy)) return -1;
    return gx;
}

/* grouped 3x3 conv, groups = F, 2 input channels per group (LGAG.W_g / W_x), pad 1, bias-free here (the bias is folded by the caller):
 * x [M][2F] -> z [M][F] + BN partial rows [pn2_pairconv_blocks(dt, N, H, W, F)][F] ; w [F][2][9] fp32 */
int pn2_pairconv3x3_fwd(int dt, const void* x, const float* w, void* z, int N, int H, int W, int F, float* psum, float* psq, void* stream) {
    if (!x || !w || !z || !psum || !psq) return -1;
    int VT, SEG, SPR, SPB, cvp, gx, gy;
    if (pc_geometry(dt, N, H, W, F, VT, SEG, SPR, SPB, cvp, gx, gy)) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL((pairconv_fwd_k<bf16_t, 4>), dim3(gx, gy), dim3(256), 256 * 4 * 4, st, (const bf16_t*)x, w, (bf16_t*)z, N, H, W, F, psum, psq, SEG, SPR, SPB, cvp);
    else if (dt == PN2_F32) hipLaunchKernelGGL((pairconv_fwd_k<float, 2>), dim3(gx, gy), dim3(256), 256 * 2 * 4, st, (const float*)x, w, (float*)z, N, H, W, F, psum, psq, SEG, SPR, SPB, cvp);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_pairconv3x3_dgrad(int dt, const void* dz, const float* w, void* dx, int N, int H, int W, int F, int accumulate, void* stream) {
    if (!dz || !w || !dx) return -1;
    int VT, SEG, SPR, SPB, cvp, gx, gy;
    if (pc_geometry(dt, N, H, W, F, VT, SEG, SPR, SPB, cvp, gx, gy)) return -2;
    const int R = 256 / cvp, nseg = N * H * SPR;
    const dim3 grid((nseg + R - 1) / R, gy);
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL((pairconv_dgrad_k<bf16_t, 4>), grid, dim3(256), 0, st, (const bf16_t*)dz, w, (bf16_t*)dx, N, H, W, F, accumulate, SEG, SPR, cvp);
    else if (dt == PN2_F32) hipLaunchKernelGGL((pairconv_dgrad_k<float, 2>), grid, dim3(256), 0, st, (const float*)dz, w, (float*)dx, N, H, W, F, accumulate, SEG, SPR, cvp);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

/* partial[pn2_pairconv_blocks(dt, N, H, W, F)][F*18] ; finish with pn2_colsum_finalize */
int pn2_pairconv3x3_wgrad(int dt, const void* dz, const void* x, float* partial, int N, int H, int W, int F, void* stream) {
    if (!dz || !x || !partial) return -1;
    int VT, SEG, SPR, SPB, cvp, gx, gy;
    if (pc_geometry(dt, N, H, W, F, VT, SEG, SPR, SPB, cvp, gx, gy)) return -2;
    hipStream_t st = (hipStream_t)stream;
    if (dt == PN2_BF16) hipLaunchKernelGGL((pairconv_wgrad_k<bf16_t, 4>), dim3(gx, gy), dim3(256), 256 * 4 * 4, st, (const bf16_t*)dz, (const bf16_t*)x, partial, N, H, W, F, SEG, SPR, SPB, cvp);
    else if (dt == PN2_F32) hipLaunchKernelGGL((pairconv_wgrad_k<float, 2>), dim3(gx, gy), dim3(256), 256 * 2 * 4, st, (const float*)dz, (const float*)x, partial, N, H, W, F, SEG, SPR, SPB, cvp);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

/* y (+)= x * gate ; mode 0: gate [N][C] (CAB), mode 1: gate [N][HW] (SAB, LGAG) ; gate fp32.  Also the data gradient (x := dy). */
int pn2_gate_mul(int dt, const void* x, const float* gate, void* y, int N, int HW, int C, int mode, int accumulate, void* stream) {
    if (!x || !gate || !y) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -2;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(gate_mul_k<T>, dim3(grid_for((size_t)N * HW * (C / V))), dim3(256), 0, (hipStream_t)stream, (const T*)x, gate, (T*)y, N, HW, C, mode, accumulate); })
    PN2_CHECK_LAUNCH();
    return 0;
}

/* gate gradient.  mode 1: dgate [N][HW] = sum_c dy*x, written directly.  mode 0: partial [pn2_gate_blocks(dt, HW, C)][N*C] rows, finish with
 * pn2_colsum_finalize(partial, nblk, N*C, N*C, dgate, acc) */
int pn2_gate_blocks(int dt, int HW, int C) {
    const int V = dt == PN2_F32 ? 4 : 8;
    if (HW < 1 || C % V) return -1;
    int cvp, pix, nblk; walk_geometry(HW, C / V, cvp, pix, nblk);
    return nblk;
}

int pn2_gate_bwd(int dt, const void* dy, const void* x, float* dgate_or_partial, int N, int HW, int C, int mode, void* stream) {
    if (!dy || !x || !dgate_or_partial) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -2;
    if (mode == 1) {
        EM_DISPATCH(dt, { hipLaunchKernelGGL(gate_dpix_k<T>, dim3(grid_for((size_t)N * HW * 64)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)x, dgate_or_partial, (size_t)N * HW, C); })
    } else {
        int cvp, pix, nblk; walk_geometry(HW, C / V, cvp, pix, nblk);
        EM_DISPATCH(dt, { hipLaunchKernelGGL(gate_dchan_k<T>, dim3(nblk, N), dim3(256), 256 * TT<T>::VEC * 4, (hipStream_t)stream, (const T*)dy, (const T*)x, dgate_or_partial, N, HW, C, pix, cvp); })
    }
    PN2_CHECK_LAUNCH();
    return 0;
}

/* nn.AdaptiveAvgPool2d(1) and nn.AdaptiveMaxPool2d(1) in one pass (CAB): avg, mx [N][C] in the compute dtype, arg [N][C] = argmax pixel */
int pn2_global_pool(int dt, const void* x, void* avg, void* mx, int* arg, int N, int HW, int C, void* stream) {
    if (!x || !avg || !mx || !arg) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -2;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(global_pool_k<T>, dim3((C / V + 7) / 8, N), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)avg, (T*)mx, arg, HW, C); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_global_pool_bwd(int dt, const void* davg, const void* dmax, const int* arg, void* dx, int N, int HW, int C, int accumulate, void* stream) {
    if (!davg || !dmax || !arg || !dx) return -1;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(global_pool_bwd_k<T>, dim3(grid_for((size_t)N * HW * C)), dim3(256), 0, (hipStream_t)stream, (const T*)davg, (const T*)dmax, arg, (T*)dx, N, HW, C, accumulate); })
    PN2_CHECK_LAUNCH();
    return 0;
}

/* SAB input: out [NP][8] = (mean over channels, max over channels, 0 x 6), arg [NP] = argmax channel */
int pn2_chan_stats(int dt, const void* x, void* out8, int* arg, long long NP, int C, void* stream) {
    if (!x || !out8 || !arg) return -1;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(chan_stats_k<T>, dim3(grid_for((size_t)NP * 64)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)out8, arg, (size_t)NP, C); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_chan_stats_bwd(int dt, const void* dout8, const int* arg, void* dx, long long NP, int C, int accumulate, void* stream) {
    if (!dout8 || !arg || !dx) return -1;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(chan_stats_bwd_k<T>, dim3(grid_for((size_t)NP * C)), dim3(256), 0, (hipStream_t)stream, (const T*)dout8, arg, (T*)dx, (size_t)NP, C, accumulate); })
    PN2_CHECK_LAUNCH();
    return 0;
}

/* nn.Upsample(scale_factor=2) (nearest) and its adjoint */
int pn2_upsample_nearest2x(int dt, const void* x, void* y, int N, int H, int W, int C, void* stream) {
    if (!x || !y) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -2;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(up2_k<T>, dim3(grid_for((size_t)N * 4 * H * W * (C / V))), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, N, H, W, C); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_upsample_nearest2x_bwd(int dt, const void* dy, void* dx, int N, int H, int W, int C, int accumulate, void* stream) {
    if (!dy || !dx) return -1;
    const int V = dt == PN2_F32 ? 4 : 8;
    if (C % V) return -2;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(up2_bwd_k<T>, dim3(grid_for((size_t)N * H * W * (C / V))), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, N, H, W, C, accumulate); })
    PN2_CHECK_LAUNCH();
    return 0;
}

/* y[p][c] = a[p][perm[c]] + b[p][perm[c]] + c[p][perm[c]]   (b, c optional): the MSDC branch sum written through channel_shuffle, and (with the
 * inverse permutation, b = c = null) its adjoint */
int pn2_gather_sum(int dt, const void* a, const void* b, const void* c, const int* perm, void* y, long long M, int C, void* stream) {
    if (!a || !perm || !y) return -1;
    EM_DISPATCH(dt, { hipLaunchKernelGGL(gather_sum_k<T>, dim3(grid_for((size_t)M * C)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, (const T*)c, perm, (T*)y, (size_t)M, C); })
    PN2_CHECK_LAUNCH();
    return 0;
}

/* y = sigmoid(x) as fp32 [n] from a [rows][ld] map with C used channels; dx (+)= dy * y * (1 - y) */
int pn2_sigmoid(int dt_in, const void* x, int ld, int C, float* y, long long n, void* stream) {
    if (!x || !y || n < 1) return -1;
    EM_DISPATCH(dt_in, { hipLaunchKernelGGL(sigmoid_k<T>, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, (const T*)x, y, (size_t)n, ld, C); })
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_sigmoid_bwd(int dt_out, const float* dy, const float* y, void* dx, int ld, int C, long long n, int accumulate, void* stream) {
    if (!dy || !y || !dx || n < 1) return -1;
    EM_DISPATCH(dt_out, { hipLaunchKernelGGL(sigmoid_bwd_k<T>, dim3(grid_for((size_t)n)), dim3(256), 0, (hipStream_t)stream, dy, y, (T*)dx, (size_t)n, ld, C, accumulate); })
    PN2_CHECK_LAUNCH();
    return 0;
}

static int ml_blocks(long long npix) { return (int)((npix + 255) / 256); }
/* rows of the `partial` scratch: one per 256 pixels + the double-precision group rows of the second reduction level */
int pn2_mutation_loss_blocks(long long npix) { return npix < 1 ? -1 : ml_blocks(npix) + 2 * ML_RG + 1; }
int pn2_mutation_loss_width(int K) { return K == 9 ? ML_NS * MLW<9>::W + 9 : -1; }

/* EMCAD/trainer.py:106-140: sum over the 15 non-empty subsets of the 4 scales of lc1*CE + lc2*Dice(softmax) + lc3*BCEWithLogits on the summed maps.
 * fg[4], bg[4]: [N][H][W][K] fp32 maps (K = 9); label [N][H][W] int64; bg_mask [N][K][H][W] fp32.  partial: [pn2_mutation_loss_blocks][pn2_mutation_loss_width]
 * scratch; sums [pn2_mutation_loss_width] is kept for the backward; loss[1]. */
int pn2_mutation_loss_fwd(const float* const* fg, const float* const* bg, const long long* label, const float* bg_mask, int N, long long HW, int K,
                          float lc1, float lc2, float lc3, float* partial, float* sums, float* loss, void* stream) {
    if (!fg || !bg || !label || !bg_mask || !partial || !sums || !loss) return -1;
    if (K != 9) return -2;
    ml_maps m;
    for (int i = 0; i < 4; ++i) { m.fg[i] = fg[i]; m.bg[i] = bg[i]; m.dfg[i] = nullptr; m.dbg[i] = nullptr; if (!fg[i] || !bg[i]) return -1; }
    const size_t NP = (size_t)N * HW;
    const int nblk = ml_blocks((long long)NP), NV = ML_NS * MLW<9>::W + 9;
    const int ngrp = nblk < 8 * ML_RG ? (nblk + 7) / 8 : ML_RG;
    double* grp = reinterpret_cast<double*>(partial + (((size_t)nblk * NV + 1) & ~(size_t)1));          // 8-byte aligned, after the block rows
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mloss_fwd_k<9>, dim3(nblk), dim3(256), 0, st, m, label, bg_mask, NP, (size_t)HW, partial);
    hipLaunchKernelGGL(mloss_reduce_k<9>, dim3(ngrp), dim3(256), 0, st, partial, nblk, grp);
    hipLaunchKernelGGL(mloss_finalize_k<9>, dim3(1), dim3(256), 0, st, grp, ngrp, sums, loss, (double)NP, lc1, lc2, lc3);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_mutation_loss_bwd(const float* const* fg, const float* const* bg, float* const* dfg, float* const* dbg, const long long* label, const float* bg_mask,
                          int N, long long HW, int K, float lc1, float lc2, float lc3, const float* sums, float gscale, void* stream) {
    if (!fg || !bg || !dfg || !dbg || !label || !bg_mask || !sums) return -1;
    if (K != 9) return -2;
    ml_maps m;
    for (int i = 0; i < 4; ++i) { m.fg[i] = fg[i]; m.bg[i] = bg[i]; m.dfg[i] = dfg[i]; m.dbg[i] = dbg[i]; if (!fg[i] || !bg[i] || !dfg[i] || !dbg[i]) return -1; }
    const size_t NP = (size_t)N * HW;
    hipLaunchKernelGGL(mloss_bwd_k<9>, dim3((unsigned)((NP + 255) / 256)), dim3(256), 0, (hipStream_t)stream, m, label, bg_mask, NP, (size_t)HW, sums, gscale, lc1, lc2, lc3);
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
