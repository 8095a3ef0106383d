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
#include <cstring>
#include "pn2_common.h"
#include "../../include/pn2.h"

namespace {

constexpr int ROWB = 128;      // bytes of K per LDS row per step (64 bf16 or 32 f32): two 64-byte MFMA sub-steps
constexpr int KSUB = ROWB / 64;
constexpr int RS = ROWB;       // LDS row stride of the gather kernels (bytes): no padding - the eight 16-byte slots of a row are XOR-swizzled with (row & 7).  ds_read_b128 is served in
                               // lane groups {0-3, 12-15, 20-27}, ... (not 16 consecutive lanes): with the 144-byte padded rows used before, a group's g = 0 and g = 1 lanes met on 7 of 8
                               // slots (SQ_LDS_BANK_CONFLICT = 30 % of SQ_LDS_IDX_ACTIVE on the fp32fast GEMMs); the swizzle is conflict-free for the reads and the stores, and 11 % smaller

// Phase stamps of the conv kernels (debug build only: make stamp -> libpn2_stamp.so, read by tools/stamp_micro.py): thread 0 of every workgroup
// leaves s_memtime at the phase boundaries, so that a launch can be taken apart into prologue / first DMA landing / K loop / C staging / stores.
#ifdef PN2_STAMP
constexpr int PN2_STAMP_SLOTS = 16, PN2_STAMP_BLOCKS = 1 << 16;
__device__ unsigned long long pn2_stamp_buf[PN2_STAMP_BLOCKS * PN2_STAMP_SLOTS];
__device__ __forceinline__ void pn2_stamp(int i) {
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < PN2_STAMP_BLOCKS) pn2_stamp_buf[blockIdx.x * PN2_STAMP_SLOTS + i] = __builtin_amdgcn_s_memtime();
}
__device__ __forceinline__ void pn2_stamp_hw() {
    if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < PN2_STAMP_BLOCKS) {
        pn2_stamp_buf[blockIdx.x * PN2_STAMP_SLOTS + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
        pn2_stamp_buf[blockIdx.x * PN2_STAMP_SLOTS + 11] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // XCC_ID
    }
}
#define PN2_STAMP_AT(i) pn2_stamp(i)
#else
#define PN2_STAMP_AT(i)
#endif

template <typename T> struct MMA;
template <> struct MMA<bf16_t> {
    static constexpr int BK = 32 * KSUB;
    static constexpr bool F64ROWS = false, DEEP = false;
    typedef f32x4_t acc_t;
    __device__ static __forceinline__ int arow(int l15) { return l15; }
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
    template <int MT, int NT> __device__ static __forceinline__ void run_block(acc_t (&acc)[MT][NT], const uint4 (&a)[MT], const uint4 (&b)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) run(acc[i][j], a[i], b[j]);
    }
};
// PN2_F32 (the parity path): fp32 storage, products and sums in DOUBLE on v_mfma_f64_16x16x4_f64, one rounding to fp32 per output - every conv
// result is the correctly rounded fp32 value of the exact contraction, so the only error left against a float64 run of the reference is the
// fp32 storage of activations (a k-ordered fp32 fma chain over K up to 6400 was 2-4x less accurate than the reference's own blocked fp32 sums).
// C/D of the f64 form is row = (lane>>4) + 4*reg (not (lane>>4)*4 + reg as in every other MFMA): A rows are fed through the 4x4 index
// transpose arow() so that the accumulators land in the common layout and the shared epilogue applies unchanged.
template <> struct MMA<float> {
    static constexpr int BK = 16 * KSUB;
    static constexpr bool F64ROWS = true, DEEP = false;
    typedef f64x4_t acc_t;
    __device__ static __forceinline__ int arow(int l15) { return ((l15 & 3) << 2) | (l15 >> 2); }
    // lane (g = lane>>4) holds k = 4g..4g+3 of this 16-deep sub-step; MFMA j contracts {j, 4+j, 8+j, 12+j}
    __device__ static __forceinline__ void run(f64x4_t& acc, const uint4& a, const uint4& b) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)__uint_as_float(a.x), (double)__uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)__uint_as_float(a.y), (double)__uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)__uint_as_float(a.z), (double)__uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)__uint_as_float(a.w), (double)__uint_as_float(b.w), acc, 0, 0, 0);
    }
    template <int MT, int NT> __device__ static __forceinline__ void run_block(acc_t (&acc)[MT][NT], const uint4 (&a)[MT], const uint4 (&b)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) run(acc[i][j], a[i], b[j]);
    }
};
// PN2_F32F ("fp32fast"): the reference's own arithmetic - fp32 operands, fp32 products and sums - on the f32 matrix pipe (v_mfma_f32_16x16x4_f32: twice the rate of
// the f64 form).  Round 1 ran ONE k-ordered MFMA chain per output and sat 2.5-3 x further from float64 than the reference's blocked fp32 sums: the matrix
// core's internal adds do not round to nearest, so a long chain drifts.  Here a chain is 4 MFMAs long (the 16 k-values of a sub-step, started from C = 0);
// the chunk sums meet in a round-to-nearest VALU add (K / 16 adds per output, unbiased), which costs 4 v_add_f32 per 4 MFMAs (128 matrix-pipe cycles).
// C/D layout is the common one (row = (lane>>4)*4 + reg): no A-row transpose.
template <> struct MMA<f32f_t> {
    static constexpr int BK = 16 * KSUB;
    static constexpr bool F64ROWS = false, DEEP = true;          // DEEP: the gather kernel prefetches two K-steps ahead (conv_gather_body)
    typedef f32x4_t acc_t;
    __device__ static __forceinline__ int arow(int l15) { return l15; }
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) {
        f32x4_t t = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), t, 0, 0, 0);
        t = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), t, 0, 0, 0);
        acc += t;
    }
    // a wave's MT x NT blocks of one sub-step, k-major: the MT * NT chains are independent, so consecutive MFMAs never wait for each other's result (issued
    // block by block as run() would, every MFMA of a chain waited for the one before it and every add for its chain: s_nop 9 per block in the ISA)
    template <int MT, int NT> __device__ static __forceinline__ void run_block(acc_t (&acc)[MT][NT], const uint4 (&a)[MT], const uint4 (&b)[NT]) {
        f32x4_t t[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].x), __uint_as_float(b[j].x), f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].y), __uint_as_float(b[j].y), t[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].z), __uint_as_float(b[j].z), t[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) t[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[i].w), __uint_as_float(b[j].w), t[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] += t[i][j];
    }
};


// 16 channels x 32 rows of a row-major bf16 tile in LDS as an MFMA operand (lane: channel l15, k-slots (g, e) <-> row (e>>2)*16 + g*4 + (e&3)):
// two transposing reads.  Used by the statistics of the swapped epilogue and by the weight-gradient kernels (same k-slot order on both operands).
typedef short s16x4_t_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 tr_frag_bf16(const char* tile, int rs, int chan0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const char* p = tile + (g * 4 + (i >> 2)) * rs + (chan0 + (i & 3) * 4) * 2;
    s16x4_t_ v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t_ __attribute__((address_space(3)))*)(p));
    s16x4_t_ v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t_ __attribute__((address_space(3)))*)(p + 16 * rs));
    uint2 lo = __builtin_bit_cast(uint2, v0), hi = __builtin_bit_cast(uint2, v1);
    return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

struct GatherGeom {
    int H, W, OH, OW, KH, KW, stride, sshift, pad_h, pad_w, dil_h, dil_w, transposed;
};

// resolve a tap of an output pixel (iy0 / ix0 precomputed) to an input pixel; returns false if padding.  The tap's dilated offsets (r * dil_h, s * dil_w)
// are carried by the caller, which steps them with the tap instead of dividing the tap index
__device__ __forceinline__ bool tap_pixel_off(const GatherGeom& g, int iy0, int ix0, int dr, int dc, int& iy, int& ix) {
    if (!g.transposed) {
        iy = iy0 + dr; ix = ix0 + dc;
        return (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
    }
    const int ty = iy0 - dr, tx = ix0 - dc;
    const int msk = g.stride - 1;
    iy = ty >> g.sshift; ix = tx >> g.sshift;
    return ty >= 0 && tx >= 0 && !(ty & msk) && !(tx & msk) && iy < g.H && ix < g.W;
}

// ------------------------------------------------------------------------------------------------
// Shared epilogue of the two GEMM kernels: accumulators -> (bias) -> LDS-staged 16-byte stores, with
//   PN2_CONV_STATS : forward BatchNorm batch statistics as one (mean, M2) pair per tile and channel; pn2_bn_finalize merges the tiles in double (Chan).
//                    bf16: of the STORED (rounded) tile, on the matrix cores (mfma_stats below) - one definition for every bf16 kernel.
//                    fp32: of the accumulators; a wave shifts its values by its own first row (no cancellation however large |mean| / sigma is),
//                    the WM wave results of a tile are merged with Chan's formula.
//   pn2_conv_ep    : BatchNorm-BACKWARD partial sums of the gradient tile this dgrad GEMM produces (see pn2.h), one or two targets: register form
//                    (fp32, wide bf16 tiles; next block) or LDS-DMA / matrix-core form (narrow bf16 tiles; ep2_* further down).
// LDS use: the C tile [BM][CRS]; fp32 statistics: + 3*WM*BN floats; register-form ep sums reuse the C tile area after it has been drained; ep2: see Ep2Layout.
// ------------------------------------------------------------------------------------------------
// BatchNorm-backward epilogue (pn2_conv_gemm_ep).  A thread owns ONE channel vector of the C tile (256 % VPR == 0) and RPT of its rows, so its
// per-channel parameters and sums stay in registers.  This code runs at the GEMM's low occupancy (1-3 workgroups per CU), where every
// instruction is exposed: a first, straightforward version executed ~1800 instructions per wave and DOUBLED the kernel time.  Hence:
//   * all global operands of a thread's rows (raw, stored y, the destination for +=) are requested in ONE batch at the top of the epilogue,
//     before the C tile is staged through LDS (a load -> use -> store walk would pay a full HBM latency per row);
//   * the ReLU mask is branch-free: keep = fmaf(ms, msc, msh) > 0 with (ms, msc, msh) = (raw, scale, shift) | (stored y, 1, 0) | (raw, 0, 1);
//   * s2 = invstd * (sum dz*raw - mean * sum dz): one fma per element in the walk;
//   * bounds checks and the += path are template parameters (full tiles / plain stores take the lean loop);
//   * the row lanes of a channel meet in LDS (one 16-byte write per lane and sum, column-parallel fixed-order adds) instead of ~100 ds_bpermute.
template <typename T, int BM, int BN> struct BnbPre {
    static constexpr int VEC = TT<T>::VEC, VPR = BN / VEC;
    static constexpr int RPT = BM * VPR / 256 > 0 ? BM * VPR / 256 : 1;
    static_assert(BM * VPR >= 256 && (BM * VPR) % 256 == 0, "tile rows must split evenly over the 256 / VPR row lanes");
    uint4 ra[RPT], ma[RPT], rb[RPT], vd[RPT];
};
constexpr int ep_lds_bytes(int vec) { return 4 * 256 * vec * 4; }      // [4 sums][256 / VPR row lanes][BN] floats

__device__ __forceinline__ void bnb_select(const pn2_bnb_target& t, int col, const void*& raw, const float*& par, bool& stat) {
    raw = t.raw; par = t.par;
    stat = (t.mode & PN2_BNB_STATS) != 0;
    if (t.split > 0 && col >= t.split) { raw = t.raw2; par = t.par2; if (!par) stat = false; }
}

// pn2_conv_ep.pool: the prior content of a += destination comes from a QUARTER-resolution tensor (row (n, y/2, x/2) for output row (n, y, x)), scaled by 1/4 -
// the backward of AvgPool2d(2, 2) (Res2Net_v1b.py:127-136: the downsample branch of a stage block) folded into the dgrad epilogue that completes the gradient
__device__ __forceinline__ int pool_row(const pn2_conv_desc& d, int m) {
    const int hw = d.OH * d.OW, n = m / hw, rem = m - n * hw, y = rem / d.OW, x = rem - y * d.OW;
    return (n * (d.OH >> 1) + (y >> 1)) * (d.OW >> 1) + (x >> 1);
}

template <typename T, int BM, int BN>
__device__ __forceinline__ void bnb_prefetch(const pn2_conv_desc& d, const pn2_conv_ep& ep, const T* out, int M, int m0, int n0, BnbPre<T, BM, BN>& P) {
    constexpr int VEC = TT<T>::VEC, VPR = BN / VEC, RPT = BnbPre<T, BM, BN>::RPT;
    const int tid = threadIdx.x, cv = tid % VPR, col = n0 + cv * VEC, rstep = 256 / VPR;
    const bool cok = col < d.Cout;
    const int colc = cok ? col : 0;
    const void *rawA, *rawB; const float *parA, *parB; bool stA, stB;
    bnb_select(ep.a, colc, rawA, parA, stA);
    bnb_select(ep.b, colc, rawB, parB, stB);
    const bool yA = stA && (ep.a.mode & PN2_BNB_MASK_Y), acc = d.flags & PN2_CONV_ACCUM;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int row = tid / VPR + u * rstep;
        const size_t mc = (cok && m0 + row < M) ? (size_t)(m0 + row) : 0;      // clamped: the loads are unconditional, results of dead rows are ignored
        if (stA) P.ra[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(rawA) + mc * ep.a.ld_raw + colc);
        if (yA) P.ma[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(ep.a.y) + mc * ep.a.ld_y + colc);
        if (stB) P.rb[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(rawB) + mc * ep.b.ld_raw + colc);
        if (acc) {
            if (ep.pool) P.vd[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const T*>(ep.pool) + (size_t)pool_row(d, (int)mc) * ep.ld_pool + colc);
            else P.vd[u] = *reinterpret_cast<const uint4*>(out + mc * d.ld_out + colc);
        }
    }
}

// mask coefficients + (mean, invstd) of this thread's channel vector
template <typename T>
__device__ __forceinline__ void bnb_params(const pn2_bnb_target& t, const float* par, int colc, bool stat, float* msc, float* msh, float* mu, float* is) {
    constexpr int VEC = TT<T>::VEC;
    const bool my = (t.mode & PN2_BNB_MASK_Y) != 0, mr = (t.mode & PN2_BNB_MASK_RAW) != 0;
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = make_float4(1.f, 1.f, 1.f, 1.f), c = a, dd = a;
        if (stat) {
            if (mr && !my) { a = *reinterpret_cast<const float4*>(par + colc + e); b = *reinterpret_cast<const float4*>(par + (size_t)t.ps + colc + e); }
            c = *reinterpret_cast<const float4*>(par + (size_t)2 * t.ps + colc + e); dd = *reinterpret_cast<const float4*>(par + (size_t)3 * t.ps + colc + e);
        }
        if (my) { a = make_float4(1.f, 1.f, 1.f, 1.f); b = make_float4(0.f, 0.f, 0.f, 0.f); }
        msc[e] = a.x; msc[e + 1] = a.y; msc[e + 2] = a.z; msc[e + 3] = a.w; msh[e] = b.x; msh[e + 1] = b.y; msh[e + 2] = b.z; msh[e + 3] = b.w;
        mu[e] = c.x; mu[e + 1] = c.y; mu[e + 2] = c.z; mu[e + 3] = c.w; is[e] = dd.x; is[e + 1] = dd.y; is[e + 2] = dd.z; is[e + 3] = dd.w;
    }
}

// One target: walk this thread's rows of the staged C tile, store them to dst (+= vd when ACC) and accumulate t1 = sum dz, t2 = sum dz * raw
template <typename T, int BM, int BN, bool ACC, bool FULL>
__device__ __forceinline__ void bnb_rows(T* __restrict__ dst, int ld_dst, const char* Cs, int M, int m0, int col, int cv, bool cok, bool stat, bool usey, bool smask,
                                         const uint4* vr, const uint4* vm, const uint4* vd, const float* msc, const float* msh, float* t1, float* t2, float ps) {
    constexpr int VEC = TT<T>::VEC, VPR = BN / VEC, CRS = BN * (int)sizeof(T) + 16, RPT = BnbPre<T, BM, BN>::RPT;
    const int r0 = threadIdx.x / VPR;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int row = r0 + u * (256 / VPR);
        const int m = m0 + row;
        if constexpr (!FULL) { if (!(cok && m < M)) continue; }
        uint4 o = *reinterpret_cast<const uint4*>(Cs + row * CRS + cv * 16);
        float x[VEC];
        TT<T>::unpack(o, x);
        if constexpr (ACC) {
            float y[VEC];
            TT<T>::unpack(vd[u], y);
#pragma unroll
            for (int e = 0; e < VEC; ++e) x[e] = __fmaf_rn(y[e], ps, x[e]);          // ps = 1 (fmaf(y, 1, x) == x + y) or 1/4 (pooled prior)
            o = TT<T>::pack(x);
            TT<T>::unpack(o, x);                  // the statistics see the STORED (rounded) gradient, as a separate reduce pass would
        }
        if (!smask) *reinterpret_cast<uint4*>(dst + (size_t)m * ld_dst + col) = o;
        if (stat) {
            float xr[VEC], ms[VEC];
            TT<T>::unpack(vr[u], xr);
            TT<T>::unpack(usey ? vm[u] : vr[u], ms);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float dz = fmaf(ms[e], msc[e], msh[e]) > 0.f ? x[e] : 0.f;
                t1[e] += dz; t2[e] = fmaf(dz, xr[e], t2[e]);
                x[e] = dz;
            }
        }
        if (smask) *reinterpret_cast<uint4*>(dst + (size_t)m * ld_dst + col) = TT<T>::pack(x);      // PN2_BNB_STORE_MASKED (only set together with statistics)
    }
}

template <typename T, int BM, int BN>
__device__ __forceinline__ void bnb_target(const pn2_bnb_target& t, T* __restrict__ dst, int ld_dst, bool accum, const char* Cs, int M, int m0, int n0, int Cout,
                                           const uint4* vr, const uint4* vm, const uint4* vd, float* s1, float* s2, float ps = 1.f) {
    constexpr int VEC = TT<T>::VEC, VPR = BN / VEC;
    const int cv = threadIdx.x % VPR, col = n0 + cv * VEC;
    const bool cok = col < Cout;
    const int colc = cok ? col : 0;
    const void* raw; const float* par; bool stat;
    bnb_select(t, colc, raw, par, stat);
    float msc[VEC], msh[VEC], mu[VEC], is[VEC], t1[VEC], t2[VEC];
    bnb_params<T>(t, par, colc, stat, msc, msh, mu, is);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { t1[e] = 0.f; t2[e] = 0.f; }
    const bool usey = (t.mode & PN2_BNB_MASK_Y) != 0;
    const bool smask = stat && (t.mode & PN2_BNB_STORE_MASKED) != 0;
    const bool full = m0 + BM <= M && n0 + BN <= Cout;
    if (full) {
        if (accum) bnb_rows<T, BM, BN, true, true>(dst, ld_dst, Cs, M, m0, col, cv, cok, stat, usey, smask, vr, vm, vd, msc, msh, t1, t2, ps);
        else bnb_rows<T, BM, BN, false, true>(dst, ld_dst, Cs, M, m0, col, cv, cok, stat, usey, smask, vr, vm, vd, msc, msh, t1, t2, ps);
    } else {
        if (accum) bnb_rows<T, BM, BN, true, false>(dst, ld_dst, Cs, M, m0, col, cv, cok, stat, usey, smask, vr, vm, vd, msc, msh, t1, t2, ps);
        else bnb_rows<T, BM, BN, false, false>(dst, ld_dst, Cs, M, m0, col, cv, cok, stat, usey, smask, vr, vm, vd, msc, msh, t1, t2, ps);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s1[e] = t1[e]; s2[e] = is[e] * (t2[e] - mu[e] * t1[e]); }
}

// The same fragment by inline asm.  hipcc orders every compiler-visible LDS access behind ALL outstanding vector-memory operations of a kernel
// that uses LDS-DMA (s_waitcnt vmcnt(0): it cannot tell a DMA landing from a global store in flight), so an LDS read after a global store costs a
// full store round trip (measured: 4 x ~2 us in a walk of 4 rows).  The epilogues therefore read LDS through asm (waits are counted by hand) and
// issue their global stores last.
typedef unsigned u32x2_t_ __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p; }
__device__ __forceinline__ void tr_frag_issue(unsigned tile, int rs, int chan0, int lane, u32x2_t_& lo, u32x2_t_& hi) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned a = tile + (g * 4 + (i >> 2)) * rs + (chan0 + (i & 3) * 4) * 2;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a + 16 * rs));
}

// ------------------------------------------------------------------------------------------------
// BatchNorm-backward epilogue of the bf16 kernels (pn2_conv_gemm_ep), second form.  Phase stamps of the first form (tools/stamp_micro.py ep; one
// thread owns a channel vector and walks its rows with every operand prefetched into registers, sums carried in registers and met in LDS) on the
// 1x1 dgrads of layer1: of a workgroup's 8.6 us, 1.2 us issuing the register prefetch, 2.3 us the walk (~100 VALU ops per 8 elements at two waves per
// SIMD), 1.3 us meeting the sums in LDS and writing them - and 184-344 registers.  Here
//   * the operand tiles (raw conv output, stored activation for the mask, the destination's prior content for +=) come by LDS-DMA into dense
//     [BM][BN] tiles behind the C tile - no registers, no 64-bit address arithmetic per row;
//   * the walk works on PACKED bf16 pairs: stored = C (+ prior), keep-mask from the sign of fmaf(m, scale, shift) as a bit mask, dz = stored & mask,
//     written back over the C tile (~45 ops per 8 elements);
//   * sum dz and sum dz * raw come from the matrix cores, from the dz tile and the raw tile read back transposed: ones x F_dz and the diagonal of
//     F_raw^T F_dz - exact products of bf16 values summed in fp32; no per-thread sums, no LDS meeting, one barrier less.
// Same definition of the sums as before (they see the STORED, rounded gradient); fp32 keeps the register form below.
// ------------------------------------------------------------------------------------------------
// Which tiles take this form: measured per launch (tools/stamp_micro.py ep, same box) it wins on the narrow tiles of the 3x3 dgrads (128 x 32 tile,
// 64 -> 32 channels at 176^2: 137.7 -> 115.7 us; registers 158 -> 118, one more workgroup per CU) and loses on the wide tiles of the 1x1 dgrads
// with += (64 x 128: 142 -> 154 us, 90.8 -> 100.3 us: three operand tiles cost ~1 us of LDS-DMA issue per workgroup and the sums + stores tail
// 2.3 us against 1.3 us of the register form).  Tiles of at most 4096 elements use it, the others keep the register form.
constexpr bool ep2_tile(int bm, int bn) { return bm * bn <= 4096; }
struct Ep2Layout { int raw_a, y_a, prior, raw_b, dz_b, raw_c, total; };
template <int BM, int BN>
__host__ __device__ inline Ep2Layout ep2_layout(bool stat_a, bool y_a, bool acc, bool stat_b, bool stat_c = false) {
    constexpr int CT = BM * (BN * 2 + 16), TB = BM * BN * 2;
    Ep2Layout L; int off = CT;
    L.raw_a = off; if (stat_a) off += TB;
    L.y_a = off; if (y_a) off += TB;
    L.prior = off; if (acc) off += TB;
    L.raw_b = off; if (stat_b) off += TB;
    L.dz_b = off; if (stat_b) off += CT;
    L.raw_c = off; if (stat_c) off += TB;
    L.total = off;
    return L;
}
__host__ __device__ inline void ep2_needs(const pn2_conv_desc& d, const pn2_conv_ep& ep, bool& stat_a, bool& y_a, bool& acc, bool& stat_b) {
    stat_a = (ep.a.mode & PN2_BNB_STATS) != 0;
    y_a = stat_a && (ep.a.mode & PN2_BNB_MASK_Y) != 0;
    acc = (d.flags & PN2_CONV_ACCUM) != 0;
    stat_b = ep.b.out != nullptr && (ep.b.mode & PN2_BNB_STATS) != 0;
}
// pn2_conv_ep.c: a second BatchNorm behind target a's MASKED gradient (the BatchNorm of a residual branch without activation: Bottle2neck's downsample,
// Res2Net_v1b.py:80,127-136 - its output gradient IS dz of bn3): sum dz and sum dz * raw_c from the dz tile this epilogue has in LDS anyway
__host__ __device__ inline bool ep2_stat_c(const pn2_conv_ep& ep) { return (ep.a.mode & PN2_BNB_STATS) && (ep.c.mode & PN2_BNB_STATS); }

// one [BM][BN] bf16 operand tile -> LDS, dense rows, by LDS-DMA.  Tensors below 2 GB without a column split go through a buffer descriptor: one
// 32-bit offset per lane, stepped by a constant per request, rows past M read as zeros (out of range); otherwise flat 64-bit addresses, rows clamped.
// Columns past Cout are clamped to the tile's first chunk.  None of the clamped / zero values ever counts: their dz is zero or their column is not written.
template <int BM, int BN>
__device__ __forceinline__ void ep2_dma_tile(char* region, const bf16_t* base, int ld, const bf16_t* base2, int split, int M, int m0, int n0, int Cout) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int CPR = BN / 8, NCH = BM * CPR, RPK = 256 / CPR;       // chunks per row / per tile, rows per round of the 4 waves
    static_assert(NCH % 256 == 0 && 256 % CPR == 0, "tile must split over 4 waves x 64 lanes");
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = wid * 64 + lane, row0 = p0 / CPR, ch = p0 - row0 * CPR;
    int col = n0 + ch * 8;
    if (col >= Cout || (split > 0 && col >= split && !base2)) col = n0;      // columns >= split without a second tensor (no BatchNorm behind them) read the tile's first chunk: `raw` need not be that wide
    const size_t extent = (size_t)M * (size_t)ld * 2;
    if (split <= 0 && extent < 0x7fffffffull) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)base >> 32)) << 32) |
                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)base)), 0, (int)extent, 0x00020000);
        unsigned vo = ((unsigned)(m0 + row0) * (unsigned)ld + (unsigned)col) * 2u;
        const unsigned step = (unsigned)RPK * (unsigned)ld * 2u;
#pragma unroll
        for (int k = 0; k < NCH / 256; ++k) {
            // (an offset that wrapped past 2^32 cannot occur: (m0 + BM) * ld * 2 < extent + BM * ld * 2 < 2^32)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)(region + (k * 4 + wid) * 1024), 16, (int)vo, 0, 0, 0);
            vo += step;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NCH / 256; ++k) {
            const int m = min(m0 + row0 + k * RPK, M - 1);
            const bf16_t* b = (split > 0 && col >= split && base2) ? base2 : base;
            __builtin_amdgcn_global_load_lds((gptr_t)(b + (size_t)m * ld + col), (lptr_t)(region + (k * 4 + wid) * 1024), 16, 0, 0);
        }
    }
}

// the prior tile of a += destination from the quarter-resolution tensor of pn2_conv_ep.pool (see pool_row): flat addresses, one source row per tile row
template <int BM, int BN>
__device__ __forceinline__ void ep2_dma_pool(char* region, const bf16_t* pool, int ld, const pn2_conv_desc& d, int M, int m0, int n0) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int CPR = BN / 8, NCH = BM * CPR, RPK = 256 / CPR;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = wid * 64 + lane, row0 = p0 / CPR, ch = p0 - row0 * CPR;
    int col = n0 + ch * 8;
    if (col >= d.Cout) col = n0;
#pragma unroll
    for (int k = 0; k < NCH / 256; ++k) {
        const int m = min(m0 + row0 + k * RPK, M - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(pool + (size_t)pool_row(d, m) * ld + col), (lptr_t)(region + (k * 4 + wid) * 1024), 16, 0, 0);
    }
}

template <int BM, int BN>
__device__ __forceinline__ void ep2_issue(char* smem, const pn2_conv_desc& d, const pn2_conv_ep& ep, const bf16_t* out, int M, int m0, int n0) {
    bool stat_a, y_a, acc, stat_b;
    ep2_needs(d, ep, stat_a, y_a, acc, stat_b);
    const Ep2Layout L = ep2_layout<BM, BN>(stat_a, y_a, acc, stat_b, ep2_stat_c(ep));
    if (ep2_stat_c(ep)) ep2_dma_tile<BM, BN>(smem + L.raw_c, (const bf16_t*)ep.c.raw, ep.c.ld_raw, nullptr, 0, M, m0, n0, d.Cout);
    if (stat_a) ep2_dma_tile<BM, BN>(smem + L.raw_a, (const bf16_t*)ep.a.raw, ep.a.ld_raw, (const bf16_t*)ep.a.raw2, ep.a.split, M, m0, n0, d.Cout);
    if (y_a) ep2_dma_tile<BM, BN>(smem + L.y_a, (const bf16_t*)ep.a.y, ep.a.ld_y, nullptr, 0, M, m0, n0, d.Cout);
    if (acc) {
        if (ep.pool) ep2_dma_pool<BM, BN>(smem + L.prior, (const bf16_t*)ep.pool, ep.ld_pool, d, M, m0, n0);
        else ep2_dma_tile<BM, BN>(smem + L.prior, out, d.ld_out, nullptr, 0, M, m0, n0, d.Cout);
    }
    if (stat_b) ep2_dma_tile<BM, BN>(smem + L.raw_b, (const bf16_t*)ep.b.raw, ep.b.ld_raw, (const bf16_t*)ep.b.raw2, ep.b.split, M, m0, n0, d.Cout);
}

// keep-mask of 8 packed bf16 values: element e is kept when fmaf(m_e, sc_e, sh_e) > 0  (m = raw with (scale, shift), or the stored activation with (1, 0))
// component-wise select on VALUES (a ?: between uint4 lvalues becomes a select between stack slots: scratch traffic with a vmcnt(0) wait each)
__device__ __forceinline__ uint4 sel4(bool c, uint4 a, uint4 b) { return make_uint4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
__device__ __forceinline__ unsigned ep2_keep2(unsigned v, unsigned m, float sc0, float sh0, float sc1, float sh1) {
    const bool k0 = fmaf(__uint_as_float(m << 16), sc0, sh0) > 0.f;
    const bool k1 = fmaf(__uint_as_float(m & 0xffff0000u), sc1, sh1) > 0.f;
    return v & ((k0 ? 0x0000ffffu : 0u) | (k1 ? 0xffff0000u : 0u));
}
__device__ __forceinline__ uint4 ep2_masked(const uint4& v, const uint4& m, const float (&sc)[8], const float (&sh)[8]) {
    return make_uint4(ep2_keep2(v.x, m.x, sc[0], sh[0], sc[1], sh[1]), ep2_keep2(v.y, m.y, sc[2], sh[2], sc[3], sh[3]),
                      ep2_keep2(v.z, m.z, sc[4], sh[4], sc[5], sh[5]), ep2_keep2(v.w, m.w, sc[6], sh[6], sc[7], sh[7]));
}

// mask coefficients of this thread's channel vector for one target (statistics columns only)
__device__ __forceinline__ void ep2_mask_params(const pn2_bnb_target& t, const float* par, int colc, bool stat, int& kind, float (&sc)[8], float (&sh)[8]) {
    kind = 0;                                        // 0: keep everything, 1: mask from raw, 2: mask from the stored activation
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
    if (!stat) return;
    if (t.mode & PN2_BNB_MASK_Y) kind = 2;
    else if (t.mode & PN2_BNB_MASK_RAW) {
        kind = 1;
        const float4 a0 = *reinterpret_cast<const float4*>(par + colc), a1 = *reinterpret_cast<const float4*>(par + colc + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(par + (size_t)t.ps + colc), b1 = *reinterpret_cast<const float4*>(par + (size_t)t.ps + colc + 4);
        sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
        sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
    }
}

// the walk: every thread owns one channel vector (8 channels) and RPT rows of the tile.  All LDS reads first (asm), then the arithmetic, then the dz
// tiles back to LDS; what goes to global memory is handed back in registers and stored after the sums (see tr_frag_issue)
template <int BM, int BN> struct Ep2Out {
    static constexpr int RPT = BM * (BN / 8) / 256;
    uint4 a[RPT], b[RPT];
};
template <int BM, int BN>
__device__ __forceinline__ void ep2_apply(char* smem, const pn2_conv_desc& d, const pn2_conv_ep& ep, int M, int m0, int n0, Ep2Out<BM, BN>& O) {
    using T = bf16_t;
    constexpr int VPR = BN / 8, CRS = BN * 2 + 16, RPT = BM * VPR / 256, RSTEP = 256 / VPR;
    bool stat_a, y_a, acc, stat_b;
    ep2_needs(d, ep, stat_a, y_a, acc, stat_b);
    const Ep2Layout L = ep2_layout<BM, BN>(stat_a, y_a, acc, stat_b, ep2_stat_c(ep));
    const int tid = threadIdx.x, cv = tid % VPR, r0 = tid / VPR, col = n0 + cv * 8;
    const bool cok = col < d.Cout, dual = ep.b.out != nullptr;
    const int colc = cok ? col : n0;
    const void* rawp; const float *parA, *parB = nullptr; bool stA, stB = false;
    bnb_select(ep.a, colc, rawp, parA, stA);
    if (dual) bnb_select(ep.b, colc, rawp, parB, stB);
    int kindA, kindB;
    float scA[8], shA[8], scB[8], shB[8];
    ep2_mask_params(ep.a, parA, colc, stA, kindA, scA, shA);
    ep2_mask_params(ep.b, parB, colc, stB, kindB, scB, shB);          // (stB is false without a second target)
    const bool smA = stA && (ep.a.mode & PN2_BNB_STORE_MASKED), smB = stB && (ep.b.mode & PN2_BNB_STORE_MASKED);
    const float ps = ep.pool ? 0.25f : 1.f;
    const unsigned base = lds_addr(smem);
    u32x4_t_ c[RPT], pr[RPT], ma[RPT], mb[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int row = r0 + u * RSTEP;
        const unsigned tslot = (unsigned)(row * VPR + cv) * 16u;
        asm volatile("ds_read_b128 %0, %1" : "=v"(c[u]) : "v"(base + row * CRS + cv * 16));
        if (acc) asm volatile("ds_read_b128 %0, %1" : "=v"(pr[u]) : "v"(base + L.prior + tslot));
        if (kindA) asm volatile("ds_read_b128 %0, %1" : "=v"(ma[u]) : "v"(base + (kindA == 2 ? L.y_a : L.raw_a) + tslot));
        if (kindB) asm volatile("ds_read_b128 %0, %1" : "=v"(mb[u]) : "v"(base + L.raw_b + tslot));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int row = r0 + u * RSTEP, m = m0 + row;
        const bool live = cok && m < M;
        const uint4 cc = __builtin_bit_cast(uint4, c[u]);
        uint4 o = cc;
        if (acc) {
            float x[8], y[8];
            TT<T>::unpack(cc, x);
            TT<T>::unpack(__builtin_bit_cast(uint4, pr[u]), y);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = __fmaf_rn(y[e], ps, x[e]);          // ps = 1 (== x + y) or 1/4 (pooled prior, pn2_conv_ep.pool)
            o = TT<T>::pack(x);                      // the statistics see the STORED (rounded) gradient, as a separate reduce pass would
        }
        uint4 dz = o;
        if (kindA) dz = ep2_masked(o, __builtin_bit_cast(uint4, ma[u]), scA, shA);
        O.a[u] = sel4(smA, dz, o);
        if (stat_a) *reinterpret_cast<uint4*>(smem + row * CRS + cv * 16) = sel4(live && stA, dz, zero);      // dz tile of target a, in place
        if (dual) {
            uint4 dzb = cc;                           // the second target takes the GEMM's own tile (never +=)
            if (kindB) dzb = ep2_masked(cc, __builtin_bit_cast(uint4, mb[u]), scB, shB);
            O.b[u] = sel4(smB, dzb, cc);
            if (stat_b) *reinterpret_cast<uint4*>(smem + L.dz_b + row * CRS + cv * 16) = sel4(live && stB, dzb, zero);
        }
    }
}
template <int BM, int BN>
__device__ __forceinline__ void ep2_store(const pn2_conv_desc& d, const pn2_conv_ep& ep, bf16_t* __restrict__ out, int M, int m0, int n0, const Ep2Out<BM, BN>& O) {
    constexpr int VPR = BN / 8, RPT = BM * VPR / 256, RSTEP = 256 / VPR;
    const int tid = threadIdx.x, cv = tid % VPR, r0 = tid / VPR, col = n0 + cv * 8;
    const bool cok = col < d.Cout, dual = ep.b.out != nullptr;
    bf16_t* outb = reinterpret_cast<bf16_t*>(ep.b.out);
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int m = m0 + r0 + u * RSTEP;
        if (cok && m < M) {
            *reinterpret_cast<uint4*>(out + (size_t)m * d.ld_out + col) = O.a[u];
            if (dual) *reinterpret_cast<uint4*>(outb + (size_t)m * ep.b.ld_out + col) = O.b[u];
        }
    }
}

// BatchNorm batch statistics of a staged bf16 C tile [BM][BN] (row stride crs bytes) on the matrix cores; rows >= nrow are zeros.
// Wave w takes the 16-channel blocks w, w + 4, ...; writes (mean, M2) of the tile per channel to psum / psq [bm][Cout].
template <int BM, int BN>
__device__ __forceinline__ void mfma_stats(const char* Cs, int crs, int nrow, int Cout, int n0, int bm, float* __restrict__ psum, float* __restrict__ psq) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const double rn = nrow == BM ? 1.0 / BM : 1.0 / (double)nrow;      // (uniform branch: the division runs in the last row block only)
    constexpr int NCB = (BN / 16 + 3) / 4, KC = BM / 32;
    const unsigned cs = lds_addr(Cs);
    // all of this wave's blocks at once: every transposing read is requested before the one wait, the 2 * NCB accumulator chains interleave
    u32x2_t_ lo[NCB][KC], hi[NCB][KC];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int cb = min(wid + 4 * c, BN / 16 - 1);                  // (a wave without a block c re-reads its last one; nothing is written for it)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) tr_frag_issue(cs + kc * 32 * crs, crs, cb * 16, lane, lo[c][kc], hi[c][kc]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t aS[NCB], aQ[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) { aS[c] = f32x4_t{0.f, 0.f, 0.f, 0.f}; aQ[c] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const uint4 F = make_uint4(lo[c][kc].x, lo[c][kc].y, hi[c][kc].x, hi[c][kc].y);
            MMA<bf16_t>::run(aQ[c], F, F);
            MMA<bf16_t>::run(aS[c], ones, F);
        }
    // column l15: every row of aS holds its sum; its sum of squares is the diagonal element of aQ, in register l15 & 3 of the lane with g == l15 >> 2:
    // that lane has both and writes the column - no cross-lane traffic
    const int rr = l15 & 3;
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int cb = wid + 4 * c, col = n0 + cb * 16 + l15;
        const float qd = rr == 0 ? aQ[c][0] : (rr == 1 ? aQ[c][1] : (rr == 2 ? aQ[c][2] : aQ[c][3]));
        if (cb < BN / 16 && (l15 >> 2) == g && col < Cout) {
            const double Sd = (double)aS[c][0], mean = Sd * rn, m2 = (double)qd - Sd * mean;
            psum[(size_t)bm * Cout + col] = (float)mean;
            psq[(size_t)bm * Cout + col] = (float)(m2 > 0.0 ? m2 : 0.0);
        }
    }
}

// p1 = sum dz, p2 = invstd * (sum dz * raw - mean * sum dz) per channel of the tile, on the matrix cores; wave w takes the 16-channel blocks w, w + 4, ...
template <int BM, int BN>
__device__ __forceinline__ void ep2_sums(const char* dz, const char* raw, const pn2_bnb_target& t, int Cout, int n0, int bm) {
    constexpr int CRS = BN * 2 + 16, NCB = (BN / 16 + 3) / 4, KC = BM / 32;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    const unsigned dza = lds_addr(dz), rawa = lds_addr(raw);
    const bool holder = (l15 >> 2) == g;             // the lane that holds the diagonal element of its column (and, like every lane of the column, its plain sum)
    float MU[NCB], IS[NCB];
    bool ST[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {                  // the holder's (mean, invstd): requested now, needed after the MFMAs
        const int cb = wid + 4 * c, col = n0 + cb * 16 + l15;
        MU[c] = 0.f; IS[c] = 0.f; ST[c] = false;
        if (cb < BN / 16 && holder && col < Cout) {
            const void* rw; const float* par;
            bnb_select(t, col, rw, par, ST[c]);
            if (ST[c]) { MU[c] = par[(size_t)2 * t.ps + col]; IS[c] = par[(size_t)3 * t.ps + col]; }
        }
    }
    u32x2_t_ dl[NCB][KC], dh[NCB][KC], rl[NCB][KC], rh[NCB][KC];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int cb = min(wid + 4 * c, BN / 16 - 1);
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) {
            tr_frag_issue(dza + kc * 32 * CRS, CRS, cb * 16, lane, dl[c][kc], dh[c][kc]);
            tr_frag_issue(rawa + kc * 32 * BN * 2, BN * 2, cb * 16, lane, rl[c][kc], rh[c][kc]);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t a1[NCB], a2[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) { a1[c] = f32x4_t{0.f, 0.f, 0.f, 0.f}; a2[c] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const uint4 Fd = make_uint4(dl[c][kc].x, dl[c][kc].y, dh[c][kc].x, dh[c][kc].y), Fr = make_uint4(rl[c][kc].x, rl[c][kc].y, rh[c][kc].x, rh[c][kc].y);
            MMA<bf16_t>::run(a1[c], ones, Fd);
            MMA<bf16_t>::run(a2[c], Fr, Fd);
        }
    const int rr = l15 & 3;
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int cb = wid + 4 * c, col = n0 + cb * 16 + l15;
        const float q = rr == 0 ? a2[c][0] : (rr == 1 ? a2[c][1] : (rr == 2 ? a2[c][2] : a2[c][3]));
        if (cb < BN / 16 && holder && col < Cout) {
            float s1 = 0.f, s2 = 0.f;
            if (ST[c]) { s1 = a1[c][0]; s2 = IS[c] * (q - MU[c] * s1); }
            t.p1[(size_t)bm * t.ldp + col] = s1;
            t.p2[(size_t)bm * t.ldp + col] = s2;
        }
    }
}

// SWP (bf16 LDS-DMA kernels): the MFMAs ran with the operands exchanged, so a lane holds FOUR CONSECUTIVE CHANNELS of one pixel
// (acc[i][j][r] = C[pixel wm*WTM + i*16 + l15][channel wn*WTN + j*16 + g*4 + r]) instead of four pixels of one channel.  Measured with in-kernel
// stamps (tools/stamp_micro.py, 64->256 1x1 conv of layer1, 128 x 128 tile): of a workgroup's 6.5 us, 1.6 us were the in-register statistics
// (192 dependent VALU ops per wave at two waves per SIMD), 1.1 us the 64 cvt + ds_write_b16 per lane (LDS store issue), 0.5 us the merge.  Here
//   - staging is 2 v_cvt_pk_bf16_f32 + ONE ds_write_b64 per 16 x 16 block (16 stores per lane instead of 64),
//   - the BatchNorm statistics come from the matrix cores: for a 16-channel block F (rows x 16, read back transposed from the staged tile),
//     ones x F gives the column sums and the diagonal of F^T F the sums of squares - 2 x BM/32 MFMAs per block, no VALU reduction, no second
//     barrier, no merge step.  They are the statistics of the bf16-ROUNDED outputs, i.e. of the tensor that is normalised afterwards (what
//     torch.autocast computes too); mean and M2 of the tile are formed in double from the two fp32 sums.
template <typename T, int BM, int BN, int WM, int WN, int MT, int NT, bool EP, bool SWP = false>
__device__ __forceinline__ void conv_epilogue(f32x4_t (&acc)[MT][NT], char* smem, const pn2_conv_desc& d, const pn2_conv_ep& ep, T* __restrict__ out,
                                              float* __restrict__ psum, float* __restrict__ psq, int M, int m0, int n0, int bm, BnbPre<T, BM, BN>& pre) {
    constexpr int VEC = TT<T>::VEC;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int CRS = BN * (int)sizeof(T) + 16;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN, l15 = lane & 15, g = lane >> 4;
    char* Cs = smem;
    float* red = reinterpret_cast<float*>(smem + BM * CRS);   // [3][WM][BN]: shifted sum, shifted sum of squares, shift
    constexpr bool EP2 = EP && sizeof(T) == 2 && ep2_tile(BM, BN);          // bf16, narrow tiles: operand tiles by LDS-DMA, sums on the matrix cores (ep2_*)
    if constexpr (EP2) ep2_issue<BM, BN>(smem, d, ep, reinterpret_cast<const bf16_t*>(out), M, m0, n0);
    if constexpr (EP && !EP2) bnb_prefetch<T, BM, BN>(d, ep, out, M, m0, n0, pre);      // in flight while the C tile is staged
    const bool full_m = m0 + BM <= M;
    if constexpr (SWP) {
        static_assert(sizeof(T) == 2, "swapped epilogue: bf16 only");
        if constexpr (EP) {
            if (d.flags & PN2_CONV_ROWGATE) {          // per-pixel gate (see below): pixel = l15 of block i
                const float* __restrict__ gate = ep.a.par;
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int m = m0 + wm * WTM + i * 16 + l15;
                    const float gg = m < M ? 1.f - 1.f / (1.f + __expf(-gate[m])) : 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[i][j][r] *= gg;
                }
            }
        }
        PN2_STAMP_AT(5);
        const int rows_ok = M - m0;                    // rows of the tile that exist (>= BM for all but the last row block)
        const bool has_bias = d.flags & PN2_CONV_BIAS;
        // PN2_CONV_AFFINE (eval-mode BatchNorm folded into the GEMM, pn2_conv_gemm_affine): y = acc * scale[c] + shift[c] (psum / psq), then the activation -
        // unless a residual is added first (copy-out below)
        const bool affine = !EP && (d.flags & PN2_CONV_AFFINE);
        const int act_pre = (affine && !ep.a.y) ? ((d.flags & PN2_CONV_RELU6) ? 2 : ((d.flags & PN2_CONV_RELU) ? 1 : 0)) : 0;
        if (full_m && !has_bias && !affine) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    *reinterpret_cast<uint2*>(Cs + (wm * WTM + i * 16 + l15) * CRS + (wn * WTN + j * 16 + g * 4) * 2) =
                        make_uint2(TT<T>::cvt2(acc[i][j][0], acc[i][j][1]), TT<T>::cvt2(acc[i][j][2], acc[i][j][3]));
        } else {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float b4[4] = {0.f, 0.f, 0.f, 0.f}, s4[4] = {1.f, 1.f, 1.f, 1.f};
                if (has_bias || affine) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cg = n0 + wn * WTN + j * 16 + g * 4 + r;
                        if (affine) { s4[r] = cg < d.Cout ? psum[cg] : 0.f; b4[r] = cg < d.Cout ? psq[cg] : 0.f; }
                        else b4[r] = cg < d.Cout ? psum[cg] : 0.f;
                    }
                }
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int row = wm * WTM + i * 16 + l15;
                    const bool live = row < rows_ok;       // rows past M are staged as zeros: they drop out of the statistics
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = affine ? fmaf(acc[i][j][r], s4[r], b4[r]) : acc[i][j][r] + b4[r];
                        if (act_pre) v[r] = fmaxf(v[r], 0.f);
                        if (act_pre == 2) v[r] = fminf(v[r], 6.f);
                        v[r] = live ? v[r] : 0.f;
                    }
                    *reinterpret_cast<uint2*>(Cs + row * CRS + (wn * WTN + j * 16 + g * 4) * 2) = make_uint2(TT<T>::cvt2(v[0], v[1]), TT<T>::cvt2(v[2], v[3]));
                }
            }
        }
        PN2_STAMP_AT(13);
        if constexpr (EP2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's operand-tile DMA has landed; the barrier publishes everyone's
        __syncthreads();
        PN2_STAMP_AT(6);
        if constexpr (EP) { if (d.flags & PN2_CONV_STATS) mfma_stats<BM, BN>(Cs, CRS, min(M - m0, BM), d.Cout, n0, bm, psum, psq); }
    } else {
    if constexpr (EP) {          // (only in the epilogue-statistics instantiations: in the plain kernels the extra live range costs a wave of occupancy)
        if (d.flags & PN2_CONV_ROWGATE) {
            // V1 reverse attention in front of a 1x1 conv (PraNet_Res2Net.py:153-155): conv((1 - sigmoid(crop)) * x) = (1 - sigmoid(crop)) * conv(x),
            // a per-pixel scale of the accumulator rows - before the statistics and the store, so the gated copy of x is never written
            const float* __restrict__ gate = ep.a.par;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WTM + i * 16 + g * 4 + r;
                    const float gg = m < M ? 1.f - 1.f / (1.f + __expf(-gate[m])) : 0.f;
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j][r] *= gg;
                }
        }
    }
    // Everything below runs once per tile at the GEMM's low occupancy, where every instruction is exposed (the 1x1 convs of layer1 / layer2 have 4
    // K-steps per tile: ~1000 epilogue instructions per wave next to 32 MFMAs; ablating the statistics alone makes those launches 5-17 % faster).
    // FULL tiles - all but the last row block of a conv - take paths without per-element row masks, without the bias add when there is no bias and
    // with a straight LDS -> global copy of the C tile (no bounds checks, no read-modify-write).
    PN2_STAMP_AT(5);
    // bf16 instantiations (the register-staged fallback of the LDS-DMA kernels): statistics from the staged, ROUNDED tile on the matrix cores like the
    // swapped epilogue - one definition of the batch statistics for every bf16 kernel (bit-identical between the kernels); fp32 stores unrounded values
    // and keeps the in-register shifted sums.
    constexpr bool MST = sizeof(T) == 2;
    if (!MST && (d.flags & PN2_CONV_STATS)) {
        const int rows_w = min(max(M - m0 - wm * WTM, 0), WTM);        // valid rows of this wave's tile (rows are ascending)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float k = __shfl(acc[0][j][0], l15);                   // row 0 of the wave tile, this lane's column
            float s = 0.f, q = 0.f;
            if (full_m) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[i][j][r] - k;
                        s += v; q = fmaf(v, v, q);
                        asm volatile("" : "+v"(s), "+v"(q));      // keeps the two chains scalar: packed-math pairing of s / q (SLP) gave run-to-run different q here
                    }
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = (i * 16 + g * 4 + r < rows_w) ? acc[i][j][r] - k : 0.f;
                        s += v; q += v * v;
                    }
            }
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            if (g == 0) {
                const int c = wn * WTN + j * 16 + l15;
                red[wm * BN + c] = s; red[(WM + wm) * BN + c] = q; red[(2 * WM + wm) * BN + c] = k;
            }
        }
    }
    PN2_STAMP_AT(12);
    if (!EP && (d.flags & PN2_CONV_AFFINE)) {       // eval-mode BatchNorm folded into the GEMM: psum / psq carry scale / shift (see the swapped epilogue)
        const int act_pre = !ep.a.y ? ((d.flags & PN2_CONV_RELU6) ? 2 : ((d.flags & PN2_CONV_RELU) ? 1 : 0)) : 0;
        float sj[NT], bj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cg = n0 + wn * WTN + j * 16 + l15;
            sj[j] = cg < d.Cout ? psum[cg] : 0.f; bj[j] = cg < d.Cout ? psq[cg] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * WTM + i * 16 + g * 4 + r, col = wn * WTN + j * 16 + l15;
                    float v = fmaf(acc[i][j][r], sj[j], bj[j]);
                    if (act_pre) v = fmaxf(v, 0.f);
                    if (act_pre == 2) v = fminf(v, 6.f);
                    TT<T>::st(reinterpret_cast<T*>(Cs + row * CRS) + col, (!MST || row < M - m0) ? v : 0.f);
                }
    } else if (d.flags & PN2_CONV_BIAS) {           // psum carries a per-output-channel fp32 bias (biased conv / nn.Linear without BN)
        float bj[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int cg = n0 + wn * WTN + j * 16 + l15;
            bj[j] = cg < d.Cout ? psum[cg] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * WTM + i * 16 + g * 4 + r, col = wn * WTN + j * 16 + l15;
                    TT<T>::st(reinterpret_cast<T*>(Cs + row * CRS) + col, (!MST || row < M - m0) ? acc[i][j][r] + bj[j] : 0.f);
                }
    } else {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * WTM + i * 16 + g * 4 + r, col = wn * WTN + j * 16 + l15;
                    TT<T>::st(reinterpret_cast<T*>(Cs + row * CRS) + col, (!MST || full_m || row < M - m0) ? acc[i][j][r] : 0.f);
                }
    }
    PN2_STAMP_AT(13);
    if constexpr (EP2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PN2_STAMP_AT(6);
    if constexpr (MST && EP) { if (d.flags & PN2_CONV_STATS) mfma_stats<BM, BN>(Cs, CRS, min(M - m0, BM), d.Cout, n0, bm, psum, psq); }
    if (!MST && (d.flags & PN2_CONV_STATS) && tid < BN) {
        const int col = n0 + tid;
        if (col < d.Cout) {
            float n = 0.f, mean = 0.f, m2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                const float nw = (float)min(max(M - m0 - w * WTM, 0), WTM);
                if (nw > 0.f) {
                    const float s = red[w * BN + tid], q = red[(WM + w) * BN + tid], k = red[(2 * WM + w) * BN + tid];
                    const float mw = k + s / nw, m2w = q - s * s / nw;
                    const float delta = mw - mean, nt = n + nw;
                    mean += delta * (nw / nt);
                    m2 += m2w + delta * delta * (n * nw / nt);
                    n = nt;
                }
            }
            psum[(size_t)bm * d.Cout + col] = mean;
            psq[(size_t)bm * d.Cout + col] = m2;
        }
    }
    }      // !SWP
    constexpr int VPR = BN / VEC;
    const bool vec_ok = (d.Cout % VEC == 0) && (d.ld_out % VEC == 0);
    const bool accum = d.flags & PN2_CONV_ACCUM;
    PN2_STAMP_AT(7);
    if constexpr (!EP) {
        // PN2_CONV_AFFINE with a residual (ep.a.y, same dtype): y = act(staged + residual) in the copy-out (vector path: the host checks the alignment)
        const T* res = (d.flags & PN2_CONV_AFFINE) ? reinterpret_cast<const T*>(ep.a.y) : nullptr;
        const int act_post = res ? ((d.flags & PN2_CONV_RELU6) ? 2 : ((d.flags & PN2_CONV_RELU) ? 1 : 0)) : 0;
        if (full_m && vec_ok && !accum && !res && n0 + BN <= d.Cout) {         // the common tile: no bounds, no read-modify-write
            T* obase = out + (size_t)m0 * d.ld_out + n0;
#pragma unroll
            for (int u = 0; u < BM * VPR / 256; ++u) {
                const int idx = tid + u * 256;
                const int row = idx / VPR, cv = idx - row * VPR;
                *reinterpret_cast<uint4*>(obase + (size_t)row * d.ld_out + cv * VEC) = *reinterpret_cast<const uint4*>(Cs + row * CRS + cv * 16);
            }
        } else
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
                if (res) {
                    float x[VEC], y[VEC];
                    TT<T>::unpack(v, x);
                    TT<T>::unpack(*reinterpret_cast<const uint4*>(res + (size_t)m * ep.a.ld_y + col), y);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        x[e] += y[e];
                        if (act_post) x[e] = fmaxf(x[e], 0.f);
                        if (act_post == 2) x[e] = fminf(x[e], 6.f);
                    }
                    v = TT<T>::pack(x);
                }
                *reinterpret_cast<uint4*>(dst) = v;
            } else {
                float x[VEC];
                TT<T>::unpack(v, x);
                for (int e = 0; e < VEC && col + e < d.Cout; ++e) TT<T>::st(dst + e, accum ? x[e] + TT<T>::ld(dst + e) : x[e]);
            }
        }
        // (swapped epilogue) statistics AFTER the stores have been issued: they drain while the matrix cores reduce the staged tile
        PN2_STAMP_AT(14);
        if constexpr (SWP || sizeof(T) == 2) { if (d.flags & PN2_CONV_STATS) mfma_stats<BM, BN>(Cs, CRS, min(M - m0, BM), d.Cout, n0, bm, psum, psq); }
    } else if constexpr (EP2) {
        Ep2Out<BM, BN> O;
        ep2_apply<BM, BN>(smem, d, ep, M, m0, n0, O);
        PN2_STAMP_AT(14);
        __syncthreads();                               // the dz tiles are complete
        PN2_STAMP_AT(15);
        bool stat_a, y_a, acc_, stat_b;
        ep2_needs(d, ep, stat_a, y_a, acc_, stat_b);
        const Ep2Layout L = ep2_layout<BM, BN>(stat_a, y_a, acc_, stat_b, ep2_stat_c(ep));
        if (stat_a) ep2_sums<BM, BN>(smem, smem + L.raw_a, ep.a, d.Cout, n0, bm);
        if (ep2_stat_c(ep)) ep2_sums<BM, BN>(smem, smem + L.raw_c, ep.c, d.Cout, n0, bm);          // the same dz tile against the second BatchNorm's raw tile
        if (stat_b) ep2_sums<BM, BN>(smem + L.dz_b, smem + L.raw_b, ep.b, d.Cout, n0, bm);
        ep2_store<BM, BN>(d, ep, reinterpret_cast<bf16_t*>(out), M, m0, n0, O);      // global stores last
    } else {
    // ---- BatchNorm-backward statistics of the produced gradient tile (vector path only: the host checks the alignment)
    float sums[4][VEC];
    const bool dual = ep.b.out != nullptr;
    if (dual) bnb_target<T, BM, BN>(ep.b, reinterpret_cast<T*>(ep.b.out), ep.b.ld_out, false, Cs, M, m0, n0, d.Cout, pre.rb, pre.rb, pre.vd, sums[2], sums[3]);
    bnb_target<T, BM, BN>(ep.a, out, d.ld_out, accum, Cs, M, m0, n0, d.Cout, pre.ra, pre.ma, pre.vd, sums[0], sums[1], ep.pool ? 0.25f : 1.f);
    PN2_STAMP_AT(14);
    __syncthreads();                                   // everyone has drained the C tile
    // the 256 / VPR row lanes of a channel vector meet in LDS: rs[sum][row lane][BN], then column-parallel adds in a fixed order
    constexpr int RL = 256 / VPR;
    float* rs = reinterpret_cast<float*>(smem);
    const int cv = tid % VPR, rl = tid / VPR, nsum = dual ? 4 : 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k < nsum) {
            float* w = rs + ((size_t)(k * RL + rl) * BN + cv * VEC);
#pragma unroll
            for (int e = 0; e < VEC; e += 4) *reinterpret_cast<float4*>(w + e) = make_float4(sums[k][e], sums[k][e + 1], sums[k][e + 2], sums[k][e + 3]);
        }
    }
    __syncthreads();
    PN2_STAMP_AT(15);
    for (int o = tid; o < nsum * BN; o += 256) {
        const int k = o / BN, c = o - k * BN;
        const float* r = rs + (size_t)k * RL * BN + c;
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll 4
        for (int q = 0; q < RL; q += 4) { v0 += r[q * BN]; v1 += r[(q + 1) * BN]; v2 += r[(q + 2) * BN]; v3 += r[(q + 3) * BN]; }
        const float v = (v0 + v1) + (v2 + v3);
        if (n0 + c < d.Cout) {
            float* pa = (k & 1) ? ep.a.p2 : ep.a.p1;
            float* pb = (k & 1) ? ep.b.p2 : ep.b.p1;
            const bool isb = k >= 2;
            float* pp = isb ? pb : pa;
            const int ldp = isb ? ep.b.ldp : ep.a.ldp;
            const int md = isb ? ep.b.mode : ep.a.mode;
            if (md & PN2_BNB_STATS) pp[(size_t)bm * ldp + n0 + c] = v;
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad gather-GEMM
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, bool PW, bool EP>
__device__ __forceinline__ void conv_gather_body(const T* __restrict__ in, const T* __restrict__ wp, T* __restrict__ out,
                                                 float* __restrict__ psum, float* __restrict__ psq, const pn2_conv_desc& d, const pn2_conv_ep& ep, int lbid, int lgrid) {
    constexpr int VEC = TT<T>::VEC, BK = MMA<T>::BK;
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int STAGE = (BM + BN) * RS;
    constexpr int NA = BM / 32, NB = BN / 32;     // 16-byte vectors per thread per step (8 vectors per 128-byte row)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN, l15 = lane & 15, g = lane >> 4;
    const int M = d.N * d.OH * d.OW;
    const int nbn = (d.Cout + BN - 1) / BN;
    const int bid = xcd_remap(lbid, lgrid);
    const int bn = bid % nbn, bm = bid / nbn;
    const int m0 = bm * BM, n0 = bn * BN;
    const int taps = d.KH * d.KW;
    const int ktot = taps * d.Cin_p;
    const int ksteps = (ktot + BK - 1) / BK;

    GatherGeom gg;
    gg.H = d.H; gg.W = d.W; gg.OH = d.OH; gg.OW = d.OW; gg.KH = d.KH; gg.KW = d.KW; gg.stride = d.stride;
    gg.sshift = __builtin_ctz(d.stride); gg.pad_h = d.pad_h; gg.pad_w = d.pad_w; gg.dil_h = d.dil_h; gg.dil_w = d.dil_w;
    gg.transposed = d.transposed;

    // ---- per-thread A rows
    const int kv = tid & 7;
    const int wslot = (kv ^ ((tid >> 3) & 7)) << 4;          // this thread's (swizzled) 16-byte slot in its LDS rows: row = (tid >> 3) + 32 i
    int rbase[NA], riy0[NA], rix0[NA]; bool rok[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int m = m0 + (tid >> 3) + 32 * i;
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
    int tcol = 0, tdr = 0, tdc = 0;          // tap column and dilated offsets, stepped with the tap (no division per K-step)
    if (!PW) { const int r0_ = tap / d.KW; tcol = tap - r0_ * d.KW; tdr = r0_ * d.dil_h; tdc = tcol * d.dil_w; }
    const T* bptr = wp + (size_t)(n0 + (tid >> 3)) * d.Kp + kv * VEC;

    // Loads are UNCONDITIONAL (out-of-range vectors read a clamped, valid address) and the zero fill is applied when the
    // registers are written to LDS: a branch around a load makes hipcc wait for it at the join, which would serialise the
    // global->register prefetch of step t+1 with the MFMAs of step t.
    // (macros, not lambdas: arrays captured by reference in a lambda ended up in scratch memory).  RA_ / RB_ / AM_: the register set a load batch lands in
    // the weight-panel vectors by literal index: with two register sets (DEEP) hipcc left an `unroll`-ed loop over RB_[i] as a loop and the arrays in scratch
#define PN2_U4(N_, M_, X_, Y_) do { if constexpr ((N_) > 0) { M_(0, X_, Y_); } if constexpr ((N_) > 1) { M_(1, X_, Y_); } if constexpr ((N_) > 2) { M_(2, X_, Y_); } if constexpr ((N_) > 3) { M_(3, X_, Y_); } } while (0)
#define PN2_GLOAD_B(i_, RB_, step_) RB_[i_] = *reinterpret_cast<const u32x4_t_*>(bptr + (size_t)(32 * (i_)) * d.Kp + (size_t)(step_) * BK)
#define PN2_LSTORE_B(i_, RB_, Bs_) *reinterpret_cast<u32x4_t_*>(Bs_ + ((tid >> 3) + 32 * (i_)) * RS + wslot) = RB_[i_]
#define PN2_GLOAD(step_, RA_, RB_, AM_)                                                                                \
    do {                                                                                                               \
        AM_ = 0;                                                                                                       \
        if (PW) {                                                                                                      \
            const int k_ = (step_) * BK + kv * VEC;                                                                    \
            const bool kok_ = k_ < d.Cin_p;                                                                            \
            const int kc_ = kok_ ? k_ : 0;                                                                             \
            _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
                if (rok[i] && kok_) AM_ |= 1u << i;                                                                    \
                RA_[i] = *reinterpret_cast<const u32x4_t_*>(in + (size_t)rbase[i] * d.ld_in + kc_);                    \
            }                                                                                                          \
        } else {                                                                                                       \
            _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
                int iy_, ix_;                                                                                          \
                const bool ok_ = rok[i] && tap < taps && tap_pixel_off(gg, riy0[i], rix0[i], tdr, tdc, iy_, ix_);      \
                const size_t off_ = ok_ ? (size_t)(rbase[i] + iy_ * d.W + ix_) * d.ld_in + ci : 0;                     \
                if (ok_) AM_ |= 1u << i;                                                                               \
                RA_[i] = *reinterpret_cast<const u32x4_t_*>(in + off_);                                                \
            }                                                                                                          \
            ci += BK;                                                                                                  \
            while (ci >= d.Cin_p) {                                                                                    \
                ci -= d.Cin_p; ++tap; ++tcol; tdc += d.dil_w;                                                          \
                if (tcol == d.KW) { tcol = 0; tdc = 0; tdr += d.dil_h; }                                               \
            }                                                                                                          \
        }                                                                                                              \
        PN2_U4(NB, PN2_GLOAD_B, RB_, step_);                                                                           \
    } while (0)
#define PN2_LSTORE(stage_, RA_, RB_, AM_)                                                                              \
    do {                                                                                                               \
        char* As_ = smem + (stage_) * STAGE;                                                                           \
        char* Bs_ = As_ + BM * RS;                                                                                     \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                               \
            const u32x4_t_ v_ = (AM_ >> i) & 1u ? RA_[i] : u32x4_t_{0u, 0u, 0u, 0u};                                   \
            *reinterpret_cast<u32x4_t_*>(As_ + ((tid >> 3) + 32 * i) * RS + wslot) = v_;                               \
        }                                                                                                              \
        PN2_U4(NB, PN2_LSTORE_B, RB_, Bs_);                                                                            \
    } while (0)
#define PN2_MFMAS(stage_)                                                                                              \
    do {                                                                                                               \
        const char* As = smem + (stage_) * STAGE;                                                                      \
        const char* Bs = As + BM * RS;                                                                                 \
        _Pragma("unroll") for (int ks = 0; ks < KSUB; ++ks) {                                                          \
            uint4 a[MT], b[NT];                                                                                        \
            _Pragma("unroll") for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const uint4*>(As + (wm * WTM + i * 16 + l15a) * RS + (((ks * 4 + g) ^ (l15a & 7)) << 4)); \
            _Pragma("unroll") for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const uint4*>(Bs + (wn * WTN + j * 16 + l15) * RS + (((ks * 4 + g) ^ (l15 & 7)) << 4));  \
            MMA<T>::template run_block<MT, NT>(acc, a, b);                                                             \
        }                                                                                                              \
    } while (0)

    BnbPre<T, BM, BN> pre;
    typename MMA<T>::acc_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = typename MMA<T>::acc_t{0, 0, 0, 0};
    const int l15a = MMA<T>::arow(l15);

    u32x4_t_ ra[NA], rb[NB], ra1[NA], rb1[NB];          // (plain vector types, not HIP's uint4 struct: with two register sets the uint4 arrays stayed in scratch memory; the second set: DEEP only)
    unsigned amask = 0, amask1 = 0;
    // Loads are UNCONDITIONAL (out-of-range vectors read a clamped, valid address) and the zero fill is applied when the
    // registers are written to LDS: a branch around a load makes hipcc wait for it at the join, which would serialise the
    // global->register prefetch of step t+1 with the MFMAs of step t.
    if constexpr (!MMA<T>::DEEP) {
        PN2_GLOAD(0, ra, rb, amask);
        PN2_LSTORE(0, ra, rb, amask);
        __syncthreads();
        for (int step = 0; step < ksteps; ++step) {
            const int cur = step & 1;
            // branch-free prefetch: the last iteration re-loads the final tile into the idle stage (never read)
            const int nxt = step + 1 < ksteps ? step + 1 : step;
            PN2_GLOAD(nxt, ra, rb, amask);
            PN2_MFMAS(cur);
            PN2_LSTORE(cur ^ 1, ra, rb, amask);
            __syncthreads();
        }
    } else {
        // fp32 on the f32 matrix pipe: a K-step is 0.4-0.9 us of MFMAs per wave, shorter than a global-load round trip under load - with the prefetch one step
        // ahead the LDS store at the end of every step waited for its loads (62 TF/s).  Two register sets: the batch of step t+2 is requested before the MFMAs
        // of step t, the batch stored to LDS at the end of step t was requested a whole step earlier (loads return in order: the compiler's vmcnt leaves the
        // younger batch in flight).  Unrolled by two so that both sets are addressed statically.
        const int last = ksteps - 1;
        PN2_GLOAD(0, ra, rb, amask);
        PN2_GLOAD(last < 1 ? last : 1, ra1, rb1, amask1);
        PN2_LSTORE(0, ra, rb, amask);
        __syncthreads();
        // Both halves of a pair run unconditionally (an odd last step is peeled): with the second half under `if (step + 1 < ksteps)` hipcc SANK the first half's
        // requests into that branch (their only use), next to the second batch - one step of lead lost - and hoisted the zero-fill selects, and the wait with
        // them, to the top of the MFMA block.  The scheduling fences pin request | MFMAs | zero-fill + LDS store inside a half.
        int step = 0;
        for (; step + 1 < ksteps; step += 2) {
            PN2_GLOAD(step + 2 < last ? step + 2 : last, ra, rb, amask);
            __builtin_amdgcn_sched_barrier(0);
            PN2_MFMAS(0);
            __builtin_amdgcn_sched_barrier(0);
            PN2_LSTORE(1, ra1, rb1, amask1);
            __syncthreads();
            PN2_GLOAD(step + 3 < last ? step + 3 : last, ra1, rb1, amask1);
            __builtin_amdgcn_sched_barrier(0);
            PN2_MFMAS(1);
            __builtin_amdgcn_sched_barrier(0);
            PN2_LSTORE(0, ra, rb, amask);
            __syncthreads();
        }
        if (step < ksteps) {
            PN2_MFMAS(0);
            __syncthreads();          // (the epilogue re-uses the stages)
        }
    }
#undef PN2_GLOAD
#undef PN2_LSTORE
#undef PN2_MFMAS
#undef PN2_U4
#undef PN2_GLOAD_B
#undef PN2_LSTORE_B

    if constexpr (sizeof(T) == 4) {
        f32x4_t accf[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) accf[i][j] = f32x4_t{(float)acc[i][j][0], (float)acc[i][j][1], (float)acc[i][j][2], (float)acc[i][j][3]};
        conv_epilogue<T, BM, BN, WM, WN, MT, NT, EP>(accf, smem, d, ep, out, psum, psq, M, m0, n0, bm, pre);
    } else {
        conv_epilogue<T, BM, BN, WM, WN, MT, NT, EP>(acc, smem, d, ep, out, psum, psq, M, m0, n0, bm, pre);
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool PW, bool EP = false>
__global__ __launch_bounds__(256) void conv_gather_gemm(const T* __restrict__ in, const T* __restrict__ wp, T* __restrict__ out,
                                                        float* __restrict__ psum, float* __restrict__ psq, pn2_conv_desc d, pn2_conv_ep ep) {
    conv_gather_body<T, BM, BN, WM, WN, PW, EP>(in, wp, out, psum, psq, d, ep, blockIdx.x, gridDim.x);
}
// table-driven launch: many convs of one kernel instantiation from a DEVICE job table (see pn2_conv_gemm_multi)
template <typename T, int BM, int BN, int WM, int WN, bool PW, bool EP>
__global__ __launch_bounds__(256) void conv_gather_gemm_tab(const pn2_conv_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_conv_job j = jobs[jb];
    conv_gather_body<T, BM, BN, WM, WN, PW, EP>((const T*)j.in, (const T*)j.wp, (T*)j.out, j.psum, j.psq, j.d, j.ep, blockIdx.x - bstart[jb], bstart[jb + 1] - bstart[jb]);
}

// ------------------------------------------------------------------------------------------------
// forward / dgrad gather-GEMM, LDS-DMA pipeline (bf16): both operands go global -> LDS with 16-byte LDS-DMA loads
// (no register staging), NS stages deep, counted vmcnt + raw s_barrier so that up to NS-1 K-steps of loads stay in
// flight across the MFMAs.  The LDS image written by the DMA is lane-linear (128-byte rows, no padding); bank
// conflicts are removed by an XOR swizzle of the 16-byte chunk index with ((row>>1)&7), applied to the per-lane SOURCE
// address and to the fragment reads.  The activation operand is addressed through a buffer descriptor: padding taps /
// out-of-range rows get an offset past num_records and the hardware writes zeros (no zero page, no branches).
// ------------------------------------------------------------------------------------------------

// NS = 3: two K-steps of loads in flight; NS = 2: one step ahead and a third less LDS, so that three (64x128) instead of two workgroups share
// a CU - the per-shape tuner picks (the K loop runs at ~27 % of the MFMA rate with two resident workgroups: barrier / wait stalls).
// KS = 2 / 4 (conv_dma_gemm_ks, 512 / 1024 threads): INTRA-workgroup split-K for launches that are one partial wave of tiles with a long K loop (the 3x3 Res2Net
// branch convs of layer3 / layer4: 121-242 tiles, 15-30 K-steps, one wave per SIMD - nothing hides the DMA round trip of a K-step, 13-15 us whatever the
// tile, tools/gemm_codes_micro.py).  Every further group of four waves runs its share of the K-steps on a ring of its own, next to waves 0-3 on the same SIMDs;
// the accumulators meet in LDS (fp32, lane-linear) and waves 0-3 alone run the unchanged epilogue (the others have ended: a barrier only counts live waves).
// Same tile, half the serial chain, no partial tiles in memory and no reduce launch (the global split-K of PN2_CONV_SPLITK needs both).
template <int BM, int BN, int WM, int WN, bool PW, int NS, bool EP, int KS = 1>
__device__ __forceinline__ void conv_dma_body(const bf16_t* __restrict__ in, const bf16_t* __restrict__ wp, bf16_t* __restrict__ out,
                                              float* __restrict__ psum, float* __restrict__ psq, const pn2_conv_desc& d, const pn2_conv_ep& ep, int lbid, int lgrid, int by) {
    using T = bf16_t;
    constexpr int VEC = 8, BK = 64, ROW = 128;
    constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int STAGE = (BM + BN) * ROW;
    constexpr int NA = BM / 32, NB = BN / 32, LPS = NA + NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    PN2_STAMP_AT(0);
    const int kg = KS > 1 ? (int)(threadIdx.x >> 8) : 0;           // K group of this wave: group k takes the k-th share of the K-steps
    const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN, l15 = lane & 15, g = lane >> 4;
    const int M = d.N * d.OH * d.OW;
    const int nbn = (d.Cout + BN - 1) / BN;
    const int bid = xcd_remap(lbid, lgrid);
    if (KS > 1) by = kg;
    // (tile -> (row block, column block): a scalar division costs ~25 dependent instructions with two VALU round trips at the very top of the kernel;
    //  one column tile - every narrow conv - or a power-of-two count need none)
    int bn, bm;
    if (nbn == 1) { bn = 0; bm = bid; }
    else if ((nbn & (nbn - 1)) == 0) { const int sh_ = __builtin_ctz(nbn); bn = bid & (nbn - 1); bm = bid >> sh_; }
    else { bn = bid % nbn; bm = bid / nbn; }
    const int m0 = bm * BM, n0 = bn * BN;
    const int taps = d.KH * d.KW;
    const int ktot = taps * d.Cin_p;
    const int ksteps_all = (ktot + BK - 1) / BK;
    // split-K (PN2_CONV_SPLITK, small-M long-K convs): blockIdx.y owns the K-steps [kt0, kt0 + ksteps) and leaves an fp32 partial tile
    const int ksplit = KS > 1 ? KS : (d.flags >> 16) & 15;
    const int kper = ksplit > 1 ? (ksteps_all + ksplit - 1) / ksplit : ksteps_all;
    const int kt0 = ksplit > 1 ? by * kper : 0;
    const int ksteps = max(0, min(ksteps_all - kt0, kper));
    const int kloop = KS > 1 ? kper : ksteps;           // (KS > 1: every group passes the same barriers; a shorter one idles through its last steps)
    char* const ring = smem + (KS > 1 ? kg * NS * STAGE : 0);

    GatherGeom gg;
    gg.H = d.H; gg.W = d.W; gg.OH = d.OH; gg.OW = d.OW; gg.KH = d.KH; gg.KW = d.KW; gg.stride = d.stride;
    gg.sshift = __builtin_ctz(d.stride); gg.pad_h = d.pad_h; gg.pad_w = d.pad_w; gg.dil_h = d.dil_h; gg.dil_w = d.dil_w;
    gg.transposed = d.transposed;

    // this thread's DMA lanes: row (tid>>3) of every 32-row group, LDS chunk slot (tid&7) which holds GLOBAL chunk cg
    const int cg = (tid & 7) ^ ((tid >> 4) & 7);
    int rbase[NA], riy0[NA], rix0[NA]; bool rok[NA];
    {
        // pixel decode of the thread's rows m, m + 32, m + 64, ...: ONE pair of integer divisions (~60 instructions on this part), the other rows
        // are stepped from it (32 pixels further: a few wrap-arounds of the column / row counters) - the decode was ~240 of the ~400 prologue instructions
        const int mf = m0 + (tid >> 3);
        int n_ = 0, oy_ = 0, ox_ = 0;
        if (!PW) {
            const unsigned hw = (unsigned)(d.OH * d.OW), mu = (unsigned)mf;
            const unsigned nq = mu / hw, rem = mu - nq * hw, oq = rem / (unsigned)d.OW;
            n_ = (int)nq; oy_ = (int)oq; ox_ = (int)(rem - oq * (unsigned)d.OW);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = mf + 32 * i;
            rok[i] = m < M;
            if (PW) { rbase[i] = rok[i] ? m : 0; riy0[i] = 0; rix0[i] = 0; }
            else {
                rbase[i] = n_ * d.H * d.W;          // (rows past M: the decode runs on, the loads are masked by rok)
                if (!d.transposed) { riy0[i] = oy_ * d.stride - d.pad_h; rix0[i] = ox_ * d.stride - d.pad_w; }
                else { riy0[i] = oy_ + d.pad_h; rix0[i] = ox_ + d.pad_w; }
                ox_ += 32;
                while (ox_ >= d.OW) { ox_ -= d.OW; ++oy_; }
                while (oy_ >= d.OH) { oy_ -= d.OH; ++n_; }
            }
        }
    }
    int ci = cg * VEC + kt0 * BK, tap = 0;
    if (!PW) { while (ci >= d.Cin_p) { ci -= d.Cin_p; ++tap; } }
    // The activation operand goes through a BUFFER descriptor: a 32-bit byte offset per lane, and a lane whose tap falls into the padding (or
    // whose row / channel chunk does not exist) gets an offset beyond num_records - the hardware then writes zeros into the LDS slot.  No 64-bit
    // address arithmetic, no select against a zero page and, above all, no branches: with flat addresses the compiler guarded every
    // address computation with exec-mask branches (~35 vector + ~40 scalar instructions per row and K-step around 8..32 MFMAs).
    // The tap's column and its signed, dilated offsets are stepped with the tap; validity is one branch-free expression for the forward
    // gather (shift 0, mask 0) and the transposed one (offsets negated; stride-divisibility through `tmsk`, source pixel through `tsh`).
    // Host side: the tensor's extent from `in` stays below 2 GB (conv_gemm_impl falls back to the register-staged kernel otherwise).
    const unsigned INV = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)in >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)in)), 0, (int)INV, 0x00020000);
    const int sgn = d.transposed ? -1 : 1;
    const int sdil_h = sgn * d.dil_h, sdil_w = sgn * d.dil_w;
    const int tsh = d.transposed ? gg.sshift : 0, tmsk = d.transposed ? d.stride - 1 : 0;
    int tcol = 0, tdr = 0, tdc = 0;
    if (!PW && tap) { const int r0_ = tap / d.KW; tcol = tap - r0_ * d.KW; tdr = r0_ * sdil_h; tdc = tcol * sdil_w; }      // (tap 0 unless a split-K slice starts later)
    const int ldb = d.ld_in * 2;                       // row pitch in bytes
    unsigned rowoff[NA];                               // PW: byte offset of the row; else: pixel index of the image origin
#pragma unroll
    for (int i = 0; i < NA; ++i) rowoff[i] = PW ? (unsigned)rbase[i] * (unsigned)ldb : (unsigned)rbase[i];
    const T* bptr = wp + (size_t)(n0 + (tid >> 3)) * d.Kp + cg * VEC;
    // wave-uniform LDS row offset of this wave inside a 32-row DMA group
    const int wrow = __builtin_amdgcn_readfirstlane(wid * 8);

// (a non-temporal policy on the activation operand's DMA wins 14-24 % on cold operands and loses inside the step, where the operand still sits in L2 / MALL: DESIGN 6)
#define PN2_ISSUE(step_, buf_)                                                                                         \
    do {                                                                                                               \
        char* sb_ = ring + (buf_) * STAGE;                                                                             \
        if (PW) {                                                                                                      \
            const int k_ = (step_) * BK + cg * VEC;                                                                    \
            const bool kok_ = k_ < d.Cin_p;                                                                            \
            _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
                const unsigned vo_ = (rok[i] && kok_) ? rowoff[i] + (unsigned)k_ * 2u : INV;                           \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(sb_ + (i * 32 + wrow) * ROW), 16, (int)vo_, 0, 0, 0); \
            }                                                                                                          \
        } else {                                                                                                       \
            const bool tok_ = tap < taps;                                                                              \
            _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                           \
                const int ty_ = riy0[i] + tdr, tx_ = rix0[i] + tdc;                                                    \
                const int iy_ = ty_ >> tsh, ix_ = tx_ >> tsh;                                                          \
                const bool ok_ = rok[i] & tok_ & ((unsigned)iy_ < (unsigned)d.H) & ((unsigned)ix_ < (unsigned)d.W) & (((ty_ | tx_) & tmsk) == 0); \
                const unsigned vo_ = ok_ ? (rowoff[i] + (unsigned)(iy_ * d.W + ix_)) * (unsigned)ldb + (unsigned)ci * 2u : INV; \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lptr_t)(sb_ + (i * 32 + wrow) * ROW), 16, (int)vo_, 0, 0, 0); \
            }                                                                                                          \
            ci += BK;                                                                                                  \
            while (ci >= d.Cin_p) {                                                                                    \
                ci -= d.Cin_p; ++tap; ++tcol; tdc += sdil_w;                                                           \
                if (tcol == d.KW) { tcol = 0; tdc = 0; tdr += sdil_h; }                                                \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                                                 \
            __builtin_amdgcn_global_load_lds((gptr_t)(bptr + (size_t)(32 * i) * d.Kp + (size_t)(step_) * BK),         \
                                             (lptr_t)(sb_ + BM * ROW + (i * 32 + wrow) * ROW), 16, 0, 0);              \
    } while (0)

    BnbPre<T, BM, BN> pre;
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets: global chunk (ks*4 + g) of row (.. + l15) sits in slot chunk ^ ((row>>1)&7); tile bases are multiples of 16
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
    const int key = (l15 >> 1) & 7;
    const unsigned so0 = ((g) ^ key) * 16, so1 = ((4 + g) ^ key) * 16;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)ring;     // LDS byte address of this K group's ring

    PN2_STAMP_AT(1);
    if (ksteps > 0) PN2_ISSUE(kt0, 0);
    if (NS == 3 && ksteps > 1) PN2_ISSUE(kt0 + 1, 1);
    PN2_STAMP_AT(2);
    for (int t = 0; t < kloop; ++t) {
        if constexpr (NS == 3) {
            // my own DMA of step t has landed once at most one later step (LPS loads) is still outstanding
            if (t + 1 < ksteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // everyone's step-t data is in LDS; everyone finished reading step t-1
#ifdef PN2_STAMP
            if (t == 0) PN2_STAMP_AT(3);
#endif
            if (t + 2 < ksteps) {
                const int nb_ = (t + 2) % NS;
                PN2_ISSUE(kt0 + t + 2, nb_);
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // only step t is outstanding
            __builtin_amdgcn_s_barrier();              // step t is in LDS for everyone; everyone finished reading the other buffer (step t-1)
#ifdef PN2_STAMP
            if (t == 0) PN2_STAMP_AT(3);
#endif
            if (t + 1 < ksteps) {
                const int nb_ = (t + 1) % NS;
                PN2_ISSUE(kt0 + t + 1, nb_);
            }
        }
        if (KS > 1 && t >= ksteps) continue;           // (a shorter K group: nothing left to multiply, the barrier above is all it owes)
        // Fragment reads are inline asm: for a compiler-visible LDS load hipcc drains ALL outstanding LDS-DMA (vmcnt(0)) first,
        // which would collapse the pipeline to depth 1.  DS operations return in order, so counted lgkmcnt waits are exact.
        const unsigned As = lds0 + (t % NS) * STAGE + (wm * WTM + l15) * ROW;
        const unsigned Bs = lds0 + (t % NS) * STAGE + BM * ROW + (wn * WTN + l15) * ROW;
        u32x4_t a0[MT], a1[MT], b0[NT], b1[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a0[i]) : "v"(As + i * 16 * ROW + so0));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(b0[j]) : "v"(Bs + j * 16 * ROW + so0));
#pragma unroll
        for (int i = 0; i < MT; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(a1[i]) : "v"(As + i * 16 * ROW + so1));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(b1[j]) : "v"(Bs + j * 16 * ROW + so1));
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(MT + NT) : "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) MMA<T>::run(acc[i][j], __builtin_bit_cast(uint4, b0[j]), __builtin_bit_cast(uint4, a0[i]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) MMA<T>::run(acc[i][j], __builtin_bit_cast(uint4, b1[j]), __builtin_bit_cast(uint4, a1[i]));
    }
#undef PN2_ISSUE
    __syncthreads();
    PN2_STAMP_AT(4);

    if constexpr (KS > 1) {      // the K groups' accumulators meet in LDS (lane-linear fp32: conflict-free); groups 1.. end here
        float* xs = reinterpret_cast<float*>(smem);
        constexpr int XG = MT * NT * 4 * 256;          // floats per group
        if (kg > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xs[(kg - 1) * XG + ((i * NT + j) * 4 + r) * 256 + tid] = acc[i][j][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int k = 0; k < KS - 1; ++k)          // fixed order: group 1, 2, 3
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] += xs[k * XG + ((i * NT + j) * 4 + r) * 256 + tid];
        __syncthreads();          // (4 live waves from here on) the exchange area is the epilogue's staging area next
    }
    if (KS == 1 && ksplit > 1) { // fp32 partial tile -> workspace [ksplit][M][Cout] (psum); pn2_conv_splitk_reduce finishes (sum, stats, bias, store)
        float* ws = psum + (size_t)by * M * d.Cout;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WTM + i * 16 + l15, col = n0 + wn * WTN + j * 16 + g * 4 + r;      // (operands exchanged: see conv_epilogue<SWP>)
                    if (m < M && col < d.Cout) ws[(size_t)m * d.Cout + col] = acc[i][j][r];
                }
        return;
    }
    conv_epilogue<T, BM, BN, WM, WN, MT, NT, EP, true>(acc, smem, d, ep, out, psum, psq, M, m0, n0, bm, pre);
#ifdef PN2_STAMP
    PN2_STAMP_AT(8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PN2_STAMP_AT(9);
    pn2_stamp_hw();
#endif
}

template <int BM, int BN, int WM, int WN, bool PW, int NS = 3, bool EP = false>
__global__ __launch_bounds__(256) void conv_dma_gemm(const bf16_t* __restrict__ in, const bf16_t* __restrict__ wp, bf16_t* __restrict__ out,
                                                     float* __restrict__ psum, float* __restrict__ psq, pn2_conv_desc d, pn2_conv_ep ep) {
    conv_dma_body<BM, BN, WM, WN, PW, NS, EP>(in, wp, out, psum, psq, d, ep, blockIdx.x, gridDim.x, blockIdx.y);
}
template <int BM, int BN, int WM, int WN, bool EP>
__global__ __launch_bounds__(512) void conv_dma_gemm_tab_ks2(const pn2_conv_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_conv_job j = jobs[jb];
    conv_dma_body<BM, BN, WM, WN, false, 3, EP, 2>((const bf16_t*)j.in, (const bf16_t*)j.wp, (bf16_t*)j.out, j.psum, j.psq, j.d, j.ep, blockIdx.x - bstart[jb], bstart[jb + 1] - bstart[jb], 0);
}
template <int BM, int BN, int WM, int WN, bool PW, int NS, bool EP, int KS>
__global__ __launch_bounds__(256 * KS) void conv_dma_gemm_ks(const bf16_t* __restrict__ in, const bf16_t* __restrict__ wp, bf16_t* __restrict__ out,
                                                             float* __restrict__ psum, float* __restrict__ psq, pn2_conv_desc d, pn2_conv_ep ep) {
    conv_dma_body<BM, BN, WM, WN, PW, NS, EP, KS>(in, wp, out, psum, psq, d, ep, blockIdx.x, gridDim.x, 0);
}
template <int BM, int BN, int WM, int WN, bool PW, int NS, bool EP>
__global__ __launch_bounds__(256) void conv_dma_gemm_tab(const pn2_conv_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_conv_job j = jobs[jb];
    conv_dma_body<BM, BN, WM, WN, PW, NS, EP>((const bf16_t*)j.in, (const bf16_t*)j.wp, (bf16_t*)j.out, j.psum, j.psq, j.d, j.ep, blockIdx.x - bstart[jb], bstart[jb + 1] - bstart[jb], 0);
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
        return tr_frag_bf16(tile, rs, chan0, lane);
    }
    __device__ static __forceinline__ void run(f32x4_t& acc, const uint4& a, const uint4& b) { MMA<bf16_t>::run(acc, a, b); }
    static constexpr int NFRAG = 1;
};
template <> struct WG<float> {
    static constexpr int PAD = 64;
    static constexpr int NFRAG = 2;  // two uint4 = 8 pixels-slots per lane per 32-pixel step
};
template <> struct WG<f32f_t> : WG<float> {};

template <typename T, int BMC, int BNK, int WM, int WN, bool PW>
__device__ __forceinline__ void conv_wgrad_body(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ slab, const pn2_wgrad_desc& d, int nsplit, int bloc) {
    // bloc: workgroup index inside this conv's launch (== blockIdx.x for a single-conv launch; table launches start every job
    // at a multiple of 8 so that bloc % 8 is still the XCD the block runs on)
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
    // XCD-aware mapping (block b runs on XCD b%8): every (co,k) tile of one pixel split goes to the SAME XCD, back to back, so the
    // dy / x slice of that split is fetched into one L2 once instead of once per XCD (PMC showed ~4x over-fetch otherwise).
    const int ntile = (d.Rp / BMC) * tk;
    const int xcd = bloc & 7, jx = bloc >> 3;
    const int split = xcd + 8 * (jx / ntile), tile = jx % ntile;
    if (split >= nsplit) return;
    const int bco = tile / tk, bk = tile % tk;
    const int co0 = bco * BMC, k0 = bk * BNK;
    const int total_steps = (M + WGP - 1) / WGP;
    const int spb = (total_steps + nsplit - 1) / nsplit;
    const int s_begin = split * spb;
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

    // unconditional loads from clamped addresses, zero fill applied at the LDS write (see conv_gather_gemm)
    uint4 ry[NY], rx[NX];
    unsigned ymask = 0, xmask = 0;
    const int yco = (y_active && yc_ok) ? co0 + ycv * VEC : 0;
    auto gload = [&](int step, bool live) {
        const int mb = step * WGP;
        ymask = 0; xmask = 0;
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int m = mb + yrow + i * RSTEPY;
            const bool ok = live && y_active && yc_ok && m < M;
            if (ok) ymask |= 1u << i;
            ry[i] = *reinterpret_cast<const uint4*>(dy + (size_t)(ok ? m : 0) * d.ld_dy + yco);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int m = mb + xrow + i * RSTEPX;
            if (PW) {
                const bool ok = live && xk_ok && m < M;
                if (ok) xmask |= 1u << i;
                rx[i] = *reinterpret_cast<const uint4*>(x + (ok ? (size_t)m * d.ld_x + kk : 0));
            } else {
                const int iy = poy[i] * d.stride - d.pad_h + xr * d.dil_h, ix = pox[i] * d.stride - d.pad_w + xs * d.dil_w;
                const bool ok = live && xk_ok && m < M && (unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W;
                if (ok) xmask |= 1u << i;
                rx[i] = *reinterpret_cast<const uint4*>(x + (ok ? ((size_t)(pn[i] * d.H + iy) * d.W + ix) * d.ld_x + xci : 0));
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
            if (y_active) *reinterpret_cast<uint4*>(Ys + (yrow + i * RSTEPY) * RSY + ycv * 16) = (ymask >> i) & 1u ? ry[i] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NX; ++i) *reinterpret_cast<uint4*>(Xs + (xrow + i * RSTEPX) * RSX + xkv * 16) = (xmask >> i) & 1u ? rx[i] : make_uint4(0, 0, 0, 0);
    };

    typename MMA<T>::acc_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = typename MMA<T>::acc_t{0, 0, 0, 0};

    if (s_begin < s_end) {
        gload(s_begin, true);
        lstore(0);
        __syncthreads();
        for (int step = s_begin; step < s_end; ++step) {
            const int cur = (step - s_begin) & 1;
            gload(step + 1, step + 1 < s_end);      // branch-free prefetch: the last one is dead (all lanes masked, address clamped)
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
                f32x4_t part[MMA<T>::F64ROWS ? 1 : MT][MMA<T>::F64ROWS ? 1 : NT];
#ifndef PN2_WCH
#define PN2_WCH 8
#endif
                constexpr int WCH = PN2_WCH;          // MFMAs per chain (x 4 pixels): 8 = the 32 pixels of a stage (4: +1.4 % step time for gradients no closer to float64 - bs32 probes: median rel-L2 1.7e-6 / 1.9e-6, reference fp32 3.3e-6)
#pragma unroll
                for (int q = 0; q < WGP / 4; ++q) {       // 4 pixels per 16x16x4 MFMA
                    float a[MT], b[NT];
#pragma unroll
                    for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const float*>(Ys + (q * 4 + g) * RSY + (wm * WTM + i * 16 + l15) * 4);
#pragma unroll
                    for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const float*>(Xs + (q * 4 + g) * RSX + (wn * WTN + j * 16 + l15) * 4);
#pragma unroll
                    for (int i = 0; i < MT; ++i)
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            if constexpr (MMA<T>::F64ROWS) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64((double)a[i], (double)b[j], acc[i][j], 0, 0, 0);
                            else {          // PN2_F32F: chains of WCH MFMAs (4 WCH pixels) from C = 0, met by round-to-nearest adds (see MMA<f32f_t>)
                                if ((q & (WCH - 1)) == 0) part[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                                else part[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], part[i][j], 0, 0, 0);
                                if ((q & (WCH - 1)) == WCH - 1) acc[i][j] += part[i][j];
                            }
                        }
                }
            }
            lstore(cur ^ 1);
            __syncthreads();
        }
    }
    float* dst = slab + ((size_t)split * d.Rp + co0) * d.Kp + k0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // f64 MFMA (fp32 path): C/D row = (lane>>4) + 4*reg; every other form: (lane>>4)*4 + reg
                const int row = MMA<T>::F64ROWS ? g + 4 * r : g * 4 + r;
                dst[(size_t)(wm * WTM + i * 16 + row) * d.Kp + wn * WTN + j * 16 + l15] = (float)acc[i][j][r];
            }
}

// ------------------------------------------------------------------------------------------------
// weight gradient, LDS-DMA pipeline (bf16): 64 pixels per step, 3-deep ring, both operands by global_load_lds.
// LDS image is chunk-linear [pixel][channel] (no padding); the 16-byte chunk index is XOR-ed with ((pixel&7)<<1) on the
// SOURCE side and in the ds_read_b64_tr_b16 addresses, which spreads the 8 pixel rows touched by one transpose-read
// over all 64 banks.  Same k-slot <-> pixel permutation for both operands as conv_wgrad.
// ------------------------------------------------------------------------------------------------
#ifndef PN2_WG_PX
#define PN2_WG_PX 32
#endif
// pixels per ring stage of the LDS-DMA wgrad kernel: 32-pixel stages (48 KB ring) leave room for three workgroups per CU
constexpr int wg_px(int bmc) { return bmc >= 64 ? PN2_WG_PX : 64; }
// BNK = 256 (co tile 128 only): every dy tile is staged once per 256 instead of once per 128 contraction columns.  The kernel runs at the
// throughput of the L2 -> LDS DMA path (~14 TB/s over the chip: 16 GB of tile traffic for 2.7 GB of operands in the pointwise launch of a step), so
// what helps is fewer bytes per flop: (BMC + BNK) / (BMC * BNK) = 1/64 -> 1/85.  The wave tile becomes 64 x 128 (128 accumulator registers).
template <int BMC, int WM, int WN, bool PW, int BNK = 128>
__device__ __forceinline__ void conv_wgrad_dma_body(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ slab,
                                                    const pn2_wgrad_desc& d, int nsplit, int bloc) {
    constexpr int PX = wg_px(BMC), NS = 3;
    constexpr int WTM = BMC / WM, WTN = BNK / WN, MT = WTM / 16, NT = WTN / 16;
    constexpr int RBY = BMC * 2, RBX = BNK * 2;              // row bytes
    constexpr int CHY = BMC / 8, CHX = BNK / 8;              // 16-byte chunks per row
    constexpr int NYI = PX * CHY / 256, NXI = PX * CHX / 256, LPS = NYI + NXI;
    constexpr int KMY = CHY / 2 - 1 < 7 ? CHY / 2 - 1 : 7, KMX = 7;
    constexpr int STAGE = PX * (RBY + RBX);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN, l15 = lane & 15, g = lane >> 4;
    const int M = d.N * d.OH * d.OW;
    const int tk = (d.Kp + BNK - 1) / BNK;
    const int ntile = (d.Rp / BMC) * tk;
    const int xcd = bloc & 7, jx = bloc >> 3;
    const int split = xcd + 8 * (jx / ntile), tile = jx % ntile;
    if (split >= nsplit) return;
    const int bco = tile / tk, bk = tile % tk;
    const int co0 = bco * BMC, k0 = bk * BNK;
    const int total_steps = (M + PX - 1) / PX;
    const int spb = (total_steps + nsplit - 1) / nsplit;
    const int s_begin = split * spb;
    int s_end = s_begin + spb; if (s_end > total_steps) s_end = total_steps;
    const int nsteps = s_end > s_begin ? s_end - s_begin : 0;
    const int taps = d.KH * d.KW;

    // ---- DMA lanes.  dy: row (tid / CHY) of each group of 256/CHY rows, LDS slot tid % CHY holding global chunk gy
    const int yrow = tid / CHY, gy = (tid % CHY) ^ ((yrow & KMY) << 1);
    const bool yc_ok = co0 + gy * 8 < d.Cout_p;
    const int xrow = tid / CHX, gx = (tid % CHX) ^ ((xrow & KMX) << 1);
    const int kk = k0 + gx * 8;
    int xtap = 0, xci = kk;
    if (!PW) { xtap = kk / d.Cin_p; xci = kk - xtap * d.Cin_p; }
    const bool xk_ok = PW ? (kk < d.Cin_p) : (xtap < taps);
    const int xr = PW ? 0 : xtap / d.KW, xs = PW ? 0 : xtap - (xtap / d.KW) * d.KW;
    int pn[NXI], poy[NXI], pox[NXI];
    if (!PW) {
#pragma unroll
        for (int i = 0; i < NXI; ++i) {
            int m = s_begin * PX + xrow + i * (256 / CHX);
            if (m >= M) m = M - 1;
            const int hw = d.OH * d.OW;
            const int n = m / hw, rem = m - n * hw;
            pn[i] = n; poy[i] = rem / d.OW; pox[i] = rem - poy[i] * d.OW;
        }
    }
    const int wch = __builtin_amdgcn_readfirstlane(wid * 64);      // this wave's first chunk inside a 256-chunk DMA instruction

    // Both operands go through buffer descriptors (see conv_dma_body): 32-bit byte offsets that advance by a constant per step, zeros from the
    // hardware for rows past M / padding taps / channel chunks past the tensor - no 64-bit address arithmetic, no zero page, no branches.
    const unsigned INV = 0x80000000u;
    auto mkrs = [&](const void* p_) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)p_ >> 32)) << 32) |
                                                         (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned long long)p_)), 0, (int)INV, 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rs_y = mkrs(dy), rs_x = mkrs(x);
    const unsigned ldyb = (unsigned)d.ld_dy * 2u, ldxb = (unsigned)d.ld_x * 2u;
    unsigned yoff[NYI], xoff[NXI];                     // byte offsets of this lane's loads at the current step (PW x: too)
    int ym[NYI], xm[NXI];                              // their pixel rows
#pragma unroll
    for (int i = 0; i < NYI; ++i) { ym[i] = s_begin * PX + yrow + i * (256 / CHY); yoff[i] = (unsigned)ym[i] * ldyb + (unsigned)(co0 + gy * 8) * 2u; }
#pragma unroll
    for (int i = 0; i < NXI; ++i) { xm[i] = s_begin * PX + xrow + i * (256 / CHX); xoff[i] = (unsigned)xm[i] * ldxb + (unsigned)kk * 2u; }
    const int tdy = xr * d.dil_h - d.pad_h, tdx = xs * d.dil_w - d.pad_w;      // this lane's tap offset (its k chunk never changes)

#define PN2_WISSUE(step_, buf_)                                                                                        \
    do {                                                                                                               \
        char* sb_ = smem + (buf_) * STAGE;                                                                             \
        _Pragma("unroll") for (int i = 0; i < NYI; ++i) {                                                              \
            const unsigned vo_ = (yc_ok && ym[i] < M) ? yoff[i] : INV;                                                 \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lptr_t)(sb_ + (i * 256 + wch) * 16), 16, (int)vo_, 0, 0, 0); \
            ym[i] += PX; yoff[i] += PX * ldyb;                                                                         \
        }                                                                                                              \
        _Pragma("unroll") for (int i = 0; i < NXI; ++i) {                                                              \
            unsigned vo_ = INV;                                                                                        \
            if (PW) {                                                                                                  \
                if (xk_ok && xm[i] < M) vo_ = xoff[i];                                                                 \
                xoff[i] += PX * ldxb;                                                                                  \
            } else {                                                                                                   \
                const int iy_ = poy[i] * d.stride + tdy, ix_ = pox[i] * d.stride + tdx;                                \
                const bool ok_ = xk_ok & (xm[i] < M) & ((unsigned)iy_ < (unsigned)d.H) & ((unsigned)ix_ < (unsigned)d.W); \
                vo_ = ok_ ? (unsigned)((pn[i] * d.H + iy_) * d.W + ix_) * ldxb + (unsigned)xci * 2u : INV;             \
                pox[i] += PX;                                                                                          \
                while (pox[i] >= d.OW) { pox[i] -= d.OW; ++poy[i]; }                                                   \
                while (poy[i] >= d.OH) { poy[i] -= d.OH; ++pn[i]; }                                                    \
            }                                                                                                          \
            xm[i] += PX;                                                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(sb_ + PX * RBY + (i * 256 + wch) * 16), 16, (int)vo_, 0, 0, 0); \
        }                                                                                                              \
    } while (0)

    // (a v_mfma_f32_32x32x16_bf16 form of this loop was built and measured in round 6: correct, 0.6 % slower in the step - profiles/r06_mfma_shape_micro.txt)
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment addressing (per lane): pixel row g*4 + (l15>>2) (+16, +32*ks), 8 bytes at channel (l15&3)*4 of a 16-channel block
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    const int prow = g * 4 + (l15 >> 2);
    const int keyy = (prow & KMY) << 1, keyx = (prow & KMX) << 1;
    unsigned offA[MT], offB[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) offA[i] = prow * RBY + ((((wm * WTM) >> 3) + i * 2 + ((l15 & 3) >> 1)) ^ keyy) * 16 + (l15 & 1) * 8;
#pragma unroll
    for (int j = 0; j < NT; ++j) offB[j] = PX * RBY + prow * RBX + ((((wn * WTN) >> 3) + j * 2 + ((l15 & 3) >> 1)) ^ keyx) * 16 + (l15 & 1) * 8;

    if (nsteps > 0) {
        PN2_WISSUE(s_begin, 0);
        if (nsteps > 1) PN2_WISSUE(s_begin + 1, 1);
        for (int t = 0; t < nsteps; ++t) {
            if (t + 1 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (t + 2 < nsteps) {
                const int nb_ = (t + 2) % NS;
                PN2_WISSUE(s_begin + t + 2, nb_);
            }
            const unsigned sb = lds0 + (t % NS) * STAGE;
#pragma unroll
            for (int ks = 0; ks < PX / 32; ++ks) {
                u32x2_t a0[MT], a1[MT], b0[NT], b1[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a0[i]) : "v"(sb + offA[i] + ks * 32 * RBY));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(a1[i]) : "v"(sb + offA[i] + (ks * 32 + 16) * RBY));
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b0[j]) : "v"(sb + offB[j] + ks * 32 * RBX));
                    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(b1[j]) : "v"(sb + offB[j] + (ks * 32 + 16) * RBX));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        MMA<bf16_t>::run(acc[i][j], make_uint4(a0[i].x, a0[i].y, a1[i].x, a1[i].y), make_uint4(b0[j].x, b0[j].y, b1[j].x, b1[j].y));
            }
        }
    }
#undef PN2_WISSUE
    float* dst = slab + ((size_t)split * d.Rp + co0) * d.Kp + k0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if (BNK > 128 && k0 + wn * WTN + j * 16 >= d.Kp) continue;          // Kp is a multiple of 128: the last 256-wide tile may be half empty
#pragma unroll
            for (int r = 0; r < 4; ++r)
                dst[(size_t)(wm * WTM + i * 16 + g * 4 + r) * d.Kp + wn * WTN + j * 16 + l15] = acc[i][j][r];
        }
}

// single-conv and table-driven (many convs, one launch) entry points of the two wgrad kernels
template <typename T, int BMC, int BNK, int WM, int WN, bool PW>
__global__ __launch_bounds__(256) void conv_wgrad(const T* __restrict__ dy, const T* __restrict__ x, float* __restrict__ slab, pn2_wgrad_desc d, int nsplit) {
    conv_wgrad_body<T, BMC, BNK, WM, WN, PW>(dy, x, slab, d, nsplit, blockIdx.x);
}
// Table launches walk the job table with a grid stride (gridDim.x == total blocks: every workgroup runs one (job, block) pair)
template <typename T, int BMC, int BNK, int WM, int WN, bool PW>
__global__ __launch_bounds__(256) void conv_wgrad_tab(const pn2_wgrad_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int total = bstart[njobs];
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        const int jb = find_job(bstart, njobs, b);
        const pn2_wgrad_job j = jobs[jb];
        const int bl = b - bstart[jb];          // (the job's XCD rotation relabels the low three bits: the body reads the split's XCD slot from them)
        conv_wgrad_body<T, BMC, BNK, WM, WN, PW>((const T*)j.dy, (const T*)j.x, j.slab, j.d, j.nsplit, (bl & ~7) | ((bl - j.rot) & 7));
        __syncthreads();            // the next pair restages LDS
    }
}
template <int BMC, int WM, int WN, bool PW, int BNK = 128>
__global__ __launch_bounds__(256) void conv_wgrad_dma(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ slab, pn2_wgrad_desc d, int nsplit) {
    conv_wgrad_dma_body<BMC, WM, WN, PW, BNK>(dy, x, slab, d, nsplit, blockIdx.x);
}
template <int BMC, int WM, int WN, bool PW, int BNK = 128>
__global__ __launch_bounds__(256) void conv_wgrad_dma_tab(const pn2_wgrad_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int total = bstart[njobs];
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        const int jb = find_job(bstart, njobs, b);
        const pn2_wgrad_job j = jobs[jb];
        const int bl = b - bstart[jb];
        conv_wgrad_dma_body<BMC, WM, WN, PW, BNK>((const bf16_t*)j.dy, (const bf16_t*)j.x, j.slab, j.d, j.nsplit, (bl & ~7) | ((bl - j.rot) & 7));
        __syncthreads();
    }
}

// One table for the pointwise AND the k x k jobs of the 128 x 256 tile (variant 14): the k x k jobs are chains of ~121 32-pixel stages that leave the memory system idle
// (1.2 TB/s over 330 us), the pointwise table streams at the HBM rate (4.7 TB/s over 670 us) - in ONE launch, job ranges interleaved by the host, a CU holds a
// workgroup of each kind at a time.  Same per-job arithmetic as the two tables: bit-identical gradients.
template <int BMC, int WM, int WN, int BNK>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_tab_mix(const pn2_wgrad_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    const int total = bstart[njobs];
    for (int b = blockIdx.x; b < total; b += gridDim.x) {
        const int jb = find_job(bstart, njobs, b);
        const pn2_wgrad_job j = jobs[jb];
        const int bl = b - bstart[jb];
        const int lb = (bl & ~7) | ((bl - j.rot) & 7);
        if (j.d.KH == 1 && j.d.KW == 1 && j.d.stride == 1 && j.d.pad_h == 0 && j.d.pad_w == 0)
            conv_wgrad_dma_body<BMC, WM, WN, true, BNK>((const bf16_t*)j.dy, (const bf16_t*)j.x, j.slab, j.d, j.nsplit, lb);
        else
            conv_wgrad_dma_body<BMC, WM, WN, false, BNK>((const bf16_t*)j.dy, (const bf16_t*)j.x, j.slab, j.d, j.nsplit, lb);
        __syncthreads();
    }
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
        TT<T>::st(wp + (size_t)row * (p.ld ? p.ld : p.Kp) + p.koff + k, v);
    }
}

__host__ __device__ inline int log2phys(int c, int gw, int gwp) { const int g = c / gw; return g * gwp + (c - g * gw); }

// tile of the table-driven repack: CO x CI logical channels, all taps, staged through <= 32 KiB of LDS
__host__ __device__ inline void pack_tile(int taps, int& CO, int& CI) {
    if (taps == 1) { CO = 64; CI = 64; }
    else if (taps <= 7) { CO = 32; CI = 32; }
    else if (taps <= 9) { CO = 16; CI = 32; }
    else if (taps <= 25) { CO = 16; CI = 16; }
    else { CO = 4; CI = 8; }      // taps <= 225
}
inline int pack_blocks(const pn2_pack_desc& p) {
    int CO, CI; pack_tile(p.KH * p.KW, CO, CI);
    return ((p.Cout + CO - 1) / CO) * ((p.Cin + CI - 1) / CI);
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi(const pn2_pack_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    // one launch refreshes every conv panel of the model (forward + dgrad layouts).  A block owns a CO x CI x taps tile of one
    // weight: the OIHW fp32 master is read in contiguous runs of CI*taps floats, the panel is written in runs along its own
    // contiguous axis (ci for the forward layout, co for the transposed one).  Pad slots are never written: they hold the
    // zeros pn2_pack_weight() put there when the panel was created.
    __shared__ float tile[7296];      // max over pack_tile() of CO * (CI * taps + 1)
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_pack_job j = jobs[jb];
    const pn2_pack_desc p = j.d;
    const float* __restrict__ w = j.w;
    T* __restrict__ wp = reinterpret_cast<T*>(j.wp);
    const int taps = p.KH * p.KW;
    int CO, CI; pack_tile(taps, CO, CI);
    const int local = blockIdx.x - bstart[jb];
    const int nci = (p.Cin + CI - 1) / CI;
    const int co0 = (local / nci) * CO, ci0 = (local - (local / nci) * nci) * CI;
    const int nco_t = min(CO, p.Cout - co0), nci_t = min(CI, p.Cin - ci0);
    const int run = nci_t * taps, rs = CI * taps + 1;     // +1: the transposed read walks LDS with stride rs
    // index math without integer divisions (the flat e / run, e / nci_t, t2 / taps form made this kernel ALU-bound at 1 TB/s):
    // a wave owns tile rows, its lanes walk (tap, channel) pairs with the power-of-two tile width peeled off by shift / mask
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int col = ty; col < nco_t; col += 4) {
        const float* src = w + ((size_t)(co0 + col) * p.Cin + ci0) * taps;
        for (int r = tx; r < run; r += 64) tile[col * rs + r] = src[r];
    }
    __syncthreads();
    const size_t ld = p.ld ? p.ld : p.Kp;
    const float inv_gi = 1.f / (float)p.gw_in, inv_go = 1.f / (float)p.gw_out;
    auto phys_in = [&](int c) { const int g = (int)(((float)c + 0.5f) * inv_gi); return g * p.gwp_in + (c - g * p.gw_in); };
    auto phys_out = [&](int c) { const int g = (int)(((float)c + 0.5f) * inv_go); return g * p.gwp_out + (c - g * p.gw_out); };
    if (!p.transposed) {
        const int sh = 31 - __clz(CI);
        for (int col = ty; col < nco_t; col += 4) {
            T* dst = wp + (size_t)phys_out(co0 + col) * ld + p.koff;
            for (int cb = tx; cb < taps * CI; cb += 64) {
                const int tap = cb >> sh, cil = cb & (CI - 1);
                if (cil < nci_t) TT<T>::st(dst + tap * p.Cin_p + phys_in(ci0 + cil), tile[col * rs + cil * taps + tap]);
            }
        }
    } else {
        const int sh = 31 - __clz(CO);
        for (int cil = ty; cil < nci_t; cil += 4) {
            T* dst = wp + (size_t)phys_in(ci0 + cil) * ld + p.koff;
            for (int cb = tx; cb < taps * CO; cb += 64) {
                const int tap = cb >> sh, col = cb & (CO - 1);
                if (col < nco_t) TT<T>::st(dst + tap * p.Cout_p + phys_out(co0 + col), tile[col * rs + cil * taps + tap]);
            }
        }
    }
}

__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slab, float* __restrict__ gw, const pn2_pack_desc& p, int nsplit, int accumulate,
                                                  int blk, int nblk) {
    // Sum the nsplit slabs and scatter into the OIHW gradient.  A lane owns 4 consecutive packed k of one row (16-byte loads, a
    // wave reads contiguous 256..1024-byte runs of every slab); the slabs are dealt round-robin to S sub-lanes (S = 1..16, more
    // when the matrix is small and the split count large) that are combined by a fixed-order butterfly -> deterministic.
    const int taps = p.KH * p.KW;
    const int ktot = taps * p.Cin_p;            // multiple of 8
    const int kv = ktot >> 2;
    const size_t total = (size_t)p.Cout_p * kv;
    const size_t sstride = (size_t)p.Rp * p.Kp;
    int S = 1;
    while (S < 16 && total * S < 32768 && 2 * S <= nsplit) S <<= 1;
    const int EPW = 64 / S;                     // element vectors per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / EPW, ei = lane - sub * EPW;
    const size_t nwaves = (size_t)nblk * 4;
    for (size_t base = ((size_t)blk * 4 + wave) * EPW; base < total; base += nwaves * EPW) {
        const size_t idx = base + ei;
        const bool live = idx < total;
        const size_t ii = live ? idx : 0;
        const int prow = (int)(ii / kv), k = (int)(ii - (size_t)prow * kv) << 2;
        const float* s = slab + (size_t)prow * p.Kp + k;
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0, v2 = v0, v3 = v0;
        int i = sub;
        for (; i + 3 * S < nsplit; i += 4 * S) {
            const float4 a0 = *reinterpret_cast<const float4*>(s + (size_t)i * sstride), a1 = *reinterpret_cast<const float4*>(s + (size_t)(i + S) * sstride);
            const float4 a2 = *reinterpret_cast<const float4*>(s + (size_t)(i + 2 * S) * sstride), a3 = *reinterpret_cast<const float4*>(s + (size_t)(i + 3 * S) * sstride);
            v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w; v1.x += a1.x; v1.y += a1.y; v1.z += a1.z; v1.w += a1.w;
            v2.x += a2.x; v2.y += a2.y; v2.z += a2.z; v2.w += a2.w; v3.x += a3.x; v3.y += a3.y; v3.z += a3.z; v3.w += a3.w;
        }
        for (; i < nsplit; i += S) { const float4 a0 = *reinterpret_cast<const float4*>(s + (size_t)i * sstride); v0.x += a0.x; v0.y += a0.y; v0.z += a0.z; v0.w += a0.w; }
        float v[4] = {(v0.x + v1.x) + (v2.x + v3.x), (v0.y + v1.y) + (v2.y + v3.y), (v0.z + v1.z) + (v2.z + v3.z), (v0.w + v1.w) + (v2.w + v3.w)};
        for (int off = EPW; off < 64; off <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += __shfl_xor(v[e], off);
        }
        if (sub == 0 && live) {
            const int tap = k / p.Cin_p, pc = k - tap * p.Cin_p;
            const int co = phys2log(prow, p.gw_out, p.gwp_out, p.Cout);
            if (co >= 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = phys2log(pc + e, p.gw_in, p.gwp_in, p.Cin);
                    if (ci >= 0) {
                        float* d = gw + ((size_t)co * p.Cin + ci) * taps + tap;
                        *d = accumulate ? *d + v[e] : v[e];
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_unpack(const float* __restrict__ slab, float* __restrict__ gw, pn2_pack_desc p, int nsplit, int accumulate) {
    wgrad_reduce_body(slab, gw, p, nsplit, accumulate, blockIdx.x, gridDim.x);
}

__global__ __launch_bounds__(256) void wgrad_reduce_multi(const pn2_reduce_job* __restrict__ jobs, const int* __restrict__ bstart, int njobs) {
    // the split-K reductions of many convs in one launch (they do not depend on each other, only on their own wgrad launch)
    const int jb = find_job(bstart, njobs, blockIdx.x);
    const pn2_reduce_job j = jobs[jb];
    wgrad_reduce_body(j.slab, j.gw, j.d, j.nsplit, j.accumulate, blockIdx.x - bstart[jb], bstart[jb + 1] - bstart[jb]);
}

inline int reduce_blocks(const pn2_pack_desc& p) {
    const size_t nthr = (size_t)p.Cout_p * p.Cin_p * p.KH * p.KW * 4;      // up to 16 sub-lanes per 4-element vector
    const size_t b = (nthr + 255) / 256;
    return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// dynamic LDS of the bf16 BatchNorm-backward epilogue for this launch (C tile + the operand tiles it needs)
template <int BM, int BN>
inline int ep2_lds_for(const pn2_conv_desc& d, const pn2_conv_ep& ep) {
    bool stat_a, y_a, acc, stat_b;
    ep2_needs(d, ep, stat_a, y_a, acc, stat_b);
    return ep2_layout<BM, BN>(stat_a, y_a, acc, stat_b, ep2_stat_c(ep)).total;
}
// table launches: the caller says what the jobs need (bits 1..4 of `ep`: statistics a, mask-from-activation a, +=, statistics b); no bits = everything
template <int BM, int BN>
inline int ep2_lds_bits(int bits) {
    if (!(bits & 15)) bits = 15;
    return ep2_layout<BM, BN>(bits & 1, bits & 2, bits & 4, bits & 8).total;
}

template <typename T, int BM, int BN, int WM, int WN, bool EP>
int launch_gemm(const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc& d, const pn2_conv_ep& ep, hipStream_t st) {
    const int M = d.N * d.OH * d.OW;
    const int grid = ((M + BM - 1) / BM) * ((d.Cout + BN - 1) / BN);
    constexpr int main_b = 2 * (BM + BN) * RS, epi_b = BM * (BN * (int)sizeof(T) + 16) + 3 * WM * BN * 4;
    constexpr bool E2 = EP && sizeof(T) == 2 && ep2_tile(BM, BN);
    constexpr int ep_b = (EP && !E2) ? ep_lds_bytes(TT<T>::VEC) : 0;
    constexpr int lds0 = (main_b > epi_b ? main_b : epi_b) > ep_b ? (main_b > epi_b ? main_b : epi_b) : ep_b;
    int lds = lds0;
    if (E2) { const int e2 = ep2_lds_for<BM, BN>(d, ep); if (e2 > lds) lds = e2; }
    if (lds > 160 * 1024) return -4;
    const bool pw = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0;
    if (lds0 > 64 * 1024 || E2) {      // opt in to more than 64 KiB of dynamic LDS once per instantiation
        static bool done = false;
        if (!done) {
            const int cap = E2 ? 160 * 1024 : lds0;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gather_gemm<T, BM, BN, WM, WN, true, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gather_gemm<T, BM, BN, WM, WN, false, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            done = true;
        }
    }
    if (pw) hipLaunchKernelGGL((conv_gather_gemm<T, BM, BN, WM, WN, true, EP>), dim3(grid), dim3(256), lds, st, (const T*)in, (const T*)wp, (T*)out, psum, psq, d, ep);
    else hipLaunchKernelGGL((conv_gather_gemm<T, BM, BN, WM, WN, false, EP>), dim3(grid), dim3(256), lds, st, (const T*)in, (const T*)wp, (T*)out, psum, psq, d, ep);
    PN2_CHECK_LAUNCH();
    return 0;
}

template <bool EP, int BM, int BN, int WM, int WN, int NS = 3>
int launch_dma(const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc& d, const pn2_conv_ep& ep, hipStream_t st) {
    const int M = d.N * d.OH * d.OW;
    const int grid = ((M + BM - 1) / BM) * ((d.Cout + BN - 1) / BN);
    constexpr int stage_b = (BM + BN) * 128, max_b = NS * stage_b, epi_b = BM * (BN * 2 + 16) + 3 * WM * BN * 4;
    // short-K convs (1-2 K-steps) only touch 1-2 ring slots: ask for less LDS so that more workgroups share a CU
    const int ksteps = (d.KH * d.KW * d.Cin_p + 63) / 64;
    const int main_b = (ksteps < NS ? ksteps : NS) * stage_b;
    int lds = main_b > epi_b ? main_b : epi_b;
    constexpr bool E2 = EP && ep2_tile(BM, BN);
    if (E2) { const int e2 = ep2_lds_for<BM, BN>(d, ep); if (e2 > lds) lds = e2; }
    else if (EP && lds < ep_lds_bytes(8)) lds = ep_lds_bytes(8);
    if (lds > 160 * 1024) return -4;
    const bool pw = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0;
    if (max_b > 64 * 1024 || epi_b > 64 * 1024 || E2) {
        static bool done = false;
        if (!done) {
            const int cap = E2 ? 160 * 1024 : (max_b > epi_b ? max_b : epi_b);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_gemm<BM, BN, WM, WN, true, NS, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_gemm<BM, BN, WM, WN, false, NS, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
            done = true;
        }
    }
    const int ksplit = (d.flags >> 16) & 15;
    const dim3 g3(grid, ksplit > 1 ? ksplit : 1);
    if (pw) hipLaunchKernelGGL((conv_dma_gemm<BM, BN, WM, WN, true, NS, EP>), g3, dim3(256), lds, st, (const bf16_t*)in, (const bf16_t*)wp, (bf16_t*)out, psum, psq, d, ep);
    else hipLaunchKernelGGL((conv_dma_gemm<BM, BN, WM, WN, false, NS, EP>), g3, dim3(256), lds, st, (const bf16_t*)in, (const bf16_t*)wp, (bf16_t*)out, psum, psq, d, ep);
    PN2_CHECK_LAUNCH();
    return 0;
}

// intra-workgroup split-K launch (conv_dma_gemm_ks): KS rings, 256 * KS threads; -4 when the rings / the exchange area / the epilogue do not fit 160 KB
template <bool EP, int BM, int BN, int WM, int WN, int NS, int KS>
int launch_dma_ks(const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc& d, const pn2_conv_ep& ep, hipStream_t st) {
    const int M = d.N * d.OH * d.OW;
    const int grid = ((M + BM - 1) / BM) * ((d.Cout + BN - 1) / BN);
    constexpr int stage_b = (BM + BN) * 128, main_b = KS * NS * stage_b, epi_b = BM * (BN * 2 + 16) + 3 * WM * BN * 4, xch_b = (KS - 1) * BM * BN * 4;
    static_assert(main_b <= 160 * 1024 && xch_b <= 160 * 1024, "tile does not fit");
    int lds = main_b > epi_b ? main_b : epi_b;
    if (xch_b > lds) lds = xch_b;
    constexpr bool E2 = EP && ep2_tile(BM, BN);
    if (E2) { const int e2 = ep2_lds_for<BM, BN>(d, ep); if (e2 > lds) lds = e2; }
    else if (EP && lds < ep_lds_bytes(8)) lds = ep_lds_bytes(8);
    if (lds > 160 * 1024 || ((d.flags >> 16) & 15) > 1) return -4;
    const bool pw = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0;
    {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_gemm_ks<BM, BN, WM, WN, true, NS, EP, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_gemm_ks<BM, BN, WM, WN, false, NS, EP, KS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            done = true;
        }
    }
    if (pw) hipLaunchKernelGGL((conv_dma_gemm_ks<BM, BN, WM, WN, true, NS, EP, KS>), dim3(grid), dim3(256 * KS), lds, st, (const bf16_t*)in, (const bf16_t*)wp, (bf16_t*)out, psum, psq, d, ep);
    else hipLaunchKernelGGL((conv_dma_gemm_ks<BM, BN, WM, WN, false, NS, EP, KS>), dim3(grid), dim3(256 * KS), lds, st, (const bf16_t*)in, (const bf16_t*)wp, (bf16_t*)out, psum, psq, d, ep);
    PN2_CHECK_LAUNCH();
    return 0;
}

// the LDS-DMA kernel addresses the activation operand with 32-bit byte offsets behind a buffer descriptor (conv_dma_body): its extent must stay below 2 GB
inline bool dma_extent_ok(const pn2_conv_desc& d) {
    return ((size_t)d.N * d.H * d.W - 1) * (size_t)d.ld_in * 2 + (size_t)d.Cin_p * 2 < 0x80000000ull;
}
// tile choice: widest N tile with the least padding, then shrink tiles until the grid can fill 256 CUs
inline void pick_tiles(int M, int cout, bool f32, int& bm, int& bn) {
    bn = pn2_conv_tile_n(cout);
    if (f32 && bn == 128) bn = 64;   // fp32 128-wide epilogue tile would exceed 64 KiB of LDS
    bm = 128;
    auto blocks = [&]() { return ((M + bm - 1) / bm) * ((cout + bn - 1) / bn); };
    if (blocks() < 384) bm = 64;
    if (blocks() < 256 && bn == 128) bn = 64;
    if (blocks() < 160 && bn == 64 && cout > 32) bn = 32;
}

template <typename T, int BMC, int WM, int WN>
int launch_wgrad(const void* dy, const void* x, float* slab, const pn2_wgrad_desc& d, int nsplit, hipStream_t st) {
    constexpr int BNK = 128;
    constexpr int lds = 2 * WGP * (BMC * (int)sizeof(T) + WG<T>::PAD + BNK * (int)sizeof(T) + WG<T>::PAD);
    dim3 grid(8 * ((nsplit + 7) / 8) * (d.Rp / BMC) * (d.Kp / BNK));
    const bool pw = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0;
    if (pw) hipLaunchKernelGGL((conv_wgrad<T, BMC, BNK, WM, WN, true>), grid, dim3(256), lds, st, (const T*)dy, (const T*)x, slab, d, nsplit);
    else hipLaunchKernelGGL((conv_wgrad<T, BMC, BNK, WM, WN, false>), grid, dim3(256), lds, st, (const T*)dy, (const T*)x, slab, d, nsplit);
    PN2_CHECK_LAUNCH();
    return 0;
}

// (kernel, BM, BN) for this desc: kern 0 register-staged, 2 LDS-DMA 3-stage, 3 LDS-DMA 2-stage.  Optional per-shape tuning code in flags bits 8..15
// (bf16 only): kernel (1 register-staged, 2 / 3 LDS-DMA) | BM (1: 64, 2: 128) << 2 | BN (1: 32, 2: 64, 3: 128) << 4
template <typename T>
void gemm_select(const pn2_conv_desc& d, int& kern, int& bm, int& bn) {
    pick_tiles(d.N * d.OH * d.OW, d.Cout, sizeof(T) == 4, bm, bn);
    const int tune = (sizeof(T) == 2 || !MMA<T>::F64ROWS) ? (d.flags >> 8) & 0xff : 0;          // (fp32fast takes the tile bits of a tuning code; its kernel is always the register-staged one)
    const int tk_ = dma_extent_ok(d) ? (tune & 3) : 1, tbm = (tune >> 2) & 3, tbn = (tune >> 4) & 3;     // > 2 GB operand: register-staged kernel (64-bit addresses)
    if (tbm) bm = tbm == 1 ? 64 : 128;
    if (tbn) bn = tbn == 1 ? 32 : (tbn == 2 ? 64 : 128);
    kern = 0;
    if (sizeof(T) == 2) kern = tk_ == 3 ? 3 : ((tk_ ? tk_ == 2 : true) ? 2 : 0);
    if (sizeof(T) == 4 && bn == 128) bn = 64;          // (fp32: 64-wide tiles at most - f64 accumulators of a 128-wide wave tile would take 128 registers; fp32fast measured slower on them: 82 -> 76 TF/s on the wide 1x1 convs, fewer workgroups per CU)
}

template <typename T, bool EP>
int gemm_dispatch(const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc& d, const pn2_conv_ep& ep, hipStream_t st) {
    int kern, bm, bn;
    gemm_select<T>(d, kern, bm, bn);
    if constexpr (sizeof(T) == 2) {
        if (((d.flags >> 8) & 0x80) && kern >= 2 && bm == 64 && bn == 64) {          // tuning-code bit 7: intra-workgroup split-K over FOUR K groups (64 x 64 tiles)
            return launch_dma_ks<EP, 64, 64, 2, 2, 2, 4>(in, wp, out, psum, psq, d, ep, st);          // (four 3-stage rings of a 64 x 64 tile would take 192 KB: 2 stages)
        }
        if (((d.flags >> 8) & 0x40) && kern >= 2 && bn >= 64) {          // tuning-code bit 6: intra-workgroup split-K over two K groups of four waves
            if (kern == 3) {
                if (bm == 128) return bn == 128 ? launch_dma_ks<EP, 128, 128, 2, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st) : launch_dma_ks<EP, 128, 64, 2, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st);
                return bn == 128 ? launch_dma_ks<EP, 64, 128, 2, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st) : launch_dma_ks<EP, 64, 64, 2, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st);
            }
            if (bm == 128) return bn == 128 ? -4 : launch_dma_ks<EP, 128, 64, 2, 2, 3, 2>(in, wp, out, psum, psq, d, ep, st);
            return bn == 128 ? launch_dma_ks<EP, 64, 128, 2, 2, 3, 2>(in, wp, out, psum, psq, d, ep, st) : launch_dma_ks<EP, 64, 64, 2, 2, 3, 2>(in, wp, out, psum, psq, d, ep, st);
        }
        if (kern == 3) {              // LDS-DMA, 2-stage ring (more workgroups per CU)
            if (bm == 128) {
                if (bn == 128) return launch_dma<EP, 128, 128, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st);
                if (bn == 64) return launch_dma<EP, 128, 64, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st);
                return launch_dma<EP, 128, 32, 4, 1, 2>(in, wp, out, psum, psq, d, ep, st);
            }
            if (bn == 128) return launch_dma<EP, 64, 128, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st);
            if (bn == 64) return launch_dma<EP, 64, 64, 2, 2, 2>(in, wp, out, psum, psq, d, ep, st);
            return launch_dma<EP, 64, 32, 4, 1, 2>(in, wp, out, psum, psq, d, ep, st);
        }
        if (kern == 2) {
            if (bm == 128) {
                if (bn == 128) return launch_dma<EP, 128, 128, 2, 2>(in, wp, out, psum, psq, d, ep, st);
                if (bn == 64) return launch_dma<EP, 128, 64, 2, 2>(in, wp, out, psum, psq, d, ep, st);
                return launch_dma<EP, 128, 32, 4, 1>(in, wp, out, psum, psq, d, ep, st);
            }
            if (bn == 128) return launch_dma<EP, 64, 128, 2, 2>(in, wp, out, psum, psq, d, ep, st);
            if (bn == 64) return launch_dma<EP, 64, 64, 2, 2>(in, wp, out, psum, psq, d, ep, st);
            return launch_dma<EP, 64, 32, 4, 1>(in, wp, out, psum, psq, d, ep, st);
        }
    }
    if (bm == 128) {
        if (bn == 128) {
            if constexpr (!MMA<T>::F64ROWS) return launch_gemm<T, 128, 128, 2, 2, EP>(in, wp, out, psum, psq, d, ep, st);
        }
        if (bn == 64) return launch_gemm<T, 128, 64, 2, 2, EP>(in, wp, out, psum, psq, d, ep, st);
        return launch_gemm<T, 128, 32, 4, 1, EP>(in, wp, out, psum, psq, d, ep, st);
    }
    if (bn == 128) {
        if constexpr (!MMA<T>::F64ROWS) return launch_gemm<T, 64, 128, 2, 2, EP>(in, wp, out, psum, psq, d, ep, st);
    }
    if (bn == 64) return launch_gemm<T, 64, 64, 2, 2, EP>(in, wp, out, psum, psq, d, ep, st);
    return launch_gemm<T, 64, 32, 4, 1, EP>(in, wp, out, psum, psq, d, ep, st);
}

// table-driven conv GEMM launches: general (non-pointwise-specialised) kernels, LDS-DMA 3-stage for bf16, register-staged for fp32
template <bool EP, int BM, int BN, int WM, int WN>
int launch_dma_tab(const pn2_conv_job* jobs, const int* bstart, int njobs, int total, int bits, hipStream_t st) {
    constexpr int stage_b = (BM + BN) * 128, max_b = 3 * stage_b, epi_b = BM * (BN * 2 + 16) + 3 * WM * BN * 4;
    int lds = max_b > epi_b ? max_b : epi_b;
    if (EP && ep2_tile(BM, BN)) { const int e2 = ep2_lds_bits<BM, BN>(bits); if (e2 > lds) lds = e2; }
    else if (EP && lds < ep_lds_bytes(8)) lds = ep_lds_bytes(8);
    if (lds > 160 * 1024) return -4;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_gemm_tab<BM, BN, WM, WN, false, 3, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, EP ? 160 * 1024 : lds);
        done = true;
    }
    hipLaunchKernelGGL((conv_dma_gemm_tab<BM, BN, WM, WN, false, 3, EP>), dim3(total), dim3(256), lds, st, jobs, bstart, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}
// table-driven launch of the intra-workgroup split-K kernel (two K groups, 3-stage rings): jobs whose shape rule asks for it share tables among themselves
template <bool EP, int BM, int BN, int WM, int WN>
int launch_dma_tab_ks2(const pn2_conv_job* jobs, const int* bstart, int njobs, int total, int bits, hipStream_t st) {
    constexpr int stage_b = (BM + BN) * 128, max_b = 2 * 3 * stage_b, epi_b = BM * (BN * 2 + 16) + 3 * WM * BN * 4, xch_b = BM * BN * 4;
    static_assert(max_b <= 160 * 1024, "tile does not fit");
    int lds = max_b > epi_b ? max_b : epi_b;
    if (xch_b > lds) lds = xch_b;
    if (EP && ep2_tile(BM, BN)) { const int e2 = ep2_lds_bits<BM, BN>(bits); if (e2 > lds) lds = e2; }
    else if (EP && lds < ep_lds_bytes(8)) lds = ep_lds_bytes(8);
    if (lds > 160 * 1024) return -4;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_dma_gemm_tab_ks2<BM, BN, WM, WN, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        done = true;
    }
    hipLaunchKernelGGL((conv_dma_gemm_tab_ks2<BM, BN, WM, WN, EP>), dim3(total), dim3(512), lds, st, jobs, bstart, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}
template <typename T, bool EP, int BM, int BN, int WM, int WN>
int launch_gather_tab_f32(const pn2_conv_job* jobs, const int* bstart, int njobs, int total, hipStream_t st) {
    constexpr int main_b = 2 * (BM + BN) * RS, epi_b = BM * (BN * 4 + 16) + 3 * WM * BN * 4, ep_b = EP ? ep_lds_bytes(4) : 0;
    constexpr int lds = (main_b > epi_b ? main_b : epi_b) > ep_b ? (main_b > epi_b ? main_b : epi_b) : ep_b;
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gather_gemm_tab<T, BM, BN, WM, WN, false, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        done = true;
    }
    hipLaunchKernelGGL((conv_gather_gemm_tab<T, BM, BN, WM, WN, false, EP>), dim3(total), dim3(256), lds, st, jobs, bstart, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}
template <bool EP>
int gemm_multi_dispatch(int dtype, int bm, int bn, int bits, const pn2_conv_job* jobs, const int* bstart, int njobs, int total, hipStream_t st) {
    if (dtype == PN2_BF16 && (bm & 0x100)) {          // tile code of pn2_conv_gemm_tile with the split-K bit: the two-K-group table kernel
        bm &= 0xff;
        if (bm == 128 && bn == 64) return launch_dma_tab_ks2<EP, 128, 64, 2, 2>(jobs, bstart, njobs, total, bits, st);
        if (bm == 64 && bn == 128) return launch_dma_tab_ks2<EP, 64, 128, 2, 2>(jobs, bstart, njobs, total, bits, st);
        if (bm == 64 && bn == 64) return launch_dma_tab_ks2<EP, 64, 64, 2, 2>(jobs, bstart, njobs, total, bits, st);
        return -2;
    }
    if (dtype == PN2_BF16) {
        if (bm == 128) {
            if (bn == 128) return launch_dma_tab<EP, 128, 128, 2, 2>(jobs, bstart, njobs, total, bits, st);
            if (bn == 64) return launch_dma_tab<EP, 128, 64, 2, 2>(jobs, bstart, njobs, total, bits, st);
            if (bn == 32) return launch_dma_tab<EP, 128, 32, 4, 1>(jobs, bstart, njobs, total, bits, st);
        } else if (bm == 64) {
            if (bn == 128) return launch_dma_tab<EP, 64, 128, 2, 2>(jobs, bstart, njobs, total, bits, st);
            if (bn == 64) return launch_dma_tab<EP, 64, 64, 2, 2>(jobs, bstart, njobs, total, bits, st);
            if (bn == 32) return launch_dma_tab<EP, 64, 32, 4, 1>(jobs, bstart, njobs, total, bits, st);
        }
        return -2;
    }
    if (dtype == PN2_F32) {
        if (bm == 128) {
            if (bn == 64) return launch_gather_tab_f32<float, EP, 128, 64, 2, 2>(jobs, bstart, njobs, total, st);
            if (bn == 32) return launch_gather_tab_f32<float, EP, 128, 32, 4, 1>(jobs, bstart, njobs, total, st);
        } else if (bm == 64) {
            if (bn == 64) return launch_gather_tab_f32<float, EP, 64, 64, 2, 2>(jobs, bstart, njobs, total, st);
            if (bn == 32) return launch_gather_tab_f32<float, EP, 64, 32, 4, 1>(jobs, bstart, njobs, total, st);
        }
        return -2;
    }
    if (dtype == PN2_F32F) {
        if (bm == 128) {
            if (bn == 128) return launch_gather_tab_f32<f32f_t, EP, 128, 128, 2, 2>(jobs, bstart, njobs, total, st);
            if (bn == 64) return launch_gather_tab_f32<f32f_t, EP, 128, 64, 2, 2>(jobs, bstart, njobs, total, st);
            if (bn == 32) return launch_gather_tab_f32<f32f_t, EP, 128, 32, 4, 1>(jobs, bstart, njobs, total, st);
        } else if (bm == 64) {
            if (bn == 128) return launch_gather_tab_f32<f32f_t, EP, 64, 128, 2, 2>(jobs, bstart, njobs, total, st);
            if (bn == 64) return launch_gather_tab_f32<f32f_t, EP, 64, 64, 2, 2>(jobs, bstart, njobs, total, st);
            if (bn == 32) return launch_gather_tab_f32<f32f_t, EP, 64, 32, 4, 1>(jobs, bstart, njobs, total, st);
        }
        return -2;
    }
    return -3;
}

template <int BMC, int WM, int WN, int BNK = 128>
int launch_wgrad_dma(const void* dy, const void* x, float* slab, const pn2_wgrad_desc& d, int nsplit, hipStream_t st) {
    constexpr int PX = wg_px(BMC), stage_b = PX * (BMC * 2 + BNK * 2), max_b = 3 * stage_b;
    const int M = d.N * d.OH * d.OW;
    const int total_steps = (M + PX - 1) / PX, spb = (total_steps + nsplit - 1) / nsplit;
    const int lds = (spb < 3 ? (spb < 1 ? 1 : spb) : 3) * stage_b;
    const int grid = 8 * ((nsplit + 7) / 8) * (d.Rp / BMC) * ((d.Kp + BNK - 1) / BNK);
    const bool pw = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0;
    if (max_b > 64 * 1024) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma<BMC, WM, WN, true, BNK>), hipFuncAttributeMaxDynamicSharedMemorySize, max_b);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma<BMC, WM, WN, false, BNK>), hipFuncAttributeMaxDynamicSharedMemorySize, max_b);
            done = true;
        }
    }
    if (pw) hipLaunchKernelGGL((conv_wgrad_dma<BMC, WM, WN, true, BNK>), dim3(grid), dim3(256), lds, st, (const bf16_t*)dy, (const bf16_t*)x, slab, d, nsplit);
    else hipLaunchKernelGGL((conv_wgrad_dma<BMC, WM, WN, false, BNK>), dim3(grid), dim3(256), lds, st, (const bf16_t*)dy, (const bf16_t*)x, slab, d, nsplit);
    PN2_CHECK_LAUNCH();
    return 0;
}

inline bool wgrad_pw(const pn2_wgrad_desc& d) { return d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad_h == 0 && d.pad_w == 0; }

// kernel instantiation a wgrad job runs on: dma * 6 + (co tile 32/64/128 -> 0/1/2) * 2 + pointwise; 12 + pointwise: the DMA kernel with 128 x 256 tiles.
// d.tune: 0 heuristic, 1 register-staged, 2 LDS-DMA (128-wide contraction tiles), 3 LDS-DMA with 128 x 256 tiles (bf16, co tile 128, Kp >= 256; else as 2)
template <typename T>
int wgrad_variant(const pn2_wgrad_desc& d) {
    const int bmc = pn2_wgrad_tile_co(d.Cout_p);
    bool dma = false, wide = false;
    if constexpr (sizeof(T) == 2) {
        dma = d.tune ? d.tune >= 2 : (wgrad_pw(d) && d.N * d.OH * d.OW >= 8192);      // default: the DMA pipeline pays off for pointwise convs with many pixels
        // the DMA kernel addresses both operands with 32-bit byte offsets (buffer descriptors): extents below 2 GB
        if ((size_t)d.N * d.OH * d.OW * d.ld_dy * 2 >= 0x80000000ull || (size_t)d.N * d.H * d.W * d.ld_x * 2 >= 0x80000000ull) dma = false;
        wide = dma && bmc == 128 && d.Kp >= 256 && d.tune == 3;
    }
    if (wide) return 12 + (wgrad_pw(d) ? 1 : 0);
    return (dma ? 6 : 0) + (bmc == 128 ? 2 : (bmc == 64 ? 1 : 0)) * 2 + (wgrad_pw(d) ? 1 : 0);
}

template <typename T>
inline int wgrad_blocks_t(const pn2_wgrad_desc& d, int nsplit) {
    const int bmc = pn2_wgrad_tile_co(d.Cout_p);
    const int bnk = wgrad_variant<T>(d) >= 12 ? 256 : 128;
    return 8 * ((nsplit + 7) / 8) * (d.Rp / bmc) * ((d.Kp + bnk - 1) / bnk);
}

template <typename T>
int wgrad_dispatch(const void* dy, const void* x, float* slab, const pn2_wgrad_desc& d, int nsplit, hipStream_t st) {
    const int bmc = pn2_wgrad_tile_co(d.Cout_p);
    if (d.Rp % bmc || d.Kp % 128) return -2;
    const int v = wgrad_variant<T>(d);
    if constexpr (sizeof(T) == 2) {
        if (v >= 12) return launch_wgrad_dma<128, 2, 2, 256>(dy, x, slab, d, nsplit, st);
        if (v >= 6) {
            if (bmc == 128) return launch_wgrad_dma<128, 2, 2>(dy, x, slab, d, nsplit, st);
            if (bmc == 64) return launch_wgrad_dma<64, 2, 2>(dy, x, slab, d, nsplit, st);
            return launch_wgrad_dma<32, 1, 4>(dy, x, slab, d, nsplit, st);
        }
    }
    if (bmc == 128) return launch_wgrad<T, 128, 2, 2>(dy, x, slab, d, nsplit, st);
    if (bmc == 64) return launch_wgrad<T, 64, 2, 2>(dy, x, slab, d, nsplit, st);
    return launch_wgrad<T, 32, 1, 4>(dy, x, slab, d, nsplit, st);
}

template <typename T, int BMC, int WM, int WN>
int launch_wgrad_tab(bool pw, const pn2_wgrad_job* jobs, const int* bstart, int njobs, int total, hipStream_t st) {
    constexpr int BNK = 128;
    constexpr int lds = 2 * WGP * (BMC * (int)sizeof(T) + WG<T>::PAD + BNK * (int)sizeof(T) + WG<T>::PAD);
    if (pw) hipLaunchKernelGGL((conv_wgrad_tab<T, BMC, BNK, WM, WN, true>), dim3(total), dim3(256), lds, st, jobs, bstart, njobs);
    else hipLaunchKernelGGL((conv_wgrad_tab<T, BMC, BNK, WM, WN, false>), dim3(total), dim3(256), lds, st, jobs, bstart, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}

template <int BMC, int WM, int WN, int BNK = 128>
int launch_wgrad_dma_tab(bool pw, const pn2_wgrad_job* jobs, const int* bstart, int njobs, int total, hipStream_t st) {
    constexpr int max_b = 3 * wg_px(BMC) * (BMC * 2 + BNK * 2);
    if (max_b > 64 * 1024) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_tab<BMC, WM, WN, true, BNK>), hipFuncAttributeMaxDynamicSharedMemorySize, max_b);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_tab<BMC, WM, WN, false, BNK>), hipFuncAttributeMaxDynamicSharedMemorySize, max_b);
            done = true;
        }
    }
    if (pw) hipLaunchKernelGGL((conv_wgrad_dma_tab<BMC, WM, WN, true, BNK>), dim3(total), dim3(256), max_b, st, jobs, bstart, njobs);
    else hipLaunchKernelGGL((conv_wgrad_dma_tab<BMC, WM, WN, false, BNK>), dim3(total), dim3(256), max_b, st, jobs, bstart, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}

template <typename T>
int wgrad_multi_dispatch(int v, const pn2_wgrad_job* jobs, const int* bstart, int njobs, int total, hipStream_t st) {
    const bool pw = v & 1;
    const int bi = (v % 6) >> 1;
    if (v == 14) {          // variants 12 + 13 in one table (conv_wgrad_dma_tab_mix)
        if constexpr (sizeof(T) == 2) {
            constexpr int max_b = 3 * wg_px(128) * (128 * 2 + 256 * 2);
            static bool done = false;
            if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_dma_tab_mix<128, 2, 2, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, max_b); done = true; }
            hipLaunchKernelGGL((conv_wgrad_dma_tab_mix<128, 2, 2, 256>), dim3(total), dim3(256), max_b, st, jobs, bstart, njobs);
            PN2_CHECK_LAUNCH();
            return 0;
        }
        return -3;
    }
    if (v >= 12) {
        if constexpr (sizeof(T) == 2) return launch_wgrad_dma_tab<128, 2, 2, 256>(pw, jobs, bstart, njobs, total, st);
        return -3;
    }
    if (v >= 6) {
        if constexpr (sizeof(T) == 2) {
            if (bi == 2) return launch_wgrad_dma_tab<128, 2, 2>(pw, jobs, bstart, njobs, total, st);
            if (bi == 1) return launch_wgrad_dma_tab<64, 2, 2>(pw, jobs, bstart, njobs, total, st);
            return launch_wgrad_dma_tab<32, 1, 4>(pw, jobs, bstart, njobs, total, st);
        }
        return -3;
    }
    if (bi == 2) return launch_wgrad_tab<T, 128, 2, 2>(pw, jobs, bstart, njobs, total, st);
    if (bi == 1) return launch_wgrad_tab<T, 64, 2, 2>(pw, jobs, bstart, njobs, total, st);
    return launch_wgrad_tab<T, 32, 1, 4>(pw, jobs, bstart, njobs, total, st);
}


// ------------------------------------------------------------------------------------------------
// Data gradient of a "patchify" conv (kernel == stride, no padding, no dilation: the spatial-reduction convs of PVTv2, pvtv2.py:70).  Every
// input pixel belongs to exactly one patch and one tap, so dX is a plain GEMM  T[(n,oy,ox)][(kh,kw,ci)] = dy[(n,oy,ox)][co] * W[co][ci][kh][kw]
// (pn2_conv_gemm, 1x1) followed by a depth-to-space copy - the generic transposed gather would walk all KH*KW taps per input pixel and find
// one of them valid (6.9 TF/s for the 8x8 stride-8 conv).
template <typename T>
__global__ __launch_bounds__(256) void pack_patch_k(const float* __restrict__ w, T* __restrict__ wp, int Cout, int Cin, int KH, int KW, int Cin_p, int Rp, int Kp) {
    const size_t total = (size_t)Rp * Kp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / Kp), k = (int)(i - (size_t)row * Kp);
        const int tap = row / Cin_p, ci = row - tap * Cin_p;
        float v = 0.f;
        if (tap < KH * KW && ci < Cin && k < Cout) v = w[((size_t)k * Cin + ci) * KH * KW + tap];
        TT<T>::st(wp + i, v);
    }
}

// dx[n][oy*S + kh][ox*S + kw][c] (+)= t[(n, oy, ox)][(kh*S + kw)*C + c] ; pixels outside the OH*S x OW*S patch area receive zero
template <typename T>
__global__ __launch_bounds__(256) void depth_to_space_k(const T* __restrict__ t, int ld_t, T* __restrict__ dx, int ld_dx, int N, int H, int W, int OH, int OW, int S, int C,
                                                        int accumulate) {
    constexpr int V = TT<T>::VEC;
    const int CV = C / V;
    const size_t total = (size_t)N * H * W * CV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cv = (int)(i % CV); const size_t pix = i / CV;
        const int ix = (int)(pix % W), iy = (int)((pix / W) % H), n = (int)(pix / ((size_t)W * H));
        const int oy = iy / S, ox = ix / S, kh = iy - oy * S, kw = ix - ox * S;
        float v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = 0.f;
        if (oy < OH && ox < OW) TT<T>::unpack(*reinterpret_cast<const uint4*>(t + (((size_t)n * OH + oy) * OW + ox) * ld_t + (size_t)(kh * S + kw) * C + cv * V), v);
        T* dst = dx + pix * ld_dx + cv * V;
        if (accumulate) {
            float o[V];
            TT<T>::unpack(*reinterpret_cast<const uint4*>(dst), o);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] += o[e];
        }
        *reinterpret_cast<uint4*>(dst) = TT<T>::pack(v);
    }
}


// ------------------------------------------------------------------------------------------------
// Data gradient of a strided conv with very few input channels (the 7x7 stride-4 patch embedding behind EMCADNet's 1 -> 3 channel stem,
// networks.py:100-102): dx[pixel][ci < 4].  The implicit GEMM would pad the 3 output columns to a 32-wide tile and walk all KH*KW taps per
// pixel although only ceil(K/S)^2 of them land on an output pixel (2.3 ms for 16x512x512).  Here a thread owns one input pixel, visits only
// its valid taps and keeps the <= 4 sums in registers; the weights sit in LDS as float4 [r][co][s] (the 4 phases of s hit distinct banks).
template <typename T>
__global__ __launch_bounds__(256) void dgrad_small_cin_k(const T* __restrict__ dy, int ld_dy, const float* __restrict__ w, T* __restrict__ dx, int ld_dx,
                                                         int N, int H, int W, int OH, int OW, int Cout, int Cin, int KH, int KW, int S, int pad, int accumulate,
                                                         int pix_per_blk) {
    constexpr int V = TT<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) float4 wl[];            // [KH][Cout][KW]
    for (int i = threadIdx.x; i < KH * Cout * KW; i += 256) {
        const int s_ = i % KW, co = (i / KW) % Cout, r = i / (KW * Cout);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        float* qq = &q.x;
        for (int ci = 0; ci < Cin; ++ci) qq[ci] = w[(((size_t)co * Cin + ci) * KH + r) * KW + s_];
        wl[i] = q;
    }
    __syncthreads();
    const size_t M = (size_t)N * H * W;
    const size_t p0 = (size_t)blockIdx.x * pix_per_blk;
    for (size_t p = p0 + threadIdx.x; p < p0 + pix_per_blk && p < M; p += 256) {
        const int ix = (int)(p % W), iy = (int)((p / W) % H), n = (int)(p / ((size_t)W * H));
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int r = (iy + pad) % S; r < KH; r += S) {
            const int oy = (iy + pad - r) / S;
            if (iy + pad - r < 0 || oy >= OH) continue;
            for (int s_ = (ix + pad) % S; s_ < KW; s_ += S) {
                const int ox = (ix + pad - s_) / S;
                if (ix + pad - s_ < 0 || ox >= OW) continue;
                const T* dp = dy + (((size_t)n * OH + oy) * OW + ox) * ld_dy;
                const float4* wp = wl + (size_t)r * Cout * KW + s_;
                for (int c0 = 0; c0 < Cout; c0 += V) {
                    float d[V];
                    TT<T>::unpack(*reinterpret_cast<const uint4*>(dp + c0), d);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const float4 q = wp[(size_t)(c0 + e) * KW];
                        a0 += d[e] * q.x; a1 += d[e] * q.y; a2 += d[e] * q.z; a3 += d[e] * q.w;
                    }
                }
            }
        }
        T* dst = dx + p * ld_dx;
        float o[V];
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = 0.f;
        o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
        if (accumulate) {
            float old[V];
            TT<T>::unpack(*reinterpret_cast<const uint4*>(dst), old);
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] += old[e];
        }
        *reinterpret_cast<uint4*>(dst) = TT<T>::pack(o);                   // the first VEC physical channels (Cin <= 4 <= VEC); pads stay zero
    }
}


// Split-K epilogue: out[m][c] (+)= sum_s ws[s][m][c] (+ bias[c]) in the conv's storage dtype, with the BatchNorm partial rows of a 64-row
// block (sums of the fp32 totals, as the GEMM epilogue takes them).  One workgroup = 64 rows; a thread owns 4 columns (16-byte loads) and
// every R-th row, the R row lanes meet in LDS in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_k(const float* __restrict__ ws, int S, int M, int C, T* __restrict__ out, int ld_out, const float* __restrict__ bias,
                                                       float* __restrict__ psum, float* __restrict__ psq, int accumulate, int LANES) {
    // grid = (64-row blocks, groups of LANES column vectors): a few-row tensor (the 11 x 11 maps this is used for have 61 row blocks) still fills the chip
    __shared__ float sh[2][256][4];
    const int CV = C >> 2, R = 256 / LANES, cl = threadIdx.x % LANES, rl = threadIdx.x / LANES;
    const int m0 = blockIdx.x * 64;
    const int cv = blockIdx.y * LANES + cl, c = cv * 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (cv < CV) {
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        if (bias) { b[0] = bias[c]; b[1] = bias[c + 1]; b[2] = bias[c + 2]; b[3] = bias[c + 3]; }
        for (int m = m0 + rl; m < m0 + 64 && m < M; m += R) {
            float4 v = *reinterpret_cast<const float4*>(ws + (size_t)m * C + c);
            for (int s = 1; s < S; ++s) {
                const float4 u = *reinterpret_cast<const float4*>(ws + ((size_t)s * M + m) * C + c);
                v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
            }
            const float vv[4] = {v.x, v.y, v.z, v.w};
            T* dst = out + (size_t)m * ld_out + c;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // bf16: the moments of the value as it is STORED (rounded), the one definition of the batch statistics for every bf16 conv kernel
                float sv = vv[e];
                if constexpr (sizeof(T) == 2) sv = bf2f(f2bf(vv[e]));
                s1[e] += sv; s2[e] += sv * sv;
                o[e] = vv[e] + b[e];
            }
            if constexpr (sizeof(T) == 2) {          // one 8-byte access per direction (ld_out and c are multiples of 4)
                if (accumulate) {
                    const uint2 p = *reinterpret_cast<const uint2*>(dst);
                    o[0] += bf2f((bf16_t)(p.x & 0xffffu)); o[1] += bf2f((bf16_t)(p.x >> 16)); o[2] += bf2f((bf16_t)(p.y & 0xffffu)); o[3] += bf2f((bf16_t)(p.y >> 16));
                }
                uint2 q;
                q.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16); q.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
                *reinterpret_cast<uint2*>(dst) = q;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) TT<T>::st(dst + e, accumulate ? o[e] + TT<T>::ld(dst + e) : o[e]);
            }
        }
    }
    if (psum) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { sh[0][threadIdx.x][e] = s1[e]; sh[1][threadIdx.x][e] = s2[e]; }
        __syncthreads();
        if (rl == 0 && cv < CV) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = 0.f, q = 0.f;
                for (int r = 0; r < R; ++r) { a += sh[0][r * LANES + cl][e]; q += sh[1][r * LANES + cl][e]; }
                psum[(size_t)blockIdx.x * C + c + e] = a; psq[(size_t)blockIdx.x * C + c + e] = q;
            }
        }
    }
}

}  // namespace

static int conv_gemm_impl(int dtype, const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc* d, const pn2_conv_ep& ep, void* stream) {
    if (!in || !wp || !out || !d) return -1;
    if (d->Cin_p % 8 || d->ld_in % 8 || d->Kp % 128 || (d->stride != 1 && d->stride != 2 && d->stride != 4 && d->stride != 8)) return -2;
    if ((d->flags & PN2_CONV_STATS) && (!psum || !psq)) return -1;
    if ((d->flags & PN2_CONV_BIAS) && (!psum || (d->flags & PN2_CONV_STATS))) return -1;
    if (((d->flags >> 16) & 15) > 1 && (dtype != PN2_BF16 || !psum || (d->flags & (PN2_CONV_STATS | PN2_CONV_BIAS | PN2_CONV_ACCUM)) || ((d->flags >> 8) & 3) < 2)) return -2;
    if (d->flags & PN2_CONV_AFFINE) return -2;                 // (pn2_conv_gemm_affine owns that flag)
    const bool gated = d->flags & PN2_CONV_ROWGATE;
    const int vec_ = dtype == PN2_BF16 ? 8 : 4;
    if (gated && (!ep.a.par || (ep.a.mode | ep.b.mode) != 0 || ep.b.out || (d->flags & PN2_CONV_BIAS) || ((d->flags >> 16) & 15) > 1 || d->Cout % vec_ || d->ld_out % vec_)) return -2;
    const bool use_ep = (ep.a.mode | ep.b.mode) != 0 || ep.b.out != nullptr || gated;      // the gate lives in the epilogue-statistics instantiations
    if (dtype == PN2_BF16) return use_ep ? gemm_dispatch<bf16_t, true>(in, wp, out, psum, psq, *d, ep, (hipStream_t)stream) : gemm_dispatch<bf16_t, false>(in, wp, out, psum, psq, *d, ep, (hipStream_t)stream);
    if (dtype == PN2_F32) return use_ep ? gemm_dispatch<float, true>(in, wp, out, psum, psq, *d, ep, (hipStream_t)stream) : gemm_dispatch<float, false>(in, wp, out, psum, psq, *d, ep, (hipStream_t)stream);
    if (dtype == PN2_F32F) return use_ep ? gemm_dispatch<f32f_t, true>(in, wp, out, psum, psq, *d, ep, (hipStream_t)stream) : gemm_dispatch<f32f_t, false>(in, wp, out, psum, psq, *d, ep, (hipStream_t)stream);
    return -3;
}

static int bnb_check(const pn2_bnb_target& t, int vec, bool is_b) {
    if (is_b && !t.out) return t.mode ? -1 : 0;
    if (is_b && t.ld_out % vec) return -2;
    if (!(t.mode & PN2_BNB_STATS)) return 0;
    if (!t.raw || !t.par || !t.p1 || !t.p2 || t.ldp < 1) return -1;
    if (t.ld_raw % vec || t.ps < 1) return -2;
    if ((t.mode & PN2_BNB_MASK_Y) && (!t.y || t.ld_y % vec)) return -1;
    if (t.split > 0 && (t.split % vec || (t.par2 && !t.raw2))) return -2;
    return 0;
}

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

int pn2_conv_tile_m(int m, int cout, int dtype) { int bm, bn; pick_tiles(m, cout, dtype != PN2_BF16, bm, bn); return bm; }

int pn2_conv_stat_blocks(int m, int cout, int dtype) { const int bm = pn2_conv_tile_m(m, cout, dtype); return (m + bm - 1) / bm; }

int pn2_conv_gemm(int dtype, const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc* d, void* stream) {
    pn2_conv_ep ep;
    memset(&ep, 0, sizeof(ep));
    return conv_gemm_impl(dtype, in, wp, out, psum, psq, d, ep, stream);
}

int pn2_conv_gemm_affine(int dtype, const void* in, const void* wp, void* out, const float* scale, const float* shift, const void* res, int ld_res, const pn2_conv_desc* d,
                         void* stream) {
    if (!d || !scale || !shift) return -1;
    if (!(d->flags & PN2_CONV_AFFINE) || (d->flags & (PN2_CONV_STATS | PN2_CONV_BIAS | PN2_CONV_ACCUM | PN2_CONV_ROWGATE)) || ((d->flags >> 16) & 15) > 1) return -2;
    const int vec_ = dtype == PN2_BF16 ? 8 : 4;
    if (res && (d->Cout % vec_ || d->ld_out % vec_ || ld_res % vec_)) return -2;
    pn2_conv_ep ep;
    memset(&ep, 0, sizeof(ep));
    ep.a.y = res; ep.a.ld_y = ld_res;
    if (!in || !wp || !out) return -1;
    if (d->Cin_p % 8 || d->ld_in % 8 || d->Kp % 128 || (d->stride != 1 && d->stride != 2 && d->stride != 4 && d->stride != 8)) return -2;
    if (dtype == PN2_BF16) return gemm_dispatch<bf16_t, false>(in, wp, out, const_cast<float*>(scale), const_cast<float*>(shift), *d, ep, (hipStream_t)stream);
    if (dtype == PN2_F32) return gemm_dispatch<float, false>(in, wp, out, const_cast<float*>(scale), const_cast<float*>(shift), *d, ep, (hipStream_t)stream);
    if (dtype == PN2_F32F) return gemm_dispatch<f32f_t, false>(in, wp, out, const_cast<float*>(scale), const_cast<float*>(shift), *d, ep, (hipStream_t)stream);
    return -3;
}

int pn2_conv_gemm_gated(int dtype, const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc* d, const float* gate, void* stream) {
    if (!d || !gate) return -1;
    pn2_conv_desc dg = *d;
    dg.flags |= PN2_CONV_ROWGATE;
    pn2_conv_ep ep;
    memset(&ep, 0, sizeof(ep));
    ep.a.par = gate;                    // no BatchNorm-backward target (mode 0): the field carries the gate logits
    return conv_gemm_impl(dtype, in, wp, out, psum, psq, &dg, ep, stream);
}

int pn2_conv_gemm_ep(int dtype, const void* in, const void* wp, void* out, const pn2_conv_desc* d, const pn2_conv_ep* ep, void* stream) {
    if (!d || !ep) return -1;
    const int vec = dtype == PN2_BF16 ? 8 : 4;
    if ((d->flags & (PN2_CONV_STATS | PN2_CONV_BIAS)) || ((d->flags >> 16) & 15) > 1) return -2;
    if (d->Cout % vec || d->ld_out % vec) return -2;                      // the statistics live in the 16-byte store path
    if (ep->c.mode) {          // second BatchNorm behind target a's masked gradient: LDS-DMA / matrix-core epilogue form only (tiles of <= 4096 elements, bf16, single launches)
        if (dtype != PN2_BF16 || ep->c.mode != PN2_BNB_STATS || !(ep->a.mode & PN2_BNB_STATS) || ep->b.out || ep->c.split) return -2;
        if (!ep->c.raw || !ep->c.par || !ep->c.p1 || !ep->c.p2 || ep->c.ldp < 1 || ep->c.ld_raw % vec || ep->c.ps < 1) return -1;
        int kern_, bm_, bn_;
        gemm_select<bf16_t>(*d, kern_, bm_, bn_);
        if (!ep2_tile(bm_, bn_)) return -2;
    }
    if (ep->pool && (!(d->flags & PN2_CONV_ACCUM) || !(ep->a.mode & PN2_BNB_STATS) || ep->b.out || (d->OH & 1) || (d->OW & 1) || ep->ld_pool % vec || ep->ld_pool < d->Cout)) return -2;
    int rc = bnb_check(ep->a, vec, false);
    if (rc) return rc;
    rc = bnb_check(ep->b, vec, true);
    if (rc) return rc;
    return conv_gemm_impl(dtype, in, wp, out, nullptr, nullptr, d, *ep, stream);
}

/* tile (bm << 8 | bn) pn2_conv_gemm would run this desc on (its tuning bits included); < 0: the launch cannot join a table (split-K, alignment) */
int pn2_conv_gemm_tile(int dtype, const pn2_conv_desc* d) {
    if (!d) return -1;
    if (d->Cin_p % 8 || d->ld_in % 8 || d->Kp % 128 || (d->stride != 1 && d->stride != 2 && d->stride != 4 && d->stride != 8)) return -2;
    if (((d->flags >> 16) & 15) > 1) return -2;
    const int ksb = (d->flags >> 8) & 0xC0;          // intra-workgroup split-K (another fp32 summation order): such jobs only share tables among themselves
    if (ksb & 0x80) return -2;
    int kern, bm, bn;
    if (dtype == PN2_BF16) { gemm_select<bf16_t>(*d, kern, bm, bn); if (!dma_extent_ok(*d)) return -2; }      // (a register-staged choice joins the table on the LDS-DMA kernel: same bits)
    else if (dtype == PN2_F32) gemm_select<float>(*d, kern, bm, bn);
    else if (dtype == PN2_F32F) gemm_select<f32f_t>(*d, kern, bm, bn);
    else return -3;
    if (ksb) {
        if (dtype != PN2_BF16 || bn < 64 || (bm == 128 && bn == 128)) return -2;
        return ((bm | 0x100) << 8) | bn;          // bm carries the split-K bit through pn2_conv_gemm_job_blocks / pn2_conv_gemm_multi
    }
    return (bm << 8) | bn;
}

int pn2_conv_gemm_job_blocks(int dtype, const pn2_conv_job* j, int bm, int bn) {
    bm &= 0xff;          // (bit 8: the split-K table kernel, same tile)
    if (!j || !j->in || !j->wp || !j->out || bm < 1 || bn < 1) return -1;
    const pn2_conv_desc& d = j->d;
    if ((d.flags & PN2_CONV_STATS) && (!j->psum || !j->psq)) return -1;
    if ((d.flags & PN2_CONV_BIAS) && !j->psum) return -1;
    if (d.flags & PN2_CONV_AFFINE) {
        const int v_ = dtype == PN2_BF16 ? 8 : 4;
        if (!j->psum || !j->psq || (d.flags & (PN2_CONV_STATS | PN2_CONV_BIAS | PN2_CONV_ACCUM | PN2_CONV_ROWGATE)) || j->ep.a.mode || j->ep.b.mode || j->ep.b.out) return -1;
        if (j->ep.a.y && (d.Cout % v_ || d.ld_out % v_ || j->ep.a.ld_y % v_)) return -2;
    }
    const bool use_ep = (j->ep.a.mode | j->ep.b.mode) != 0 || j->ep.b.out != nullptr;
    const int vec = dtype == PN2_BF16 ? 8 : 4;
    if (use_ep && (d.Cout % vec || d.ld_out % vec || (d.flags & (PN2_CONV_STATS | PN2_CONV_BIAS)))) return -2;
    return ((d.N * d.OH * d.OW + bm - 1) / bm) * ((d.Cout + bn - 1) / bn);
}

int pn2_conv_gemm_multi(int dtype, int bm, int bn, int ep, const pn2_conv_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    return (ep & 1) ? gemm_multi_dispatch<true>(dtype, bm, bn, ep >> 1, jobs_dev, block_start_dev, njobs, total_blocks, (hipStream_t)stream)
                    : gemm_multi_dispatch<false>(dtype, bm, bn, 0, jobs_dev, block_start_dev, njobs, total_blocks, (hipStream_t)stream);
}

int pn2_conv_wgrad(int dtype, const void* dy, const void* x, float* slab, const pn2_wgrad_desc* d, int nsplit, void* stream) {
    if (!dy || !x || !slab || !d || nsplit < 1) return -1;
    if (d->Cin_p % 8 || d->ld_x % 8 || d->ld_dy % 8 || d->Cout_p % 8) return -2;
    if (dtype != PN2_BF16 && d->tune >= 2) return -2;           // the LDS-DMA kernels (and pn2_conv_wgrad_blocks' tile for them) are bf16 only
    if (dtype == PN2_BF16) return wgrad_dispatch<bf16_t>(dy, x, slab, *d, nsplit, (hipStream_t)stream);
    if (dtype == PN2_F32) return wgrad_dispatch<float>(dy, x, slab, *d, nsplit, (hipStream_t)stream);
    if (dtype == PN2_F32F) return wgrad_dispatch<f32f_t>(dy, x, slab, *d, nsplit, (hipStream_t)stream);
    return -3;
}

int pn2_conv_wgrad_variant(int dtype, const pn2_wgrad_desc* d) {
    if (!d) return -1;
    return dtype == PN2_BF16 ? wgrad_variant<bf16_t>(*d) : ((dtype == PN2_F32 || dtype == PN2_F32F) ? wgrad_variant<float>(*d) : -3);
}

int pn2_conv_wgrad_blocks(const pn2_wgrad_desc* d, int nsplit) {
    if (!d || nsplit < 1) return -1;
    const int bmc = pn2_wgrad_tile_co(d->Cout_p);
    if (d->Rp % bmc || d->Kp % 128) return -2;
    return wgrad_blocks_t<bf16_t>(*d, nsplit);      // (d->tune 2 / 3 are bf16 kernels: pn2_conv_wgrad rejects them for fp32, so an fp32 job never has the wide tile)
}

int pn2_conv_wgrad_multi(int dtype, int variant, const pn2_wgrad_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1 || variant < 0 || variant >= 15) return -1;
    if (dtype == PN2_BF16) return wgrad_multi_dispatch<bf16_t>(variant, jobs_dev, block_start_dev, njobs, total_blocks, (hipStream_t)stream);
    if (dtype == PN2_F32) return wgrad_multi_dispatch<float>(variant, jobs_dev, block_start_dev, njobs, total_blocks, (hipStream_t)stream);
    if (dtype == PN2_F32F) return wgrad_multi_dispatch<f32f_t>(variant, jobs_dev, block_start_dev, njobs, total_blocks, (hipStream_t)stream);
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

int pn2_pack_blocks(const pn2_pack_desc* p) { return (p && p->KH * p->KW <= 225) ? pack_blocks(*p) : -1; }

int pn2_pack_weights_multi(int dtype, const pn2_pack_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    if (dtype == PN2_BF16) hipLaunchKernelGGL(pack_weight_multi<bf16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else if (dtype == PN2_F32) hipLaunchKernelGGL(pack_weight_multi<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_wgrad_reduce(const float* slab, float* gw, const pn2_pack_desc* p, int nsplit, int accumulate, void* stream) {
    if (!slab || !gw || !p) return -1;
    const int grid = reduce_blocks(*p);
    hipLaunchKernelGGL(wgrad_reduce_unpack, dim3(grid), dim3(256), 0, (hipStream_t)stream, slab, gw, *p, nsplit, accumulate);
    PN2_CHECK_LAUNCH();
    return 0;
}

int pn2_wgrad_reduce_blocks(const pn2_pack_desc* p) { return p ? reduce_blocks(*p) : -1; }

int pn2_wgrad_reduce_multi(const pn2_reduce_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream) {
    if (!jobs_dev || !block_start_dev || njobs < 1 || total_blocks < 1) return -1;
    hipLaunchKernelGGL(wgrad_reduce_multi, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, block_start_dev, njobs);
    PN2_CHECK_LAUNCH();
    return 0;
}


/* panel of the patchify data gradient: wp[(tap*Cin_p + ci)][co] = w[co][ci][tap] (zero pads), [Rp][Kp] in the compute dtype */
int pn2_pack_patch_weight(int dtype, const float* w_oihw, void* wp, int Cout, int Cin, int KH, int KW, int Cin_p, int Rp, int Kp, void* stream) {
    if (!w_oihw || !wp || Cin_p < Cin || Rp < KH * KW * Cin_p || Kp < Cout) return -1;
    const size_t total = (size_t)Rp * Kp;
    const unsigned grid = (unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (dtype == PN2_BF16) hipLaunchKernelGGL(pack_patch_k<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (bf16_t*)wp, Cout, Cin, KH, KW, Cin_p, Rp, Kp);
    else if (dtype == PN2_F32) hipLaunchKernelGGL(pack_patch_k<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, w_oihw, (float*)wp, Cout, Cin, KH, KW, Cin_p, Rp, Kp);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

/* dx[n][oy*S+kh][ox*S+kw][c] (+)= t[(n,oy,ox)][(kh*S+kw)*C + c]: scatters the GEMM result of the patchify data gradient back to NHWC */
int pn2_depth_to_space(int dtype, const void* t, int ld_t, void* dx, int ld_dx, int N, int H, int W, int OH, int OW, int S, int C, int accumulate, void* stream) {
    if (!t || !dx || N < 1 || S < 1) return -1;
    const int V = dtype == PN2_F32 ? 4 : 8;
    if (C % V || ld_t % V || ld_dx % V || OH * S > H || OW * S > W) return -2;
    const size_t total = (size_t)N * H * W * (C / V);
    const unsigned grid = (unsigned)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    if (dtype == PN2_BF16) hipLaunchKernelGGL(depth_to_space_k<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)t, ld_t, (bf16_t*)dx, ld_dx, N, H, W, OH, OW, S, C, accumulate);
    else if (dtype == PN2_F32) hipLaunchKernelGGL(depth_to_space_k<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)t, ld_t, (float*)dx, ld_dx, N, H, W, OH, OW, S, C, accumulate);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}


/* dx[n][iy][ix][ci] (+)= sum over the taps (r, s) that land on an output pixel and over co of dy[n][oy][ox][co] * w[co][ci][r][s]   (Cin <= 4,
 * dilation 1): data gradient of a strided few-channel conv without the zero taps / padded columns of the implicit GEMM.  w: fp32 OIHW master.
 * dx rows are ld_dx wide with at least 4 (fp32) / 8 (bf16) physical channels; channels Cin.. of that first vector are written as zero. */
int pn2_conv_dgrad_small_cin(int dtype, const void* dy, int ld_dy, const float* w_oihw, void* dx, int ld_dx, int N, int H, int W, int OH, int OW,
                             int Cout, int Cin, int KH, int KW, int stride, int pad, int accumulate, void* stream) {
    if (!dy || !w_oihw || !dx || N < 1 || stride < 1) return -1;
    const int V = dtype == PN2_F32 ? 4 : 8;
    const size_t lds = (size_t)KH * KW * Cout * sizeof(float4);
    if (Cin < 1 || Cin > 4 || Cout % V || ld_dy % V || ld_dx % V || ld_dx < V || lds > 64 * 1024) return -2;
    const size_t M = (size_t)N * H * W;
    const int pix = 1024;
    const unsigned grid = (unsigned)((M + pix - 1) / pix);
    if (dtype == PN2_BF16) hipLaunchKernelGGL(dgrad_small_cin_k<bf16_t>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const bf16_t*)dy, ld_dy, w_oihw, (bf16_t*)dx, ld_dx,
                                              N, H, W, OH, OW, Cout, Cin, KH, KW, stride, pad, accumulate, pix);
    else if (dtype == PN2_F32) hipLaunchKernelGGL(dgrad_small_cin_k<float>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const float*)dy, ld_dy, w_oihw, (float*)dx, ld_dx,
                                                  N, H, W, OH, OW, Cout, Cin, KH, KW, stride, pad, accumulate, pix);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}


/* finishes a PN2_CONV_SPLITK launch: out (+)= sum of the ksplit fp32 partial tiles in ws (+ bias); psum / psq (optional): BatchNorm partial rows
 * [ceil(M / 64)][Cout] */
int pn2_conv_splitk_reduce(int dtype, const float* ws, int ksplit, int M, int Cout, void* out, int ld_out, const float* bias, float* psum, float* psq, int accumulate,
                           void* stream) {
    if (!ws || !out || ksplit < 1 || M < 1 || Cout < 1 || ((psum == nullptr) != (psq == nullptr))) return -1;
    if (Cout % 4) return -2;
    int lanes = 1; while (lanes < Cout / 4 && lanes < 16) lanes <<= 1;          // <= 16 column lanes (64 channels) per workgroup -> >= 16 row lanes
    if (dtype == PN2_BF16 && ld_out % 4) return -2;
    const dim3 grid((M + 63) / 64, (Cout / 4 + lanes - 1) / lanes);
    if (dtype == PN2_BF16) hipLaunchKernelGGL(splitk_reduce_k<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, ws, ksplit, M, Cout, (bf16_t*)out, ld_out, bias, psum, psq, accumulate, lanes);
    else if (dtype == PN2_F32) hipLaunchKernelGGL(splitk_reduce_k<float>, grid, dim3(256), 0, (hipStream_t)stream, ws, ksplit, M, Cout, (float*)out, ld_out, bias, psum, psq, accumulate, lanes);
    else return -3;
    PN2_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"

#ifdef PN2_STAMP
extern "C" int pn2_debug_stamps(unsigned long long* dst_host, int nblk) {
    if (nblk > PN2_STAMP_BLOCKS) nblk = PN2_STAMP_BLOCKS;
    return (int)hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(pn2_stamp_buf), (size_t)nblk * PN2_STAMP_SLOTS * sizeof(unsigned long long));
}
#endif
