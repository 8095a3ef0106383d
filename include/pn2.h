/* pn2.h — C ABI of libpn2_hip.so: the MI355X (gfx950) PraNet-V2 hot-path kernels.
 *
 * The reference (ai4colonoscopy/PraNet-V2) has no native code and no FFI: its operator API is the
 * nn.Module surface of binary_seg/lib and the torch ops those modules dispatch.  Each entry point
 * below replaces the torch op(s) cited next to it (file:line under /root/reference/binary_seg);
 * pranet-v2_amd/pn2/capi.py is the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates; the library never frees);
 *   - activations are NHWC views: base pointer at the view's first channel, `ld` = elements between pixels;
 *   - dtype: PN2_F32 (fp32 parity path: fp32 storage, conv products and sums in double on v_mfma_f64_16x16x4_f64, one rounding per output)
 *     or PN2_BF16 (bf16 storage, v_mfma_f32_16x16x32_bf16, f32 accumulate);
 *   - every call is asynchronous on `stream` (a hipStream_t); returns 0 on success, <0 for argument
 *     errors (-1 null pointer, -2 unsupported geometry/alignment, -3 unknown dtype), >0 = hipError_t;
 *   - no global state; safe to call from any thread / any stream; graph-capture safe (no sync, no malloc).
 */
#ifndef PN2_H
#define PN2_H
#ifdef __cplusplus
extern "C" {
#endif

#define PN2_F32 0
#define PN2_BF16 1
#define PN2_CONV_STATS 1   /* emit per-channel sum / sum-of-squares partials (fused BN batch statistics) */
#define PN2_CONV_ACCUM 2   /* out += result (gradient accumulation) */
#define PN2_CONV_SPLITK(n) ((n) << 16)   /* bf16 LDS-DMA kernels only (tuning code kernel 2 / 3): n = 2..15 workgroups share the K loop of a tile and
                                            write fp32 partial tiles to `psum` = workspace [n][M][Cout]; finish with pn2_conv_splitk_reduce.  For convs with
                                            few output rows and a long contraction (5x5 on 11x11 maps).  Excludes STATS / BIAS / ACCUM (the reduce does those). */
#define PN2_CONV_ROWGATE 8 /* set by pn2_conv_gemm_gated: accumulator rows are scaled by 1 - sigmoid(gate[m]) before statistics / store */
#define PN2_CONV_AFFINE 16 /* pn2_conv_gemm_affine: out = act(acc * scale[c] + shift[c] (+ residual)) - eval-mode BatchNorm (+ ReLU / ReLU6) (+ residual add) folded into the GEMM epilogue */
#define PN2_CONV_RELU 32   /* ... with ReLU */
#define PN2_CONV_RELU6 64  /* ... with ReLU6 */
#define PN2_CONV_BIAS 4    /* `psum` is a [Cout] fp32 bias (physical columns) added in the epilogue; excludes PN2_CONV_STATS */

/* ---------------------------------------------------------------------------------------------- conv
 * F.conv2d / nn.Conv2d forward and its autograd backward:
 *   lib/Res2Net_v1b.py:32,44,49,102-108,133 ; lib/pranet.py:34-36 (BasicConv2d), :52-73 (RFB), :94-104 (aggregation),
 *   :303-325 (DSRA stacks) ; backward via MyTrain_med.py:84 (loss.backward()).                      */
typedef struct pn2_conv_desc {
    int N, H, W;            /* gathered tensor (x for forward, dy for dgrad) */
    int OH, OW;             /* produced tensor (y for forward, dx for dgrad) */
    int Cin_p, ld_in;       /* gathered physical channels (multiple of 8) and its pixel stride */
    int Cout, ld_out;       /* produced channels to store and pixel stride */
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;   /* of the FORWARD convolution */
    int transposed;         /* 0 forward gather, 1 dgrad gather */
    int Kp;                 /* packed-weight row length, multiple of 128 */
    int flags;              /* PN2_CONV_* ; bits 8..15 optional tuning code (bf16): kernel | BM<<2 | BN<<4, see pn2_conv_tile_m */
} pn2_conv_desc;

typedef struct pn2_wgrad_desc {
    int N, H, W, OH, OW;    /* x spatial, dy spatial */
    int Cin_p, ld_x;        /* x physical channels / stride */
    int Cout_p, ld_dy;      /* dy physical channels / stride */
    int KH, KW, stride, pad_h, pad_w, dil_h, dil_w;
    int Rp, Kp;             /* slab rows (multiple of the co tile, normally 128) and row length (multiple of 128) */
    int tune;               /* 0 heuristic, 1 register-staged kernel, 2 LDS-DMA kernel (bf16), 3 LDS-DMA kernel with 128 x 256 tiles (bf16; co tile 128 and Kp >= 256, else as 2) */
} pn2_wgrad_desc;

typedef struct pn2_pack_desc {
    int Cout, Cin, KH, KW;              /* logical OIHW shape */
    int Cout_p, gw_out, gwp_out;        /* physical output channels; groups of gw logical stored in gwp slots */
    int Cin_p, gw_in, gwp_in;
    int Rp, Kp;                         /* packed panel rows / row length */
    int transposed;                     /* 0: [co][tap*Cin_p+ci] (forward, wgrad slabs) ; 1: [ci][tap*Cout_p+co] (dgrad) */
    int ld, koff;                       /* destination row stride (0 = Kp) and column offset: lets several convs that share an
                                           input write side by side into ONE panel (fused 1x1 reducers of the RFB / RA stages) */
} pn2_pack_desc;

int pn2_conv_tile_n(int cout);          /* N tile the forward/dgrad kernel will pick for `cout` */
int pn2_wgrad_tile_co(int cout_p);      /* co tile of the wgrad kernel */
int pn2_conv_tile_m(int m, int cout, int dtype);        /* M tile (128 or 64) chosen for m output pixels x cout channels */
int pn2_conv_stat_blocks(int m, int cout, int dtype);   /* rows of the psum/psq partial buffers = ceil(m / tile_m) */
/* bf16: the LDS-DMA kernels (tuning code kernel 2 / 3, and the default choice) address `in` through a buffer descriptor with 32-bit byte offsets - padding taps,
 * rows past M and channel chunks past Cin_p are zero-filled by the hardware's out-of-range rule.  A gathered tensor whose extent from `in` reaches 2 GB is served by
 * the register-staged kernel instead (64-bit addresses; same results bit for bit, tests/test_gpu_convkernels.py). */
int pn2_conv_gemm(int dtype, const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc* d, void* stream);
/* pn2_conv_gemm with the BatchNorm-BACKWARD statistics taken in the GEMM epilogue (autograd of nn.BatchNorm2d + ReLU behind a conv,
 * Res2Net_v1b.py:60-63,70-72,84-89 ; pranet.py:40-43).  A dgrad GEMM whose result completes the gradient dy of a BatchNorm output y = act(BN(raw))
 * has every dy tile in registers: the epilogue loads the matching `raw` (and, for BN + residual + ReLU, the stored y) tile, forms
 * dz = dy * [y > 0] and leaves the per-channel partial sums  p1[tile][c] = sum dz,  p2[tile][c] = sum dz * (raw - mean) * invstd  of its rows -
 * the separate pn2_bn_bwd_reduce pass (three full-tensor reads) disappears.  The stored result stays UNMASKED (pn2_bn_bwd_apply masks) unless
 * PN2_BNB_STORE_MASKED is set.
 * Target `a` describes the GEMM's own destination `out`; target `b` (optional, b.out != NULL) receives a second copy of the plain result
 * (never accumulated) with its own statistics: the gradient of Bottle2neck's  sp + spx[i]  (Res2Net_v1b.py:66-68) flows to both operands,
 * which sit behind two different BatchNorms.  16-byte aligned rows only; excludes PN2_CONV_STATS / PN2_CONV_BIAS / split-K.               */
#define PN2_BNB_STATS 1      /* emit p1 / p2 */
#define PN2_BNB_MASK_RAW 2   /* ReLU mask recomputed as fmaf(raw, scale, shift) > 0 (bit-identical to the forward's affine pass) */
#define PN2_BNB_MASK_Y 4     /* ReLU mask from the stored activation y > 0 (BN + residual + ReLU) */
#define PN2_BNB_STORE_MASKED 8   /* store dz (the masked gradient) instead of the plain one: it IS the gradient of the residual branch, so the caller
                                    can alias that buffer and pn2_bn_bwd_apply neither re-reads y nor writes dres */
/* V1 reverse attention fused into the 1x1 conv behind it (PraNet_Res2Net.py:153-155,166-168,177-179:  x = -1*sigmoid(crop)+1 ; x = x.expand(-1, C, -1, -1).mul(x_l) ;
 * x = ra*_conv1(x)).  The gate is one number per pixel and ra*_conv1 is 1x1, so conv(gate * x_l) = gate * conv(x_l): out[m][:] = (1 - sigmoid(gate[m])) * conv(in)[m][:],
 * applied to the accumulator rows in the GEMM epilogue; PN2_CONV_STATS statistics are those of the gated result.  The gated copy of x_l (512..2048 channels)
 * is never materialised.  Not with PN2_CONV_BIAS / split-K.  Backward: pn2_ra_gate_post_bwd on (out, d out), then plain dgrad / wgrad.                         */
int pn2_conv_gemm_gated(int dtype, const void* in, const void* wp, void* out, float* psum, float* psq, const pn2_conv_desc* d, const float* gate, void* stream);
/* Eval-mode conv + BatchNorm (+ ReLU / ReLU6) (+ residual) in ONE launch (MyTest_med.py:98-104, the in-training evaluation MyTrain_med.py:163-164; BasicConv2d
 * pranet.py:40-43, Bottle2neck Res2Net_v1b.py:60-63,70-72,84-89 with running statistics): scale / shift [Cout] are the folded BatchNorm rows of pn2_bn_eval_prepare
 * (physical columns), d->flags carries PN2_CONV_AFFINE (| PN2_CONV_RELU | PN2_CONV_RELU6); `res` (optional, same dtype, 16-byte aligned rows) is added before the
 * activation.  No raw conv output is written and no separate pn2_affine_act pass runs.  fp32: bit-identical to the two launches; bf16: one rounding fewer.  Table-driven
 * form: a pn2_conv_job with psum = scale, psq = shift, ep.a.y / ep.a.ld_y = res, every other ep field zero. */
int pn2_conv_gemm_affine(int dtype, const void* in, const void* wp, void* out, const float* scale, const float* shift, const void* res, int ld_res,
                         const pn2_conv_desc* d, void* stream);
typedef struct pn2_bnb_target {
    void* out; int ld_out;          /* target b only: second destination (same dtype / column range as the GEMM's out) */
    int mode;                       /* PN2_BNB_* ; 0 = no statistics */
    const void* raw; int ld_raw;    /* raw conv output behind the BatchNorm, columns aligned with the destination's */
    const void* y; int ld_y;        /* PN2_BNB_MASK_Y: stored activation */
    const float* par; int ps;       /* rows scale, shift, mean, invstd of that BatchNorm, row stride ps (elements) */
    int split;                      /* > 0: columns >= split use raw2 / par2 instead (a concat buffer whose tail belongs to another BatchNorm); */
    const void* raw2; const float* par2;   /* par2 == NULL: no statistics for those columns (zeros are written) */
    float* p1; float* p2; int ldp;  /* partial rows [pn2_conv_stat_blocks or ceil(M / tile_m)][ldp] */
} pn2_bnb_target;
/* pool (optional, with PN2_CONV_ACCUM, target a with statistics, no target b, even OH / OW): the prior content of the += destination is NOT read from `out`; it is
 * 1/4 of row (n, y/2, x/2) of this quarter-resolution tensor [N][OH/2][OW/2][ld_pool] for output row (n, y, x) - the backward of AvgPool2d(2, 2) in front of the
 * downsample conv of a Res2Net stage block (Res2Net_v1b.py:127-136,80) folded into the dgrad that completes the block input's gradient: out = dgrad + pool/4 (the bits a
 * separate pn2_avgpool_bwd + += would leave: 1/4 is exact), no pool-backward launch, no full-resolution write + re-read of that gradient. */
/* c (optional; mode == PN2_BNB_STATS, bf16 launches whose tile has at most 4096 elements, no target b): a SECOND BatchNorm whose output gradient is target a's masked
 * gradient dz - the BatchNorm of a residual branch without activation (Bottle2neck's downsample: out = relu(bn3(..) + dbn(dconv(pool(x)))), Res2Net_v1b.py:80-89,127-136).
 * The epilogue that forms dz for bn3 leaves p1 = sum dz, p2 = invstd_c * (sum dz * raw_c - mean_c * sum dz) for it as well (fields raw / ld_raw / par / ps / p1 / p2 / ldp):
 * the pn2_bn_bwd_reduce pass of that BatchNorm (two full-tensor reads) disappears. */
typedef struct pn2_conv_ep { pn2_bnb_target a, b; const void* pool; int ld_pool; int pad_; pn2_bnb_target c; } pn2_conv_ep;
int pn2_conv_gemm_ep(int dtype, const void* in, const void* wp, void* out, const pn2_conv_desc* d, const pn2_conv_ep* ep, void* stream);
/* Many conv GEMMs (forward / dgrad, with or without the epilogue above) of ONE tile shape in one launch, from a DEVICE job table: the convs at the same
 * position of independent chains (the RFB branches of the three RFB modules, pranet.py:46-83; the parallel 3x3 convs of a Res2Net stage block,
 * Res2Net_v1b.py:66-69) run in lock step.  pn2_conv_gemm_tile = (bm << 8 | bn) pn2_conv_gemm would pick for a desc (the partial-row counts of its
 * statistics depend on bm, so a job must run on its own tile); jobs of equal tile and equal `ep` use (any target mode / b.out set) share a launch.
 * The general (not pointwise-specialised) kernels serve every job; split-K jobs cannot join.  Bit-identical to the single launches.
 * `ep`: bit 0 = the jobs carry a pn2_conv_ep; for bf16 its operand tiles are staged in LDS and the launch is sized for what the jobs need -
 * bit 1: a.mode has PN2_BNB_STATS, bit 2: ... and PN2_BNB_MASK_Y, bit 3: PN2_CONV_ACCUM, bit 4: b.out with PN2_BNB_STATS (OR over the jobs; no
 * bits = all four).  Only tiles of at most 4096 elements stage operand tiles (the 3x3 dgrads); wider tiles keep their operands in registers.  -4: LDS exceeded. */
typedef struct pn2_conv_job { const void* in; const void* wp; void* out; float* psum; float* psq; pn2_conv_desc d; int pad_; pn2_conv_ep ep; } pn2_conv_job;
int pn2_conv_gemm_tile(int dtype, const pn2_conv_desc* d);
int pn2_conv_gemm_job_blocks(int dtype, const pn2_conv_job* j, int bm, int bn);
int pn2_conv_gemm_multi(int dtype, int bm, int bn, int ep, const pn2_conv_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
int pn2_conv_splitk_reduce(int dtype, const float* ws, int ksplit, int M, int Cout, void* out, int ld_out, const float* bias, float* psum, float* psq,
                           int accumulate, void* stream);   /* psum / psq: BatchNorm partial rows [ceil(M / 64)][Cout] of the summed result, or NULL */
int pn2_conv_wgrad(int dtype, const void* dy, const void* x, float* slab, const pn2_wgrad_desc* d, int nsplit, void* stream);
/* Many weight gradients in ONE launch.  A wgrad only feeds the optimizer, so a training step may defer them (keeping dy / x alive)
 * and run all convs that share a kernel instantiation (pn2_conv_wgrad_variant) together, from a DEVICE job table.
 * block_start_dev: njobs + 1 prefix sums of pn2_conv_wgrad_blocks(&job.d, job.nsplit).  Same arithmetic, bit for bit, as
 * pn2_conv_wgrad on each job.  variant 14 (bf16): jobs of variants 12 AND 13 (the 128 x 256 LDS-DMA tile, k x k and pointwise) in one table - the k x k chains leave the
 * memory system idle, the pointwise jobs stream; the caller interleaves the job ranges (pn2/core.py GradQueue._build). */
typedef struct pn2_wgrad_job { const void* dy; const void* x; float* slab; pn2_wgrad_desc d; int nsplit;
    int rot;      /* 0..7: XCD rotation of this job's pixel splits (split s runs on XCD (s + rot) % 8): the caller advances it by nsplit % 8 from job to job, so the
                   * partial split groups of a table (nsplit = 3, 6, 12 ... on the long-contraction layers) spread over the 8 XCDs instead of piling onto the first ones */
} pn2_wgrad_job;
int pn2_conv_wgrad_variant(int dtype, const pn2_wgrad_desc* d);
int pn2_conv_wgrad_blocks(const pn2_wgrad_desc* d, int nsplit);
int pn2_conv_wgrad_multi(int dtype, int variant, const pn2_wgrad_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
int pn2_pack_weight(int dtype, const float* w_oihw, void* wp, const pn2_pack_desc* p, void* stream);
int pn2_wgrad_reduce(const float* slab, float* gw_oihw, const pn2_pack_desc* p, int nsplit, int accumulate, void* stream);
/* Data gradient of a "patchify" conv (kernel == stride, pad 0: the spatial-reduction convs, pvtv2.py:70) as GEMM + depth-to-space instead of the
 * transposed gather: pn2_pack_patch_weight builds wp[(tap*Cin_p + ci)][co] = w[co][ci][tap] ([Rp][Kp], zero pads), pn2_conv_gemm (1x1) gives
 * t[(n,oy,ox)][(tap, ci)], pn2_depth_to_space writes dx[n][oy*S+kh][ox*S+kw][c] (+)= t[..][(kh*S+kw)*C + c] (zero outside the patch area). */
int pn2_pack_patch_weight(int dtype, const float* w_oihw, void* wp, int Cout, int Cin, int KH, int KW, int Cin_p, int Rp, int Kp, void* stream);
int pn2_depth_to_space(int dtype, const void* t, int ld_t, void* dx, int ld_dx, int N, int H, int W, int OH, int OW, int S, int C, int accumulate, void* stream);
/* Data gradient of a strided conv with <= 4 input channels (the 7x7 stride-4 patch embedding behind EMCADNet's 1 -> 3 channel stem,
 * EMCAD/lib/networks.py:100-102): a thread per input pixel walks only the taps that land on an output pixel; w is the fp32 OIHW master. */
int pn2_conv_dgrad_small_cin(int dtype, const void* dy, int ld_dy, const float* w_oihw, void* dx, int ld_dx, int N, int H, int W, int OH, int OW,
                             int Cout, int Cin, int KH, int KW, int stride, int pad, int accumulate, void* stream);
/* one launch that repacks many weights (all convs of a model, forward and dgrad panels) from a DEVICE job table */
typedef struct pn2_pack_job { const float* w; void* wp; pn2_pack_desc d; } pn2_pack_job;
/* block_start_dev[j] = first workgroup of job j (njobs + 1 entries, prefix sums of pn2_pack_blocks(&job.d)).  Only real
 * (non-pad) elements are rewritten: the panels must have been created by pn2_pack_weight(), which also zero-fills the pads. */
int pn2_pack_blocks(const pn2_pack_desc* p);
int pn2_pack_weights_multi(int dtype, const pn2_pack_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
/* the split-K reductions (pn2_wgrad_reduce) of many convs in ONE launch, from a DEVICE job table: they only depend on their own
 * pn2_conv_wgrad launch, so a training step defers them and runs them together (prefix sums of pn2_wgrad_reduce_blocks). */
typedef struct pn2_reduce_job { const float* slab; float* gw; pn2_pack_desc d; int nsplit; int accumulate; } pn2_reduce_job;
int pn2_wgrad_reduce_blocks(const pn2_pack_desc* p);
int pn2_wgrad_reduce_multi(const pn2_reduce_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);

/* ---------------------------------------------------------------------------------------------- batch norm
 * nn.BatchNorm2d train/eval forward + backward (lib/pranet.py:37,41-42 ; lib/Res2Net_v1b.py:33,45,50,103,106,110,135),
 * fused with the ReLU / residual add that follows it (Res2Net_v1b.py:63,72,88-89 ; pranet.py:82,358-360).   */
typedef struct pn2_bn_desc {
    int M;                  /* pixels */
    int Cp;                 /* physical channels of the raw conv output */
    int C, gw, gwp;         /* logical channels and group-padding map of gamma/beta/running stats */
    float eps, momentum;
    int ldp;                /* row stride of the partial buffers and plane stride of `coef` (0 = Cp): a BN that owns a channel
                               slice [c0, c0+Cp) of a wider fused conv output passes base pointers offset by c0 and ldp = total */
    int tile_rows;          /* pn2_bn_finalize: 0 = the partial rows hold raw moments (sum x, sum x^2) of their row block; > 0 = they hold
                               (mean, M2 = sum (x - mean)^2) of blocks of tile_rows rows (the last one shorter), as the GEMM epilogue leaves
                               them (PN2_CONV_STATS): merged with Chan's formula in double, no E[x^2] - mean^2 cancellation */
} pn2_bn_desc;
/* batch statistics from the conv epilogue partials -> scale/shift (physical), saved mean/invstd, running-stat update */
int pn2_bn_finalize(const float* psum, const float* psq, int nblk, const pn2_bn_desc* d, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float* scale, float* shift, float* mean, float* invstd, void* stream);
/* eval mode: scale/shift from running statistics */
int pn2_bn_eval_prepare(const pn2_bn_desc* d, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float* scale, float* shift, void* stream);
/* the same for many BatchNorms in one launch, from a DEVICE job table (inference: every BatchNorm of a model folded by one graph node); block_start_dev:
 * njobs + 1 prefix sums of ceil(Cp / 256) */
typedef struct pn2_bnprep_job { const float* gamma; const float* beta; const float* running_mean; const float* running_var; float* scale; float* shift; pn2_bn_desc d; int pad_; } pn2_bnprep_job;
int pn2_bn_eval_prepare_multi(const pn2_bnprep_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
/* y[m][c] = act(x[m][c]*scale[c] + shift[c] + res[m][c]) for c < Cout ; scale==NULL -> identity affine */
int pn2_affine_act(int dt_in, const void* x, int ld_x, int dt_out, void* y, int ld_y, int M, int Cout,
                   const float* scale, const float* shift, const void* res, int ld_res, int relu, void* stream);
/* pn2_affine_act (same dtype, no residual) with a second output y2 = y + add: Bottle2neck's branch sum sp + spx[i] (Res2Net_v1b.py:66-68)
 * written by the pass that produces sp.  16-byte aligned rows only (-2 otherwise). */
int pn2_affine_act_sum(int dt, const void* x, int ld_x, void* y, int ld_y, int M, int C, const float* scale, const float* shift, int relu,
                       const void* add, int ld_add, void* y2, int ld_y2, void* stream);
/* pn2_affine_act (same dtype, no residual) whose channels >= c_lo are ALSO written to y3[m][c - c_lo]: conv1 + bn1 + ReLU of a Bottle2neck writes its last
 * slice spx[3] straight into the concat buffer (Res2Net_v1b.py:78-79: `out = torch.cat((out, spx[self.nums]), 1)`), no copy launch.  16-byte aligned rows / c_lo only. */
int pn2_affine_act_tee(int dt, const void* x, int ld_x, void* y, int ld_y, int M, int C, const float* scale, const float* shift, int relu,
                       void* y3, int ld_y3, int c_lo, void* stream);
/* BatchNorm + ReLU + MaxPool2d(3, 2, 1) forward as ONE pass (the stem of Res2Net_v1b.py:137-139: self.bn1 -> self.relu -> self.maxpool): the normalised full-resolution activation feeds
 * the pool alone, so it is never written.  pooled[n][oy][ox][c] = max over the window of T(relu(raw*scale + shift)), idx = argmax tap (0..8), exactly as pn2_affine_act followed by
 * pn2_maxpool3x3s2_fwd give them.  The backward is pn2_maxpool3x3s2_bwd + the BatchNorm passes with the ReLU mask recomputed from raw. */
int pn2_bn_relu_maxpool_fwd(int dt, const void* raw, int ld_raw, const float* scale, const float* shift, void* y, int ld_y, unsigned char* idx,
                            int N, int H, int W, int C, int OH, int OW, void* stream);
/* The backward of that op without the full-resolution gradient tensor (even H, W; C / vector a power of two <= 256): the two BatchNorm passes with the incoming gradient formed per
 * 2 x 2 input quad from the pooled gradient dpool [N][OH][OW][C] and idx - each pixel's value is the sum pn2_maxpool3x3s2_bwd forms, in its tap order, rounded to T.  p1 / p2: nblk
 * partial rows for pn2_bn_bwd_finalize (any nblk >= 1: the launch has nblk workgroups); dz as pn2_bn_bwd_apply writes it.  mask_scale / mask_shift: the forward's scale / shift. */
int pn2_pool_bn_bwd_reduce(int dt, const void* dpool, int ld_dp, const unsigned char* idx, const void* raw, int ld_raw, int N, int H, int W, int C, int OH, int OW,
                           const float* mean, const float* invstd, const float* mask_scale, const float* mask_shift, float* p1, float* p2, int nblk, void* stream);
int pn2_pool_bn_bwd_apply(int dt, const void* dpool, int ld_dp, const unsigned char* idx, const void* raw, int ld_raw, int N, int H, int W, int C, int OH, int OW,
                          const float* mean, const float* invstd, const float* coef, const float* mask_scale, const float* mask_shift, void* dz, int ld_dz, void* stream);
/* backward pass 1: per-channel partials of sum(dz) and sum(dz*xhat), dz = dy*(relu mask) ; dy has Cdy valid channels.
 * ReLU mask: y>0 when y is given; else recomputed as fmaf(x,mask_scale,mask_shift)>0 when mask_scale is given (saves reading y). */
int pn2_bn_bwd_reduce(int dt, int dt_dy, const void* dy, int ld_dy, int Cdy, const void* y, int ld_y, int dt_y, const void* x, int ld_x,
                      int M, int Cp, const float* mean, const float* invstd, float* p1, float* p2, int nblk,
                      const float* mask_scale, const float* mask_shift, int relu6, void* stream);
int pn2_bn_bwd_blocks(int M, int Cp, int dtype);   /* rows of the p1/p2 partial buffers */
/* pass 1b: dgamma/dbeta (logical, optionally accumulated) + per-channel coefficients for pass 2 */
int pn2_bn_bwd_finalize(const float* p1, const float* p2, int nblk, const pn2_bn_desc* d, const float* gamma, const float* invstd,
                        float* dgamma, float* dbeta, int accumulate, float* coef, void* stream);
/* pn2_bn_bwd_finalize when the partial rows of the BatchNorm's channels come from up to 4 different producers (GEMM epilogues, pn2_conv_gemm_ep,
 * or pn2_bn_bwd_reduce on a channel slice): segment s covers physical channels [c0[s], c0[s+1]) (the last one up to d->Cp), p1[s] / p2[s] point at
 * its first column, nblk[s] rows of stride ldp[s].  d->ldp is the plane stride of `coef` only. */
typedef struct pn2_bn_segs { int nseg; int c0[4]; int nblk[4]; int ldp[4]; const float* p1[4]; const float* p2[4]; } pn2_bn_segs;
int pn2_bn_bwd_finalize_seg(const pn2_bn_segs* segs, const pn2_bn_desc* d, const float* gamma, const float* invstd,
                            float* dgamma, float* dbeta, int accumulate, float* coef, void* stream);
/* ---- table-driven launches of the BatchNorm family.  Independent chains of a model (the three RFB modules and their three branches each, pranet.py:46-83;
 * the three parallel 3x3 convs of a Res2Net stage block, Res2Net_v1b.py:66-69) advance in LOCK STEP: one launch per kernel kind and position in the
 * chain instead of one per chain.  pn2_*_job_blocks (host) fills the derived geometry of a job and returns its workgroup count (< 0: not batchable,
 * launch it on its own); pn2_*_multi runs the jobs from a DEVICE table (block_start_dev: njobs + 1 prefix sums).  Bit-identical to the single launches. */
typedef struct pn2_bnfin_job { const float* psum; const float* psq; const float* gamma; const float* beta; float* running_mean; float* running_var;
                               float* scale; float* shift; float* mean; float* invstd; pn2_bn_desc d; int nblk; int cpb; int pad_; } pn2_bnfin_job;
int pn2_bn_finalize_job_blocks(pn2_bnfin_job* j);
int pn2_bn_finalize_multi(const pn2_bnfin_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
typedef struct pn2_affine_job { const void* x; void* y; const float* scale; const float* shift; const void* res; const void* add; void* y2;
                                int ld_x, ld_y, ld_res, ld_add, ld_y2, M, C, relu, rows_per_blk, cvp; } pn2_affine_job;     /* pn2_affine_act / _sum, 16-byte rows */
int pn2_affine_job_blocks(int dt, pn2_affine_job* j);
int pn2_affine_multi(int dt, const pn2_affine_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
typedef struct pn2_bnbfin_job { pn2_bn_segs sg; const float* gamma; const float* invstd; float* dgamma; float* dbeta; float* coef; pn2_bn_desc d;
                                int accumulate; int cpb; int pad_; } pn2_bnbfin_job;   /* pn2_bn_bwd_finalize(_seg): plain = one segment */
int pn2_bn_bwd_finalize_job_blocks(pn2_bnbfin_job* j);
int pn2_bn_bwd_finalize_multi(const pn2_bnbfin_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
typedef struct pn2_bnapply_job { const void* dy; const void* y; const void* x; const float* mean; const float* invstd; const float* coef; void* dx; void* dres;
                                 const float* msc; const float* msh; int ld_dy, ld_y, ld_x, ld_dx, ld_dres, M, Cp, dres_accum, r6, rows_per_blk, cvp, pad_; } pn2_bnapply_job;
int pn2_bn_bwd_apply_job_blocks(int dt, pn2_bnapply_job* j);
#define PN2_MULTI_F32DY 0x200  /* OR into the dt (PN2_BF16) of pn2_bn_bwd_apply / _reduce job_blocks + _multi: the jobs' dy is fp32 with job.pad_ channels (the K-channel head maps of bf16 layers: element-wise / scalar kernels, as pn2_bn_bwd_apply / _reduce launch for that dtype pair) */
#define PN2_MULTI_F32OUT 0x200 /* ... and of pn2_affine_job_blocks / pn2_affine_multi: bf16 in, fp32 out (element-wise kernel) */
#define PN2_MULTI_LEAN 0x100   /* OR into pn2_bn_bwd_apply_multi's dt when NO job of the table has y or dres: the launch runs the register-lean instantiation */
int pn2_bn_bwd_apply_multi(int dt, const pn2_bnapply_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
typedef struct pn2_bnreduce_job { const void* dy; const void* y; const void* x; const float* mean; const float* invstd; float* p1; float* p2;
                                  const float* msc; const float* msh; int ld_dy, ld_y, ld_x, M, Cp, nblk, rows_per_blk, cvp, r6, pad_; } pn2_bnreduce_job;
int pn2_bn_bwd_reduce_job_blocks(int dt, pn2_bnreduce_job* j);
int pn2_bn_bwd_reduce_multi(int dt, const pn2_bnreduce_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
/* pass 2: dx = g*invstd*(dz - c1 - xhat*c2) ; optional dres (+)= dz.  coef = [gscale|c1|c2] each Cp long.
 * coef==NULL: pure activation backward (dx = dz), used for eval-mode / affine-only layers.                */
int pn2_bn_bwd_apply(int dt, int dt_dy, const void* dy, int ld_dy, int Cdy, const void* y, int ld_y, int dt_y, const void* x, int ld_x,
                     int M, int Cp, const float* mean, const float* invstd, const float* coef, void* dx, int ld_dx,
                     void* dres, int ld_dres, int dres_accum, const float* mask_scale, const float* mask_shift, int relu6, void* stream);

/* ---------------------------------------------------------------------------------------------- pooling
 * nn.MaxPool2d(3,2,1) Res2Net_v1b.py:112 ; nn.AvgPool2d(3,stride,1) :40,80 ; AvgPool2d(s,s,ceil,count_include_pad=False) :131-132 */
int pn2_maxpool3x3s2_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, unsigned char* idx, int N, int H, int W, int C, int OH, int OW, void* stream);
int pn2_maxpool3x3s2_bwd(int dt, const void* dy, int ld_dy, const unsigned char* idx, void* dx, int ld_dx, int N, int H, int W, int C, int OH, int OW, void* stream);
int pn2_avgpool_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, int OH, int OW,
                    int k, int stride, int pad, int count_include_pad, void* stream);
int pn2_avgpool_bwd(int dt, const void* dy, int ld_dy, void* dx, int ld_dx, int N, int H, int W, int C, int OH, int OW,
                    int k, int stride, int pad, int count_include_pad, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------- resampling
 * nn.Upsample(x2, bilinear, align_corners=True) pranet.py:93 ; F.interpolate(scale_factor=s, mode='bilinear')
 * (align_corners=False, given scale used) pranet.py:349-354,370-376,392-398,414-415.  rh/rw = source step per output pixel. */
int pn2_bilinear_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, int N, int H, int W, int C, int OH, int OW,
                     int align_corners, float rh, float rw, void* stream);
int pn2_bilinear_bwd(int dt, const void* dy, int ld_dy, void* dx, int ld_dx, int N, int H, int W, int C, int OH, int OW,
                     int align_corners, float rh, float rw, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------------------- DSRA / RA
 * V2 fusion fg + fg*softmax(crop_fg - crop_bg, dim=1) (or *(crop_fg-crop_bg)) pranet.py:365-368,385-389,407-411 ;
 * all maps fp32 [M][K].  V1 gate (1 - sigmoid(crop)).expand(C) * x  PraNet_Res2Net.py:153-154,166-167,177-178. */
int pn2_dsra_fuse_fwd(const float* fg, const float* crop_fg, const float* crop_bg, float* out, int M, int K, int use_softmax, void* stream);
int pn2_dsra_fuse_bwd(const float* fg, const float* crop_fg, const float* crop_bg, const float* dout, float* dfg, float* dcrop_fg,
                      float* dcrop_bg, int M, int K, int use_softmax, void* stream);
int pn2_ra_gate_fwd(int dt, const void* x, int ld_x, const float* crop, void* out, int ld_out, int M, int C, void* stream);
int pn2_ra_gate_bwd(int dt, const void* x, int ld_x, const float* crop, const void* dout, int ld_dout, void* dx, int ld_dx, int dx_accum,
                    float* dcrop, int M, int C, void* stream);
/* backward of the gate applied BEHIND the conv (pn2_conv_gemm_gated): raw = gate * u is the stored (gated) conv output, dz its gradient;
 * dzg = (1 - s) * dz  (gradient of the un-gated GEMM result, feeds dgrad and wgrad),  dcrop[m] = -s * sum_c raw[m][c] * dz[m][c],  s = sigmoid(crop[m]). */
int pn2_ra_gate_post_bwd(int dt, const void* raw, int ld_raw, const float* crop, const void* dz, int ld_dz, void* dzg, int ld_dzg, float* dcrop,
                         int M, int C, void* stream);

/* ---------------------------------------------------------------------------------------------- structure loss
 * MyTrain_med.py:19-38 applied to the P (fg,bg) pairs of :78-82 in one pass.  preds = P fg maps then P bg maps,
 * each [N][HW] fp32 ; mask [N][HW].  weit/wsum are `weit` is produced once per batch by pn2_loss_weights.   */
int pn2_loss_weights(const float* mask, float* weit, int N, int H, int W, int ksize, void* stream);
/* the same launch also zeroes `nclear` 64-bit words at `clear` (the image-sum accumulators `isum` of pn2_dsra_tail_fwd_bwd, which runs right behind it) */
int pn2_loss_weights_clear(const float* mask, float* weit, int N, int H, int W, int ksize, long long* clear, int nclear, void* stream);
int pn2_loss_blocks(int HW);            /* row-chunks per image of the `partial` scratch: [P][N][blocks][5] floats */
/* preds: 2P maps laid out at preds + j*map_stride (j<P: fg of pair j, j>=P: bg of pair j-P), each [N][HW] fp32.
 * Outputs: sums [P][N][4] and wsum [N] (kept for the backward), loss [P+1] = per-pair losses then their total. */
int pn2_structure_loss_fwd(const float* preds, long long map_stride, int P, const float* mask, const float* weit, float* partial,
                           float* sums, float* wsum, float* loss, int N, int HW, void* stream);
int pn2_structure_loss_bwd(const float* preds, float* dpreds, long long map_stride, int P, const float* mask, const float* weit,
                           const float* wsum, const float* sums, float gscale, int N, int HW, void* stream);
/* The same backward for logits that live where the caller's framework put them (the nn.Module surface: autograd hands the loss the maps the model returned):
 * dpreds has its own map stride (dmap_stride, elements), and the upstream gradient of pair p is read from the DEVICE (gscale_dev[p], may be NULL) times gscale -
 * no host synchronisation, no separate scaling pass over the gradient maps (MyTrain_med.py:78-84: loss = loss5 + loss3 + loss2 + loss1; loss.backward()). */
int pn2_structure_loss_bwd_dev(const float* preds, float* dpreds, long long map_stride, long long dmap_stride, int P, const float* mask, const float* weit,
                               const float* wsum, const float* sums, const float* gscale_dev, float gscale, int N, int HW, void* stream);

/* ---------------------------------------------------------------------------------------------- fused DSRA tail (K = 1)
 * The memory-bound end of the training step as two kernels: lateral = bilinear(low-res fg|bg map, x8/x16/x32) for all 2P maps
 * (pranet.py:354,371,393,415) + the dual structure loss of MyTrain_med.py:19-38,78-82 in ONE pass that writes the 2P full-resolution
 * maps once; and its backward, which recomputes the logits from the low-res maps and applies the bilinear adjoint on the fly, so the
 * 2P full-resolution gradient maps are never materialised.  Algorithmic traffic per image (S = OH*OW*4 B): forward 2P*S written +
 * 2*S read (mask, weit); backward ~2*S*(#distinct scales)*2 read.                                                               */
#define PN2_TAIL_MAX_MAPS 16
typedef struct pn2_tail_map {
    const float* src;       /* low-res K=1 logits [N][h][w] fp32 */
    float* dsrc;            /* their gradient, same layout (backward only) */
    int h, w;
    float rh, rw;           /* source-index scales of the resize (1/scale_factor; (h-1)/(OH-1) with align_corners) */
    int accumulate;         /* backward: dsrc += instead of = */
    int pad_;
} pn2_tail_map;
typedef struct pn2_tail_desc {
    int N, OH, OW, P, align_corners, pad_;
    pn2_tail_map maps[PN2_TAIL_MAX_MAPS];       /* maps[j] j<P: fg logits of pair j ; maps[P+j]: bg logits of pair j */
} pn2_tail_desc;
int pn2_dsra_tail_blocks(int OH);               /* row bands per image of `partial`: [P][N][blocks][5] floats */
/* lat: [2P][N][OH*OW] fp32 written ; sums/wsum/loss as pn2_structure_loss_fwd */
int pn2_dsra_tail_fwd(const pn2_tail_desc* d, float* lat, const float* mask, const float* weit, float* partial,
                      float* sums, float* wsum, float* loss, void* stream);
/* floats of caller-owned scratch the band-wise backward needs ([N][blocks][3 rows of every map]); 0 = geometry served by the row kernels */
int pn2_dsra_tail_scratch(const pn2_tail_desc* d);
/* scratch may be NULL / short: the backward then runs one block per low-res row (slower, same result up to summation order) */
int pn2_dsra_tail_bwd(const pn2_tail_desc* d, const float* mask, const float* weit, const float* wsum, const float* sums,
                      float gscale, float* scratch, long long scratch_floats, void* stream);
/* Forward + backward of the tail in ONE pass over the pixels (pranet.py:354,371,393,415 + MyTrain_med.py:19-38,78-84): the gradient of the structure
 * loss is linear in three per-logit quantities whose coefficients alone need the image-wide sums, so the forward walk also pushes them through the
 * bilinear adjoint and leaves band partials; a second small kernel combines them with the coefficients it forms from the loss partials, a third
 * finishes loss[P+1].  No backward walk, no single-workgroup finalize between two big kernels.  Served geometry (pn2_dsra_tail_fused_ok() == 1, else -2:
 * use _fwd + _bwd): align_corners = 0, every map magnified by a power of two >= 8 in both directions (all scales of MyTrain_med.py:55,70-73), pair p's
 * fg and bg map of one geometry, OW <= 512.  per: [P][N] floats of caller-owned scratch; sums / wsum / loss as pn2_dsra_tail_fwd (valid when the call has
 * run); scratch: pn2_dsra_tail_fused_scratch() floats.  PN2_TAIL_BAND=0|1 (row / band kernels of the two-call path) also switches this path off.
 * isum (optional): P*N*5 sums of TWO 64-bit words each (P*N*10 words), ZERO on entry (pn2_loss_weights_clear) - the walk adds every image's five loss sums into
 * them as two-word fixed point (value * 2^30: integer part, 50 fractional bits) with integer atomics: integer addition is associative, so the result does not
 * depend on the order in which the workgroups arrive - the step is bit-reproducible across replays and ranks (tests/test_gpu_determinism.py).  The second kernel
 * reads them instead of reducing the partial rows, and one of its workgroups finishes loss[P+1]: two launches instead of three (P*N <= 1024).
 * NULL: the second kernel reduces `partial` itself and a third launch forms the loss. */
int pn2_dsra_tail_fused_ok(const pn2_tail_desc* d);
int pn2_dsra_tail_fused_scratch(const pn2_tail_desc* d);
int pn2_dsra_tail_fwd_bwd(const pn2_tail_desc* d, float* lat, const float* mask, const float* weit, float* partial, float* sums, float* wsum,
                          float* per, float* loss, float gscale, float* scratch, long long scratch_floats, long long* isum, void* stream);

/* ---------------------------------------------------------------------------------------------- PVTv2 encoder (lib/pvtv2.py)
 * Tokens [B, N, C] of the reference are NHWC pixels here.  The nn.Linear layers run as 1x1 pn2_conv_gemm / pn2_conv_wgrad.      */
/* nn.LayerNorm over the C channels of each of M rows (pvtv2.py:71,119,126,169,224-247); mean/rstd [M] fp32 are kept for the backward */
int pn2_layernorm_fwd(int dt, const void* x, int ld_x, void* y, int ld_y, int M, int C, const float* gamma, const float* beta, float eps,
                      float* mean, float* rstd, void* stream);
int pn2_ln_slots(int dt, int C);                /* rows a workgroup handles at a time (row unit of the partial buffers) */
int pn2_rows_blocks(int M, int unit);           /* partial rows (= workgroups) of the column-reduction kernels below */
/* dx (+)= LN backward ; pg/pb [nblk][C] partial dgamma/dbeta rows, nblk = pn2_rows_blocks(M, pn2_ln_slots(dt, C)); finish with pn2_colsum_finalize */
int pn2_layernorm_bwd(int dt, const void* dy, int ld_dy, const void* x, int ld_x, int M, int C, const float* gamma, const float* mean, const float* rstd,
                      void* dx, int ld_dx, int accumulate_dx, float* pg, float* pb, int nblk, void* stream);
/* bias gradients: partial[nblk][C] = per-block column sums of dy [M][C] (nblk = pn2_rows_blocks(M, pn2_colsum_unit(dt, C))), then
 * out[c] (+)= sum_b partial[b*ld + c] in a fixed order */
int pn2_colsum_unit(int dt, int C);
int pn2_colsum(int dt, const void* dy, int ld, int M, int C, float* partial, int nblk, void* stream);
int pn2_colsum_finalize(const float* partial, int nblk, int C, int ld, float* out, int accumulate, void* stream);
/* many pn2_colsum_finalize in ONE launch from a DEVICE job table (prefix sums of pn2_colsum_finalize_blocks(C) per job): the bias / LayerNorm /
 * depth-wise parameter-gradient sums of a training step only feed the optimizer, so the backward pass queues them and runs them together */
/* the column sums themselves (pn2_colsum) for many tensors of one dtype in ONE launch; pn2_colsum_job_blocks fills rows / cvp and returns the job's
 * workgroup count = rows of its partial buffer */
typedef struct pn2_colsum_in_job { const void* dy; float* partial; int ld, M, C, rows, cvp, pad_; } pn2_colsum_in_job;
int pn2_colsum_job_blocks(int dt, pn2_colsum_in_job* job);
int pn2_colsum_multi(int dt, const pn2_colsum_in_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
typedef struct pn2_colsum_job { const float* partial; float* out; int nblk, C, ld, accumulate; } pn2_colsum_job;
int pn2_colsum_finalize_blocks(int C);
int pn2_colsum_finalize_multi(const pn2_colsum_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
/* DWConv (pvtv2.py:363-374, groups = C, 3x3, pad 1) [+ bias] [+ nn.GELU of Mlp.forward :45]: z = dw(x) + b (kept for the backward),
 * y_gelu = gelu(z) when non-null.  flip=1 correlates with the mirrored kernel = data gradient of the same conv.  w [C][9] fp32. */
int pn2_dwconv3x3(int dt, const void* x, const float* w, const float* b, void* z, void* y_gelu, int N, int H, int W, int C, int flip, int accumulate, void* stream);
/* The data-gradient form (no GELU, no accumulate; bf16 window kernels only) that also leaves cpart[nblk][C]: per-workgroup column sums of the stored result - the
 * bias gradient of the nn.Linear whose output gradient this is (Mlp.fc1, pvtv2.py:49-56) without a second read of that tensor.  nblk = pn2_dwconv3x3_colsum_blocks()
 * (< 0: geometry not served).  Finish with pn2_colsum_finalize. */
int pn2_dwconv3x3_colsum_blocks(int dt, int N, int H, int W, int C);
int pn2_dwconv3x3_colsum(int dt, const void* x, const float* w, const float* b, void* z, int N, int H, int W, int C, int flip, float* cpart, int nblk, void* stream);
int pn2_gelu_bwd(int dt, const void* dy, const void* z, void* dz, long long n, void* stream);      /* dz = dy * gelu'(z), exact erf form */
/* partial[nblk][C*10]: columns c*9+tap = dW, C*9+c = dbias ; nblk = pn2_dwconv3x3_wgrad_blocks(dt, N, H, W, C) ; finish with pn2_colsum_finalize.
 * zpre non-null: `dz` holds dy (the gradient of the GELU output), dz = dy * gelu'(zpre) is formed in the same pass and written to dz_out
 * (the pn2_gelu_bwd pass fused in; dz_out then feeds the flip=1 data-gradient launch). */
int pn2_dwconv3x3_wgrad_blocks(int dt, int N, int H, int W, int C);
int pn2_dwconv3x3_wgrad(int dt, const void* dz, const void* x, float* partial, int nblk, int N, int H, int W, int C, const void* zpre, void* dz_out, void* stream);
/* Spatial-reduction attention (Attention.forward pvtv2.py:90-111), head_dim 64, Nkv <= 256:
 * q [B][Nq][heads*64] ; kv [B][Nkv][2*heads*64] (k then v, heads inner, as the reference's reshape(B,-1,2,heads,hd)) ;
 * out = softmax(q k^T * scale) v, heads concatenated ; lse [B][heads][Nq] fp32 saved for the backward.
 * Backward: dq, dkv written (not accumulated); `partial` is scratch (see pn2_attn_bwd_blocks). */
/* DropPath (stochastic depth, pvtv2.py:125,148-149): y[n] = x[n] * scale[n] (+ res[n]: the residual add of Block.forward :148-149 in the same pass),
 * scale[n] = bernoulli(keep)/keep drawn by the caller; with res == NULL it is its own adjoint */
int pn2_scale_samples(int dt, const void* x, void* y, const float* scale, const void* res, int N, long long per_sample, void* stream);
int pn2_attn_fwd(int dt, const void* q, int ld_q, const void* kv, int ld_kv, void* out, int ld_o, float* lse, int B, int Nq, int Nkv, int heads, int head_dim,
                 float scale, void* stream);
int pn2_attn_bwd_blocks(int dt, int B, int heads, int Nq);   /* partial slots per (b, head): partial holds [B][heads][slots][2][roundup(Nkv,64)][64] fp32 */
/* out = the forward result (delta = rowsum(dO * out) lets the bf16 MFMA path treat key ranges independently); delta: [B][heads][Nq] fp32 scratch */
int pn2_attn_bwd(int dt, const void* q, int ld_q, const void* kv, int ld_kv, const void* out, int ld_o, const void* dout, int ld_do, const float* lse, void* dq, int ld_dq,
                 void* dkv, int ld_dkv, float* partial, float* delta, int B, int Nq, int Nkv, int heads, int head_dim, float scale, void* stream);

/* ---------------------------------------------------------------------------------------------- EMCAD decoder (multiclass_seg/EMCAD/lib/decoders.py)
 * Memory-bound pieces of config 5; the 1x1 / 3x3 / 7x7 convs are pn2_conv_*, BatchNorm is pn2_bn_* fed by the partial rows below.          */
/* rows of the BatchNorm partial buffers of pn2_dwconv (wgrad = 0) / of the partial buffer of pn2_dwconv_wgrad (wgrad = 1) */
int pn2_dwconv_blocks(int dt, int N, int H, int W, int C, int K, int wgrad);
/* depth-wise K x K conv (K = 1, 3, 5), pad K/2, stride 1, no bias: z (+)= dw(x); flip = mirrored kernel (data gradient).  psum/psq non-null:
 * per-block partial sums of z, z^2 as [pn2_dwconv_blocks(dt, N, H, W, C, K, 0)][C] rows for pn2_bn_finalize */
int pn2_dwconv(int dt, const void* x, const float* w, void* z, int N, int H, int W, int C, int K, int flip, int accumulate, float* psum, float* psq, void* stream);
/* partial[pn2_dwconv_blocks(.., 1)][C*K*K] of the depth-wise weight gradient; finish with pn2_colsum_finalize(partial, nblk, C*K*K, C*K*K, dW, acc) */
int pn2_dwconv_wgrad(int dt, const void* dz, const void* x, float* partial, int N, int H, int W, int C, int K, void* stream);
int pn2_pairconv_blocks(int dt, int N, int H, int W, int F);   /* rows of the BN partial buffers of _fwd and of the partial buffer of _wgrad */
/* grouped 3x3 conv, groups = F, 2 input channels per group (LGAG.W_g / W_x), pad 1, bias-free here (the bias is folded by the caller):
 * x [M][2F] -> z [M][F] + BN partial rows [pn2_pairconv_blocks(dt, N, H, W, F)][F] ; w [F][2][9] fp32 */
int pn2_pairconv3x3_fwd(int dt, const void* x, const float* w, void* z, int N, int H, int W, int F, float* psum, float* psq, void* stream);
int pn2_pairconv3x3_dgrad(int dt, const void* dz, const float* w, void* dx, int N, int H, int W, int F, int accumulate, void* stream);
/* partial[pn2_pairconv_blocks(dt, N, H, W, F)][F*18] ; finish with pn2_colsum_finalize */
int pn2_pairconv3x3_wgrad(int dt, const void* dz, const void* x, float* partial, int N, int H, int W, int F, void* stream);
/* y (+)= x * gate ; mode 0: gate [N][C] (CAB), mode 1: gate [N][HW] (SAB, LGAG) ; gate fp32.  Also the data gradient (x := dy). */
int pn2_gate_mul(int dt, const void* x, const float* gate, void* y, int N, int HW, int C, int mode, int accumulate, void* stream);
/* gate gradient.  mode 1: dgate [N][HW] = sum_c dy*x, written directly.  mode 0: partial [pn2_gate_blocks(dt, HW, C)][N*C] rows, finish with
 * pn2_colsum_finalize(partial, nblk, N*C, N*C, dgate, acc) */
int pn2_gate_blocks(int dt, int HW, int C);
int pn2_gate_bwd(int dt, const void* dy, const void* x, float* dgate_or_partial, int N, int HW, int C, int mode, void* stream);
/* nn.AdaptiveAvgPool2d(1) and nn.AdaptiveMaxPool2d(1) in one pass (CAB): avg, mx [N][C] in the compute dtype, arg [N][C] = argmax pixel */
int pn2_global_pool(int dt, const void* x, void* avg, void* mx, int* arg, int N, int HW, int C, void* stream);
int pn2_global_pool_bwd(int dt, const void* davg, const void* dmax, const int* arg, void* dx, int N, int HW, int C, int accumulate, void* stream);
/* SAB input: out [NP][8] = (mean over channels, max over channels, 0 x 6), arg [NP] = argmax channel */
int pn2_chan_stats(int dt, const void* x, void* out8, int* arg, long long NP, int C, void* stream);
int pn2_chan_stats_bwd(int dt, const void* dout8, const int* arg, void* dx, long long NP, int C, int accumulate, void* stream);
/* nn.Upsample(scale_factor=2) (nearest) and its adjoint */
int pn2_upsample_nearest2x(int dt, const void* x, void* y, int N, int H, int W, int C, void* stream);
int pn2_upsample_nearest2x_bwd(int dt, const void* dy, void* dx, int N, int H, int W, int C, int accumulate, void* stream);
/* y[p][c] = a[p][perm[c]] + b[p][perm[c]] + c[p][perm[c]]   (b, c optional): the MSDC branch sum written through channel_shuffle, and (with the
 * inverse permutation, b = c = null) its adjoint */
int pn2_gather_sum(int dt, const void* a, const void* b, const void* c, const int* perm, void* y, long long M, int C, void* stream);
/* y = sigmoid(x) as fp32 [n] from a [rows][ld] map with C used channels; dx (+)= dy * y * (1 - y) */
int pn2_sigmoid(int dt_in, const void* x, int ld, int C, float* y, long long n, void* stream);
int pn2_sigmoid_bwd(int dt_out, const float* dy, const float* y, void* dx, int ld, int C, long long n, int accumulate, void* stream);

/* EMCAD/trainer.py:106-140 ("mutation" supervision, dual): sum over the 15 non-empty subsets s of the 4 scales of
 *   lc1 * CrossEntropy(sum_{i in s} fg_i, label) + lc2 * DiceLoss(softmax(sum fg_i), label) (utils/utils.py:102-138) + lc3 * BCEWithLogits(sum_{i in s} bg_i, bg_mask)
 * in ONE pass over the 8 maps.  fg[4] / bg[4] (host arrays of device pointers): [N][H][W][K] fp32, K = 9; label [N][H][W] int64; bg_mask [N][K][H][W] fp32.
 * partial [pn2_mutation_loss_blocks(N*HW)][pn2_mutation_loss_width(K)] scratch; sums [width] kept for the backward; loss[1].
 * Backward: dfg[i] / dbg[i] = gscale * d loss / d map, same layout (written, not accumulated). */
int pn2_mutation_loss_blocks(long long npix);
int pn2_mutation_loss_width(int K);
int pn2_mutation_loss_fwd(const float* const* fg, const float* const* bg, const long long* label, const float* bg_mask, int N, long long HW, int K,
                          float lc1, float lc2, float lc3, float* partial, float* sums, float* loss, void* stream);
int pn2_mutation_loss_bwd(const float* const* fg, const float* const* bg, float* const* dfg, float* const* dbg, const long long* label, const float* bg_mask,
                          int N, long long HW, int K, float lc1, float lc2, float lc3, const float* sums, float gscale, void* stream);

/* ---------------------------------------------------------------------------------------------- element-wise / layout */
int pn2_binary(int dt, int op /*0 add,1 mul*/, const void* a, int ld_a, const void* b, int ld_b, void* out, int ld_out, int M, int C, int accumulate, void* stream);
/* backward of out = a * b in one pass (autograd of the aggregation's products x2_1 = conv_upsample1(up(x1)) * x2 ..., pranet.py:111-119): ga (+)= g * b, gb (+)= g * a -
 * the bits two pn2_binary launches leave; ga != gb */
int pn2_mul_bwd(int dt, const void* g, int ld_g, const void* a, int ld_a, const void* b, int ld_b, void* ga, int ld_ga, int acc_a, void* gb, int ld_gb, int acc_b,
                int M, int C, void* stream);
int pn2_copy(int dt_in, const void* src, int ld_s, int dt_out, void* dst, int ld_d, int M, int C, int accumulate, void* stream);
/* Many same-dtype copies (16-byte aligned rows) in ONE launch from a DEVICE job table - the pass-through slices of independent chains at one lock-step position
 * (pranet.py:77-79: branch0 of the three RFB modules into their concat buffers; the same slices of the gradient in the backward pass).  Bit-identical to pn2_copy per job. */
typedef struct pn2_copy_job { const void* src; void* dst; int ld_s, ld_d, M, C, accumulate, pad_; } pn2_copy_job;
int pn2_copy_job_blocks(int dt, const pn2_copy_job* j);
int pn2_copy_multi(int dt, const pn2_copy_job* jobs_dev, const int* block_start_dev, int njobs, int total_blocks, void* stream);
int pn2_nchw_to_nhwc(int dt_out, const float* x, void* y, int ld_y, int N, int C, int HW, int Cp, void* stream);   /* pad channels zeroed */
int pn2_bias_grad(const float* dy, int M, int K, float* db, int accumulate, void* stream);   /* db[k] = sum_m dy[m][k] (fp32 head maps) */

/* ---------------------------------------------------------------------------------------------- optimiser
 * clip_gradient (utils/utils.py:7-17: per-element clamp) + torch.optim.Adam step (MyTrain_med.py:149,85-86), one launch over
 * the flat parameter arena.  step_ptr: device int64 step counter incremented by the kernel launch before (graph friendly). */
int pn2_clamp_adam(float* param, float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1, float beta2,
                   float eps, float clip, float grad_scale, const float* bias_corr /* device [8]: 1-b1^t, 1-b2^t, b1^t, b2^t (pn2_adam_tick), then lr, clip,
                                    weight_decay and a flag: flag != 0 -> the kernel takes these three from the device (LR schedules under hipGraph replay) */,
                   float weight_decay /* decoupled (torch.optim.AdamW, EMCAD/trainer.py:75): p *= 1 - lr*wd ; 0 = Adam */, void* stream);
int pn2_adam_tick(float* bias_corr /* updates [0..3]: bc1, bc2, b1^t, b2^t */, float beta1, float beta2, void* stream);

/* MyTest_med.py:104-111 tail on device: sum of 4 maps already resized -> sigmoid -> min-max -> uint8 */
int pn2_eval_tail(const float* res, unsigned char* out, float* minmax /* scratch [2 + 2*512] */, long long n, void* stream);
/* MyTrain_med.py:163-164 / eval.py:22-50 threshold sweep without a numpy round trip of the maps: hist[0..255] = pixel counts per
 * uint8 prediction value, hist[256..511] = the same restricted to gt > 0.5 (integer atomics: deterministic).  All 256-threshold
 * metrics of Fmeasure_calu (eval_functions.py:131-166) and the MAE follow from these counts (pn2/evaltail.py).                    */
int pn2_eval_hist(const unsigned char* pred_u8, const float* gt, long long n, unsigned* hist, void* stream);
/* The remaining metrics of eval_for_testAllInOne (binary_seg/eval.py:18-66), finished on the host in float64 with the reference's expressions (pn2/evaltail.py):
 * Sm (StructureMeasure, utils/eval_functions.py:5-94): out25[0..2] = sum of rows, sum of columns, count of the foreground (gt > 0.5) pixels; out25[3 + q*5 ..] =
 *   {pixels, sum k, sum k^2, sum g, sum k*g} of centroid quadrant q = (row >= X) + 2*(col >= Y) (k = prediction byte, g in {0,1}); out25[23], [24] = X, Y.
 *   All integers: exact.  S_Object comes from the pn2_eval_hist histograms.
 * wFm (original_WFb, :96-129): exact Euclidean feature transform with scipy.ndimage.distance_transform_edt's order of preference among equidistant pixels,
 *   7x7 Gaussian (K49, row-major, computed by the caller as fspecial_gauss(7, 5)) of the propagated error with edge replication, c5 = log(0.5)/5;
 *   part[pn2_eval_wfm_blocks(H, W)][2] = per-block sums of Ew over the foreground / background (fixed order).  work_i: H*W ints, work_d: 2*H*W doubles.
 * meanEm (EnhancedMeasure, :168-192) needs no kernel: per threshold it is a function of the pn2_eval_hist histograms. */
int pn2_eval_region_sums(const unsigned char* pred_u8, const float* gt, int H, int W, unsigned long long* out25, void* stream);
int pn2_eval_wfm_blocks(int H, int W);
int pn2_eval_wfm(const unsigned char* pred_u8, const float* gt, int H, int W, const double* K49, double c5, int* work_i, double* work_d, double* part, void* stream);

/* ---------------------------------------------------------------------------------------------- input transform
 * binary_seg/utils/dataloader.py:104-111 (PolypDataset) / :176-181 (test_dataset): transforms.Resize((S, S)) on the decoded PIL image
 * (= PIL.Image.resize(BILINEAR): separable antialiased triangle filter, uint8 after each pass, 22-bit fixed-point taps; Pillow Resample.c),
 * transforms.ToTensor() and transforms.Normalize(mean, std), on device uint8 HWC images.  Bit-exact with Pillow: pn2_resize_coeffs is HOST
 * code computing the taps of one axis in double exactly as Pillow does; copy them to the device and run the width pass, then the height pass. */
int pn2_resize_ksize(int in_size, int out_size);
int pn2_resize_coeffs(int in_size, int out_size, int* xmin, int* count, int* kk /* [out_size][pn2_resize_ksize] */);
int pn2_resize_u8_pass(const unsigned char* src, unsigned char* dst, int H, int W, int C, int out_size, int axis /* 1: width, 0: height */,
                       const int* xmin_dev, const int* count_dev, const int* kk_dev, int ksize, void* stream);
int pn2_u8_to_tensor(const unsigned char* src, float* dst_chw, int H, int W, int C, const float* mean_dev, const float* std_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif
