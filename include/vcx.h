/*
 * vcx.h — C ABI of libvcx.so, the MI355X (gfx950) kernel library behind the
 * ViewCrafter DDIM denoising hot path.
 *
 * The reference (Drexubery/ViewCrafter) is pure PyTorch on this path; it has no FFI of
 * its own.  Each entry point below names the reference call site(s) (file:line under the
 * reference tree) whose arithmetic it replaces.  A maintainer binds these with ctypes
 * (see INTEGRATION.md for the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - activations are fp16, channels-last: [B, T, H, W, C] (C contiguous). Spatial ops see
 *     it as [(B T), H, W, C], token ops as [(B T H W), C];
 *   - weights are fp16 [N_out][K] row-major with K ordered (ky, kx, cin) for convolutions;
 *   - bias / per-image add vectors / norm affine parameters / statistics are fp32;
 *   - all functions are stream-ordered on `stream` (a hipStream_t passed as void*), never
 *     synchronise, never allocate, and are capturable into a hipGraph;
 *   - return 0 on success, a negative VCX_E* code on failure; vcx_last_error() gives text.
 */
#ifndef VCX_H
#define VCX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VCX_OK 0
#define VCX_EINVAL (-1)   /* bad argument / unsupported shape            */
#define VCX_ELAUNCH (-2)  /* HIP launch or runtime error                 */
#define VCX_ENODEV (-3)   /* no gfx950 device                            */

/* 2: groupnorm stats are (mean, biased variance); vcx_tune_*.  3: vcx_gemm_desc grows ln_stats / ln_colsum (VCX_GEMM_LNFOLD*),
 * vcx_rowstats_f16.  4: colstats (VCX_GEMM_COLSTATS), vcx_groupnorm_stats_from_colstats_f32.  5: vcx_gemm_desc starts with
 * struct_size - a descriptor of another layout is rejected instead of read past its end; ldcs; vcx_clip_preprocess_f32,
 * vcx_add_nchw_f32_to_nhwc_f16.  6: vcx_ddim_ws_bytes, ws_bytes argument of the DDIM steps.  7: vcx_groupnorm_fold_linear_f16,
 * vcx_gemm_units_f16, vcx_attn_flash_d512_f16.  8: vcx_gemm_desc grows rowstats / rowstats_eps (VCX_GEMM_ROWSTATS) and
 * tail_a0 / tail_a1 (K tail of a convolution from linear sources); vcx_groupnorm_apply2_f16. */
#define VCX_ABI_VERSION 9

int vcx_abi_version(void);
const char* vcx_last_error(void);
/* Fills name[0..len) with the gcnArchName of the current device. */
int vcx_device_arch(char* name_host, int len);

/* ------------------------------------------------------------------------------------
 * GEMM / implicit-GEMM convolution engine (MFMA f16 -> f32 accumulate).
 *
 *   out[m, n] = epilogue( alpha * sum_k X[m, k] * W[n, k] )
 *
 * X rows are gathered either linearly (mode 0: X[m, k] = A[m*lda + k]) or as an im2col
 * view of a channels-last image (mode 1), so that one kernel serves
 *   nn.Linear            lvdm/modules/attention.py:53-57,269,290,336,362,418,438
 *   nn.Conv2d 3x3 / 1x1  lvdm/modules/networks/openaimodel3d.py:143-147,174-186,69-71,93-106
 *                        lvdm/modules/networks/ae_modules.py:33-52,99-106,118-127,162-197
 *   nn.Conv3d (3,1,1)    lvdm/modules/networks/openaimodel3d.py:255-266
 *   nn.Conv1d k=1        lvdm/modules/attention.py:332-334 (init_attn proj)
 * Mode 1 geometry: image [n_img, in_h, in_w, cin] with pixel stride lda (elements);
 * K = kh*kw*cin ordered (ky, kx, c) - or, with VCX_GEMM_CONV_SLABK (cin % 64 == 0), in
 * slabs of 64 input channels, (c / 64, ky, kx, c % 64): all taps of one channel slab are
 * consecutive K-steps, so the tap re-reads of a tile's input hit L2; output pixel (oy, ox)
 * reads input pixel
 * ((oy*stride + ky - pad_h) >> ups, (ox*stride + kx - pad_w) >> ups), zero outside
 * [0, in_h<<ups) x [0, in_w<<ups)  (ups=1 fuses F.interpolate(scale_factor=2,'nearest'),
 * openaimodel3d.py:100-103, ae_modules.py:124).  A temporal (3,1,1) convolution is the
 * same gather with in_h = T, in_w = H*W, kh = 3, kw = 1, pad_h = 1, pad_w = 0.
 * ---------------------------------------------------------------------------------- */
#define VCX_GEMM_BIAS_N 0x1     /* + bias[n]                                              */
#define VCX_GEMM_BIAS_M 0x2     /* + bias[m]  (used for the transposed V projection)     */
#define VCX_GEMM_ROWADD 0x4     /* + rowadd[(m / rowadd_div) * rowadd_ld + n]  (ResBlock emb add,
                                   openaimodel3d.py:216-226)                              */
#define VCX_GEMM_RESIDUAL 0x8   /* + residual[m*ldr + n]                                  */
#define VCX_GEMM_GEGLU 0x10     /* out[m, j] = x * gelu_erf(gate), attention.py:415-422;
                                   W/bias rows pre-interleaved in blocks of 32: rows
                                   [64b, 64b+32) are x for output cols [32b, 32b+32),
                                   rows [64b+32, 64b+64) their gates; out has N/2 columns */
#define VCX_GEMM_OUT_F32 0x20   /* store fp32 instead of fp16                             */
#define VCX_GEMM_CONV_SLABK 0x40 /* mode 1: W rows are ordered (c / 64, ky, kx, c % 64)     */
/* LayerNorm folded into the projection that consumes it (nn.LayerNorm -> nn.Linear pairs of BasicTransformerBlock,
 * attention.py:226-246): with W' = gamma o W (fp16), colsum = sum_k W'[., k] (fp32, of the ROUNDED W') and
 * bias' = bias + W beta, LN(x) W^T + bias = rstd (x W'^T - mean colsum) + bias' exactly; the GEMM reads the un-normalised
 * rows and the epilogue applies out = alpha rstd (acc - mean colsum) + bias'.  ln_stats = (mean, rstd) pairs from
 * vcx_rowstats_f16.  Linear mode, K % 64 == 0 only.
 *   VCX_GEMM_LNFOLD    X rows are the normalised operand: ln_stats[m], ln_colsum[n], bias'[n] (BIAS_N); with GEGLU too
 *   VCX_GEMM_LNFOLD_T  W rows are the normalised operand (the transposed V projection out[d, token]): ln_stats[n],
 *                      ln_colsum[m], bias'[m] (BIAS_M)                                                                   */
#define VCX_GEMM_LNFOLD 0x80
#define VCX_GEMM_LNFOLD_T 0x100
/* GroupNorm statistics from the producing layer: besides out, the epilogue writes colstats[M / 64][ldcs][2] fp32 - for every
 * 64-row strip of output rows and every output column the (mean, M2 = sum of squared deviations) of the fp16-rounded outputs,
 * accumulated around a per-strip shift (robust to |mean| >> std).  vcx_groupnorm_stats_from_colstats_f32 merges them into the
 * (mean, variance) pairs vcx_groupnorm_apply_f16 takes, so the GroupNorm behind the layer needs no statistics pass over the
 * tensor: ResBlock out_layers and the norms of TemporalConvBlock behind their convolutions (openaimodel3d.py:174-186,255-266),
 * and - round 4, linear mode - the norms fed by a transformer's proj_out + residual (attention.py:290,362 -> the next block's
 * GroupNorm) or by a concatenated skip (openaimodel3d.py:596: the two producers write their columns of ONE moment buffer,
 * ldcs = C1 + C2).  DMA kernel only (K % 64 == 0 / cin % 64 == 0), fp16 output, M % 64 == 0, no GEGLU, no LNFOLD. */
#define VCX_GEMM_COLSTATS 0x200
/* LayerNorm statistics from the producing layer (round 6): besides out, the layer writes rowstats[m] = (mean, rstd = 1 / sqrt(var +
 * rowstats_eps)) of the fp16-ROUNDED output row m over its N columns - what vcx_rowstats_f16 computes from the stored tensor, so the
 * LayerNorm-folded projection behind it (VCX_GEMM_LNFOLD: nn.LayerNorm -> nn.Linear of BasicTransformerBlock, attention.py:226-246)
 * needs no statistics pass.  Sums on the matrix pipe (fp32 accumulation of exact fp16 products): the row sum as ones x X, the sum of
 * squares as the diagonal of D^T D with D = X minus a per-strip shift - 0, or one of the row's values where the strip is offset
 * (|value| > 4 x the spread of eight samples: X - shift is then exact in fp16) - per 80-column strip; the four strips of a row
 * merged Chan-style in a fixed order: robust to |mean| >> std, bit-reproducible, independent of M; agrees with vcx_rowstats_f16
 * to fp32 rounding of the sums (mean 2e-6 of the row's magnitude, rstd 6e-5 relative), not bit for bit (another summation order).  Only where ONE block owns whole rows:
 * the weight-stationary kernel, linear mode, N = K = 320, M >= 8192, BIAS_N / RESIDUAL at most, fp16 output (the attention output
 * projections and proj_in of the C = 320 level); with vcx_gemm_units_f16 too.  Any other shape: VCX_EINVAL. */
#define VCX_GEMM_ROWSTATS 0x400

typedef struct vcx_gemm_desc {
    size_t struct_size;   /* = sizeof(vcx_gemm_desc) of the header the caller was compiled against; anything else is VCX_EINVAL */
    const void* A;        /* fp16 activations                                             */
    const void* W;        /* fp16 weights [N][ldw]                                        */
    void* C;              /* fp16 (or fp32) output [M][ldc]                               */
    const float* bias;    /* fp32 [N] or [M]                                              */
    const float* rowadd;  /* fp32 [ceil(M/rowadd_div)][rowadd_ld >= N]                    */
    const void* residual; /* fp16 [M][ldr]                                                */
    int64_t lda;          /* linear: row stride; conv: pixel stride (elements)            */
    int32_t M, N, K;
    int32_t ldw, ldc, ldr;
    int32_t mode;         /* 0 linear, 1 conv gather                                      */
    int32_t in_h, in_w, out_h, out_w, cin, kh, kw, stride, pad_h, pad_w, ups;
    int32_t rowadd_div;
    int32_t flags;
    float alpha;
    const float* ln_stats;  /* VCX_GEMM_LNFOLD[_T]: fp32 (mean, rstd) per normalised row            */
    const float* ln_colsum; /* VCX_GEMM_LNFOLD[_T]: fp32 row sums of the folded weight              */
    float* colstats;        /* VCX_GEMM_COLSTATS: out, fp32 [M / 64][ldcs][2] (first of this call's N columns) */
    int64_t ldcs;           /* columns between consecutive strips of colstats; 0 = N                */
    float* rowstats;        /* VCX_GEMM_ROWSTATS: out, fp32 [M][2] = (mean, rstd) of every output row, 8-byte aligned */
    float rowstats_eps;     /* VCX_GEMM_ROWSTATS: the LayerNorm's eps                                */
    int32_t rowadd_ld;      /* VCX_GEMM_ROWADD: elements between consecutive rows of rowadd; 0 = N (round 6: the addends of all
                             * ResBlocks come from ONE projection of the embedding, each block reads its columns of that matrix) */
    /* K tail of a convolution (mode 1, round 6): K = kh*kw*cin + tail_k0 + tail_k1, and the last tail_k0 + tail_k1 columns of every W
     * row multiply, for output row m, the first tail_k0 elements of row m of tail_a0 and then the first tail_k1 of row m of tail_a1
     * (fp16 [M][tail_lda*], row-for-row like a linear layer).  A ResBlock's 1x1 skip convolution folded into its second 3x3
     * convolution - `return self.skip_connection(x) + h`, openaimodel3d.py:228-235 - with x in one piece or, on the up path, as the
     * two halves of `torch.cat([h, hs.pop()], dim=1)` (:596) that are then never concatenated.  tail_k* % 64 == 0, DMA kernel only
     * (cin % 64 == 0, 32-bit extents), fp16 output, no GEGLU.  0 / null: none. */
    const void* tail_a0;
    const void* tail_a1;
    int64_t tail_lda0;
    int64_t tail_lda1;
    int32_t tail_k0;
    int32_t tail_k1;
} vcx_gemm_desc;

int vcx_gemm_f16(const vcx_gemm_desc* desc_host, void* stream);
/* The same linear layer with ONE weight / bias set per `unit_rows` consecutive rows: rows [u unit_rows, (u + 1) unit_rows) use
 * W + u w_unit_stride and bias + u bias_unit_stride (element strides) - the (Wn, bn) sets of vcx_groupnorm_fold_linear_f16, one per
 * frame (SpatialTransformer.norm -> proj_in, attention.py:265-269,299) or per video (TemporalTransformer, attention.py:331-336,369-372).
 * Linear mode, VCX_GEMM_BIAS_N at most, M a whole number of units.  N = K = 320 with unit_rows % 32 == 0 runs as ONE launch of the
 * weight-stationary kernel (a block keeps its unit's weights in registers); every other shape unit by unit through vcx_gemm_f16. */
int vcx_gemm_units_f16(const vcx_gemm_desc* desc_host, int unit_rows, int64_t w_unit_stride, int64_t bias_unit_stride, void* stream);

/* ------------------------------------------------------------------------------------
 * GroupNorm(32 groups) on channels-last fp16 with fp32 statistics.
 *   per-frame   GroupNormSpecific lvdm/basics.py:76-81 (ResBlock in/out layers, UNet out),
 *               SpatialTransformer.norm attention.py:265, VAE Normalize ae_modules.py:15-16
 *   per-video   TemporalConvBlock GN openaimodel3d.py:256-266, TemporalTransformer.norm
 *               attention.py:331   (statistics span T*H*W)
 * x is [n_outer][pixels][C]; statistics are taken over (pixels, C/groups) per (n, group).
 * stats: fp32 [n_outer][groups][2] = (mean, biased variance); written by _stats, which reduces
 * Welford/Chan-style ((count, mean, M2) partials merged in a fixed order: robust to |mean| >> std
 * like torch's GroupNorm, deterministic, no atomics) through the caller's workspace `ws` of
 * vcx_groupnorm_ws_bytes() bytes; consumed by _apply which computes
 *   y = (x - mean) * rsqrt(var + eps) * gamma[c] + beta[c], then x*sigmoid(x) if silu.
 * ---------------------------------------------------------------------------------- */
size_t vcx_groupnorm_ws_bytes(int n_outer, int64_t pixels, int groups);
int vcx_groupnorm_stats_f16(const void* x, float* stats, void* ws, int n_outer, int64_t pixels,
                            int C, int groups, void* stream);
int vcx_groupnorm_apply_f16(const void* x, void* y, const float* stats, const float* gamma,
                            const float* beta, int n_outer, int64_t pixels, int C, int groups,
                            float eps, int silu, void* stream);
/* The same pass over a channel concat that is never materialised (round 6): channels [0, c1) of a pixel are read from x1
 * [n_outer][pixels][c1], channels [c1, C) from x2 [n_outer][pixels][C - c1]; y [n_outer][pixels][C] as above.  The in_layers norm of
 * an up-path ResBlock over `torch.cat([h, hs.pop()], dim=1)` (openaimodel3d.py:596 -> :174-186).  c1 % 8 == 0. */
int vcx_groupnorm_apply2_f16(const void* x1, int c1, const void* x2, void* y, const float* stats, const float* gamma, const float* beta,
                             int n_outer, int64_t pixels, int C, int groups, float eps, int silu, void* stream);
/* stats[n_outer][groups][2] from the column moments a VCX_GEMM_COLSTATS convolution wrote (colstats[n_outer * pixels / 64][C][2];
 * pixels % 64 == 0): the strips and then the columns of a group are merged Chan-style in a fixed order (bit-reproducible,
 * independent of the batch size); ws as for vcx_groupnorm_stats_f16 (vcx_groupnorm_ws_bytes). */
int vcx_groupnorm_stats_from_colstats_f32(const float* colstats, float* stats, void* ws, int n_outer, int64_t pixels, int C,
                                          int groups, void* stream);

/* GroupNorm folded into the nn.Linear / 1x1 Conv1d behind it - TemporalTransformer.norm -> proj_in (attention.py:331-336,369-372; the
 * same pair in SpatialTransformer, attention.py:265-269,299): no SiLU sits between them, so GroupNorm-apply is an affine map per
 * (statistics unit n, channel) and  Linear(GroupNorm(x_n)) = x_n Wn[n]^T + bn[n]  with
 *   Wn[n][o][c] = fp16(W[o][c] gamma[c] rstd[n, g(c)]),   bn[n][o] = bias[o] + sum_c (W[o][c] beta[c] - float(Wn[n][o][c]) mean[n, g(c)])
 * (the mean term on the ROUNDED weight: a common offset of a group's channels cancels exactly).  The caller then runs vcx_gemm_f16 on
 * the un-normalised rows of unit n with (Wn[n], bn[n]): the normalised copy of the tensor is neither written nor re-read.
 * W fp32 [N][C] (master weights), bias fp32 [N] or NULL, stats [n_outer][groups][2] = (mean, variance) as vcx_groupnorm_apply_f16
 * takes them; out: Wn fp16 [n_outer][N][C], bn fp32 [n_outer][N].  C % 4 == 0. */
int vcx_groupnorm_fold_linear_f16(const float* W, const float* bias, const float* gamma, const float* beta, const float* stats,
                                  void* Wn, float* bn, int n_outer, int N, int C, int groups, float eps, void* stream);

/* LayerNorm over the last dim (nn.LayerNorm, attention.py:226-228), fp32 statistics. */
/* (mean, rstd) of every row, stats[rows][2] fp32: the read-only half of LayerNorm in front of a VCX_GEMM_LNFOLD projection
 * (same summation order as vcx_layernorm_f16: the statistics are bit-identical to the ones it normalises with). */
int vcx_rowstats_f16(const void* x, float* stats, int64_t rows, int C, float eps, void* stream);
int vcx_layernorm_f16(const void* x, void* y, const float* gamma, const float* beta,
                      int64_t rows, int C, float eps, void* stream);

/* ------------------------------------------------------------------------------------
 * Flash attention, head dim 64, no mask:  O = softmax(scale * Q K^T) V
 *   spatial self-attention / text and image cross-attention
 *   (CrossAttention.forward / efficient_forward, lvdm/modules/attention.py:81-209)
 * Problem p = (group g, head h): g in [0, n_groups), h in [0, heads).
 *   Q rows   : q  + (g*nq)*ldq + h*64          nq rows, row stride ldq
 *   K rows   : k  + ((g / kv_div) * kv_rows)*ldk + h*64      nk valid rows
 *   V^T rows : vt + (h*64)*ldvt + (g / kv_div)*kv_rows       64 rows (d) x nk cols (keys)
 *   O rows   : o  + (g*nq)*ldo + h*64
 * kv_rows is the row count between consecutive K/V groups (>= nk, multiple of 8).
 * flags: VCX_ATTN_ACCUMULATE adds into O (second softmax of the text (+) image
 * cross-attention, attention.py:129-142); VCX_ATTN_LOG2_LOGITS says the caller folded
 * scale * log2(e) into Q and/or K (e.g. as the alpha of the projection GEMM), so that
 * Q K^T is already the base-2 logit - `scale` is then ignored and the kernel saves one
 * multiply-add per score.
 * Two kernels sit behind this entry point: with VCX_ATTN_LOG2_LOGITS, nk a multiple of 64,
 * nk >= 4096 and no VCX_ATTN_ACCUMULATE the software-pipelined one-wave-per-SIMD kernel
 * (csrc/attention_v2.hip), otherwise the phased kernel (csrc/attention.hip); same contract,
 * results agree to fp16 rounding (different summation order).  Knob VCX_TUNE_FLASH_IMPL.
 * ---------------------------------------------------------------------------------- */
#define VCX_ATTN_ACCUMULATE 1
#define VCX_ATTN_LOG2_LOGITS 2
int vcx_attn_flash_d64_f16(const void* q, const void* k, const void* vt, void* o, int n_groups,
                           int heads, int nq, int nk, int kv_rows, int kv_div, int64_t ldq,
                           int64_t ldk, int64_t ldvt, int64_t ldo, float scale, int flags,
                           void* stream);

/* Two key/value sets in one pass:  O = softmax(scale Q K1^T) V1 + softmax(scale Q K2^T) V2
 * - the text (+) image cross-attention of attention.py:129-142 (image_cross_attention_scale
 * = 1) without reading Q twice and without writing O, reading it back and writing it again.
 * Same addressing as above per set (kv_rows / kv_div / ldk / ldvt / nk with suffix 1, 2);
 * the two partial results are added in fp32 and rounded once.  flags: VCX_ATTN_LOG2_LOGITS. */
int vcx_attn_flash_dual_d64_f16(const void* q, const void* k1, const void* vt1, const void* k2,
                                const void* vt2, void* o, int n_groups, int heads, int nq,
                                int nk1, int kv_rows1, int kv_div1, int64_t ldk1, int64_t ldvt1,
                                int nk2, int kv_rows2, int kv_div2, int64_t ldk2, int64_t ldvt2,
                                int64_t ldq, int64_t ldo, float scale, int flags, void* stream);

/* Flash attention with ONE head of dim 512, no mask: O = softmax(scale * Q K^T) V - the AttnBlock of the VAE
 * (lvdm/modules/networks/ae_modules.py:26-78 at 9216 tokens per 576x1024 frame); the score matrix is never materialised.
 * Group g (a frame): Q rows q + (g*nq)*ldq (512 columns), K rows k + (g*kv_rows)*ldk (nk valid), V^T rows vt + d*ldvt + g*kv_rows
 * (512 rows d x nk columns), O rows o + (g*nq)*ldo.  kv_rows >= nk, multiple of 8 (rows [nk, kv_rows) must be readable). */
int vcx_attn_flash_d512_f16(const void* q, const void* k, const void* vt, void* o, int n_groups, int nq, int nk, int kv_rows,
                            int64_t ldq, int64_t ldk, int64_t ldvt, int64_t ldo, float scale, void* stream);

/* Temporal self-attention over T <= 64 frames per pixel (one 32 x 32 score tile per (pixel, head) up to 32 frames, 2 x 2 tiles beyond), head dim 64
 * (TemporalTransformer -> CrossAttention.forward, attention.py:365-412, 81-126).
 * qkv is [(b t p)][ld] with q at col 0, k at col k_off, v at col v_off (+ h*64);
 * token (b, t, p) is row (b*T + t)*P + p.  o is [(b t p)][ldo]. */
int vcx_attn_temporal_d64_f16(const void* qkv, void* o, int B, int T, int64_t P, int heads,
                              int64_t ld, int k_off, int v_off, int64_t ldo, float scale,
                              void* stream);
/* ABI 9.  The same with flags: VCX_ATTN_CAUSAL masks the keys of later frames (frame t attends to frames <= t) -
 * TemporalTransformer(causal_attention=True): the lower-triangular mask of attention.py:343-345, 377-384 applied at :111-115
 * (`use_causal_attention`; not used by the ViewCrafter YAMLs).  flags = 0 is vcx_attn_temporal_d64_f16. */
#define VCX_ATTN_CAUSAL 4
int vcx_attn_temporal_d64_masked_f16(const void* qkv, void* o, int B, int T, int64_t P, int heads,
                                     int64_t ld, int k_off, int v_off, int64_t ldo, float scale,
                                     int flags, void* stream);
/* ABI 9.  ... with relative position (CrossAttention(relative_position=True), attention.py:20-40, 59-62, 104-108, 120-123; `use_relative_position`,
 * not used by the ViewCrafter YAMLs; T <= 32, 1 <= R <= 31):  logits = scale (q_t . k_s + q_t . Ek[c(s - t)]),  out_t = sum_s P[t, s] (v_s + Ev[c(s - t)]),
 * c(d) = clamp(d, -R, R) + R, tables of 2R + 1 <= 64 rows.  The two table contractions are 64-wide GEMMs of the CALLER:
 *   relg [(b t p)][heads][64] fp16 (in)  = q Ek^T per query row and head (slots 2R + 1 .. 63 unused) - added to the scores before scale and softmax;
 *   relp [(b t p)][heads][64] fp16 (out) = the probabilities of a query by clipped distance (keys beyond +-R summed into slots 0 / 2R); the caller
 *   ZEROES it before the call and adds relp Ev to o afterwards.  flags: VCX_ATTN_CAUSAL. */
int vcx_attn_temporal_d64_rel_f16(const void* qkv, void* o, const void* relg, void* relp, int R, int B, int T, int64_t P,
                                  int heads, int64_t ld, int k_off, int v_off, int64_t ldo, float scale, int flags,
                                  void* stream);

/* Row softmax in place on fp16 [rows][ld] over the first n columns (fp32 math): VAE AttnBlock, ae_modules.py:66-69.
 * ld % 8 == 0; when n is not a multiple of 8 the columns up to the next multiple of 8 (ld must cover them) are written as zeros. */
int vcx_softmax_rows_f16(void* x, int64_t rows, int n, int64_t ld, void* stream);

/* ------------------------------------------------------------------------------------
 * Element-wise / layout helpers
 * ---------------------------------------------------------------------------------- */
/* x*sigmoid(x) on fp32 -> fp32 (emb_layers SiLU, openaimodel3d.py:158-164). */
int vcx_silu_f32(const float* x, float* y, int64_t n, void* stream);
/* exact (erf) GELU on fp16 -> fp16, y may alias x (nn.GELU of the Resampler feed-forward,
 * lvdm/modules/encoders/resampler.py:27-34). */
int vcx_gelu_f16(const void* x, void* y, int64_t n, void* stream);
/* Image pre-processing of the OpenCLIP vision tower, lvdm/modules/encoders/condition.py:322-329:
 * kornia.geometry.resize(x, (size, size), 'bicubic', align_corners=True, antialias) -> (x + 1) / 2 -> (x - mean) / std.
 * x fp32 [B, C, H, W] in [-1, 1] (C <= 4), y fp32 [B, C, size, size]; mean_host / std_host: C host floats.  kornia's anti-aliasing
 * (Gaussian, sigma = (factor - 1) / 2 per axis, kernel int(max(4 sigma, 3)) made odd, mirror border; only when an axis shrinks)
 * and torch's bicubic (A = -0.75, clamped neighbours) are evaluated in one pass, the blurred image is never written. */
int vcx_clip_preprocess_f32(const float* x, float* y, int B, int C, int H, int W, int size, int antialias,
                            const float* mean_host, const float* std_host, void* stream);
/* sinusoidal embedding, lvdm/models/utils_diffusion.py:8-28: out[b] = [cos(t f) | sin(t f)] */
int vcx_timestep_embedding_f32(const int64_t* t, float* out, int B, int dim, float max_period,
                               void* stream);
/* fp32 -> fp16 / fp16 -> fp32 casts (n elements). */
int vcx_cast_f32_to_f16(const float* x, void* y, int64_t n, void* stream);
int vcx_cast_f16_to_f32(const void* x, float* y, int64_t n, void* stream);
/* strided 2-D copy of fp16: dst[r*ldd + c] = src[r*lds + c], c < cols (skip concat,
 * openaimodel3d.py:596). cols, ldd, lds multiples of 8. */
int vcx_copy2d_f16(const void* src, void* dst, int64_t rows, int cols, int64_t lds, int64_t ldd,
                   void* stream);
/* ABI 9.  2x2 average pool / nearest 2x of a channels-last fp16 image batch: y[n][H/2][W/2][C] = mean of the 2x2 window (fp32 sum, one
 * rounding; odd H / W drop the last row / column) and y[n][2H][2W][C] = x[n][y/2][x/2][C].  C % 8 == 0.  The sampling halves of the
 * reference's ResBlock(up / down) - `h_upd` / `x_upd`, openaimodel3d.py:160-165, 210-215 (`resblock_updown: true`) - and of Downsample /
 * Upsample without a convolution (`conv_resample: false`, openaimodel3d.py:70-72, 98-103); neither is used by the ViewCrafter YAMLs. */
int vcx_avgpool2x2_f16(const void* x, void* y, int n, int H, int W, int C, void* stream);
int vcx_upsample2x_f16(const void* x, void* y, int n, int H, int W, int C, void* stream);
/* h[n][p][c] += src[n][c][p] (h fp16 channels-last [n, HW, C], src fp32 [n, C, HW]): the adapter feature maps the reference adds
 * behind every third input block when `features_adapter` is given (openaimodel3d.py:582-585). */
int vcx_add_nchw_f32_to_nhwc_f16(const float* src, void* h, int n, int C, int64_t HW, void* stream);
/* fp32 [B, C, T, H, W] -> fp16 channels-last [B, T, H, W, ldc] at channel offset c_off
 * (DiffusionWrapper concat, ddpm3d.py:1437-1443; 'b c t h w -> (b t) c h w',
 * openaimodel3d.py:566); `scale` multiplies (VAE 1/scale_factor, ddpm3d.py:657-661). */
int vcx_ncthw_f32_to_nthwc_f16(const float* src, void* dst, int B, int C, int T, int64_t HW,
                               int ldc, int c_off, float scale, void* stream);
/* channels-last [B, T, HW, ldc] (fp16 or fp32 per src_f32) -> fp32 [B, C, T, H, W]
 * ('(b t) c h w -> b c t h w', openaimodel3d.py:601-602). */
int vcx_nthwc_to_ncthw_f32(const void* src, float* dst, int B, int C, int T, int64_t HW, int ldc,
                           int src_f32, void* stream);

/* ------------------------------------------------------------------------------------
 * DDIM step (v-prediction, classifier-free guidance, guidance rescale, dynamic rescale):
 * DDIMSampler.p_sample_ddim lvdm/models/samplers/ddim.py:208-281,
 * rescale_noise_cfg utils_diffusion.py:147-158, predict_*_from_z_and_v ddpm3d.py:239-251.
 * All tensors fp32 [B][n] (n = C*T*H*W).  coef_host[8] =
 *   { sqrt_acp_t, sqrt_1m_acp_t, a_prev, sigma_t, scale_ratio(prev/t), cfg_scale,
 *     guidance_rescale, parameterization_is_v }.
 * v_uncond may be NULL (no guidance).  noise may be NULL when sigma_t == 0.
 * ws / ws_bytes: device workspace (per-block fp64 partial sums, reduced in fixed order), 8-byte aligned, at least
 * vcx_ddim_ws_bytes(B, n) bytes - the callee checks the size it is told (ABI 6; never more than 8192 * B).
 * ---------------------------------------------------------------------------------- */
size_t vcx_ddim_ws_bytes(int B, int64_t n);
int vcx_ddim_step_f32(const float* x, const float* v_cond, const float* v_uncond,
                      const float* noise, float* x_prev, float* pred_x0, void* ws,
                      size_t ws_bytes, int B, int64_t n, const float* coef_host, void* stream);
/* Multi-condition guidance (lvdm/models/samplers/ddim_multiplecond.py:220-236, `--multiple_cond_cfg`): v_img is the
 * prediction under (empty text, image) conditioning and coef_host[8] = cfg_img:
 *   v = v_uncond + cfg_img (v_img - v_uncond) + cfg (v_cond - v_img).  v_img == NULL reduces to vcx_ddim_step_f32. */
int vcx_ddim_step3_f32(const float* x, const float* v_cond, const float* v_uncond,
                       const float* v_img, const float* noise, float* x_prev, float* pred_x0,
                       void* ws, size_t ws_bytes, int B, int64_t n, const float* coef_host,
                       void* stream);

/* ------------------------------------------------------------------------------------
 * Lightweight per-kernel-family profiling with HIP events (used by bench.py to fill
 * `roofline`): between begin/end every launch of a family is bracketed by an event pair on
 * its own stream.  family ids: 0 gemm/conv, 1 flash attention, 2 temporal attention,
 * 3 groupnorm, 4 layernorm, 5 elementwise.  out_host[6][4] = {launches, total_ms,
 * total_flops, total_bytes}.  vcx_profile_end synchronises the recorded events.
 * ---------------------------------------------------------------------------------- */
#define VCX_PROF_FAMILIES 6
int vcx_profile_begin(int max_records);
int vcx_profile_end(double* out_host);

/* ------------------------------------------------------------------------------------
 * Experiment knobs (debug / A-B tooling only; the defaults ARE the product).  A knob is a
 * process-wide integer read by the dispatchers at launch time; its initial value comes
 * from the environment variable VCX_TUNE_<NAME> read ONCE at the first use, never per
 * launch.  vcx_tune_set returns the previous value.  bench.py prints every knob that is
 * not at its default next to the numbers it measured.
 * ---------------------------------------------------------------------------------- */
#define VCX_TUNE_GEMM_CFG 0        /* -1 auto | 0..3 force tile config 128x128 / 128x160 / 256x256 / 256x320 */
#define VCX_TUNE_GEMM_DMA 1        /* 1 | 0 = register-staged kernel everywhere                              */
#define VCX_TUNE_FLASH_QB 2        /* 0 auto | 1 | 2 query blocks of 32 rows per wave (v1 kernel)            */
#define VCX_TUNE_XATTN_RESIDENT 3  /* 1 | 0 = never use the LDS-resident cross-attention kernel | 2 = its first form */
#define VCX_TUNE_FLASH_IMPL 4      /* 0 auto | 1 phased v1 kernel | 2 software-pipelined v2 kernel           */
#define VCX_TUNE_EXP0 5            /* free for one-off experiments (0)                                       */
#define VCX_TUNE_EXP1 6
#define VCX_TUNE_GEMM_WS 7         /* 1 | 0 = never use the weight-stationary kernels (K = 320) | 2 = not for the GEGLU projection | 3 = that one without its cross-XCD streams | 4 = not for the LayerNorm-folded projections | 5 = those for every N % 64 == 0 (tests) */
#define VCX_TUNE_COUNT 8
int vcx_tune_set(int knob, int value);
int vcx_tune_get(int knob);

#ifdef __cplusplus
}
#endif
#endif /* VCX_H */
