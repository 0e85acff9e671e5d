/* libpsld_hip — C ABI of the MI355X (gfx950) PSLD hot path.
 *
 * Drop-in boundary.  The reference's only native FFI is the pybind module
 * `upfirdn2d_op.upfirdn2d(...)` (main/models/score_fn/song_sde/op/upfirdn2d.cpp:12-22) and
 * `fused.fused_bias_act(...)` (op/fused_bias_act.cpp:11-20); everything else on the path is
 * eager ATen called from Python.  This header is what a maintainer of the reference binds
 * instead (ctypes stub in INTEGRATION.md): every entry point takes raw device pointers, sizes
 * and a hipStream_t, never allocates, never synchronises, is re-entrant per stream, and
 * returns 0 on success (non-zero -> psld_last_error() has the message; the Python side raises
 * RuntimeError, or ValueError for PSLD_ERR_NUMERIC to mirror psld.py:166-171).
 *
 * Activations inside the network are NHWC ("channels-last": [B][H][W][C], C contiguous);
 * the public tensors (x, eps, samples) stay NCHW like the reference and are converted at the
 * network boundary by psld_nchw_to_nhwc_f32 / psld_nhwc_to_nchw_f32.
 */
#ifndef PSLD_HIP_H
#define PSLD_HIP_H

#if defined(__HIP__) || defined(__HIPCC__)
#include <hip/hip_runtime_api.h>
#else
typedef struct ihipStream_t* hipStream_t; /* opaque: pass the raw HIP stream handle */
#endif

#ifdef __cplusplus
extern "C" {
#endif


#define PSLD_ABI_VERSION 14 /* 2: limb-MFMA convolutions, math mode, bias-gradient / batched-copy entry points; 3: pointwise weight gradient and batched activation GEMM on limb kernels; 4: GroupNorm statistics from the limb kernels' epilogue; 5: per-sample-time reverse SDE, ScoreLoss nll / l1; 6: limb-plane activations; 7: device-resident dropout seed / Adam scalars (captured training step), GroupNorm-backward sums from the producing epilogue; 8: Winograd F(2x2,3x3) limb convolution; 9: launch tape; 10: GroupNorm apply (+SiLU) fused into the Winograd convolution's input staging; 11: GroupNorm backward kernel selector; 12: column sums of dx from the GroupNorm backward; 13: GroupNorm backward returns per-image sums (dgamma / dbeta / bias gradients and split-K slab reductions of a whole backward pass in batched launches), the GroupNorm-backward by-product of the limb epilogue and the launch tape (9) removed; 14: the optimiser kernels take the device error word (a refused step is a no-op + NaN loss), Winograd-domain 3x3 weight gradient (psld_conv3x3_wgrad_wino_*) */
#define PSLD_COEFF_STRIDE 12

int psld_version(void);
const char* psld_last_error(void);

/* Arithmetic of the MFMA tile kernels (GEMM / conv / wgrad).  Both take and return fp32 and accumulate in fp32.
 *   PSLD_MATH_F32    v_mfma_f32_32x32x2_f32 on the operands as they are (one fmaf per k);
 *   PSLD_MATH_BF16X6 every operand is split exactly into three bf16 limbs (hi + mid + lo == x, all 24 mantissa
 *                    bits) and each product is the six leading limb products on v_mfma_f32_32x32x16_bf16; the three
 *                    dropped terms are < 2^-23 of the product, i.e. below fp32 rounding of the product itself.
 * Process-wide; the initial value comes from the environment variable PSLD_MATH ("f32" | "bf16x6"). */
#define PSLD_MATH_F32 0
#define PSLD_MATH_BF16X6 1
int psld_set_math_mode(int mode);
int psld_get_math_mode(void);

/* ---- fused epilogue of every MFMA tile kernel ------------------------------------------
 * value = ((alpha * acc + bias[n] + rowbias[m / rows_per_img][n] + residual[m][n]) * out_scale)
 *         (+ C[m][n] if accumulate)
 * Used for: conv bias (layers.py:103-109), the time-embedding add `h += Dense_0(act(temb))`
 * (layerspp.py:262-263), the residual `(x + h) / sqrt(2)` (layerspp.py:271-274, :88-91) and the
 * attention scale C^-1/2 (layerspp.py:82). */
typedef struct psld_epilogue {
    float alpha;
    const float* bias;
    const float* rowbias;
    int ld_rowbias;
    int rows_per_img;
    const float* residual;
    int ld_residual;
    long long residual_stride_batch;
    float out_scale;
    int accumulate;
    /* Optional (limb kernels psld_conv3x3_split_f32 / _limb_f32 / _wino_f32 / psld_gemm_split_f32 only; a call that splits
     * its K range over workgroups forms them in its reduction + epilogue pass instead: rows, N and gn_hw multiples of 64):
     * GroupNorm statistics of the OUTPUT as a by-product, for the GroupNorm that reads it next
     * (layerspp.py:258,264 GroupNorm_0/1; :77 of the attention block).  gn_part[((img*chunks + chunk)*(N/8) + f)*2 + {0,1}]
     * = sum / sum of squares of the 8 channels 8f..8f+7 over the chunk-th run of 64 rows of image img
     * (chunks = gn_hw / 64, gn_hw = rows per image, a multiple of 64; N a multiple of 128).  gn_fine = 4: sums of FOUR
     * channels 4f..4f+3 instead ((N/4) entries per run: tensors whose groups are 4 channels wide - 128 channels in 32
     * groups, CelebA-64's first level); 0 or 8: eight.
     * psld_gn_stats_from_partials_f32 turns them into the statistics of any group size that is a multiple of 8. */
    double* gn_part;
    int gn_hw;
    int gn_fine;
} psld_epilogue_t;

/* C[b] = epilogue(op(A[b]) * op(B[b])), fp32 MFMA (v_mfma_f32_32x32x2_f32), batched.
 * trans_a = 0: A is [M][K] (lda >= K); 1: A is [K][M].   trans_b = 0: B is [K][N]; 1: B is [N][K].
 * Replaces nn.Linear / NIN / 1x1 conv / attention einsums (layers.py:531-540, layerspp.py:82-87). */
int psld_gemm_f32(int trans_a, int trans_b, int M, int N, int K,
                  const float* A, int lda, long long stride_a,
                  const float* B, int ldb, long long stride_b,
                  float* C, int ldc, long long stride_c, int batch,
                  const psld_epilogue_t* epi, hipStream_t stream);

/* slabs[s] = A[ks:ke]^T B[ks:ke] for nsplit K-ranges (weight-gradient GEMMs with K = B*H*W). */
int psld_gemm_tn_splitk_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                            float* slabs, int nsplit, hipStream_t stream);

/* NHWC implicit-GEMM convolution; input = concat(x1[c1], x2[c2]) along channels (x2 may be
 * NULL with c2 = 0: the up-path `torch.cat([h, hs.pop()], dim=1)` of ncsnpp.py:374 is never
 * materialised).  Weights are [cout][kh][kw][c1+c2].
 * transposed_stride = 1: y[n,oy,ox,:] = sum x[n, oy*stride+ky-pad, ox*stride+kx-pad, :] w[:,ky,kx,:]
 * transposed_stride = s>1 (data-gradient of a stride-s conv): input index (oy+ky-pad)/s when divisible.
 * Replaces nn.Conv2d 3x3/1x1 (layers.py:85-109) and F.conv2d(stride=2) of
 * up_or_down_sampling.py:177. */
int psld_conv2d_nhwc_f32(const float* x1, int c1, const float* x2, int c2,
                         int batch, int ih, int iw,
                         const float* w_ohwi, int cout, int kh, int kw,
                         int stride, int pad, int transposed_stride,
                         int oh, int ow, float* y, int ldy,
                         const psld_epilogue_t* epi, hipStream_t stream);

/* Same convolution with an optional scratch buffer (>= psld_conv2d_workspace_bytes): layers whose output
 * grid cannot fill the 256 CUs (small batches, 8x8 / 16x16 maps) split the K range over extra workgroups and
 * apply the epilogue while summing the partial slabs.  workspace == NULL behaves like psld_conv2d_nhwc_f32. */
long long psld_conv2d_workspace_bytes(int batch, int oh, int ow, int cout);
int psld_conv2d_nhwc_ws_f32(const float* x1, int c1, const float* x2, int c2,
                            int batch, int ih, int iw,
                            const float* w_ohwi, int cout, int kh, int kw,
                            int stride, int pad, int transposed_stride,
                            int oh, int ow, float* y, int ldy,
                            const psld_epilogue_t* epi, void* workspace, long long ws_bytes,
                            hipStream_t stream);

/* ---- 3x3 stride-1 pad-1 convolutions on bf16 limb MFMA (PSLD_MATH_BF16X6) --------------------------------
 * Direct convolution: a (rows+2) x (W+2) halo tile of the input is split into limbs once per 32-channel chunk and
 * read from LDS at the nine tap offsets; the weights come pre-split in MFMA fragment order
 * (psld_pack_conv3x3_frag, once per optimizer step; psld_conv3x3_frag_bytes bytes).  Same contract as
 * psld_conv2d_nhwc_ws_f32 for kh = kw = 3, stride = pad = 1 (input = concat(x1[c1], x2[c2]), fused epilogue,
 * optional split-K workspace of psld_conv2d_workspace_bytes).  Shapes: c1, c2 multiples of 32, cout a multiple
 * of 128, w in {8,16,32,64} with h*w dividing or divisible by 128 (psld_conv3x3_split_supported).
 * dgrad = 1 packs the weights of the data-gradient (taps flipped, channel roles swapped: "cout" of the call that
 * consumes them is the layer's cin).  Replaces nn.Conv2d 3x3 of every ResBlock (layerspp.py:29-39). */
long long psld_conv3x3_frag_bytes(int cout, int cin);
int psld_conv3x3_split_supported(int c1, int c2, int batch, int h, int w, int cout);
int psld_pack_conv3x3_frag(const float* w_oihw, void* wfrag, int cout, int cin, int dgrad, hipStream_t stream);
/* All weight tensors of a network in ONE launch (the per-tensor packs above are launch-bound: 228 x 7 us per step).
 * table_dev: device array of `entries` x 8 int64: {src pointer, dst pointer, n_out, k_in, taps | flip << 32,
 * stride_n, stride_k, first work item}; an entry has n_out*k_in/8 work items (one per lane slot, all taps and limbs)
 * and produces exactly what psld_pack_conv3x3_frag (taps 9) / psld_pack_gemm_frag (taps 1) would;
 * total_items = sum over entries.  k_in may carry (chunk0 << 20) | (chunks_total << 40) in its upper bits: the entry then
 * fills the 32-wide K chunks chunk0 .. chunk0 + k_in/32 of a fragment set whose K dimension has chunks_total chunks
 * (several parameters concatenated along K into one set - the q | k | v projections of an attention block read by ONE
 * data-gradient GEMM); both zero: the entry is the whole K dimension. */
int psld_pack_frag_batch(const long long* table_dev, int entries, long long total_items, hipStream_t stream);
int psld_conv3x3_split_f32(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w,
                           const void* wfrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                           void* workspace, long long ws_bytes, hipStream_t stream);

/* Winograd F(2x2, 3x3) form of psld_conv3x3_split_f32 (forward and, with dgrad fragments, data gradient): the same
 * convolution, contract and fused epilogue (no gnb_* by-product, no split-K workspace) with 2.25x fewer matrix
 * instructions.  Y = A^T[(G g G^T) (.) (B^T d B)]A per 2x2 output tile; the transformed operands are formed in fp32 and
 * carried as three exact bf16 limbs each, six limb products per product, fp32 accumulation: fp32-equivalent arithmetic
 * re-associated (what cuDNN may pick for the reference's nn.Conv2d 3x3, song_sde/layers.py:103-109), within ~2x of
 * the direct kernel's rounding error against fp64.  Weights come pre-transformed and pre-split in MFMA operand order
 * (psld_pack_conv3x3_wino, once per optimizer step; psld_conv3x3_wino_frag_bytes = 16/9 of the direct fragments).
 * Shapes: c1, c2 multiples of 32, cout a multiple of 128, h even, w in {8,16,32,64}, h*w dividing or divisible by 128. */
long long psld_conv3x3_wino_frag_bytes(int cout, int cin);
int psld_conv3x3_wino_supported(int c1, int c2, int batch, int h, int w, int cout);
int psld_pack_conv3x3_wino(const float* w_oihw, void* ufrag, int cout, int cin, int dgrad, hipStream_t stream);
/* All Winograd fragment sets of a network in ONE launch.  table_dev: `entries` x 8 int64 {src pointer (OIHW weights),
 * dst pointer, n_out, k_in, flip, stride_n, stride_k, first work item}: forward fragments of a [cout][cin][3][3] weight
 * are {w, dst, cout, cin, 0, cin*9, 9, first}, data-gradient fragments {w, dst, cin, cout, 1, 9, cin*9, first}; an entry
 * has n_out*k_in/8 work items and produces exactly what psld_pack_conv3x3_wino would. */
int psld_pack_wino_batch(const long long* table_dev, int entries, long long total_items, hipStream_t stream);
int psld_conv3x3_wino_f32(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w,
                          const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                          hipStream_t stream);
/* The same with a workspace for small grids (one workgroup per CU: a launch below half a round - the 8x8 level at training
 * batches - leaves CUs idle): the channel chunks are split over psld_conv3x3_wino_ksplit workgroups per tile, plain partial
 * outputs go to the workspace (psld_conv3x3_wino_ws_bytes) and one reduction pass applies the epilogue, as
 * psld_conv3x3_split_f32 does for such shapes (epi->gn_part: formed by that pass).  Without a workspace or with ksplit == 1
 * it is psld_conv3x3_wino_f32. */
int psld_conv3x3_wino_ksplit(int c1, int c2, int batch, int h, int w, int cout);
long long psld_conv3x3_wino_ws_bytes(int c1, int c2, int batch, int h, int w, int cout);
int psld_conv3x3_wino_ws_f32(const float* x1, int c1, const float* x2, int c2, int batch, int h, int w,
                             const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                             void* workspace, long long ws_bytes, hipStream_t stream);
/* The same convolution applied to act(GroupNorm(x)) without materialising it: x1 / x2 are the RAW tensors, scale* /
 * shift* the per-(image, channel) rows psld_gn_stats_nhwc_f32 / psld_gn_stats_from_partials_f32 produce ([batch][c1],
 * [batch][c2]; a two-source input is normalised per source, as the executor's concat-free up path does), act = 1: SiLU.
 * The activation a = act(x * scale + shift) is formed on the float4 a thread has just loaded for the kernel's raw halo
 * image; pixels outside the image stay zero (the reference pads the activated tensor).  For passes that need the
 * activated tensor nowhere else - the inference forward the samplers drive (GroupNorm_0/1 + act + Conv_0/1,
 * layerspp.py:245-263; no dropout in eval mode) - this removes psld_gn_apply_nhwc_f32's round trip through HBM.
 * Shapes: psld_conv3x3_wino_supported and h*w >= 128 (one image per workgroup region). */
int psld_conv3x3_wino_gn_supported(int c1, int c2, int batch, int h, int w, int cout);
int psld_conv3x3_wino_gn_f32(const float* x1, int c1, const float* scale1, const float* shift1, const float* x2, int c2,
                             const float* scale2, const float* shift2, int act, int batch, int h, int w,
                             const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                             hipStream_t stream);
/* ... with the workspace of psld_conv3x3_wino_ws_f32: small grids split their channel chunks the same way (bitwise the
 * unfused psld_gn_apply + psld_conv3x3_wino_ws_f32 pair). */
int psld_conv3x3_wino_gn_ws_f32(const float* x1, int c1, const float* scale1, const float* shift1, const float* x2,
                                int c2, const float* scale2, const float* shift2, int act, int batch, int h, int w,
                                const void* ufrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                                void* workspace, long long ws_bytes, hipStream_t stream);

/* "Limb planes": an NHWC activation [rows][c] (c a multiple of 32) stored already decomposed, as bf16
 * [rows][c/32 chunks][3 limbs hi|mid|lo][32 channels] (6 bytes per element; hi + mid + lo == x bit for bit).  The
 * producers of a 3x3 convolution's input write this form (psld_gn_apply_nhwc_f32 with y_limb) so that the convolution
 * stages its halo tile by LDS-DMA without splitting: psld_conv3x3_limb_f32 = psld_conv3x3_split_f32 with x1 / x2
 * given as limb planes (same weights fragments, same epilogue, bitwise the same result). */
long long psld_limb_bytes(long long rows, int c);
int psld_f32_to_limb(const float* x, long long rows, int c, void* y_limb, hipStream_t stream);
int psld_limb_to_f32(const void* y_limb, long long rows, int c, float* x, hipStream_t stream);
int psld_conv3x3_limb_f32(const void* x1_limb, int c1, const void* x2_limb, int c2, int batch, int h, int w,
                          const void* wfrag, int cout, float* y, int ldy, const psld_epilogue_t* epi,
                          void* workspace, long long ws_bytes, hipStream_t stream);

/* Pointwise sibling of the kernel above: C = epilogue(A * B^T), A = concat(a1[m][k1], a2[m][k2]) row-major fp32,
 * B given as limb fragments of an [n][k] matrix (psld_pack_gemm_frag: element (n, k) is read at
 * b[n*stride_n + k*stride_k], so a [k][n] matrix such as NIN.W packs with stride_n = 1, stride_k = n;
 * psld_gemm_frag_bytes bytes).  Shapes: k1, k2 multiples of 32, k1 + k2 a multiple of 64, n a multiple of 128.
 * Replaces the 1x1 shortcut convolution (layerspp.py:268-270) and the NIN projections of the attention block
 * (layers.py:531-540, layerspp.py:78-88) and their data gradients. */
long long psld_gemm_frag_bytes(int n, int k);
int psld_gemm_split_supported(int k1, int k2, int m, int n);
int psld_pack_gemm_frag(const float* b, void* bfrag, int n, int k, long long stride_n, long long stride_k,
                        hipStream_t stream);
int psld_gemm_split_f32(const float* a1, int k1, const float* a2, int k2, int m, const void* bfrag, int n,
                        float* y, int ldy, const psld_epilogue_t* epi, void* workspace, long long ws_bytes,
                        hipStream_t stream);

/* Fused single-head spatial self-attention, forward: o[b][i][:] = sum_j softmax_j(scale * q[b][i] . k[b][j]) v[b][j][:] in
 * one kernel, the hw x hw score matrix never written (unless p != NULL: the probabilities [batch][hw][hw] fp32, what the
 * backward pass reads).  q / k / v: [batch*hw][c] fp32 with a common row stride ld (the fused q|k|v buffer: ld = 3c); o:
 * [batch*hw][c], row stride ldo.  Limb-MFMA arithmetic (three exact bf16 limbs, six products, fp32 accumulation), fp32 softmax.
 * hw in {256, 64} (16x16 / 8x8 maps), c in {256, 128}.  Replaces einsum -> softmax -> einsum of AttnBlockpp.forward
 * (song_sde/layerspp.py:82-86). */
int psld_attn_fwd_split_supported(int hw, int c);
int psld_attn_fwd_split_f32(const float* q, const float* k, const float* v, int ld, int batch, int hw, int c, float scale,
                            float* o, int ldo, float* p, hipStream_t stream);

/* Weight gradient of the same convolution, same slab contract as psld_conv2d_wgrad_nhwc_f32 (kh = kw = 3,
 * stride = pad = 1): slabs[s][cout][9][cin_total] restricted to columns [col0, col0 + cin), one slab per K range
 * of ceil(batch*h*w/32 / nsplit) 32-pixel tiles (every slab must be non-empty); the caller reduces
 * (psld_reduce_slabs_f32).  x2 / cin2 (null / 0 for none): second source of a channel concatenation, filling columns
 * [col0 + cin, col0 + cin + cin2).  Shapes: cout, cin, cin2 multiples of 64, w in {8,16,32,64}, h*w a multiple of 32. */
int psld_conv3x3_wgrad_split_supported(int cout, int cin, int batch, int h, int w);
/* Output channels per workgroup the kernel will use for this cout (64 or 128); a launch has
 * (cout / tile) * (cin / 64) * 3 * nsplit workgroups, 3 (tile 64) or 2 (tile 128) resident per CU: what the caller
 * needs to pick nsplit. */
int psld_conv3x3_wgrad_split_cout_tile(int cout);
int psld_conv3x3_wgrad_split_f32(const float* dy, int lddy, int cout, const float* x, int cin,
                                 const float* x2, int cin2, int batch, int h, int w, float* slabs,
                                 int cin_total, int col0, int nsplit, hipStream_t stream);
/* The same weight gradient with x (and x2) given as limb planes (psld_gn_apply_limb_nhwc / psld_f32_to_limb): the x
 * operand is staged without a split; dy stays fp32.  Bitwise the result of psld_conv3x3_wgrad_split_f32. */
int psld_conv3x3_wgrad_xlimb_f32(const float* dy, int lddy, int cout, const void* x_limb, int cin,
                                 const void* x2_limb, int cin2, int batch, int h, int w, float* slabs,
                                 int cin_total, int col0, int nsplit, hipStream_t stream);
/* Weight gradient of the same convolution in the Winograd F(2x2, 3x3) domain (wgrad_wino.hip; replaces the backward of
 * nn.Conv2d, reference main/models/score_fn/song_sde/layers.py:103-109): dU[xi] = sum over 2x2 output tiles of
 * (A dY A^T)[xi] (x) (B^T d B)[xi] for the 16 positions - 16 limb-MFMA products per 4 pixels instead of 36 - then
 * alpha * G^T dU G written (accumulate = 0) or added (1) to the OIHW gradient dw_oihw[cout][cin + cin2][3][3].  fp32 in, fp32
 * accumulate, exact three-limb operands like every limb kernel.  slabs: workspace of psld_conv3x3_wgrad_wino_ws_bytes;
 * nsplit: K splits of ceil(batch*h*w/128 / nsplit) 32-tile K tiles, none empty (psld_conv3x3_wgrad_wino_nsplit fills the chip
 * with one round of one-workgroup-per-CU tiles).
 * Shapes: cout % 128 == 0 (a workgroup owns 256 output channels of one position, 128 when cout is not a multiple of 256),
 * cin % 128 == 0, cin2 % 128 == 0 (second source of a channel concatenation; 0 / null for none),
 * h == w in {8,16,32,64}, batch*h*w % 128 == 0; dy rows of lddy floats. */
int psld_conv3x3_wgrad_wino_supported(int cout, int cin, int cin2, int batch, int h, int w);
int psld_conv3x3_wgrad_wino_nsplit(int cout, int cin_total, int batch, int h, int w);
long long psld_conv3x3_wgrad_wino_ws_bytes(int cout, int cin_total, int nsplit);
int psld_conv3x3_wgrad_wino_f32(const float* dy, int lddy, int cout, const float* x, int cin, const float* x2, int cin2,
                                int batch, int h, int w, float* slabs, int nsplit, float* dw_oihw, int accumulate,
                                float alpha, hipStream_t stream);
/* Pointwise weight gradient on the limb kernels: slabs[s][i][j] (row stride ldc) = sum over the s-th range of
 * ceil(k/32 / nsplit) 32-row tiles of a[p][i] * b[p][j]  (a: [k][m] rows of lda floats, b: [k][n] rows of ldb floats;
 * b2 / ldb2 / n2 (null / 0 / 0 for none): further columns [n, n + n2) of B from a second tensor (concatenation);
 * m, n, n2 multiples of 128, k of 32; every slab non-empty).  Replaces the dW of the 1x1 convolutions and NIN
 * projections that autograd computes in the reference (layerspp.py:235,268-270 Conv_2; layers.py:531-540 NIN). */
int psld_gemm_tn_split_supported(int m, int n, int k);
int psld_gemm_tn_split_f32(int m, int n, int k, const float* a, int lda, const float* b, int ldb,
                           const float* b2, int ldb2, int n2, float* slabs, int ldc, int nsplit, hipStream_t stream);

/* Batched GEMM on the limb kernels with BOTH operands fp32 activations (split inside the kernel):
 * c[b][i][j] = alpha * sum_p A(i,p) B(p,j);  ta: a is stored [k][m] (else [m][k]); tb: b is stored [n][k] (else [k][n]);
 * same operand conventions as psld_gemm_f32, not both transposed.  m, n multiples of 128, k of 32.  Replaces the
 * einsum contractions of the attention block, layerspp.py:82-86 (QK^T, softmax(QK^T) V) and their gradients. */
int psld_bgemm_split_supported(int ta, int tb, int m, int n, int k);
int psld_bgemm_split_f32(int ta, int tb, int m, int n, int k, const float* a, int lda, long long stride_a,
                         const float* b, int ldb, long long stride_b, float* c, int ldc, long long stride_c,
                         int batch, float alpha, hipStream_t stream);

/* Weight gradient of the convolution above for one input source:
 * slabs[s][co][tap][col0 + ci] = sum over the s-th range of output pixels of dy[pix][co] * x[pix+tap][ci]. */
int psld_conv2d_wgrad_nhwc_f32(const float* dy, int lddy, int cout,
                               const float* x, int cin, int batch, int ih, int iw,
                               int kh, int kw, int stride, int pad, int oh, int ow,
                               float* slabs, int cin_total, int col0, int nsplit,
                               hipStream_t stream);

/* out[perm(i)] = alpha * sum_s slabs[s][i]; layout: 0 = keep [co][tap][ci], 1 = write OIHW [co][ci][tap]. */
int psld_reduce_slabs_f32(const float* slabs, int nsplit, long long n, float* out,
                          int layout, int cout, int taps, int cin, float alpha, hipStream_t stream);

/* ---- weight layout (state_dict keeps the reference's OIHW / [in,out] shapes) ------------ */
/* OIHW -> [co][tap][ci] (forward operand). */
int psld_pack_oihw_to_ohwi_f32(const float* w, float* out, int cout, int cin, int taps, hipStream_t stream);
/* OIHW -> [ci][flip(tap)][co] (data-gradient operand: the conv of dy with the flipped, transposed filter). */
int psld_pack_oihw_to_dgrad_f32(const float* w, float* out, int cout, int cin, int taps, hipStream_t stream);

/* ---- layout at the network boundary -------------------------------------------------------*/
int psld_nchw_to_nhwc_f32(const float* x, float* y, int batch, int c, int hw, hipStream_t stream);
int psld_nhwc_to_nchw_f32(const float* x, float* y, int batch, int c, int hw, hipStream_t stream);

/* ---- GroupNorm (+SiLU), NHWC (nn.GroupNorm(min(C//4,32), eps=1e-6) + nn.SiLU:
 *      layerspp.py:67,219,231,243,264; ncsnpp.py:276-280,427) ------------------------------- */
/* workspace: psld_gn_workspace_bytes(batch, hw, c, groups). Outputs per-(n,g) mean/rstd and the
 * per-(n,c) affine scale = rstd*gamma, shift = beta - mean*rstd*gamma. */
long long psld_gn_workspace_bytes(int batch, int hw, int c, int groups);
int psld_gn_stats_nhwc_f32(const float* x, int batch, int hw, int c, int groups, float eps,
                           const float* gamma, const float* beta,
                           float* mean, float* rstd, float* scale, float* shift,
                           void* workspace, hipStream_t stream);
/* Second half of psld_gn_stats_nhwc_f32 on partial sums a limb kernel's epilogue produced (psld_epilogue_t.gn_part,
 * fine_width = its gn_fine: 8 or 4 channels per sum; channels per group must be a multiple of it). */
int psld_gn_stats_from_partials_f32(const double* gn_part, int fine_width, int batch, int hw, int c, int groups, float eps,
                                    const float* gamma, const float* beta, float* mean, float* rstd,
                                    float* scale, float* shift, hipStream_t stream);
/* y = dropout(act(x*scale[n,c] + shift[n,c])); act: 0 = identity, 1 = SiLU.  Dropout
 * (nn.Dropout, layerspp.py:265): element i of the NHWC tensor is kept iff
 * psld_dropout_keep(seed, i, p) (counter-based hash, reproducible in the backward pass, no mask
 * tensor) and scaled by 1/(1-p); drop_p = 0 disables it. */
/* seed_dev (may be NULL): one device word added to `seed` by the kernel - the per-step part of the dropout seed lives
 * in device memory so that a hipGraph-captured training step draws a fresh mask on every replay. */
int psld_gn_apply_nhwc_f32(const float* x, const float* scale, const float* shift, float* y,
                           int batch, int hw, int c, int act, float drop_p, unsigned long long seed,
                           const unsigned long long* seed_dev, hipStream_t stream);
/* The same pass writing bf16 limb planes (see psld_conv3x3_limb_f32) instead of fp32: y_limb holds
 * psld_limb_bytes(batch*hw, c) bytes; c a multiple of 32.  The dropout mask is the one psld_gn_apply_nhwc_f32 and
 * psld_gn_bwd_nhwc_f32 derive from (seed, element index). */
int psld_gn_apply_limb_nhwc(const float* x, const float* scale, const float* shift, void* y_limb, int batch,
                            int hw, int c, int act, float drop_p, unsigned long long seed,
                            const unsigned long long* seed_dev, hipStream_t stream);
/* Which kernel psld_gn_bwd_nhwc_f32 takes when the backward runs in one pass over (dy, x):
 *   PSLD_GN_BWD_AUTO      the resident-workgroup kernel (gn_bwd_pipe_kernel: the next slab's x lands in LDS by global_load_lds
 *                         while the workgroup reduces and stores) where its shape rules hold and there is no third operand, else
 *   PSLD_GN_BWD_ONE_SLAB  the register-resident one-slab kernel (gn_bwd_fused_kernel) for every shape.
 * Both give bitwise equal dx / sums / colsum_img (tests/test_kernels_gpu.py compares them through this switch).  Process-wide;
 * the initial value comes from the environment variable PSLD_GN_BWD_PIPE ("0" = one slab). */
#define PSLD_GN_BWD_AUTO 0
#define PSLD_GN_BWD_ONE_SLAB 1
int psld_set_gn_bwd_kernel(int kind);
int psld_get_gn_bwd_kernel(void);

/* Backward of y = dropout(act(GN(x))) (autograd of nn.GroupNorm + nn.SiLU + nn.Dropout, layerspp.py:256-265):
 *   dx = d/dx (+ add_scale * add when add != NULL: the gradient of an identity branch parallel to the normalisation,
 *        e.g. the residual `(x + h) / sqrt(2)` of layerspp.py:271-274) (+ the previous dx when accumulate_dx);
 *   sums [batch][2][c] (written): per image and channel sum_p dz and sum_p dz * xhat, dz = dy * mask/keep * act'(.).  The
 *        parameter gradients are their sums over the batch - dbeta[c] = sum_n sums[n][0][c], dgamma[c] = sum_n sums[n][1][c] -
 *        formed by psld_param_reduce2_f32 for one layer or, for all layers of a backward pass in ONE launch, by
 *        psld_param_reduce_batch_f32 (the reference computes them inside one autograd graph; here they leave the
 *        dependency chain of the backward pass);
 *   colsum_img (may be NULL; only where psld_gn_bwd_colsum_supported): [batch] rows of ld_img floats, columns [0, c)
 *        written: the column sums over the pixels of image n of the dx values THIS call stores (add / previous dx included) -
 *        the bias gradient and per-image time-embedding gradient of the layer whose output gradient this dx is
 *        (layerspp.py:258-263: Conv_0 + Dense_0(act(temb))[:, :, None, None]; :268-274: Conv_1 / Conv_2 when this call is
 *        the last writer of the residual stream's gradient) without a pass over dx.  Without a third operand from a
 *        closed form, per channel sum_p dx = k0 sum_p dz - hw k1 - k2 sum_p xhat, of sums the kernel reduces anyway; with
 *        one by summing the stored values (fp32 per thread, fp64 across threads, fixed order). */
int psld_gn_bwd_nhwc_f32(const float* dy, const float* x, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, int batch, int hw, int c, int groups,
                         int act, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                         float* dx, int accumulate_dx, const float* add, float add_scale, float* sums,
                         float* colsum_img, int ld_img, void* workspace, hipStream_t stream);
int psld_gn_bwd_colsum_supported(int batch, int hw, int c, int groups);
/* The same backward on WHOLE NHWC ROWS (round 5): a workgroup owns 64 pixels x 128 channels, the K = hw / 64 workgroups of an
 * (image, 128-channel block) exchange their per-group partial sums through `sync` and write dx from their registers - the
 * reads are 512-byte runs of consecutive rows instead of one 128-byte segment per row (5.9-6.4 against 4.3-4.6 TB/s).  Maps
 * above 32x32 take 128 pixels per workgroup (K = hw / 128): there the alternative is the three-pass form.
 *   psld_gn_bwd_team_rows: K (> 1) when the form takes the shape - c a multiple of 128, groups of 4 | cpg | 128 channels,
 *        128 <= hw <= 1024 a multiple of 64 or 1024 < hw <= 4096 a multiple of 128 - and the kernel selector is
 *        PSLD_GN_BWD_AUTO; else 0.
 *   sums [batch * K][2][c]: sum_p dz and sum_p dz * xhat per (image, team member, channel): dbeta / dgamma are the sums over
 *        ALL batch * K rows (psld_param_reduce*_f32 with rows = batch * K).
 *   colsum_rows (may be NULL): [batch * K] rows of ld_rows floats: column sums of the stored dx over the member's pixels
 *        (a bias gradient is alpha * the sum over all rows; per-IMAGE sums - the time-embedding gradient - need the one-slab
 *        kernels, psld_gn_bwd_nhwc_f32).
 *   sync: psld_gn_bwd_team_sync_bytes() of device memory, zeroed ONCE by the caller and then left to the kernels (slots,
 *        tag counter, error word 0: non-zero after a member gave up waiting - several seconds - for its team); one stream at a
 *        time may use a given buffer.  The launch is one RESIDENT grid; results are deterministic (fixed summation order) but
 *        not bitwise those of the one-slab kernels (group terms from fp32-rounded member sums). */
int psld_gn_bwd_team_rows(int batch, int hw, int c, int groups);
long long psld_gn_bwd_team_sync_bytes(void);
int psld_gn_bwd_team_f32(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                         const float* beta, int batch, int hw, int c, int groups, int act, float drop_p, unsigned long long seed,
                         const unsigned long long* seed_dev, float* dx, int accumulate_dx, const float* add, float add_scale,
                         float* sums, float* colsum_rows, int ld_rows, void* sync, hipStream_t stream);

/* ---- parameter gradients that are sums over the batch (or over split-K slabs), off the backward chain ----------------
 * dst1[col] (and dst2[col] when non-NULL) = alpha * sum over r in [0, rows) of src[r * ld + col], col in [0, c): fp64
 * accumulation in a fixed order (16 row lanes r, r + 16, ... combined in lane order): bitwise repeatable.  What
 * nn.GroupNorm's dgamma / dbeta, the bias gradients and nn.Linear's bias gradient are in the reference's autograd graph.
 * psld_param_reduce2_f32: two jobs of the same shape in one launch (dbeta = rows of `sums`, dgamma = rows of `sums + c`).
 * psld_param_reduce_batch_f32: `jobs` rows of a DEVICE table of 8 int64
 *   [src pointer, rows, ld, c, dst1 pointer, dst2 pointer or 0, bit pattern of (float) alpha, first 64-column block],
 * blocks = sum over the jobs of ceil(c / 64). */
int psld_param_reduce2_f32(const float* src_a, const float* src_b, int rows, int ld, int c, float* dst_a, float* dst_b,
                           float alpha, hipStream_t stream);
int psld_param_reduce_batch_f32(const long long* table_dev, int jobs, int blocks, hipStream_t stream);
/* Split-K slab reductions of MANY weight gradients in one launch (psld_reduce_slabs_f32 per job, bitwise the same sums): rows of
 * a DEVICE table of 10 int64 [slabs pointer, nsplit, n, out pointer, layout, taps, cin, bit pattern of (float) alpha, first
 * unit, units]; a job has psld_reduce_slabs_batch_units(n, layout, taps, cin) units (0: the job does not qualify - n and cin
 * must be multiples of 4, taps <= 9; slabs 16-byte aligned); units = their sum.  With layout 1 the OIHW scatter goes through
 * LDS and leaves as contiguous 16-byte stores.  layout 2: a column block of wider slabs - element (r, j) of the
 * [n / cols][cols] result is the sum over slabs of slabs[s][r * ld + j] with cols passed as `taps`, ld as `cin`, a slab
 * (n / cols) * ld floats long and the slabs pointer at the block's first column (the q | k | v weight gradients of
 * AttnBlockpp from ONE [c][3c] GEMM, layerspp.py:78-80). */
int psld_reduce_slabs_batch_units(long long n, int layout, int taps, int cin);
int psld_reduce_slabs_batch_f32(const long long* table_dev, int jobs, long long units, hipStream_t stream);


/* ---- FIR resampling: the replacement of the pybind op upfirdn2d_op.upfirdn2d
 *      (op/upfirdn2d.cpp:12-22, op/upfirdn2d_kernel.cu:209-369).  Same semantics: zero-insert
 *      upsample by `up`, pad (negative = crop), convolve with `kernel` (i.e. correlate with the
 *      flipped kernel), decimate by `down`.  layout 0: NCHW ([N*C][H][W], the reference op's view),
 *      1: NHWC.  out_h = (in_h*up + pad0 + pad1 - kh)/down + 1.  Backward = same entry with the
 *      flipped kernel, up<->down swapped and the g_pad of op/upfirdn2d.py:111-116. */
int psld_upfirdn2d_f32(const float* x, float* y, int batch, int c, int in_h, int in_w,
                       const float* kernel_host, int kh, int kw,
                       int up_x, int up_y, int down_x, int down_y,
                       int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                       int layout, int accumulate, hipStream_t stream);
/* The (compiled-but-unused) second native op of the reference, for inventory parity:
 * y = act(x + b[(i / step_b) % size_b]) * scale, act 1 = linear, 3 = leaky-relu(alpha)
 * (op/fused_bias_act_kernel.cu:18-49, forward only). */
int psld_fused_bias_act_f32(const float* x, const float* b, float* y, long long n, int size_b, int step_b,
                            int act, float alpha, float scale, hipStream_t stream);
/* The op with all seven arguments of the reference's binding (op/fused_bias_act.cpp:11-20: input, bias, refer, act,
 * grad, alpha, scale): grad = 1 is the first derivative - x is the incoming gradient, `refer` the forward output, as
 * FusedLeakyReLUFunctionBackward calls it (op/fused_act.py:27-33) - grad = 2 the second (zero for both activations). */
int psld_fused_bias_act_grad_f32(const float* x, const float* b, const float* refer, float* y, long long n, int size_b,
                                 int step_b, int act, int grad, float alpha, float scale, hipStream_t stream);

/* ---- pointwise / reductions ---------------------------------------------------------------*/
/* y = (a*sa + b*sb) (b may be NULL); accumulate: y += ... */
int psld_axpby_f32(const float* a, float sa, const float* b, float sb, float* y, long long n,
                   int accumulate, hipStream_t stream);
int psld_silu_f32(const float* x, float* y, long long n, hipStream_t stream);
int psld_silu_bwd_f32(const float* x, const float* dy, float* dx, long long n, hipStream_t stream);
/* out[b][c] = alpha * sum over the hw rows of image b of x[(b*hw+p)*ld + c] (bias / temb-bias gradients). */
long long psld_colsum_workspace_bytes(int batch, int hw, int c);
int psld_colsum_f32(const float* x, int ld, int batch, int hw, int c, float* out, float alpha,
                    void* workspace /* >= psld_colsum_workspace_bytes, NULL = slow scalar path */,
                    hipStream_t stream);
/* Bias gradient in three launches (two chained psld_colsum_f32 calls take four): out[c] = alpha * sum over (batch, hw) of x[b][p][0..c) (row stride ld), and, when
 * per_image is not NULL, per_image[b*ld_per_image + c] = the unscaled per-image sums (ld_per_image = 0 means c; the
 * time-embedding gradient needs them, layerspp.py:262-263).  Same workspace as psld_colsum_f32; c % 4 == 0, c <= 1024, 16-byte aligned x. */
int psld_bias_grad_f32(const float* x, int ld, int batch, int hw, int c, float* per_image, int ld_per_image,
                       float* out, float alpha, void* workspace, hipStream_t stream);
/* The same over a [batch*hw][3*seg] buffer (row stride ld) whose three column segments are the output gradients of three
 * layers computed by one GEMM (the q | k | v projections of AttnBlockpp, layerspp.py:78-80): out0 / out1 / out2 [seg]. */
int psld_bias_grad_seg_f32(const float* x, int ld, int batch, int hw, int seg, float* out0, float* out1, float* out2,
                           float alpha, void* workspace, hipStream_t stream);
/* Many contiguous copies in one launch: table_dev = `entries` x 4 int64 {src pointer, dst pointer, float4 count, first
 * float4 index}; total4 = sum of the counts.  Used to gather the Dense_0 (time-embedding projection,
 * layerspp.py:225-228) weights of all ResBlocks into one [sum C_out][4*nf] matrix per optimizer step, so that their
 * 57 forward Linear calls and 57 data gradients become one GEMM each. */
int psld_copy_batch_f32(const long long* table_dev, int entries, long long total4, hipStream_t stream);

/* dst[r][0:cols] (+)= src[r][0:cols] with row strides: channel concat (ncsnpp.py:374) and its split. */
int psld_copy2d_f32(const float* src, int ld_src, float* dst, int ld_dst, long long rows, int cols,
                    int accumulate, hipStream_t stream);
/* 3x3 im2col of a few-channel NHWC tensor (9*c <= ld_out): cols[m][ch*9 + tap] (zero-filled to ld_out columns), so
 * that the 6-channel stem / first-pyramid convolutions (ncsnpp.py:317, layerspp.py:149-163) and the data / weight
 * gradients of the 6-channel head (ncsnpp.py:430) run as K = 64 GEMMs on the tile engine.  flip = 1 (stride 1 only)
 * stores tap 8 - t in the column of tap t (what a data gradient reads).  The column order ch*9 + tap is OIHW's. */
int psld_im2col3x3_small_f32(const float* x, int batch, int ih, int iw, int c, int oh, int ow, int stride, int pad,
                             int flip, float* out, int ld_out, hipStream_t stream);
/* 3x3 im2col / col2im of an NHWC tensor with c % 4 == 0 channels, K order (tap, channel) = the OHWI weight order of
 * psld_pack_oihw_to_ohwi_f32: cols[m][tap*c + ch] = x[n, oy*stride + ky - pad, ox*stride + kx - pad, ch] (zero outside);
 * col2im is its adjoint (written, fixed summation order).  They turn the stride-2 convolution of the input pyramid
 * (`Downsample` with fir: up_or_down_sampling.py:177 `F.conv2d(x, w, stride=2)`, layerspp.py:149-163) and its data
 * gradient into psld_gemm_split_f32 calls. */
int psld_im2col3x3_f32(const float* x, int batch, int ih, int iw, int c, int oh, int ow, int stride, int pad,
                       float* cols, hipStream_t stream);
int psld_col2im3x3_f32(const float* dcols, int batch, int ih, int iw, int c, int oh, int ow, int stride, int pad,
                       float* dx, hipStream_t stream);
/* y[n,oy,ox,o] = bias[o] + sum_{tap,c} x[n,oy+ky-1,ox+kx-1,c] * w_ohwi[o][tap][c]: 3x3 stride-1 pad-1 convolution with
 * 3 or 6 output channels (the network head `conv3x3(in_ch, channels)`, ncsnpp.py:430), exact fp32 fmaf chains + a
 * fixed cross-lane sum; cin % 4 == 0, cin <= 256, weights in the OHWI order of psld_pack_oihw_to_ohwi_f32. */
int psld_conv3x3_fewout_supported(int cin, int cout);
int psld_conv3x3_fewout_f32(const float* x, const float* w_ohwi, const float* bias, float* y, int batch,
                            int h, int w, int cin, int cout, hipStream_t stream);
/* dst[r][j] = alpha * src[r][j], j < cols, arbitrary cols / leading dimensions (pads and un-pads the small K = 54
 * weight matrices of the calls above). */
int psld_scale_copy2d_f32(const float* src, int ld_src, float* dst, int ld_dst, long long rows, int cols, float alpha,
                          hipStream_t stream);

/* rows of length L: y = softmax(x) ; dx = y * (dy - sum(y*dy)) (layerspp.py:84). */
int psld_softmax_rows_f32(const float* x, float* y, long long rows, int L, hipStream_t stream);
int psld_softmax_rows_bwd_f32(const float* y, const float* dy, float* dx, long long rows, int L, hipStream_t stream);

/* ---- time embedding (layerspp.py:39-41, layers.py:500-514) ----------------------------------*/
/* out[b] = cat[sin(p), cos(p)], p = ((logf(t[b]) * W[e]) * 2) * pi   (use_log = 1, Fourier)
 *                               p = t[b] * W[e]                      (use_log = 0, positional) */
int psld_time_embed_f32(const float* t, const float* W, float* out, int batch, int e, int use_log,
                        hipStream_t stream);

/* ---- PSLD SDE (main/models/sde/psld.py) -----------------------------------------------------*/
typedef struct psld_sde_params {
    double beta_0, beta_1, nu, gamma, m_inv, numerical_eps;
    int decomp_lower;      /* 1 = 'lower' (Cholesky), 0 = 'upper' */
} psld_sde_params_t;

/* Per-sample f64 scalars of the perturbation kernel, PSLD_COEFF_STRIDE doubles per sample:
 * [0] = b_t (psld.py:42-44), [1] = exp(-(nu+gamma)/4 * b_t) (psld.py:65-67), [4..7] = c11,c12,c21,c22
 * of get_coeff (psld.py:154-186), [8..10] = (xx_t, xm_t, mm_t) of _cov (psld.py:86-152).  nan_flag (device int, caller-zeroed) is set to 1 if any
 * coefficient is NaN (psld.py:166-171 raises ValueError). */
int psld_perturb_coeffs_f64(const double* t, int batch, const psld_sde_params_t* p,
                            double xx_0, double mm_0, double* coeffs, int* nan_flag, hipStream_t stream);
/* u_t = mu_t + L_t eps (psld.py:262-287), NCHW.  x0,m0 [B,C,H,W] f32 (m0 NULL = zeros: HSM),
 * eps [B,2C,H,W] f32; writes z_t f32 (losses.py:114) and optionally u_t / mu_t f64. */
int psld_perturb_f32(const float* x0, const float* m0, const float* eps, const double* coeffs,
                     const psld_sde_params_t* p, int batch, int c, int hw,
                     float* z_f32, double* u_f64, double* mu_f64, hipStream_t stream);
/* loss = mean|sum (eps - eps_pred)^2 over n elements (losses.py:118-129); partials workspace
 * >= psld_reduce_workspace_bytes(n); also writes d(loss)/d(eps_pred) * upstream if grad != NULL. */
long long psld_reduce_workspace_bytes(long long n);
int psld_sqerr_loss_f32(const float* eps, const float* eps_pred, long long n, int reduce_mean,
                        float* loss, float* grad, float grad_scale, void* workspace, hipStream_t stream);

/* One Euler-Maruyama predictor update of the reverse SDE (samplers/sde.py:16-26 +
 * psld.py:230-260,330-364), NCHW, float64 state:
 *   f = drift(x, beta), g = diffusion;  score = -L^{-T} eps_pred (f32, coefficients cast to f32);
 *   f_bar = -f + g^2 * score (x0.5 if probability_flow); x_mean = x + f_bar*dt;
 *   x = x_mean + g*sqrt(dt)*z   (z NULL -> drift-only "denoise" step, sde.py:28-36)
 * Also writes x as f32 for the next network call (psld.py:354). */
typedef struct psld_em_coeffs {
    double beta;                 /* beta(T - t) */
    double m_inv, gamma, nu, m;  /* SDE constants */
    float c11, c12, c21, c22;    /* get_inv_coeff(...) cast to f32 (psld.py:253-258) */
    double dt;
    int score_mode;              /* 0 = score_xm (6-ch eps), 1 = score_m lower (3-ch eps -> momentum only),
                                    2 = score_x upper */
    int probability_flow;
} psld_em_coeffs_t;
int psld_em_step_f64(double* x, const float* eps_pred, const double* z, const psld_em_coeffs_t* k,
                     int batch, int c, int hw, float* x_f32_out, hipStream_t stream);
/* reverse_sde outputs themselves (f_bar, g_bar) for callers that want them (psld.py:345-364). */
int psld_reverse_sde_f64(const double* x, const float* eps_pred, const psld_em_coeffs_t* k,
                         int batch, int c, int hw, double* f_bar, double* g_bar, hipStream_t stream);
/* The same outputs with PER-SAMPLE times, as PSLD.sde / reverse_sde take them (psld.py:330-364, t[B]): t_rev[b] is the
 * already reversed time T - t of sample b (device f64); beta_t, _cov and get_inv_coeff are evaluated per sample on the
 * device (no host read of t).  eps_pred NULL: the forward SDE's own (f, g) of psld.py:330-343 (t_rev = t).  nan_flag
 * as in psld_perturb_coeffs_f64 (psld.py:214-219 raises ValueError).  batch <= 65535. */
int psld_reverse_sde_rows_f64(const double* x, const float* eps_pred, const double* t_rev,
                              const psld_sde_params_t* p, double xx_0, double mm_0, int score_mode,
                              int probability_flow, int batch, int c, int hw, double* f_bar, double* g_bar,
                              int* nan_flag, hipStream_t stream);
/* Symmetric-splitting (SSCS) sampler, SURVEY.md 8(f) rank 1 (samplers/sde.py:227-370):
 * analytic half step u <- M u + L z (M = exp-scaled 2x2 mean matrix of :236-263, L = factor of the
 * transition covariance :265-291 through get_coeff) and the Euler score step of :313-329. */
typedef struct psld_sscs_coeffs {
    double a_xx, a_xm, a_mx, a_mm;
    double c11, c12, c21, c22;
} psld_sscs_coeffs_t;
int psld_sscs_analytic_f64(double* x, const double* z, const psld_sscs_coeffs_t* k,
                           int batch, int c, int hw, float* x_f32_out, hipStream_t stream);
int psld_sscs_score_step_f64(double* x, const float* eps_pred, const psld_em_coeffs_t* k,
                             int batch, int c, int hw, hipStream_t stream);
/* Edges of the path.  Writer: position half of the f64 [B,c_total,H,W] state ->
 * uint8 [B,H,W,c] = trunc(clip((x*0.5+0.5)*255, 0, 255)) (callbacks.py:103-107, util.py:147-158).
 * Loader: uint8 [B,H,W,c] -> f32 [B,c,H,W] = img/127.5-1 (norm) or img/255, optional per-image
 * horizontal flip (util.py:25-30, datasets/cifar10.py:33-46). */
int psld_samples_to_uint8(const double* x, unsigned char* out, int batch, int c, int c_total, int hw,
                          int denorm, hipStream_t stream);
int psld_uint8_to_images_f32(const unsigned char* img, float* out, const unsigned char* flip,
                             int batch, int c, int h, int w, int norm, hipStream_t stream);
/* Adaptive Runge-Kutta building blocks for the black-box probability-flow ODE sampler (SURVEY 8(f)
 * rank 2, samplers/ode.py:40-76; the reference round-trips every RHS evaluation through host numpy via
 * torchdiffeq's scipy bridge).  v / coef are HOST arrays of nv <= 8 device pointers / doubles.
 *   out = base + sum_j coef[j]*v[j]            (base may be NULL; optional f32 copy)
 *   out[0] = sum_i ((sum_j coef[j]*v[j][i]) / (atol + rtol*max(|p_i|,|q_i|)))^2   (device scalar) */
int psld_lincomb_f64(double* out, const double* base, const double* const* v, const double* coef, int nv,
                     long long n, float* out_f32, hipStream_t stream);
int psld_scaled_norm_sq_f64(const double* const* v, const double* coef, int nv, const double* p,
                            const double* q, double atol, double rtol, long long n, double* out,
                            void* workspace /* >= psld_reduce_workspace_bytes(n) */, hipStream_t stream);
/* VP-SDE baseline (main/models/sde/vpsde.py:9-99): perturbation kernel and the reverse
 * drift / Euler-Maruyama update (mode 0: f_bar = -f + g^2*score; mode 1: x <- x + f_bar*dt + g*sqrt(dt)*z). */
/* ScoreLoss beyond the eps-MSE (main/losses.py:38-39, 55-63).  mode 1: L1 criterion, loss f32 = mean|sum |eps - eps_pred|;
 * mode 2: weighting 'nll', loss f64 = mean|sum (score(eps_pred) - score(eps))^2 * beta(t), score = -eps / std(t)
 * (vpsde.py:26-27, 97-99), t [batch] f64, per = C*H*W.  grad (optional): d loss / d eps_pred * grad_scale, f32.
 * workspace >= psld_reduce_workspace_bytes(batch*per). */
int psld_vp_score_loss(const float* eps, const float* eps_pred, const double* t, double beta0, double beta1,
                       int batch, long long per, int mode, int reduce_mean, void* loss, float* grad,
                       float grad_scale, void* workspace, hipStream_t stream);
int psld_vp_perturb_f32(const float* x0, const float* eps, const double* t, double beta0, double beta1,
                        int batch, long long per_image, float* z_f32, double* u_f64, hipStream_t stream);
int psld_vp_reverse_f64(double* x, const float* eps_pred, const double* z, double beta, double std, double dt,
                        int probability_flow, int mode, long long n, double* f_bar, float* x_f32_out,
                        hipStream_t stream);
/* Classifier guidance (ClassCondEulerMaruyamaSampler.predictor_update_fn, samplers/sde.py:90-97): x += coef * grad
 * with coef_x on the position half and coef_m on the momentum half of the [B,2C,HW] f64 state (coef = g^2 * dt,
 * the classifier temperature already folded into grad); optionally refreshes the f32 copy. */
int psld_guide_f64(double* x, const float* grad, double coef_x, double coef_m, int batch, int c, int hw,
                   float* x_f32_out, hipStream_t stream);
/* Softmax cross entropy of [rows][n] logits vs int64 labels (nn.CrossEntropyLoss in PSLDTimeCELoss,
 * losses.py:147-173): *loss = loss_scale * sum_r (logsumexp(z_r) - z_r[y_r]); dlogits (optional) =
 * grad_scale * (softmax(z_r) - onehot(y_r)); *correct (optional) = number of rows whose argmax is the label. */
int psld_softmax_xent_f32(const float* logits, const long long* labels, int rows, int n, float loss_scale,
                          float grad_scale, float* loss, float* dlogits, float* correct, hipStream_t stream);
/* Inpainting combine (ES3EulerMaruyamaInpainter.inpaint_update_fn, samplers/sde.py:161-181): for the state
 * x = [x | m] and the re-perturbed known image u = [x_k | m_k], both [B,2C,HW] f64, x <- x*(1-mask) + u*mask with
 * mask [B,C,HW] f32 in {0,1} applied to both halves; optionally refreshes the f32 copy the network reads. */
int psld_mask_combine_f64(double* x, const double* u, const float* mask, int batch, int c, int hw,
                          float* x_f32_out, hipStream_t stream);
int psld_f64_to_f32(const double* x, float* y, long long n, hipStream_t stream);
int psld_f32_to_f64(const float* x, double* y, long long n, hipStream_t stream);

/* ---- per-step parameter maintenance over flat buffers (wrapper.py:82-89,128-155;
 *      callbacks.py:57-64): 97.6 M parameters in ONE launch each --------------------------------*/
/* norm_out[0] = sqrt(sum g^2) (double); workspace >= psld_reduce_workspace_bytes(n). */
int psld_grad_norm_f32(const float* g, long long n, double* norm_out, void* workspace, hipStream_t stream);
/* clip_coef = min(max_norm / (norm + 1e-6), 1) computed on device from norm_out (no host sync);
 * g *= clip_coef; Adam (torch.optim.Adam semantics, step is 1-based); optional EMA of p into ema.
 * max_norm <= 0 disables clipping (norm may be NULL).
 * err_word (may be NULL): a device error word of this step's backward (first 8 bytes of the team kernels' slot buffer,
 * psld_gn_bwd_team_f32): when it is non-zero at execution time the launch leaves p / m / v / ema untouched and writes NaN to
 * poison[0] (may be NULL; the loss scalar the caller logs) - bad gradients never reach the parameters, with no host read. */
int psld_adam_ema_f32(float* p, const float* g, float* m, float* v, float* ema, long long n,
                      const double* norm, double max_norm, double lr, double beta1, double beta2,
                      double eps, double weight_decay, int step, double ema_tau, int write_clipped_grad,
                      float* g_mut, const float* hyper_dev, const unsigned long long* err_word, float* poison,
                      hipStream_t stream);
/* The two per-step scalars of the kernel above, lr / (1 - beta1^step) and 1 / sqrt(1 - beta2^step), formed in
 * double like the launcher does: what a caller writes into hyper_dev[0..1] (device floats) before replaying a
 * captured training step (then lr / step of the captured call are ignored). */
void psld_adam_step_scalars(double lr, double beta1, double beta2, int step, float* out2_host);
/* The same two scalars written to a 2-float DEVICE buffer by a kernel that takes them by value (captured training step). */
int psld_adam_step_scalars_dev(double lr, double beta1, double beta2, int step, float* out2_dev, hipStream_t stream);
/* target = target*tau + src*(1-tau) (callbacks.py:62-64); hyper-parameters are doubles so that
 * (1 - tau), (1 - beta) are formed in double and rounded once, as torch does for python floats.
 * err_word as in psld_adam_ema_f32: non-zero -> no-op. */
int psld_ema_f32(float* target, const float* src, long long n, double tau, const unsigned long long* err_word,
                 hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PSLD_HIP_H */
