/*
 * fw_mi355x.h -- C ABI of libfw_mi355x.so: the MI355X (gfx950 / CDNA4) kernels behind
 * FantasyWorld's per-step denoising forward (FantasyWorldFusionModel.joint_forward).
 *
 * The reference (Fantasy-AMAP/fantasy-world) has no FFI: every hot op is a call into PyTorch
 * (SURVEY.md 2.3).  Each entry point below replaces one family of those call sites; the file:line
 * next to it is the reference code whose arithmetic it implements (paths relative to the reference
 * repo root; DIT21 = FantasyWorld/diffsynth_wan21/models/wan_video_dit.py, IRG =
 * FantasyWorld/fusion/layer/block.py, VB/VA/VR = FantasyWorld/vggt/layers/{block,attention,rope}.py,
 * CAM = FantasyWorld/diffsynth_wan21/models/camera_control.py, M21 = FantasyWorld/fusion/model_wan21.py).
 *
 * Conventions:
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch); the library never allocates,
 *     frees or synchronises; every call only enqueues work on `stream` (a hipStream_t passed as void*).
 *   - bf16 = raw IEEE bfloat16 bits (uint16_t); "ld*" arguments are leading dimensions in ELEMENTS.
 *   - every function returns 0 on success, a positive hipError_t, or a negative FW_E_* code.
 *   - no C++ types, no torch types, no exceptions cross this boundary.
 */
#ifndef FW_MI355X_H
#define FW_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FW_ABI_VERSION 12

/* error codes (negative; positive values are hipError_t) */
#define FW_E_BADARG   (-1)   /* shape / alignment / enum violates the documented contract */
#define FW_E_UNSUPPORTED (-2)

/* dtype enums for residual / output tensors */
#define FW_DT_NONE 0
#define FW_DT_BF16 1
#define FW_DT_F32  2

/* activation applied to (acc + bias) in fw_gemm_bf16 / fw_gemv_f32 */
#define FW_ACT_NONE      0
#define FW_ACT_RELU      1   /* CAM:29-33,47-49 (adapter MLPs) */
#define FW_ACT_GELU_TANH 2   /* DIT21:274-275 FFN, DIT21:388-392 text_embedding */
#define FW_ACT_GELU_ERF  3   /* DIT21:330 img_emb, vggt/layers/mlp.py:22 */
#define FW_ACT_SILU      4   /* DIT21:393-399 time_embedding / time_projection */

/* q/k normalisation modes of fw_qk_prep */
#define FW_NORM_NONE      0  /* IRG:547-551 (bicross q,k: no norm) */
#define FW_NORM_RMS_FULL  1  /* DIT21:135-146,170-171: RMSNorm over the FULL width (all heads), weight[width] */
#define FW_NORM_LN_HEAD   2  /* VA:43-44,55: LayerNorm(head_dim) per head, affine weight/bias[head_dim] */

/* rotary modes of fw_qk_prep */
#define FW_ROPE_NONE        0
#define FW_ROPE_INTERLEAVED 1 /* DIT21:97-102: complex multiply on pairs (2i,2i+1) */
#define FW_ROPE_HALF2D      2 /* VR:120-131,154-188: two hd/2 halves (y,x), rotate-half pairs (i,i+hd/4) */

int fw_abi_version(void);

/*
 * Kernel-selection knobs for A/B measurements (tools/microbench.py); results never depend on them.
 * Each slot is initialised once from the environment variable of the same name.
 */
#define FW_OPT_GEMM_TILE   0   /* FW_GEMM_TILE: 0 = auto, 128 / 256 = force the tile family */
#define FW_OPT_GEMM_KERNEL 1   /* FW_GEMM_KERNEL: 9 (default) = 8-wave TWO-slot ping-pong kernel (round 4, gemm_pp.hip; the implicit-GEMM
                                  convolutions and operands with a row stride >= 2^21 elements stay on 4); 4 = 8-wave four-slot
                                  ping-pong kernel (the default of rounds 2-3); 5 = four-wave 128x128-wave-tile kernel (independent
                                  implementation, A/B).  All three keep the same k order per output element: bit-identical results */
#define FW_OPT_GEMM_VAR    2   /* FW_GEMM_VAR: 0 (default); bit 1 = TIMING build of the ping-pong kernel (4: phase + tile stamps; 9: phase stamps),
                                  bit 2 = tile stamps only (tools/gemm_timeline.py); value >> 4 (if non-zero) = M-tiles per group of
                                  the tile order (default 4 for outputs >= 20 column tiles wide, else 8; tools/gemm_ab.py) */
#define FW_OPT_ATTN_VAR    3   /* FW_ATTN_VAR: 192 (default) = per-head-dim choice among the log2-domain kernels that take q
                                   already multiplied by scale*log2(e) (FW_ATTN_Q_PRESCALED).  Round 6: for hd 128 / 96 / 64 the
                                   single-stream kernel with its tile loop unrolled by the LDS ring depth (193; ring slots are
                                   compile-time, no address VALU in the loop) and its row sums on the matrix pipe.  Bit 11 (2048, added
                                   to any value): round 5's choice -- fp32 row sums on the vector pipe, hd 96 on the two-segment
                                   ping-pong kernel (64) -- for the A/B.  129 = single-stream, run-time ring slots (the round-2
                                   default; what the split-KV tails run); 131 / 195 = the same two with one 64-row wave per SIMD;
                                   196 = 193 without the issue-order pins; 66 = TIMING build of the ping-pong kernel; 0 = the generic
                                   first kernel (also what calls WITHOUT the pre-scaled flag get).  Bits 8-9: static wave priority
                                   before the tile loop, 256 = waves 4..7, 512 = waves 0..3 (measured +-0, default off); bit 10
                                   (1024): tile requests in the pointer form.
                                   fw_attention_fp8: default (round 6) = the single-stream kernel with linear-byte probabilities, the
                                   two-block tile, requests between the PV MFMAs, one barrier per two tiles, the steady loop unrolled
                                   by the ring depth and row sums by a 16x16x128 MFMA; 17 / 16 / 15 / 11 = the same arithmetic (same
                                   bits) without the last one / two / three / four of those steps; 14 = linear-byte probabilities in
                                   round 5's tile body; 12 / 13 = round 5's kernel (exact exponential), the two waves of a SIMD half
                                   a tile apart / in phase; 9 = the two-group ping-pong kernel (rounds 2-4; also serves views of 4 GiB
                                   or more); 8 = the round-2 in-phase kernel */
#define FW_OPT_COUNT       4
int fw_set_option(int opt, int value);

/* Measurement hook: shader-clock timestamps [wave 0..7][8] of work-group 0 at KV tile 100, written by the TIMING build of the
 * ping-pong attention kernel (FW_ATTN_VAR = 66); synchronous copy to host memory. */
int fw_debug_gemm_timestamps(unsigned long long* host_out, int n);   /* same, GEMM ping-pong kernel (FW_GEMM_KERNEL=4, var bit 1) */
int fw_debug_gemm_pp_timestamps(unsigned long long* host_out, int n);   /* same, two-slot GEMM kernel (FW_GEMM_KERNEL=9, var bit 1): [group 0..1][slab 16..19][8] */
int fw_debug_attention_timestamps(unsigned long long* host_out, int n);

/* Human-readable description of the last negative error on this thread (never NULL). */
const char* fw_last_error(void);

/*
 * C[M,N] = epilogue( A[M,K] @ W[N,K]^T )   -- every nn.Linear on the hot path
 * (DIT21:166-169,216-225,274-275,357,388-392; IRG:340-346; VA:42,46; mlp.py:29-31; CAM:27-51;
 *  VGGT.projection_head = 1x1x1 Conv3d = Linear, vggt/models/vggt.py:32; patch_embedding Conv3d
 *  k=s=(1,2,2) = Linear over gathered 144-wide patches, DIT21:386,424-435).
 *   y = acc + bias[n]                      (bias may be NULL)
 *   y = act(y)
 *   y = y * g1[n] + g0[n]                  (g1/g0 may be NULL -> 1 / 0): gates, LayerScale, VGGT post-MLP modulation
 *   y = y + res[m,n]                       (res_dtype FW_DT_NONE / BF16 / F32; ldr leading dim)
 *   C = (out_dtype) y                      (FW_DT_BF16 or FW_DT_F32; C may alias res)
 * A, W: bf16 row-major, K % 64 == 0 (callers zero-pad), lda/ldw % 8 == 0, 16-byte aligned bases.
 * bf16 MFMA 32x32x16, fp32 accumulate.
 */
int fw_gemm_bf16(const uint16_t* A, int64_t lda, const uint16_t* W, int64_t ldw,
                 void* C, int64_t ldc, int out_dtype,
                 int M, int N, int K,
                 const float* bias, int act, const float* g1, const float* g0,
                 const void* res, int64_t ldr, int res_dtype,
                 void* stream);

/*
 * Non-causal softmax(Q K^T * scale) V for head_dim in {64, 96, 128}
 * (DIT21:28-66 flash_attention; IRG:598-605 the two bidirectional SDPA calls; VA:61).
 *   Q : [batch][Lq][heads*hd] bf16, row stride ldq, batch stride bsq (elements); same for K (ldk, bsk).
 *   Vt: [batch][heads][hd][Lk_pad] bf16, produced by fw_v_transpose (keys permuted inside each 32-block,
 *       zero padded to Lk_pad = roundup(Lk, 64)).
 *   O : [batch][Lq][heads*hd] bf16 (ldo, bso).
 *   flags: FW_ATTN_ACCUMULATE -> O += result (cross-attn text+image sum, DIT21:197-200);
 *          FW_ATTN_Q_PRESCALED -> Q already carries the factor scale*log2(e) (fw_qk_prep out_scale): `scale` is ignored, the
 *          scores are used in the log2 domain as they come out of the MFMA (saves one multiply-add per score and lets the
 *          running max be folded into the accumulator input of QK^T) -- the fast ping-pong kernel needs this form.
 *   workspace / workspace_bytes: optional caller-owned scratch (the library never allocates).  With at least
 *          fw_attention_workspace_bytes(...) bytes and FW_ATTN_Q_PRESCALED, a tail q-block (Lq % 256 != 0) that would cost the
 *          grid a whole extra round of the 256 CUs runs as split-KV instead (its keys cut into runs processed side by side,
 *          un-normalised partial outputs merged by a second kernel); NULL / 0 = single launch.  Same result up to fp32
 *          summation order.
 * fp32 online softmax, bf16 MFMA 32x32x16 for QK^T and PV.
 */
#define FW_ATTN_ACCUMULATE  1
#define FW_ATTN_Q_PRESCALED 2
int fw_attention_bf16(const uint16_t* Q, int64_t ldq, int64_t bsq,
                      const uint16_t* K, int64_t ldk, int64_t bsk,
                      const uint16_t* Vt, int64_t Lk_pad,
                      uint16_t* O, int64_t ldo, int64_t bso,
                      int batch, int heads, int head_dim, int Lq, int Lk,
                      float scale, int flags, void* workspace, int64_t workspace_bytes, void* stream);
/* Scratch size with which fw_attention_bf16 takes the split-KV route for this shape; 0 = it would not use a workspace. */
int64_t fw_attention_workspace_bytes(int batch, int heads, int head_dim, int Lq, int Lk);

/*
 * V[batch][Lk][heads*hd] (ldv, bsv) -> Vt[batch][heads][hd][Lk_pad] in the key order fw_attention_bf16 expects.
 */
int fw_v_transpose(const uint16_t* V, int64_t ldv, int64_t bsv, uint16_t* Vt, int64_t Lk_pad,
                   int batch, int heads, int head_dim, int Lk, void* stream);

/*
 * LayerNorm over the last dim + optional affine + optional AdaLN modulation, bf16 out
 * (DIT21:267-273,301,305,310 norm1/2/3 + modulate; DIT21:350-357 Head; IRG:164-167,199 bicross norms;
 *  VB:45,63,73-81 norm1/norm2; DIT21:324-341 img_emb LayerNorms).
 *   y = (x - mean) * rsqrt(var + eps); y = y*w + b (w,b may be NULL); y = y*(1+scale) + shift (may be NULL)
 * x: [rows][C] f32 or bf16 (x_dtype), ldx; y: bf16 [rows][C], ldy. C % 8 == 0, C <= 8192.
 */
int fw_layernorm_mod(const void* x, int64_t ldx, int x_dtype, uint16_t* y, int64_t ldy,
                     int rows, int C, const float* w, const float* b,
                     const float* scale, const float* shift, float eps, void* stream);

/* fw_layernorm_mod with the rounding remainder kept: y = y_hi + y_lo, y_hi = bf16(y), y_lo = bf16(y - y_hi) (round 6).  For the ONE
 * LayerNorm whose bf16 rounding reaches the output without a residual stream behind it -- the one in front of the output head
 * (DIT21:350-357): a third of the whole forward's bf16 floor (docs/parity.md, per-site ablation).  The head's 5120 -> 64 GEMM then
 * runs on both parts.  x fp32 [rows][C] (ldx), y_hi / y_lo bf16 (ldy); C % 256 == 0. */
int fw_layernorm_mod_split(const float* x, int64_t ldx, uint16_t* y_hi, uint16_t* y_lo, int64_t ldy, int rows, int C,
                           const float* w, const float* b, const float* scale, const float* shift, float eps, void* stream);

/*
 * In-place q/k post-projection: normalisation + rotary embedding on a [rows][heads*hd] bf16 slice (ldx).
 *   norm_mode: FW_NORM_*; norm_w (and norm_b for LN_HEAD) fp32.
 *   out_scale: the result is multiplied by out_scale in fp32 before the single bf16 rounding (1.0f = none); the engine
 *   folds softmax_scale*log2(e) into q here (see FW_ATTN_Q_PRESCALED).
 *   rope_mode: FW_ROPE_*; rope_tab = fp32 [tab_rows][hd/2][2] (cos,sin), row used = row % tab_rows
 *   (DIT21:97-102,170-182 RMSNorm+RoPE3D; VA:55-59 + VR:154-188; IRG:547-551 with the identity rows of
 *    DIT21:105-132 baked into the table).
 */
int fw_qk_prep(uint16_t* x, int64_t ldx, int rows, int heads, int head_dim,
               int norm_mode, const float* norm_w, const float* norm_b, float eps,
               int rope_mode, const float* rope_tab, int tab_rows, float out_scale, void* stream);

/*
 * Head-sharded tensor parallelism (north_star's partition; fantasy_world_amd/tensor_parallel.py): a rank holds only a column slice
 * [rows][heads_local*head_dim] of q / k, but the DiT's RMSNorm spans ALL heads (DIT21:135-146,170-171).
 *   fw_row_sumsq : out[r] = sum_c x[r][c]^2 of the local slice (fp32) -- all-reduced across the ranks by the caller;
 *   fw_qk_prep_tp: fw_qk_prep(FW_NORM_RMS_FULL) with the row statistic supplied: r = rsqrt(row_sumsq[r] / norm_width + eps),
 *                  norm_w = the local slice of the weight.
 *   fw_residual_add: x[r][c] += (y[r][c] + bias[c]) * g1[c] + g0[c] (fp32 stream x; y bf16 or f32; bias / g1 / g0 optional) -- the
 *                  epilogue of a row-parallel GEMM (o-projection, FFN / MLP second layer) applied AFTER the all-reduce of its
 *                  partial sums (DIT21:246-251,311-313; VB:73-81).
 */
int fw_row_sumsq(const uint16_t* x, int64_t ldx, int rows, int width, float* out, void* stream);
int fw_qk_prep_tp(uint16_t* x, int64_t ldx, int rows, int heads, int head_dim, const float* norm_w, float eps,
                  int rope_mode, const float* rope_tab, int tab_rows, float out_scale,
                  const float* row_sumsq, int norm_width, void* stream);
int fw_residual_add(float* x, int64_t ldx, const void* y, int64_t ldy, int y_dtype, int rows, int C,
                    const float* bias, const float* g1, const float* g0, void* stream);

/*
 * Per-forward modulation tables for ALL blocks of a kind in one launch (round 6; replaces ~330 single-work-group tensor-op launches
 * per forward): every DiT block adds its learned [6][C] modulation to the time projection (DIT21:296-297 `self.modulation + t_mod`),
 * every VGGT block does the same and gates its MLP with LayerScale (VB:73-81: x += (ls2 (mlp) (1 + e4) + e3) e5), the head adds the
 * time embedding to its [2][C] table (DIT21:352-353).
 *   table[b][r][c] = mod[b][r][c] + t[r % t_rows][c]                    b < nblk, r < rows, c < C   (fp32, contiguous)
 *   ls2 != NULL (rows == 6):  g1[b][c] = ls2[b][c] * (1 + table[b][4][c]) * table[b][5][c]
 *                             g0[b][c] = ls2[b][c] * table[b][3][c] * table[b][5][c]
 * (the per-column scale / offset the fc2 GEMM's epilogue applies).  Same operation order and rounding as the tensor ops it replaces.
 */
int fw_modulation_tables(const float* mod, const float* t, int t_rows, const float* ls2, float* table, float* g1, float* g0,
                         int nblk, int rows, int C, void* stream);

/*
 * Sampler step on the device: out = latents + (neg + cfg_scale * (pos - neg)) * dsigma -- the CFG combine
 * (M21:318-319) and FlowMatchScheduler.step (diffsynth_wan21/schedulers/flow_match.py:43-53) in one launch, BIT-identical to the
 * reference's five PyTorch tensor ops (every intermediate rounded to the tensors' dtype).  n elements, dtype FW_DT_BF16 / FW_DT_F32;
 * out may alias latents.  dev_params (optional, device float[2] = {cfg_scale, dsigma}) overrides the two host values, so a
 * captured HIP graph replays with per-step values.
 */
int fw_cfg_euler_step(const void* pos, const void* neg, const void* latents, void* out, int64_t n, int dtype,
                      float cfg_scale, float dsigma, const float* dev_params, void* stream);

/*
 * out[n] = act( sum_k x[k]*W[n,k] + bias[n] ), all fp32, M = 1 (time embeddings:
 * DIT21:393-399, M21:119-121, vggt/models/vggt.py:126-130 fp32 island).
 */
int fw_gemv_f32(const float* x, const float* W, int64_t ldw, const float* bias, float* out,
                int N, int K, int act_in_silu, int act_out, void* stream);

/*
 * out[0:dim/2] = cos(t * 10000^(-i/(dim/2))), out[dim/2:] = sin(...), computed in fp64 then cast to fp32
 * (DIT21:73-77, wan/modules/model.py:17-27).  t is read from device memory as t_dtype (bf16 or f32).
 */
int fw_sinusoid(const void* t, int t_dtype, float* out, int dim, void* stream);

/*
 * Patchify gather: latents x[Cx][F][H2][W2] and y[Cy][F][H2][W2] (bf16 or f32, dtype) ->
 * P[L][Kpad] bf16 with L = F*(H2/2)*(W2/2), column (c*4 + dy*2 + dx) = in[c][f][2h+dy][2w+dx], zero pad
 * to Kpad (Conv3d k=s=(1,2,2) as a GEMM, DIT21:386,424-435; channel concat M21:125-126).
 */
int fw_patchify(const void* x, int Cx, const void* y, int Cy, int dtype,
                uint16_t* P, int64_t ldp, int F, int H2, int W2, void* stream);

/*
 * Unpatchify scatter: head output Hd[L][64] (f32) -> out[16][F][2h][2w] (out_dtype), channel order
 * (x y z c) of DIT21:437-442.
 */
int fw_unpatchify(const float* Hd, int64_t ldh, void* out, int out_dtype, int F, int Hh, int Ww, void* stream);

/*
 * VGGT token assembly (vggt/models/aggregator.py:261-306): tokens[S][P][C] f32 with P = n_special + hw:
 * rows [0,n_special) <- special[(s==0?0:1)][n_special][C] (camera token then register tokens),
 * rows [n_special, P) <- patch[s*hw + i][C] (bf16, ldp).
 */
int fw_assemble_tokens(const uint16_t* patch, int64_t ldp, const float* special, float* tokens,
                       int S, int hw, int n_special, int C, void* stream);

/*
 * y[rows][C] (bf16) = x[rows][C] (f32): the DiT->VGGT bridge feeds the raw residual stream to
 * projection_head (M21:170-173, vggt/models/vggt.py:123).
 */
int fw_cast_f32_bf16(const float* x, int64_t ldx, uint16_t* y, int64_t ldy, int rows, int C, void* stream);

/*
 * Wan2.2 control adapter (FantasyWorld/diffsynth_wan22/models/wan_video_camera_controller.py:8-44, called from
 * WanModel.patchify, diffsynth_wan22/models/wan_video_dit.py:390-396): PixelUnshuffle(8) + Conv2d(k=2,s=2) as a GEMM over
 * 16x16 pixel patches.  in [C][F][16*Hh][16*Ww] (bf16 or f32) -> P[L][ldp] bf16, column (c*64 + dy*8 + dx)*4 + ky*2 + kx.
 */
int fw_control_patchify(const void* in, int dtype, uint16_t* P, int64_t ldp, int C, int F, int Hh, int Ww, void* stream);

/*
 * im2col of a 3x3 / pad 1 / stride 1 Conv2d on token-major activations (ResidualBlock,
 * wan_video_camera_controller.py:64-76): x [F*Hh*Ww][C] bf16 (ldx) -> out [L][9*C] (ldo), column c*9 + ky*3 + kx.
 */
int fw_im2col3x3(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int C, int F, int Hh, int Ww, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * VGGT geometry heads (VGGT._head_predction, FantasyWorld/vggt/models/vggt.py:134-154): the pieces around
 * fw_gemm_bf16.  Feature maps are channels-last matrices [frames*H*W][C] in bf16, C a multiple of 8, 16-byte aligned rows;
 * a k x k (x k) convolution is fw_im2col followed by fw_gemm_bf16 with the weight flattened tap-major ([N][kt][kh][kw][C]).
 * ------------------------------------------------------------------------------------------------------------- */

/*
 * Gather for nn.Conv2d 3x3 (stride 1 or 2, padding 1: vggt/heads/dpt_head.py:82-87, 352-397, 412-428) and CausalConv3d
 * (wan/modules/vae_modified.py:17-36; kernels (3,1,1) and (3,3,3)).  x [T*H*W][C] -> out rows (t - t0, yo, xo) for output
 * frames t0..t0+nt-1, column ((dt*kh + dy)*kw + dx)*C + c = x[t + dt - (kt-1)][yo*sh + dy - ph][xo*sw + dx - pw][c], zero
 * outside the volume (causal in time; ph / pw = k/2 gives 'same', 0 the patch embedding Conv3d(k = s = (1,2,2)) of
 * CameraPoseEncoder, pose_adaptor_ac3d.py:40-41).  relu_in applies ReLU to the gathered values.  up = 2: the convolution runs
 * on the nearest-neighbour x2 up-sampling of x (Resample 'upsample2d/3d' of the Wan VAE, diffsynth_wan21/models/
 * wan_video_vae.py:92-99: nn.Upsample(scale 2, 'nearest-exact') + Conv2d 3x3), gathered on the fly.
 */
int fw_im2col(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int C, int T, int H, int W, int kt, int kh, int kw,
              int sh, int sw, int ph, int pw, int up, int t0, int nt, int relu_in, void* stream);

/*
 * The same convolutions as ONE kernel (implicit GEMM): out[(t - t0, yo, xo)][n] = epi(sum_{tap, c} x[..tap..][c] * Wt[n][tap*C + c])
 * with fw_im2col's geometry and fw_gemm_bf16's epilogue (bias, act, per-column affine, residual with `ldr`, bf16 / f32 output).
 * The gathered matrix is never written: the k-slab (64 channels of one tap) of a tile row is DMA'd to LDS straight from the row of
 * x that tap points at (128 B of zeros outside the volume), so the convolution reads x (L2-resident neighbours) instead of
 * writing and re-reading taps x C values per pixel.  Bit-identical to fw_im2col + fw_gemm_bf16 (same kernel, same k-order).
 * Requires C % 64 == 0 (the feature maps of the geometry heads / VAE decoder are padded that way), T <= 255 and
 * (Ho-1)*sh, (Wo-1)*sw <= 4095; Wt [N][ldw >= kt*kh*kw*C] tap-major as for fw_im2col.  nn.Conv2d / CausalConv3d call sites:
 * vggt/heads/dpt_head.py:82-87, 352-397, 412-428; wan/modules/vae_modified.py:17-36; diffsynth_wan21/models/wan_video_vae.py:92-99.
 */
int fw_conv_gemm_bf16(const uint16_t* x, int64_t ldx, int C, int T, int H, int W, int kt, int kh, int kw,
                      int sh, int sw, int ph, int pw, int up, int t0, int nt,
                      const uint16_t* Wt, int64_t ldw, void* out, int64_t ldo, int out_dtype, int N,
                      const float* bias, int act, const float* g1, const float* g0,
                      const void* res, int64_t ldr, int res_dtype, void* stream);

/* F.interpolate(mode="bilinear", align_corners=True) (custom_interpolate, vggt/heads/dpt_head.py:538-566):
 * x [N*h*w][C] -> out [N*H*W][C]. */
int fw_resize_bilinear(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int N, int h, int w, int H, int W, int C,
                       void* stream);

/* SiLU(RMS_norm(x)) over channels (ResidualBlock_Half, wan/modules/vae_modified.py:39-54, 201-203):
 * out = silu(x / max(|x|_2, 1e-12) * sqrt(c_true) * gamma[c]); channels [c_true, C) are padding (zero in, zero out).
 * silu = 0: the bare RMS_norm of the VAE's AttentionBlock (diffsynth_wan21/models/wan_video_vae.py:246,256). */
int fw_chan_rmsnorm_silu(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int64_t rows, int C, int c_true,
                         const float* gamma, int silu, void* stream);

/* nn.ConvTranspose2d with kernel = stride = k (vggt/heads/dpt_head.py:68-81) after its GEMM:
 * y[(n, yy, xx)][(dy*k + dx)*C + c] -> out[(n, yy*k + dy, xx*k + dx)][c]. */
int fw_depth_to_space(const uint16_t* y, int64_t ldy, uint16_t* out, int64_t ldo, int N, int h, int w, int k, int C, void* stream);

/* x[(n, p)][c] += table[p][c] (fp32 [hw][C]): UV positional embedding (_apply_pos_embed, vggt/heads/dpt_head.py:262-283). */
int fw_add_table(uint16_t* x, int64_t ldx, const float* table, int64_t rows, int hw, int C, void* stream);

/* Resample 'upsample3d' (wan/modules/vae_modified.py:121-124): y[(i, p)][j*C + c] -> out[(2i + j, p)][c], j in {0, 1}. */
int fw_unfold_time2(const uint16_t* y, int64_t ldy, uint16_t* out, int64_t ldo, int n, int hw, int C, void* stream);

/* out = relu?(a + b) on contiguous bf16 tensors of n elements (b may be NULL): FeatureFusionBlock's skip sum, which the
 * following ResidualConvUnit's in-place ReLU rewrites (vggt/heads/dpt_head.py:440, 517-522). */
int fw_add_act(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n, int relu, void* stream);

/* CameraHead.trunk_fn modulation (vggt/heads/camera_head.py:124-128): out = gate * (LN(x) * (1 + scale) + shift) + x with
 * mod[row] = shift | scale | gate ([rows][3C]), LayerNorm without affine, fp32. */
int fw_adaln_rows(const float* x, const float* mod, float* out, int rows, int C, float eps, void* stream);

/* activate_head / activate_pose (vggt/heads/head_act.py:11-33, 61-125) on y [rows][n] fp32:
 * mode 0 "exp": pts[rows][n-1] = exp(.), conf = 1 + exp(y[:, n-1]); mode 1 "inv_log": pts = sign(v) expm1(|v|), same conf;
 * mode 2 "pose": pts[rows][n] = y with ReLU on columns >= 7 (conf unused). */
int fw_head_activation(const float* y, int64_t rows, int n, int mode, float* pts, float* conf, void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * CameraPoseEncoder (FantasyWorld/diffsynth_wan21/models/pose_adaptor_ac3d.py:83-118, called once per
 * generation through CameraConditionModel.get_pose_fea, camera_control.py:233-234): the pieces around fw_gemm_bf16.
 * ------------------------------------------------------------------------------------------------------------- */

/* nn.PixelUnshuffle(r) (pose_adaptor_ac3d.py:26,91): in [F][H][W][C] (bf16 or f32, the Pluecker embedding as the caller
 * holds it) -> out [(f, y, x)][c*r*r + dy*r + dx] = in[f][y*r + dy][x*r + dx][c], bf16 rows (ldo). */
int fw_pixel_unshuffle(const void* in, int dtype, uint16_t* out, int64_t ldo, int F, int H, int W, int C, int r, void* stream);

/* nn.GroupNorm(groups, C) (+ ReLU) on channels-last rows [frames*hw][C] (pose_adaptor_ac3d.py:30-40): statistics per
 * (frame, group) over hw x C/groups values, biased variance, affine w/b per channel. */
int fw_group_norm_rows(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int frames, int hw, int C, int groups,
                       const float* w, const float* b, float eps, int relu, void* stream);

/* CameraPoseEncoder.compress_time (pose_adaptor_ac3d.py:61-76): rows [frames*hw][C] -> [frames'*hw][C]; an odd frame count keeps
 * frame 0 and averages the rest in pairs (frames' = 1 + (frames-1)/2), an even one averages all pairs. */
int fw_time_avg_pool(const uint16_t* x, int64_t ldx, uint16_t* out, int64_t ldo, int frames, int hw, int C, void* stream);

/* out = act(x) (FW_ACT_*) on a contiguous bf16 tensor of n elements: the GELU between LayerNorm and Linear in
 * CameraPoseEncoder.fc (pose_adaptor_ac3d.py:43-48). */
int fw_activation(const uint16_t* x, uint16_t* out, int64_t n, int act, void* stream);

/* Row softmax for the Wan VAE's AttentionBlock (diffsynth_wan21/models/wan_video_vae.py:235-273: one head of width 384 over the
 * h*w positions of a frame, outside the flash kernel's head sizes): out[r][c] = softmax_c(s[r][c] * scale) for c < cols, 0 for
 * cols <= c < cols_pad; s fp32 (lds), out bf16 (ldo).  QK^T and PV around it are fw_gemm_bf16 calls. */
int fw_softmax_rows(const float* s, int64_t lds, uint16_t* out, int64_t ldo, int rows, int cols, int cols_pad, float scale,
                    void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * fp8 linear: AutoWrappedLinear.fp8_linear, FantasyWorld/diffsynth_wan22/vram_management/layers.py:115-151 --
 * the only fp8 definition the reference has (OCP e4m3fn, gfx950's native fp8).
 * ------------------------------------------------------------------------------------------------------------- */

/* raw = 0: scale[m] = max(bf16(max_k |x[m][k]| / 448), 1); q[m][k] = e4m3(x[m][k] / (scale[m] + 1e-8))   (layers.py:126-136)
 * raw = 1: q = e4m3(x), scale untouched                                                                  (weight cast, layers.py:137)
 * x bf16 [M][K] (ldx), q bytes [M][K] (ldq), round-to-nearest-even. */
int fw_fp8_quant_rows(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, float* scale, int M, int K, int raw, void* stream);

/* Row-parallel fp8 linears under head / FFN-column tensor parallelism (round 6): a rank holds a K-slice [M][K_local] of the
 * activation, but scale_a[m] of layers.py:126-133 is a property of the FULL row.  fw_row_absmax: out[m] = max_k |x[m][k]| of the
 * local slice (fp32; exact, the inputs are bf16) -- all-reduced (MAX) across the ranks by the caller; fw_fp8_quant_rows_amax: the
 * raw = 0 quantiser above with the row maximum SUPPLIED (amax[m] over the full row), so that every rank divides by the same scale. */
int fw_row_absmax(const uint16_t* x, int64_t ldx, int rows, int width, float* out, void* stream);
int fw_fp8_quant_rows_amax(const uint16_t* x, int64_t ldx, uint8_t* q, int64_t ldq, const float* amax, float* scale, int M, int K,
                           void* stream);

/* ---------------------------------------------------------------------------------------------------------------
 * fp8 attention (BASELINE.json configs[4]: "CDNA4 fp8 attention + FFN"), head_dim 128 = the DiT self-attention.
 * PARITY UNPINNED: the reference has no fp8 attention (its fp8 entry is the nn.Linear swap above), so these semantics are this
 * library's own: Q8 = e4m3(q * softmax_scale * log2(e) * 2^q_exp), K8 = e4m3(k) (both: fw_qk_prep, then fw_fp8_quant_rows with
 * raw = 1), Vt8 = fw_v_transpose_fp8(v); scores in fp32 (scaled MFMA, the 2^q_exp undone by the operand's block scale), online
 * softmax in fp32, O accumulated in fp32, written as bf16.  P (round 6): the e4m3 BYTE of the probability is
 * round_to_nearest_even(8 (s - m + 7) + 56), saturated at 0 -- the e4m3 encoding of 2^(s - m + 7) with 2^f taken as 1 + f inside a
 * binade (exact at the binade ends, at most 6.1 % high in between; the row sum is taken from the same bytes, so a constant factor
 * cancels) -- one integer conversion per score instead of an exponential; FW_ATTN_VAR = 12 keeps P = e4m3(2^(s - m + 7)) for the A/B,
 * and views of 4 GiB or more (served by the round-2..4 ping-pong kernel) still use it.
 * m is a shift by a WHOLE NUMBER OF BINADES that keeps the row's largest byte in (104, 120] (set from the first tile, it moves only
 * when a later score would pass 2^8.06): the byte -> value map is exponential from binade to binade only, so with whole-binade shifts
 * the probabilities of a row are the same numbers up to one common power of two whatever the shift's history -- the kernel agrees with
 * the CPU statement of these semantics (oracle/ref_ops.py::attention_fp8: true row maximum) to the bf16 rounding of its output
 * (1.6-1.9e-3 rel-L2).  head_dim 128 (DiT self-attention; the bicross attention on heads zero-padded from 96) and 64 (VGGT).
 * Opt-in (FusionEngine(fp8_attention=True)).
 * ------------------------------------------------------------------------------------------------------------- */

/* V [batch][Lk][heads*hd] bf16 (row stride ldv, batch stride bsv, elements) -> Vt8 [batch][heads][hd_out][lkp] e4m3 bytes,
 * lkp % 64 == 0, keys >= Lk zero; inside each 64-key tile the keys sit in the order the PV operand of fw_attention_fp8 reads them.
 * hd_out (0 = hd): rows per head of Vt8; rows hd .. hd_out-1 are written as zeros -- a head_dim-96 operand (the bicross attention)
 * laid out for the head_dim-128 kernel. */
int fw_v_transpose_fp8(const uint16_t* V, int64_t ldv, int64_t bsv, uint8_t* Vt8, int64_t lkp, int batch, int heads, int hd, int Lk,
                       int hd_out, void* stream);

/* The same layout from a V that is ALREADY e4m3 (one byte per element, raw cast of the bf16 V by fw_fp8_quant_rows(raw = 1)): what a
 * rank holds after the sequence shard's head exchange has carried q | k | v as bytes (fantasy_world_amd/parallel.py).  A pure byte
 * gather: fw_v_transpose_e4m3(cast(V)) == fw_v_transpose_fp8(V) bit for bit.  ldv / bsv in bytes (= elements), % 8 == 0. */
int fw_v_transpose_e4m3(const uint8_t* V8, int64_t ldv, int64_t bsv, uint8_t* Vt8, int64_t lkp, int batch, int heads, int hd, int Lk,
                        int hd_out, void* stream);

/* fw_qk_prep / fw_qk_prep_tp writing e4m3 instead of bf16 (round 6: removes the two cast passes between the q/k pass and
 * fw_attention_fp8): x is READ ONLY; out8[r][c] = e4m3(bf16(result[r][c])) -- the value fw_qk_prep would have stored, rounded to
 * bf16 first, then cast raw -- so fw_qk_prep_fp8(x) == fw_fp8_quant_rows(fw_qk_prep(x), raw = 1) bit for bit.  row_sumsq /
 * norm_width as in fw_qk_prep_tp (NULL / 0: the statistic is taken over this call's width).  ld8 in bytes, % 8 == 0.
 * head_stride8 (0 = head_dim): bytes per head in out8; > head_dim pads every head with zeros ([head_dim, head_stride8)) -- the
 * head_dim-96 q / k of the bicross attention laid out for the head_dim-128 kernel (zeros add nothing to q k^T). */
int fw_qk_prep_fp8(const uint16_t* x, int64_t ldx, int rows, int heads, int head_dim,
                   int norm_mode, const float* norm_w, const float* norm_b, float eps,
                   int rope_mode, const float* rope_tab, int tab_rows, float out_scale,
                   const float* row_sumsq, int norm_width, uint8_t* out8, int64_t ld8, int head_stride8, void* stream);

/* O[b][q][h*head_dim + d] = softmax_k(Q K^T) V per (batch, head), head_dim 128 or 64; strides of Q8 / K8 in BYTES (= elements), of O in
 * bf16 elements. */
int fw_attention_fp8(const uint8_t* Q8, int64_t ldq, int64_t bsq, const uint8_t* K8, int64_t ldk, int64_t bsk,
                     const uint8_t* Vt8, int64_t lkp, uint16_t* O, int64_t ldo, int64_t bso,
                     int batch, int heads, int head_dim, int Lq, int Lk, int q_exp, void* stream);

/* C[M][N] = epi((A[M][K] W[N][K]^T) * scale_a[m] + bias[n])  -- torch._scaled_mm(xq, wq^T, scale_a, 1, bias, out_dtype)
 * (layers.py:141-148), fp32 accumulation; A, W e4m3 bytes (K % 64 == 0), bias fp32 holding bf16-rounded values (or NULL).
 * The epilogue after the scaled product is fw_gemm_bf16's (act -> per-column affine g1/g0 -> + residual -> out_dtype), so the
 * fp8 linear drops into every place the engine calls a bf16 linear: this is the module swap enable_vram_management performs
 * (diffsynth_wan21/vram_management/layers.py:145-166, module_map = {nn.Linear: AutoWrappedLinear}) with the surrounding
 * gate / residual arithmetic of the block fused in.  M >= 2048, N >= 1024, K % 128 == 0: 256x256x128 ping-pong kernel on
 * v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (the fp8-rate instruction of gfx950); otherwise a 128x128 kernel on
 * v_mfma_f32_32x32x16_fp8_fp8. */
int fw_gemm_fp8(const uint8_t* A, int64_t lda, const uint8_t* W, int64_t ldw, const float* scale_a,
                void* C, int64_t ldc, int out_dtype, int M, int N, int K,
                const float* bias, int act, const float* g1, const float* g0,
                const void* res, int64_t ldr, int res_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FW_MI355X_H */
