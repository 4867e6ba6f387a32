/* hcp_mi355x.h — C ABI of libhcp_mi355x.so: the MI355X (gfx950) kernels behind HCP-Diffusion's fine-tuning hot path.
 *
 * The reference (IrisRainbowNeko/HCP-Diffusion v0.9.1) is pure Python; its hot path
 *   Trainer.train_one_step            hcpdiff/train_ac.py:467-504
 *   -> TEUnetWrapper.forward          hcpdiff/models/wrapper.py:14-30   (self.unet(...).sample, :29)
 *   -> LoraPatchContainer.forward     hcpdiff/models/lora_base_patch.py:20-35
 * bottoms out in torch / diffusers ops.  Each entry point below replaces one of those ops; the comment on each
 * names the reference call it stands in for.  Binding a maintainer would add: ctypes, see INTEGRATION.md.
 *
 * Conventions: every function returns 0 on success, <0 on error (message: hcp_last_error(), thread-local);
 * raw DEVICE pointers, explicit shapes/strides (in elements), a hipStream_t; no allocation (workspace sizes are
 * queryable), no stream synchronisation, re-entrant, callable from any thread (autograd runs backward on its own).
 * bf16 tensors are raw uint16 bit patterns; "NHWC" = channels-last [B,H,W,C] == token-major [B,H*W,C].
 */
#ifndef HCP_MI355X_H
#define HCP_MI355X_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hcpStream_t; /* == hipStream_t */

const char* hcp_last_error(void);
int hcp_is_emulated(void); /* 0 for the product library */
/* Revision of this header.  hcp_abi_version() returns the value the library was built with; a binding made for another revision must
 * refuse to call it (hcp_diffusion_amd/_lib.py does).  1: rounds 1-5.  2: round 6 — workspace arguments on the LoRA weight-gradient
 * entry points (slabs + ordered reduce instead of fp32 atomics), 152-byte grouped descriptors; earlier in-place argument insertions
 * (ldt on hcp_gemm_lora_bf16 / hcp_gemm_geglu_bwd_bf16, l_lo, ldu / ldt) are covered by the same bump.  3: round 6 — the (hi | lo)
 * residual stream: residual_lo / D_lo on hcp_gemm_bf16 / hcp_gemm_lora_bf16, x_lo / addend_lo / dx_lo on hcp_layernorm_fwd / _bwd; the
 * GEGLU-forward output gact on hcp_gemm_bf16 / hcp_gemm_lora_bf16. */
#define HCP_ABI_VERSION 3
int hcp_abi_version(void);
/* Device self-check of the fp32 atomic path (no reference counterpart: the reference's sums are torch's).  workgroups x 256 threads add
 * small integers into line[16] and into bucket[i * stride], i < nb (both cleared here first); exact expected values:
 * line[j] = workgroups * 16 * ((j & 3) + 1), bucket[i * stride] = sum of ((t & 3) + 1) over (w, t) with (37 w + 101 t) % nb == i. */
int hcp_selfcheck_atomics(float* line, float* bucket, int nb, int stride, int workgroups, hcpStream_t stream);

/* D[M,N] = alpha*(A[M,K] B[N,K]^T + A2[M,K2] B2[N,K2]^T) + bias[N] + rowbias[m/rows_per_group, :] + residual[M,N]
 * Replaces nn.Linear / 1x1 nn.Conv2d forward and input-gradient of the UNet, and the LoRA container's
 * torch.mm(x, (W_host + alpha*W_up@W_down)^T) + bias (lora_layers_patch.py:50-57, lora_base_patch.py:68-74):
 * (A2,B2) is the rank-r side path (x W_down^T, alpha*W_up) appended to the reduction. */
int hcp_gemm_bf16(const void* A, int lda, const void* B, int ldb, void* D, int ldd, int M, int N, int K, const void* A2,
                  int lda2, const void* B2, int ldb2, int K2, const float* bias, const float* rowbias, int rowbias_ld,
                  int rows_per_group, const void* residual, int ldr, const void* residual_lo, void* D_lo, void* gact, float alpha,
                  int out_f32, void* workspace, size_t workspace_bytes, hcpStream_t stream);
/* (hi | lo) residual stream, ABI 3: residual_lo / D_lo (both optional, bf16, leading dimensions ldr / ldd).  A transformer block's
 * residual stream x is carried as hi = bf16(x) and lo = bf16(x - hi): the epilogue adds residual + residual_lo in fp32 and writes
 * D = bf16(v), D_lo = bf16(v - D) — the 16 mantissa bits the reference's LoRA layers keep on that stream by returning
 * mm(x, W^T) [bf16 under autocast] + bias [fp32] = fp32 (lora_layers_patch.py:50-57).  NULL, NULL = the plain bf16 epilogue.
 * GEGLU forward, ABI 3: gact (optional, bf16 [M, N/2]).  D = (h | g) is the output of diffusers' GEGLU projection (FeedForward.net[0].proj,
 * cfgs/unet_struct.txt:27-30): gact = bf16(h * gelu(g)) is formed in the same epilogue from the fp32 values (the stand-alone hcp_geglu_fwd
 * pass and one rounding disappear: the reference's LoRA layer hands GEGLU an fp32 (h | g) too).  Needs a contiguous bf16 D (ldd = N),
 * N % 16 == 0, no residual / row bias / alpha; where the dispatched tile cannot pair the halves the library runs hcp_geglu_fwd itself. */
/* Fused LoRA linear, forward and input-gradient: T = A L^T (rank slot 32, written to Tout if non-NULL),
 * D = A B^T + T E^T + bias + residual in ONE launch (the block accumulates its T tile from the A tiles it streams);
 * deep-K / small-M shapes run as two launches (T GEMM, then split-K K-extension GEMM) and then require Tout.
 * ldt = 32: T is rounded to bf16 (Tout [M,32]).  ldt = 64: "split" T — the fp32 accumulator leaves as T_hi = bf16(T) and
 * T_lo = bf16(T - T_hi), the product takes both (D += T_hi E^T + T_lo E^T) and Tout [M,64] = (T_hi | T_lo) for hcp_lora_wgrad*:
 * the rank-r intermediate keeps 16 mantissa bits, as the reference's merged-weight form never rounds it at all
 * (the two-launch form keeps the rounded T and zeroes the residual half).
 * Replaces LoraPatchContainer.forward's weight merge + mm (lora_base_patch.py:20-35,61-74). */
int hcp_gemm_lora_bf16(const void* A, int lda, const void* B, int ldb, const void* L, const void* E, void* Tout, int ldt, void* D,
                       int ldd, int M, int N, int K, const float* bias, const void* residual, int ldr, const void* residual_lo, void* D_lo,
                       void* gact, void* workspace, size_t workspace_bytes, hcpStream_t stream);
/* FF-out input-gradient with the GEGLU backward in its epilogue: dY_ff = A B^T (+ the LoRA side path as hcp_gemm_lora_bf16's backward
 * form: L = W_up^T, E = alpha W_down^T, Tout = dY W_up; L = E = NULL for a plain host) is never written; with (h | g) = HG[M, 2F] saved
 * by the forward, DHG[m, n] = dY_ff gelu(g), DHG[m, F + n] = dY_ff h gelu'(g).  Replaces the dX GEMM of FeedForward.net[2] + the
 * GEGLU backward pass (diffusers GEGLU in BasicTransformerBlock.ff, reference cfgs/unet_struct.txt:27-33 / autograd of F.gelu). */
int hcp_gemm_geglu_bwd_bf16(const void* A, int lda, const void* B, int ldb, const void* L, const void* E, void* Tout, int ldt,
                            const void* HG, void* DHG, int M, int F, int K, void* workspace, size_t workspace_bytes, hcpStream_t stream);
/* fp32 split-K scratch (optional: workspace may be NULL, then small-M problems run unsplit). */
size_t hcp_gemm_workspace_bytes(int M, int N);

/* 3x3 convolution over NHWC bf16 as an implicit GEMM.  mode 0: forward, Wp = [Cout][3][3][C1+C2];
 * mode 1: data gradient, X1 = dY, Wp = [Cin][3][3][Cout].  pad 1 = padding 1 (every conv of the UNet); pad 0 (forward only) = taps
 * 0..+2 with zeros past the bottom/right edge, i.e. F.pad(x,(0,1,0,1)) + padding 0: the VAE encoder's Downsample2D.
 * Options: stride 1|2, nearest-2x upsampled source,
 * second source tensor (channel concat), bias, per-sample row bias (time embedding), residual, and a rank-32 K-extension
 * D += A2[M,32] B2[Cout,32]^T — the conv (LoCon) LoRA side path T (alpha W_up)^T, T = conv3x3(x, W_down)
 * (reference lora_layers_patch.py:64-100 merges W + alpha * einsum(W_up, W_down) into the conv weight instead).
 * Replaces F.conv2d in diffusers ResnetBlock2D / Downsample2D / Upsample2D (reference cfgs/unet_struct.txt:92-114,390-393). */
int hcp_conv3x3_bf16(const void* X1, int C1, const void* X2, int C2, int B, int Hs, int Ws, int Ho, int Wo, int mode,
                     int stride, int upsample, int pad, const void* Wp, int Cout, void* D, int ldd, const float* bias,
                     const float* rowbias, int rowbias_ld, const void* residual, int ldr, int out_f32, const void* A2,
                     const void* B2, void* workspace, size_t workspace_bytes, hcpStream_t stream);

/* Fused attention, element (b,n,h,c) at base + b*bs + n*rs + h*D + c; lse[B,H,Nq] = logsumexp(scale*QK^T).
 * Replaces diffusers CrossAttention/AttnProcessor2_0 (SDPA) or xformers (reference train_ac.py:258-260). D in {40,64,80,160}. */
/* key_bias (optional, fp32 [B,Nk], batch stride key_bias_bs): additive bias on the SCALED scores of every head/query —
 * diffusers' encoder_attention_mask -> (1 - mask) * -10000 (the attn_mask the reference passes at models/wrapper.py:22-29).
 * `causal` is a flag word.  bit 0 (self-attention only, Nq == Nk): key k contributes to query q only if k <= q — the CLIP text encoder's
 * causal_attention_mask (cfgs/te_struct.txt CLIPAttention; text-encoder LoRA, cfgs/train/examples/lora_conventional.yaml:14-19). 
 * bit 1 (unmasked problems only): Q already holds Q * scale * log2(e) — the caller folded the factor into the weights of the Q
 * projection (and into the LoRA alpha of that layer), so the score accumulator is the exp2 argument and no multiply precedes the
 * exponential; `scale` still names the softmax scale.  hcp_attention_bwd then returns dQ as the gradient w.r.t. THAT tensor. */
int hcp_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H, int Nq, int Nk, int D,
                      long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                      const float* key_bias, long key_bias_bs, int causal, hcpStream_t stream);
/* workspace (optional, 2*B*Nk*H*D floats): lets short-key problems split the dK/dV query loop across workgroups */
int hcp_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                      float* delta_ws /* [B,H,Nq] */, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int D,
                      long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                      const float* key_bias, long key_bias_bs, int causal, void* workspace, size_t workspace_bytes,
                      hcpStream_t stream);


/* GroupNorm (+SiLU) over NHWC; stats[B,G,2] = (mean, rstd).  Replaces F.group_norm + SiLU
 * (unet_struct.txt:13,93,97,929).  Backward returns dx only (affine parameters frozen). */
size_t hcp_groupnorm_workspace_bytes(int B, int HW, int C, int G);
int hcp_groupnorm_silu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, void* workspace,
                           int B, int HW, int C, int G, float eps, int silu, hcpStream_t stream);
int hcp_groupnorm_silu_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* stats,
                           const void* addend /* optional residual-path gradient, added in the same pass */, void* dx,
                           void* workspace, int B, int HW, int C, int G, int silu, hcpStream_t stream);

/* LayerNorm over the last dim; stats[M,2] = (mean, rstd).  Replaces F.layer_norm (unet_struct.txt:44-46).
 * x_lo / addend_lo / dx_lo (optional, ABI 3): the lo images of a (hi | lo) residual stream (see hcp_gemm_bf16) — the row normalised is
 * x + x_lo, the gradient arriving on the residual path is addend + addend_lo, and the result leaves as dx = bf16(g), dx_lo = bf16(g - dx). */
int hcp_layernorm_fwd(const void* x, const void* x_lo, const float* gamma, const float* beta, void* y, float* stats, int M, int C,
                      float eps, hcpStream_t stream);
int hcp_layernorm_bwd(const void* x, const void* x_lo, const void* dy, const float* gamma, const float* stats, const void* addend,
                      const void* addend_lo, void* dx, void* dx_lo, int M, int C, hcpStream_t stream);

/* y[M,F] = h[:, :F] * gelu(h[:, F:])   (diffusers GEGLU, unet_struct.txt:28-30) */
int hcp_geglu_fwd(const void* h, void* y, long M, int F, hcpStream_t stream);
int hcp_geglu_bwd(const void* h, const void* dy, void* dh, long M, int F, hcpStream_t stream);

int hcp_add_bf16(const void* a, const void* b, void* out, long n, hcpStream_t stream);
int hcp_copy2d_bf16(const void* src, int sld, void* dst, int dld, long M, int C, hcpStream_t stream); /* skip concat/split */
/* torch.cat([h, skip], dim=1) of the up-block ResnetBlock2D and its gradient split in one launch: joint [M][c1+c2] <-> a [M][c1], b [M][c2];
 * split = 0 writes joint, split = 1 writes a and b (c1, c2 multiples of 8). */
int hcp_concat2_bf16(void* a, int c1, void* b, int c2, void* joint, long M, int split, hcpStream_t stream);
int hcp_silu_fwd(const void* x, void* y, long n, hcpStream_t stream);
int hcp_silu_bwd(const void* x, const void* dy, void* dx, long n, hcpStream_t stream);
int hcp_nchw_to_nhwc_bf16(const void* src, int src_is_f32, void* dst, int B, int C, int HW, int Cpad, hcpStream_t stream);
int hcp_nhwc_to_nchw_f32(const float* src, float* dst, int B, int C, int HW, int Csrc, hcpStream_t stream);
int hcp_upsample2x_bwd(const void* dup, void* dx, int B, int H, int W, int C, hcpStream_t stream);

/* ---- host-layer weight gradients (full fine-tuning: cfgs/train/examples/DreamBooth.yaml:6-10 trains every UNet
 * parameter; autograd's dW of nn.Linear / nn.Conv2d [ext]).  fp32 `+=` into the gradient buffer (atomics). ---- */
/* dW[N,K] += dY[M,N]^T X[M,K]; workspace = optional fp32 scratch for token-split partial sums */
int hcp_wgrad_linear_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw, int M, int N, int K,
                          float* workspace, size_t workspace_bytes, hcpStream_t stream);
/* dW[Cout][3][3][Cw] += dY^T im2col(X1|X2): same gather as hcp_conv3x3_bf16 (stride, nearest-2x upsample, concat);
 * layout = torch channels_last storage of the diffusers weight [Cout,Cin,3,3] */
int hcp_wgrad_conv3x3_bf16(const void* dY, int ldy, const void* X1, int C1, const void* X2, int C2, float* dW, int Cw, int B,
                           int Hs, int Ws, int Ho, int Wo, int Cout, int stride, int upsample, float* workspace,
                           size_t workspace_bytes, hcpStream_t stream);
/* out[g][n] += sum of rows of group g of Y (bias gradients; per-sample time-embedding row-bias gradient) */
int hcp_colsum_bf16(const void* Y, int ldy, float* out, int ldo, int M, int N, int rows_per_group, hcpStream_t stream);
/* After the optimizer step of a full fine-tune: refresh every layer's bf16 operand copies (row-major + transposed) from
 * the fp32 masters in ONE grouped launch.  pieces = device array of 56-byte descriptors
 * {const float* src; bf16* dst_rm; bf16* dst_tr; int rows, cols, src_ld, rm_ld, tr_ld, tile0, tiles_c; float scale;} sorted by tile0. */
int hcp_pack_piece_bytes(void);
int hcp_pack_weights(const void* pieces, int count, int total_tiles, hcpStream_t stream);
/* norm weight / bias gradients: dgamma[c] += sum dz * xhat, dbeta[c] += sum dz (dz includes the fused SiLU'), stats
 * as saved by the forward entry points */
int hcp_groupnorm_affine_grad(const void* x, const void* dy, const float* gamma, const float* beta, const float* stats,
                              float* dgamma, float* dbeta, int B, int HW, int C, int G, int silu, hcpStream_t stream);
int hcp_layernorm_affine_grad(const void* x, const void* dy, const float* stats, float* dgamma, float* dbeta, int M, int C,
                              hcpStream_t stream);

/* Timesteps(flip_sin_to_cos=True, freq_shift=0) (unet_struct.txt:3): emb[b] = [cos(t f_i) | sin(t f_i)] */
int hcp_timestep_embedding(const long long* timesteps, void* emb, int B, int dim, float max_period, hcpStream_t stream);
/* same for fp32 inputs: SDXL's add_time_proj over added_cond_kwargs["time_ids"] (= crop_info, models/wrapper.py:66) */
int hcp_timestep_embedding_f32(const float* values, void* emb, int B, int dim, float max_period, hcpStream_t stream);
/* DDPMScheduler.add_noise as called by train_ac.py:447 */
int hcp_add_noise(const float* x0, const float* noise, const long long* timesteps, const float* alphas_cumprod, float* xt,
                  int B, long per_sample, hcpStream_t stream);
/* --- VAE encode (AutoencoderKL.encode(...).latent_dist.sample() * scaling_factor, reference
 * data/pair_dataset.py:72-75 and train_ac.py:428-435).  The encoder reuses the conv / GroupNorm / GEMM entry points above; these
 * three finish it: the single-head d=512 mid-block attention runs as GEMM -> hcp_softmax_rows -> GEMM (V transposed by
 * hcp_transpose_bf16), and hcp_vae_latent_sample folds quant_conv (1x1), the logvar clamp [-30,20] and the reparameterised
 * draw (noise == NULL: the mode) into one pass over the fp32 moments [B,2L,hw]. */
int hcp_transpose_bf16(const void* src, void* dst, int batch, int R, int C, hcpStream_t stream);
int hcp_softmax_rows(const float* S, long lds, void* P, long ldp, int M, int N, float scale, hcpStream_t stream);
int hcp_vae_latent_sample(const float* moments, const float* Wq, const float* bq, const float* noise, float* latents, int B, int L,
                          long hw, float scale, hcpStream_t stream);
/* --- text encoder (CLIP, cfgs/te_struct.txt; text-encoder LoRA of cfgs/train/examples/lora_conventional.yaml:14-19): its linears,
 * LayerNorms and (causal) attention are the entry points above; these two are its embedding lookup and MLP activation. */
int hcp_quick_gelu(const void* x, const void* dy, void* out, long n, hcpStream_t stream);   /* dy NULL: forward; else dx */
int hcp_embedding_bf16(const float* token_table, const long long* ids, const float* position_table, const long long* position_ids,
                       void* out, long n, int C, int L, hcpStream_t stream);
/* per-sample loss weights of the reference's timestep-aware criteria (hcpdiff/loss/min_snr_loss.py): kind 0 MinSNRLoss :21-25,
 * 1 SoftMinSNRLoss :31-35, 2 KDiffMinSNRLoss :39-43, 3 EDMLoss :47-52; snr = acp/(1-acp) as in :14-19.  w: float[B]. */
int hcp_snr_loss_weight(const long long* timesteps, const float* alphas_cumprod, float* w, int B, int kind, float gamma,
                        hcpStream_t stream);
/* (criterion(pred, target[, timesteps]) * mask).mean() * weight and its gradient (train_ac.py:506-515); criterion = MSE
 * (reduction none) times the optional per-sample weight sample_weight[B] (null: plain MSELoss) */
int hcp_mse_masked_mean(const float* pred, const float* target, const float* mask, int mask_channels,
                        const float* sample_weight, float* loss, float* grad, int B, int C, int HW, float weight,
                        hcpStream_t stream);

/* out[p,q] (+)= scale * sum_m L[m,p] R[m,q]  (fp32 atomics): the rank-r LoRA weight gradients
 * dW_down = alpha (dY W_up)^T x, dW_up = alpha dY^T (x W_down^T) (autograd of lora_base_patch.py:61-74). */
/* l_lo: 0, or the column offset of the residual half of a split L = (L_hi | L_lo) (hcp_gemm_lora_bf16 with ldt = 64: l_lo = 32): the
 * product is then (L_hi + L_lo)^T R. */
int hcp_lora_wgrad(const void* L, int ldl, int l_lo, const void* R, int ldr, float* out, int ldo, int M, int P, int Q, float scale,
                   int transpose_out, void* workspace, size_t workspace_bytes, hcpStream_t stream);
/* both gradients of one LoRA layer in one launch: grad_down[r,K] += s U^T x ; grad_up[N,r] += s dY^T T  (ldu / ldt: 32, or 64 = split) */
int hcp_lora_wgrad_pair(const void* U, int ldu, const void* x, int ldx, int K, float* grad_down, const void* T, int ldt, const void* dY,
                        int ldy, int N, float* grad_up, int M, int r, float scale, void* workspace, size_t workspace_bytes,
                        hcpStream_t stream);
/* dst[M,2C] bf16 = (bf16(src) | bf16(src - bf16(src))), src [M,C] fp32: the split form of a T / U produced by a GEMM of its own */
int hcp_split_hi_lo_bf16(const float* src, void* dst, long M, int C, hcpStream_t stream);
/* the weight gradients of MANY LoRA layers in two launches: partial tiles, ordered reduce (descriptor layout: csrc/lora.hip
 * WgradGroupDesc, 152 B; total_tiles = sum of qt_down + qt_up; slab_units = sum of blocks * P over the layers with more than one
 * token range: the workspace holds slab_units * 512 bytes; a layer with one range adds its tile itself).
 * No atomics anywhere on this path: every gradient element is summed in a fixed order by one thread (ABI 2; ABI 1 accumulated the
 * partials with fp32 atomics and took no workspace). */
int hcp_lora_wgrad_group_geometry(int M, int K, int N, int P, int target_blocks, int* qt_down, int* qt_up, int* splits,
                                  int* rows_per_split);
int hcp_lora_wgrad_group_desc_bytes(void);
int hcp_lora_wgrad_grouped(const void* descs, int count, int total_blocks, int total_tiles, long slab_units, void* workspace,
                           size_t workspace_bytes, hcpStream_t stream);
/* fp32 LoRA factors -> the four bf16 operand layouts, all layers in one launch (descs: device array, 80 B each, see
 * hcp_lora_pack_desc_bytes(): {const float* w_down; const float* w_up; bf16* ad; bf16* adt; bf16* bu; bf16* but; int K; int N; int r;
 * float alpha; int slot0; int n0; int Ntot; int bu_ld;} — slot0 / n0 / Ntot place a layer inside operand images shared by a fused
 * group; bu_ld = row stride of bu in elements (0 = 32); adt / but may be NULL when that image is not wanted) */
int hcp_lora_pack(const void* descs, int count, hcpStream_t stream);
int hcp_lora_pack_desc_bytes(void);

/* accelerator.clip_grad_norm_ + torch.optim.AdamW.step + zero_grad (train_ac.py:485-494) on one flat fp32 bucket. */
int hcp_sumsq_f32(const float* g, long n, float* out, hcpStream_t stream);
int hcp_adamw_clip_fused(float* p, float* g, float* m, float* v, long n, const float* lr, float beta1, float beta2,
                         float eps, float weight_decay, const float* sumsq, float grad_scale, float max_norm, int* step,
                         hcpStream_t stream);
/* ModelEMA.update (reference hcpdiff/utils/ema.py:17-27) over a flat bucket: ema <- lerp(ema, p, 1 - decay(step)) */
int hcp_ema_update(float* ema, const float* p, long n, const int* step, float inv_gamma, float power, float decay_max,
                   hcpStream_t stream);
/* bf16 wire formats of the sharded gradient / parameter exchange (what torch DDP's bf16_compress_hook does around its all-reduce,
 * torch/distributed/algorithms/ddp_comm_hooks/default_hooks.py; the reference reaches DDP through accelerate, train_ac.py:117-123):
 * dst_bf16[i] = bf16(src[i] * scale), src cleared when zero_src != 0 (the bucket's zero_grad);  dst_f32[i] = float(src_bf16[i]). */
int hcp_cast_f32_bf16(float* src, void* dst_bf16, long n, float scale, int zero_src, hcpStream_t stream);
int hcp_cast_bf16_f32(const void* src_bf16, float* dst, long n, hcpStream_t stream);

/* One DDIM (eta = 0) sampler step with classifier-free guidance fused in (reference utils/pipe_hook.py:120-140 + scheduler.step):
 * eps = eps2[:n] + g (eps2[n:] - eps2[:n]) when guided (eps2 = UNet output on [uncond ; cond]), else eps2[:n];
 * out = sqrt(a_prev) (x - sqrt(1 - a_t) eps) / sqrt(a_t) + sqrt(1 - a_prev) eps.  fp32, out may alias x. */
int hcp_cfg_ddim_step(const float* x, const float* eps2, float* out, long n, int guided, float guidance_scale,
                      float alpha_cumprod_t, float alpha_cumprod_prev, hcpStream_t stream);

/* (the tuning / ablation hooks of the -DHCP_TOOLS build are declared in include/hcp_mi355x_tools.h: they are not part of this ABI) */

/* ---- data-parallel exchange: RCCL over xGMI on flat buffers (csrc/comm.hip).  Replaces accelerator.backward's DDP gradient
 * all-reduce (reference train_ac.py:117-123,175,482).  One process per GPU; `comm` is an opaque handle owned by the caller;
 * every collective is stream-ordered, allocates nothing, never synchronises and can be captured into a hipGraph.
 * dtype: 0 = fp32, 1 = bf16.  RCCL is dlopen'ed on first use (the copy PyTorch-ROCm already mapped, if any). */
#define HCP_COMM_UNIQUE_ID_BYTES 128
int hcp_comm_unique_id(void* out128);                                   /* rank 0: create the 128-byte rendezvous token */
int hcp_comm_init(int rank, int world, const void* unique_id128, void** comm_out);   /* collective over all ranks */
int hcp_comm_destroy(void* comm);
int hcp_comm_rank(const void* comm);
int hcp_comm_world(const void* comm);
/* buf[i] <- sum over ranks, in place (one call per flat gradient bucket; 1/world is applied by hcp_adamw_clip_fused) */
int hcp_allreduce_flat(void* comm, void* buf, size_t count, int dtype, hcpStream_t stream);
/* recv[0..n) <- sum over ranks of send[rank*n .. rank*n+n)   (send: world*n elements) */
int hcp_reduce_scatter_flat(void* comm, const void* send, void* recv, size_t recv_count, int dtype, hcpStream_t stream);
/* recv[r*n .. r*n+n) <- rank r's send[0..n)   (recv: world*n elements; send may be its own slot of recv) */
int hcp_allgather_flat(void* comm, const void* send, void* recv, size_t send_count, int dtype, hcpStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
