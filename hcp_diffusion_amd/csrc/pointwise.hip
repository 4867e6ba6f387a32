// pointwise.hip — HBM-bound elementwise / layout / loss kernels of the SD fine-tuning inner loop.
//
//  * GEGLU  hidden*gelu(gate)             (diffusers GEGLU, reference cfgs/unet_struct.txt:28-30)
//  * residual adds, SiLU                  (ResnetBlock2D / BasicTransformerBlock skip paths)
//  * NCHW fp32 <-> NHWC bf16 at the UNet boundary (conv_in / conv_out are the only NCHW points)
//  * nearest-2x upsample backward         (Upsample2D, unet_struct.txt:390-393)
//  * sinusoidal timestep embedding        (Timesteps(flip_sin_to_cos=True, freq_shift=0), unet_struct.txt:3)
//  * DDPM add_noise and masked-MSE loss   (reference train_ac.py:437-447, 506-515)
// All kernels are grid-stride over 16-byte (8 x bf16) vectors.
#include "hcp_common.h"

namespace {

constexpr int PW_THREADS = 256;
inline int pw_grid(long nvec) {
    long g = (nvec + PW_THREADS - 1) / PW_THREADS;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

HCP_DEVICE float gelu_erf(float x) { return hcp_gelu_erf(x); }
HCP_DEVICE float gelu_erf_grad(float x) { return hcp_gelu_erf_grad(x); }

HCP_KERNEL(256) geglu_fwd_kernel(const hcp_bf16* h, hcp_bf16* y, long M, int F) {
    const int fv = F / 8;
    const long total = M * fv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / fv; int j = (int)(i - m * fv) * 8;
        hcp_bf16x8 a = *(const hcp_bf16x8*)(h + m * 2 * F + j);
        hcp_bf16x8 g = *(const hcp_bf16x8*)(h + m * 2 * F + F + j);
        hcp_bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (short)hcp_f2bf(hcp_bf2f((unsigned short)a[q]) * gelu_erf(hcp_bf2f((unsigned short)g[q])));
        *(hcp_bf16x8*)(y + m * F + j) = o;
    }
}

HCP_KERNEL(256) geglu_bwd_kernel(const hcp_bf16* h, const hcp_bf16* dy, hcp_bf16* dh, long M, int F) {
    const int fv = F / 8;
    const long total = M * fv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / fv; int j = (int)(i - m * fv) * 8;
        hcp_bf16x8 a = *(const hcp_bf16x8*)(h + m * 2 * F + j);
        hcp_bf16x8 g = *(const hcp_bf16x8*)(h + m * 2 * F + F + j);
        hcp_bf16x8 d = *(const hcp_bf16x8*)(dy + m * F + j);
        hcp_bf16x8 da, dg;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float af = hcp_bf2f((unsigned short)a[q]), gf = hcp_bf2f((unsigned short)g[q]), df = hcp_bf2f((unsigned short)d[q]);
            da[q] = (short)hcp_f2bf(df * gelu_erf(gf));
            dg[q] = (short)hcp_f2bf(df * af * gelu_erf_grad(gf));
        }
        *(hcp_bf16x8*)(dh + m * 2 * F + j) = da;
        *(hcp_bf16x8*)(dh + m * 2 * F + F + j) = dg;
    }
}

HCP_KERNEL(256) add_kernel(const hcp_bf16* a, const hcp_bf16* b, hcp_bf16* o, long nvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        hcp_bf16x8 x = *(const hcp_bf16x8*)(a + i * 8), y = *(const hcp_bf16x8*)(b + i * 8), r;
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = (short)hcp_f2bf(hcp_bf2f((unsigned short)x[q]) + hcp_bf2f((unsigned short)y[q]));
        *(hcp_bf16x8*)(o + i * 8) = r;
    }
}

// mode 0: y = silu(x); mode 1: dx = dy * silu'(x)
HCP_KERNEL(256) silu_kernel(const hcp_bf16* x, const hcp_bf16* dy, hcp_bf16* o, long nvec, int mode) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(x + i * 8), r;
        hcp_bf16x8 d = mode ? *(const hcp_bf16x8*)(dy + i * 8) : hcp_zero8();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float z = hcp_bf2f((unsigned short)v[q]);
            float sg = hcp_sigmoid(z);
            r[q] = (short)hcp_f2bf(mode ? hcp_bf2f((unsigned short)d[q]) * sg * (1.f + z * (1.f - sg)) : z * sg);
        }
        *(hcp_bf16x8*)(o + i * 8) = r;
    }
}

// src NCHW (fp32 or bf16) -> dst NHWC bf16 with channels zero-padded to Cp
HCP_KERNEL(256) nchw_to_nhwc_kernel(const void* src, int src_f32, hcp_bf16* dst, int B, int C, int HW, int Cp) {
    const long total = (long)B * HW * Cp;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % Cp); long r = i / Cp; int hw = (int)(r % HW); int b = (int)(r / HW);
        float v = 0.f;
        if (c < C) {
            size_t s = ((size_t)b * C + c) * HW + hw;
            v = src_f32 ? ((const float*)src)[s] : hcp_bf2f(((const hcp_bf16*)src)[s]);
        }
        dst[i] = hcp_f2bf(v);
    }
}

// src NHWC fp32 [B,HW,Cs] (first C channels used) -> dst NCHW fp32 [B,C,HW]
HCP_KERNEL(256) nhwc_to_nchw_kernel(const float* src, float* dst, int B, int C, int HW, int Cs) {
    const long total = (long)B * C * HW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int hw = (int)(i % HW); long r = i / HW; int c = (int)(r % C); int b = (int)(r / C);
        dst[i] = src[((size_t)b * HW + hw) * Cs + c];
    }
}

// dx[b,y,x,:] = sum of the 2x2 block of dup[b,2y+i,2x+j,:]
HCP_KERNEL(256) upsample2x_bwd_kernel(const hcp_bf16* dup, hcp_bf16* dx, int B, int H, int W, int C) {
    const int cv = C / 8;
    const long total = (long)B * H * W * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % cv) * 8; long r = i / cv; int x = (int)(r % W); r /= W; int y = (int)(r % H); int b = (int)(r / H);
        float acc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = 0.f;
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                hcp_bf16x8 v = *(const hcp_bf16x8*)(dup + (((size_t)b * 2 * H + 2 * y + dyy) * 2 * W + 2 * x + dxx) * C + c);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] += hcp_bf2f((unsigned short)v[q]);
            }
        hcp_bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (short)hcp_f2bf(acc[q]);
        *(hcp_bf16x8*)(dx + (((size_t)b * H + y) * W + x) * C + c) = o;
    }
}

// emb[b, 0:half] = cos(t * f_i), emb[b, half:] = sin(t * f_i), f_i = exp(-ln(max_period) * i / half)
// t is int64 (scheduler timesteps) or, when tf != nullptr, fp32 (SDXL micro-conditioning scalars).
HCP_KERNEL(256) timestep_embedding_kernel(const long long* t, const float* tf, hcp_bf16* emb, int B, int dim, float max_period) {
    const int half = dim / 2;
    const int total = B * half;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int b = i / half, k = i - b * half;
        float f = expf(-logf(max_period) * (float)k / (float)half);
        float a = (tf ? tf[b] : (float)t[b]) * f;
        emb[(size_t)b * dim + k] = hcp_f2bf(cosf(a));
        emb[(size_t)b * dim + half + k] = hcp_f2bf(sinf(a));
    }
}

// x_t = sqrt(acp[t]) * x0 + sqrt(1 - acp[t]) * noise      (NCHW fp32 in, NCHW fp32 out)
HCP_KERNEL(256) add_noise_kernel(const float* x0, const float* noise, const long long* t, const float* acp, float* xt,
                                 int B, long per) {
    const long total = (long)B * per;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int b = (int)(i / per);
        float a = acp[t[b]];
        xt[i] = sqrtf(a) * x0[i] + sqrtf(1.0f - a) * noise[i];
    }
}

// One sampler step with classifier-free guidance, fused (inference / in-training previews; reference utils/pipe_hook.py:120-140:
// noise_pred = uncond + scale (text - uncond), then scheduler.step):
//   eps = eps2[b] + g (eps2[B + b] - eps2[b])            eps2 = the UNet's output on the batch [uncond ; cond] (g = 1, eps2 of B rows: no CFG)
//   x0  = (x - sqrt(1 - a_t) eps) / sqrt(a_t)            (epsilon prediction)
//   out = sqrt(a_prev) x0 + sqrt(1 - a_prev) eps         (DDIM, eta = 0; a_prev = 1 on the final step)
HCP_KERNEL(256) cfg_ddim_kernel(const float* x, const float* eps2, float* out, long n, long cond_off, float guidance, float a_t, float a_prev) {
    const float sa = sqrtf(a_t), sb = sqrtf(1.0f - a_t), pa = sqrtf(a_prev), pb = sqrtf(1.0f - a_prev);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float eu = eps2[i];
        const float e = cond_off ? eu + guidance * (eps2[cond_off + i] - eu) : eu;
        const float x0 = (x[i] - sb * e) / sa;
        out[i] = pa * x0 + pb * e;
    }
}

// Per-sample loss weight from the noise level of the sample's timestep (reference hcpdiff/loss/min_snr_loss.py):
//   snr = acp/(1-acp) (= (alpha/sigma)^2, :14-19), sigma^2 = 1-acp
//   kind 0 MinSNRLoss      min(gamma/snr, 1)                                   (:21-25)
//   kind 1 SoftMinSNRLoss  gamma^3 / (snr^2 + gamma^3)                         (:31-35)
//   kind 2 KDiffMinSNRLoss 4 (gamma snr)^2 / (snr^2 + gamma^2)^2               (:39-43)
//   kind 3 EDMLoss         (sigma^2 + gamma^2) / (snr (sigma gamma)^2)         (:47-52)
HCP_KERNEL(64) snr_weight_kernel(const long long* t, const float* acp, float* w, int B, int kind, float gamma) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float a = acp[t[b]];
    const float alpha = sqrtf(a), sigma = sqrtf(1.0f - a);
    const float r = alpha / sigma;
    const float snr = r * r;
    float v;
    if (kind == 0) { v = gamma / snr; v = v > 1.0f ? 1.0f : v; }
    else if (kind == 1) { const float g3 = gamma * gamma * gamma; v = g3 / (snr * snr + g3); }
    else if (kind == 2) { const float gs = gamma * snr, d = snr * snr + gamma * gamma; v = 4.0f * (gs * gs / (d * d)); }
    else { const float sg = sigma * gamma; v = (sigma * sigma + gamma * gamma) / (snr * (sg * sg)); }
    w[b] = v;
}

// loss += sum((pred-target)^2 * mask * w[b]) * scale ; grad = 2*(pred-target)*mask*w[b]*scale
HCP_KERNEL(256) mse_kernel(const float* pred, const float* target, const float* mask, int mask_c, const float* sw, float* loss,
                           float* grad, int B, int C, int HW, float scale) {
    const long total = (long)B * C * HW;
    float acc = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float m = 1.0f;
        if (mask) {
            int hw = (int)(i % HW); long r = i / HW; int c = (int)(r % C); int b = (int)(r / C);
            m = mask[((size_t)b * mask_c + (mask_c == 1 ? 0 : c)) * HW + hw];
        }
        if (sw) m *= sw[i / ((long)C * HW)];
        float d = pred[i] - target[i];
        acc += d * d * m;
        if (grad) grad[i] = 2.0f * d * m * scale;
    }
    acc = hcp_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) hcp_atomic_add(loss, acc * scale);
}

// dst[m, 0:C] = src[m, 0:C] with independent row strides (channel concat / split of channels-last tensors)
HCP_KERNEL(256) copy2d_kernel(const hcp_bf16* src, int sld, hcp_bf16* dst, int dld, long M, int C) {
    const int cv = C / 8;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long m = i / cv; int c = (int)(i - m * cv) * 8;
        *(hcp_bf16x8*)(dst + m * dld + c) = *(const hcp_bf16x8*)(src + m * sld + c);
    }
}

// CLIP text encoder MLP activation: quick_gelu(x) = x * sigmoid(1.702 x)   (cfgs/te_struct.txt QuickGELUActivation)
HCP_KERNEL(256) quick_gelu_kernel(const hcp_bf16* x, const hcp_bf16* dy, hcp_bf16* out, long nvec) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(x + i * 8);
        hcp_bf16x8 d = dy ? *(const hcp_bf16x8*)(dy + i * 8) : hcp_zero8();
        hcp_bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float f = hcp_bf2f((unsigned short)v[q]);
            const float sg = hcp_sigmoid(1.702f * f);
            o[q] = (short)hcp_f2bf(dy ? hcp_bf2f((unsigned short)d[q]) * (sg + 1.702f * f * sg * (1.f - sg)) : f * sg);
        }
        *(hcp_bf16x8*)(out + i * 8) = o;
    }
}

// out[i, :] = tok[ids[i], :] + pos[pos_ids ? pos_ids[i] : i % L, :]   (CLIPTextEmbeddings, te_struct.txt; fp32 tables -> bf16 rows)
HCP_KERNEL(256) embedding_kernel(const float* tok, const long long* ids, const float* pos, const long long* pos_ids, hcp_bf16* out,
                                 long n, int C, int L) {
    const int cv = C / 8;
    const long total = n * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cv; const int c = (int)(i - r * cv) * 8;
        const float* t = tok + (size_t)ids[r] * C + c;
        const float* q = pos + (size_t)(pos_ids ? pos_ids[r] : r % L) * C + c;
        hcp_bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)hcp_f2bf(t[j] + q[j]);
        *(hcp_bf16x8*)(out + r * C + c) = o;
    }
}

// dst[b][c][r] = src[b][r][c]  (bf16, per-sample 2-D transpose through a 64x64 LDS tile, padded against bank conflicts)
HCP_KERNEL(256) transpose_kernel(const hcp_bf16* src, hcp_bf16* dst, int R, int C) {
    HCP_DYN_SMEM(smem);
    hcp_bf16* tile = (hcp_bf16*)smem;                 // [64][66]
    const int b = blockIdx.z, r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const hcp_bf16* s = src + (size_t)b * R * C;
    hcp_bf16* d = dst + (size_t)b * R * C;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (r0 + i < R && c0 + tx < C) tile[i * 66 + tx] = s[(size_t)(r0 + i) * C + c0 + tx];
    HCP_SYNC();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < C && r0 + tx < R) d[(size_t)(c0 + i) * R + r0 + tx] = tile[tx * 66 + i];
}

// p[m, :N] = softmax(scale * s[m, :N])  (fp32 scores in, bf16 probabilities out; one workgroup per row, three passes over an
// L2-resident row).  The single-head d=512 attention of the VAE encoder's mid block runs as GEMM -> this -> GEMM.
HCP_KERNEL(256) softmax_rows_kernel(const float* s, long lds_, hcp_bf16* p, long ldp, int N, float scale) {
    HCP_DYN_SMEM(smem);
    float* red = (float*)smem;                         // [4]
    const float* row = s + (size_t)blockIdx.x * lds_;
    hcp_bf16* out = p + (size_t)blockIdx.x * ldp;
    const int tid = threadIdx.x;
    float mx = -3.0e38f;
    for (int i = tid; i < N; i += 256) { float v = row[i] * scale; mx = v > mx ? v : mx; }
    mx = hcp_wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    HCP_SYNC();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    HCP_SYNC();
    float sum = 0.f;
    for (int i = tid; i < N; i += 256) sum += expf(row[i] * scale - mx);
    sum = hcp_wave_sum(sum);
    if ((tid & 63) == 0) red[tid >> 6] = sum;
    HCP_SYNC();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    for (int i = tid; i < N; i += 256) out[i] = hcp_f2bf(expf(row[i] * scale - mx) * inv);
}

// AutoencoderKL.encode(...).latent_dist.sample() * scaling_factor (reference data/pair_dataset.py:72-75, train_ac.py:431-432) from
// the encoder's conv_out moments [B, 2L, hw] (fp32 NCHW): quant_conv (1x1, 2L -> 2L) per pixel, mean | logvar = chunk, logvar
// clamped to [-30, 20], latent = (mean + exp(0.5 logvar) * noise) * scale; noise == null -> the distribution's mode (mean).
HCP_KERNEL(256) vae_sample_kernel(const float* mom, const float* Wq, const float* bq, const float* noise, float* out, int B, int L,
                                  long hw, float scale) {
    const long total = (long)B * hw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / hw, px = i - b * hw;
        const float* m = mom + (size_t)b * 2 * L * hw + px;
        float in[16];
        for (int c = 0; c < 2 * L; ++c) in[c] = m[(size_t)c * hw];
        for (int l = 0; l < L; ++l) {
            float mean = bq[l], logvar = bq[L + l];
            for (int c = 0; c < 2 * L; ++c) { mean += Wq[l * 2 * L + c] * in[c]; logvar += Wq[(L + l) * 2 * L + c] * in[c]; }
            logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
            float v = mean;
            if (noise) v += expf(0.5f * logvar) * noise[((size_t)b * L + l) * hw + px];
            out[((size_t)b * L + l) * hw + px] = v * scale;
        }
    }
}

}  // namespace

HCP_API int hcp_geglu_fwd(const void* h, void* y, long M, int F, hipStream_t stream) {
    HCP_REQUIRE(h && y && M > 0 && F > 0 && F % 8 == 0, "hcp_geglu_fwd: bad arguments");
    HCP_LAUNCH(geglu_fwd_kernel, dim3(pw_grid(M * (F / 8))), dim3(PW_THREADS), 0, stream, (const hcp_bf16*)h, (hcp_bf16*)y, M, F);
    HCP_LAUNCH_CHECK("geglu_fwd");
}
HCP_API int hcp_geglu_bwd(const void* h, const void* dy, void* dh, long M, int F, hipStream_t stream) {
    HCP_REQUIRE(h && dy && dh && M > 0 && F > 0 && F % 8 == 0, "hcp_geglu_bwd: bad arguments");
    HCP_LAUNCH(geglu_bwd_kernel, dim3(pw_grid(M * (F / 8))), dim3(PW_THREADS), 0, stream, (const hcp_bf16*)h,
               (const hcp_bf16*)dy, (hcp_bf16*)dh, M, F);
    HCP_LAUNCH_CHECK("geglu_bwd");
}
HCP_API int hcp_add_bf16(const void* a, const void* b, void* out, long n, hipStream_t stream) {
    HCP_REQUIRE(a && b && out && n > 0 && n % 8 == 0, "hcp_add_bf16: n must be a positive multiple of 8");
    HCP_LAUNCH(add_kernel, dim3(pw_grid(n / 8)), dim3(PW_THREADS), 0, stream, (const hcp_bf16*)a, (const hcp_bf16*)b,
               (hcp_bf16*)out, n / 8);
    HCP_LAUNCH_CHECK("add_bf16");
}
HCP_API int hcp_silu_fwd(const void* x, void* y, long n, hipStream_t stream) {
    HCP_REQUIRE(x && y && n > 0 && n % 8 == 0, "hcp_silu_fwd: n must be a positive multiple of 8");
    HCP_LAUNCH(silu_kernel, dim3(pw_grid(n / 8)), dim3(PW_THREADS), 0, stream, (const hcp_bf16*)x, (const hcp_bf16*)nullptr,
               (hcp_bf16*)y, n / 8, 0);
    HCP_LAUNCH_CHECK("silu_fwd");
}
HCP_API int hcp_silu_bwd(const void* x, const void* dy, void* dx, long n, hipStream_t stream) {
    HCP_REQUIRE(x && dy && dx && n > 0 && n % 8 == 0, "hcp_silu_bwd: n must be a positive multiple of 8");
    HCP_LAUNCH(silu_kernel, dim3(pw_grid(n / 8)), dim3(PW_THREADS), 0, stream, (const hcp_bf16*)x, (const hcp_bf16*)dy,
               (hcp_bf16*)dx, n / 8, 1);
    HCP_LAUNCH_CHECK("silu_bwd");
}
HCP_API int hcp_nchw_to_nhwc_bf16(const void* src, int src_is_f32, void* dst, int B, int C, int HW, int Cpad,
                                  hipStream_t stream) {
    HCP_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && Cpad >= C, "hcp_nchw_to_nhwc_bf16: bad arguments");
    HCP_LAUNCH(nchw_to_nhwc_kernel, dim3(pw_grid((long)B * HW * Cpad)), dim3(PW_THREADS), 0, stream, src, src_is_f32,
               (hcp_bf16*)dst, B, C, HW, Cpad);
    HCP_LAUNCH_CHECK("nchw_to_nhwc");
}
HCP_API int hcp_nhwc_to_nchw_f32(const float* src, float* dst, int B, int C, int HW, int Csrc, hipStream_t stream) {
    HCP_REQUIRE(src && dst && B > 0 && C > 0 && HW > 0 && Csrc >= C, "hcp_nhwc_to_nchw_f32: bad arguments");
    HCP_LAUNCH(nhwc_to_nchw_kernel, dim3(pw_grid((long)B * HW * C)), dim3(PW_THREADS), 0, stream, src, dst, B, C, HW, Csrc);
    HCP_LAUNCH_CHECK("nhwc_to_nchw");
}
HCP_API int hcp_upsample2x_bwd(const void* dup, void* dx, int B, int H, int W, int C, hipStream_t stream) {
    HCP_REQUIRE(dup && dx && B > 0 && H > 0 && W > 0 && C % 8 == 0, "hcp_upsample2x_bwd: bad arguments");
    HCP_LAUNCH(upsample2x_bwd_kernel, dim3(pw_grid((long)B * H * W * (C / 8))), dim3(PW_THREADS), 0, stream,
               (const hcp_bf16*)dup, (hcp_bf16*)dx, B, H, W, C);
    HCP_LAUNCH_CHECK("upsample2x_bwd");
}
HCP_API int hcp_timestep_embedding(const long long* timesteps, void* emb, int B, int dim, float max_period,
                                   hipStream_t stream) {
    HCP_REQUIRE(timesteps && emb && B > 0 && dim > 0 && dim % 2 == 0, "hcp_timestep_embedding: bad arguments");
    HCP_LAUNCH(timestep_embedding_kernel, dim3(pw_grid((long)B * dim / 2)), dim3(PW_THREADS), 0, stream, timesteps,
               (const float*)nullptr, (hcp_bf16*)emb, B, dim, max_period);
    HCP_LAUNCH_CHECK("timestep_embedding");
}
HCP_API int hcp_timestep_embedding_f32(const float* values, void* emb, int B, int dim, float max_period, hipStream_t stream) {
    HCP_REQUIRE(values && emb && B > 0 && dim > 0 && dim % 2 == 0, "hcp_timestep_embedding_f32: bad arguments");
    HCP_LAUNCH(timestep_embedding_kernel, dim3(pw_grid((long)B * dim / 2)), dim3(PW_THREADS), 0, stream,
               (const long long*)nullptr, values, (hcp_bf16*)emb, B, dim, max_period);
    HCP_LAUNCH_CHECK("timestep_embedding_f32");
}
HCP_API int hcp_add_noise(const float* x0, const float* noise, const long long* timesteps, const float* alphas_cumprod,
                          float* xt, int B, long per_sample, hipStream_t stream) {
    HCP_REQUIRE(x0 && noise && timesteps && alphas_cumprod && xt && B > 0 && per_sample > 0, "hcp_add_noise: bad arguments");
    HCP_LAUNCH(add_noise_kernel, dim3(pw_grid((long)B * per_sample)), dim3(PW_THREADS), 0, stream, x0, noise, timesteps,
               alphas_cumprod, xt, B, per_sample);
    HCP_LAUNCH_CHECK("add_noise");
}
// x [n] fp32, eps2 [2n] (classifier-free guidance: uncond rows first) or [n] (guided == 0), out [n]; may run in place (out == x).
HCP_API int hcp_cfg_ddim_step(const float* x, const float* eps2, float* out, long n, int guided, float guidance_scale,
                              float alpha_cumprod_t, float alpha_cumprod_prev, hipStream_t stream) {
    HCP_REQUIRE(x && eps2 && out && n > 0, "hcp_cfg_ddim_step: bad arguments");
    HCP_REQUIRE(alpha_cumprod_t > 0.f && alpha_cumprod_t <= 1.f && alpha_cumprod_prev > 0.f && alpha_cumprod_prev <= 1.f,
                "hcp_cfg_ddim_step: alphas_cumprod must lie in (0, 1]");
    HCP_LAUNCH(cfg_ddim_kernel, dim3(pw_grid(n)), dim3(PW_THREADS), 0, stream, x, eps2, out, n, guided ? n : 0L, guidance_scale,
               alpha_cumprod_t, alpha_cumprod_prev);
    HCP_LAUNCH_CHECK("cfg_ddim_step");
}
// w[b] = loss weight of sample b from its timestep (kinds above); w feeds hcp_mse_masked_mean's sample_weight.
HCP_API int hcp_snr_loss_weight(const long long* timesteps, const float* alphas_cumprod, float* w, int B, int kind, float gamma,
                                hipStream_t stream) {
    HCP_REQUIRE(timesteps && alphas_cumprod && w && B > 0 && kind >= 0 && kind <= 3 && gamma > 0.f, "hcp_snr_loss_weight: bad arguments");
    HCP_LAUNCH(snr_weight_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, timesteps, alphas_cumprod, w, B, kind, gamma);
    HCP_LAUNCH_CHECK("snr_loss_weight");
}
// loss (device scalar, zeroed here) = mean((pred-target)^2 * mask * sample_weight[b]) * weight; grad (optional) = d loss / d pred.
// sample_weight: null or float[B] (train_ac.py:506-515 with a need_timesteps criterion).
HCP_API int hcp_mse_masked_mean(const float* pred, const float* target, const float* mask, int mask_channels,
                                const float* sample_weight, float* loss, float* grad, int B, int C, int HW, float weight,
                                hipStream_t stream) {
    HCP_REQUIRE(pred && target && loss && B > 0 && C > 0 && HW > 0, "hcp_mse_masked_mean: bad arguments");
    HCP_REQUIRE(!mask || mask_channels == 1 || mask_channels == C, "hcp_mse_masked_mean: mask channels must be 1 or C");
    if (hcp_memset_async(loss, 0, sizeof(float), stream)) return hcp_set_error("hcp_mse_masked_mean: memset failed");
    float scale = weight / (float)((long)B * C * HW);
    HCP_LAUNCH(mse_kernel, dim3(pw_grid((long)B * C * HW)), dim3(PW_THREADS), 0, stream, pred, target, mask, mask_channels,
               sample_weight, loss, grad, B, C, HW, scale);
    HCP_LAUNCH_CHECK("mse_masked_mean");
}

// dst[m, :C] = src[m, :C]; row strides sld/dld in elements (all multiples of 8). Used for the up-block skip concat
// (torch.cat(dim=1) in diffusers' UpBlock2D/CrossAttnUpBlock2D) and its backward split.
// The up-path skip concat (torch.cat([h, skip], dim=1) in every up-block ResnetBlock2D) and its gradient split as ONE launch each:
// joint [M][C1+C2] <-> a [M][C1], b [M][C2]; split = 0: joint <- (a | b), split = 1: (a, b) <- joint.
HCP_KERNEL(256) concat2_kernel(hcp_bf16* a, int c1, hcp_bf16* b, int c2, hcp_bf16* joint, long M, int split) {
    const int cv = (c1 + c2) / 8, c1v = c1 / 8;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / cv; const int c = (int)(i - r * cv);
        hcp_bf16* part = c < c1v ? a + r * c1 + (long)c * 8 : b + r * c2 + (long)(c - c1v) * 8;
        hcp_bf16* j = joint + r * (c1 + c2) + (long)c * 8;
        if (split) *(hcp_bf16x8*)part = *(const hcp_bf16x8*)j;
        else *(hcp_bf16x8*)j = *(const hcp_bf16x8*)part;
    }
}
HCP_API int hcp_concat2_bf16(void* a, int c1, void* b, int c2, void* joint, long M, int split, hipStream_t stream) {
    HCP_REQUIRE(a && b && joint && M > 0 && c1 > 0 && c2 > 0 && c1 % 8 == 0 && c2 % 8 == 0, "hcp_concat2_bf16: bad arguments");
    HCP_LAUNCH(concat2_kernel, dim3(pw_grid(M * ((c1 + c2) / 8))), dim3(PW_THREADS), 0, stream, (hcp_bf16*)a, c1, (hcp_bf16*)b, c2,
               (hcp_bf16*)joint, M, split);
    HCP_LAUNCH_CHECK("concat2");
}

HCP_API int hcp_copy2d_bf16(const void* src, int sld, void* dst, int dld, long M, int C, hipStream_t stream) {
    HCP_REQUIRE(src && dst && M > 0 && C > 0 && C % 8 == 0 && sld % 8 == 0 && dld % 8 == 0, "hcp_copy2d_bf16: bad arguments");
    HCP_LAUNCH(copy2d_kernel, dim3(pw_grid(M * (C / 8))), dim3(PW_THREADS), 0, stream, (const hcp_bf16*)src, sld,
               (hcp_bf16*)dst, dld, M, C);
    HCP_LAUNCH_CHECK("copy2d");
}

// dst[b][c][r] = src[b][r][c] for `batch` row-major bf16 matrices
HCP_API int hcp_transpose_bf16(const void* src, void* dst, int batch, int R, int C, hipStream_t stream) {
    HCP_REQUIRE(src && dst && batch > 0 && R > 0 && C > 0 && batch < 65536, "hcp_transpose_bf16: bad arguments");
    HCP_LAUNCH(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64, batch), dim3(256), 64 * 66 * sizeof(hcp_bf16), stream,
               (const hcp_bf16*)src, (hcp_bf16*)dst, R, C);
    HCP_LAUNCH_CHECK("transpose_bf16");
}
// P[m, :N] (bf16, row stride ldp) = softmax(scale * S[m, :N]) (fp32, row stride lds)
HCP_API int hcp_softmax_rows(const float* S, long lds, void* P, long ldp, int M, int N, float scale, hipStream_t stream) {
    HCP_REQUIRE(S && P && M > 0 && N > 0 && lds >= N && ldp >= N, "hcp_softmax_rows: bad arguments");
    HCP_LAUNCH(softmax_rows_kernel, dim3(M), dim3(256), 4 * sizeof(float), stream, S, lds, (hcp_bf16*)P, ldp, N, scale);
    HCP_LAUNCH_CHECK("softmax_rows");
}
// latents[B, L, hw] = (mean + std * noise) * scale from conv_out moments [B, 2L, hw] and quant_conv (Wq [2L,2L], bq [2L]); L <= 8
HCP_API int hcp_vae_latent_sample(const float* moments, const float* Wq, const float* bq, const float* noise, float* latents, int B,
                                  int L, long hw, float scale, hipStream_t stream) {
    HCP_REQUIRE(moments && Wq && bq && latents && B > 0 && L > 0 && L <= 8 && hw > 0, "hcp_vae_latent_sample: bad arguments");
    HCP_LAUNCH(vae_sample_kernel, dim3(pw_grid((long)B * hw)), dim3(PW_THREADS), 0, stream, moments, Wq, bq, noise, latents, B, L, hw, scale);
    HCP_LAUNCH_CHECK("vae_latent_sample");
}

// y = quick_gelu(x) (dy == NULL) or dx = dy * quick_gelu'(x); n bf16 elements, n % 8 == 0
HCP_API int hcp_quick_gelu(const void* x, const void* dy, void* out, long n, hipStream_t stream) {
    HCP_REQUIRE(x && out && n > 0 && n % 8 == 0, "hcp_quick_gelu: bad arguments");
    HCP_LAUNCH(quick_gelu_kernel, dim3(pw_grid(n / 8)), dim3(PW_THREADS), 0, stream, (const hcp_bf16*)x, (const hcp_bf16*)dy, (hcp_bf16*)out, n / 8);
    HCP_LAUNCH_CHECK("quick_gelu");
}
// out[n, C] (bf16) = token_table[ids] + position_table[position_ids or (row % L)]; fp32 tables, int64 ids, C % 8 == 0
HCP_API int hcp_embedding_bf16(const float* token_table, const long long* ids, const float* position_table, const long long* position_ids,
                               void* out, long n, int C, int L, hipStream_t stream) {
    HCP_REQUIRE(token_table && ids && position_table && out && n > 0 && C > 0 && C % 8 == 0 && L > 0, "hcp_embedding_bf16: bad arguments");
    HCP_LAUNCH(embedding_kernel, dim3(pw_grid(n * (C / 8))), dim3(PW_THREADS), 0, stream, token_table, ids, position_table, position_ids,
               (hcp_bf16*)out, n, C, L);
    HCP_LAUNCH_CHECK("embedding_bf16");
}
