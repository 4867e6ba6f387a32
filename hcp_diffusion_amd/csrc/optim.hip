// optim.hip — global-norm clip + AdamW over ONE flat fp32 bucket (all trainable parameters).
//
// Replaces accelerator.clip_grad_norm_ + torch.optim.AdamW.step + zero_grad
// (reference hcpdiff/train_ac.py:485-494, cfgs/train/train_base.yaml:17,37-40).  The reference
// launches hundreds of tiny per-tensor kernels for 320 LoRA tensors; here it is three launches,
// no host synchronisation (norm, lr and step count stay in device memory so the step can be
// captured in a hipGraph).
#include "hcp_common.h"

namespace {

// One atomic per BLOCK (wave sums meet in LDS first): same-address device atomics retire at ~13 ns each, so the first version's
// one-per-wave (8192 of them for the 3 M-element LoRA bucket) cost 107 us for a 12 MB read.
HCP_KERNEL(256) sumsq_kernel(const float* g, long n, float* out) {
    HCP_DYN_SMEM(smem);
    float* part = (float*)smem;
    float acc = 0.f;
    const long n4 = ((((uintptr_t)g) & 15) == 0) ? n / 4 : 0;
    const hcp_f32x4* g4 = (const hcp_f32x4*)g;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const hcp_f32x4 v = g4[i];
        acc += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += g[i] * g[i];
    acc = hcp_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    HCP_SYNC();
    if (threadIdx.x == 0) hcp_atomic_add(out, part[0] + part[1] + part[2] + part[3]);
}

HCP_KERNEL(64) step_inc_kernel(int* step) {
    if (threadIdx.x == 0) *step += 1;
}

HCP_KERNEL(256) adamw_kernel(float* p, float* g, float* m, float* v, long n, const float* lr_p, float beta1, float beta2,
                             float eps, float wd, const float* sumsq, float grad_scale, float max_norm, const int* step_p) {
    const float lr = *lr_p;
    const int t = *step_p;
    float clip = grad_scale;
    if (sumsq && max_norm > 0.f) {
        float norm = sqrtf(*sumsq) * grad_scale;
        float c = max_norm / (norm + 1e-6f);
        if (c < 1.f) clip *= c;
    }
    const float bc1 = 1.f - powf(beta1, (float)t), bc2 = 1.f - powf(beta2, (float)t);
    const float step_size = lr / bc1;
    const float inv_sqrt_bc2 = 1.f / sqrtf(bc2);
    // 16 bytes per lane where the four arrays allow it (the flat buckets do: 27.5 GB of traffic per step for a full fine-tune)
    const long n4 = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) ? n / 4 : 0;
    hcp_f32x4* p4 = (hcp_f32x4*)p; hcp_f32x4* g4 = (hcp_f32x4*)g; hcp_f32x4* m4 = (hcp_f32x4*)m; hcp_f32x4* v4 = (hcp_f32x4*)v;
    const float decay = 1.f - lr * wd;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        hcp_f32x4 gq = g4[i], pq = p4[i], mq = m4[i], vq = v4[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float gi = gq[q] * clip;
            float pi = pq[q] * decay;
            float mi = beta1 * mq[q] + (1.f - beta1) * gi;
            float vi = beta2 * vq[q] + (1.f - beta2) * gi * gi;
            pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
            pq[q] = pi; mq[q] = mi; vq[q] = vi;
        }
        p4[i] = pq; m4[i] = mq; v4[i] = vq; g4[i] = hcp_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * clip;
        float pi = p[i] * decay;
        float mi = beta1 * m[i] + (1.f - beta1) * gi;
        float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        pi -= step_size * mi / (sqrtf(vi) * inv_sqrt_bc2 + eps);
        p[i] = pi; m[i] = mi; v[i] = vi; g[i] = 0.f;
    }
}

// ema <- lerp(ema, p, 1 - decay), decay = clip(1 - (1 + step / inv_gamma)^-power, 0, decay_max)   (reference utils/ema.py:18-27)
HCP_KERNEL(256) ema_kernel(float* ema, const float* p, long n, const int* step_p, float inv_gamma, float power, float decay_max) {
    float decay = 1.f - powf(1.f + (float)(*step_p) / inv_gamma, -power);
    decay = fminf(fmaxf(decay, 0.f), decay_max);
    const float w = 1.f - decay;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float e = ema[i];
        ema[i] = e + w * (p[i] - e);
    }
}

// Wire formats of the sharded exchange: dst_bf16[i] = bf16(src[i] * scale) (optionally clearing src: the zero_grad of a bucket whose
// gradients leave as bf16), and back.  8 elements per thread-iteration where the pointers allow it.
HCP_KERNEL(256) cast_f32_bf16_kernel(float* src, unsigned short* dst, long n, float scale, int zero_src) {
    const bool vec = (((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0;
    const long n4 = vec ? n / 4 : 0;
    hcp_f32x4* s4 = (hcp_f32x4*)src;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const hcp_f32x4 v = s4[i];
        unsigned short o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = hcp_f2bf(v[j] * scale);
        *(uint64_t*)(dst + 4 * i) = *(const uint64_t*)o;
        if (zero_src) s4[i] = hcp_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        dst[i] = hcp_f2bf(src[i] * scale);
        if (zero_src) src[i] = 0.f;
    }
}

HCP_KERNEL(256) cast_bf16_f32_kernel(const unsigned short* src, float* dst, long n) {
    const bool vec = (((uintptr_t)dst) & 15) == 0 && (((uintptr_t)src) & 7) == 0;
    const long n4 = vec ? n / 4 : 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        unsigned short in[4];
        *(uint64_t*)in = *(const uint64_t*)(src + 4 * i);
        ((hcp_f32x4*)dst)[i] = hcp_f32x4{hcp_bf2f(in[0]), hcp_bf2f(in[1]), hcp_bf2f(in[2]), hcp_bf2f(in[3])};
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = hcp_bf2f(src[i]);
}

inline int opt_grid(long n) { long g = (n + 255) / 256; if (g > 2048) g = 2048; if (g < 1) g = 1; return (int)g; }

}  // namespace

// out[0] = sum(g^2)  (device scalar, zeroed here)
HCP_API int hcp_sumsq_f32(const float* g, long n, float* out, hipStream_t stream) {
    HCP_REQUIRE(g && out && n > 0, "hcp_sumsq_f32: bad arguments");
    if (hcp_memset_async(out, 0, sizeof(float), stream)) return hcp_set_error("hcp_sumsq_f32: memset failed");
    long grid = (n + 4095) / 4096;                                  // >= 16 elements per thread before another block is worth an atomic
    if (grid > 2048) grid = 2048;
    HCP_LAUNCH(sumsq_kernel, dim3((int)grid), dim3(256), 4 * sizeof(float), stream, g, n, out);
    HCP_LAUNCH_CHECK("sumsq");
}

// One fused optimizer step on a flat bucket: g *= grad_scale (e.g. 1/world after an all-reduce SUM);
// global-norm clip to max_norm using *sumsq (sum of squares of the UNSCALED g; pass null / max_norm<=0 to skip);
// AdamW (torch semantics: decoupled weight decay, bias correction with the device-resident step count,
// which this call increments first); finally g = 0.
HCP_API int hcp_adamw_clip_fused(float* p, float* g, float* m, float* v, long n, const float* lr, float beta1, float beta2,
                                 float eps, float weight_decay, const float* sumsq, float grad_scale, float max_norm,
                                 int* step, hipStream_t stream) {
    HCP_REQUIRE(p && g && m && v && lr && step && n > 0, "hcp_adamw_clip_fused: bad arguments");
    HCP_LAUNCH(step_inc_kernel, dim3(1), dim3(64), 0, stream, step);
    HCP_LAUNCH(adamw_kernel, dim3(opt_grid(n)), dim3(256), 0, stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay,
               sumsq, grad_scale, max_norm, (const int*)step);
    HCP_LAUNCH_CHECK("adamw_clip_fused");
}

// ModelEMA.update for a flat bucket (reference hcpdiff/utils/ema.py:17-27, train_ac.py:503,517-521): one launch, the decay
// schedule evaluated on the device from the optimizer's own step counter (graph-capturable, no host sync).
HCP_API int hcp_ema_update(float* ema, const float* p, long n, const int* step, float inv_gamma, float power, float decay_max,
                           hipStream_t stream) {
    HCP_REQUIRE(ema && p && step && n > 0 && inv_gamma > 0.f, "hcp_ema_update: bad arguments");
    HCP_LAUNCH(ema_kernel, dim3(opt_grid(n)), dim3(256), 0, stream, ema, p, n, step, inv_gamma, power, decay_max);
    HCP_LAUNCH_CHECK("ema_update");
}

// Sharded exchange, bf16 on the wire (trainer.py `grad_wire` / `param_wire`): dst = bf16(src * scale), src cleared when zero_src.
HCP_API int hcp_cast_f32_bf16(float* src, void* dst, long n, float scale, int zero_src, hipStream_t stream) {
    HCP_REQUIRE(src && dst && n > 0, "hcp_cast_f32_bf16: bad arguments");
    HCP_LAUNCH(cast_f32_bf16_kernel, dim3(opt_grid((n + 3) / 4)), dim3(256), 0, stream, src, (unsigned short*)dst, n, scale, zero_src);
    HCP_LAUNCH_CHECK("cast_f32_bf16");
}

HCP_API int hcp_cast_bf16_f32(const void* src, float* dst, long n, hipStream_t stream) {
    HCP_REQUIRE(src && dst && n > 0, "hcp_cast_bf16_f32: bad arguments");
    HCP_LAUNCH(cast_bf16_f32_kernel, dim3(opt_grid((n + 3) / 4)), dim3(256), 0, stream, (const unsigned short*)src, dst, n);
    HCP_LAUNCH_CHECK("cast_bf16_f32");
}
