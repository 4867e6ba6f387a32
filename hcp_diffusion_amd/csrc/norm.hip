// norm.hip — GroupNorm(+SiLU) and LayerNorm, forward and input-gradient, NHWC / token-major bf16.
//
// Replaces torch.nn.functional.group_norm + SiLU (diffusers ResnetBlock2D.norm1/norm2,
// conv_norm_out; Transformer2DModel.norm without SiLU — reference cfgs/unet_struct.txt:13,93,97,929)
// and F.layer_norm (BasicTransformerBlock.norm1-3, unet_struct.txt:44-46).
// HBM-bound: every pass streams the tensor with 16-byte loads; statistics are fp32,
// combined across row-chunks with Chan's parallel-variance formula (deterministic, no atomics).
#include "hcp_common.h"

namespace {

constexpr int GN_MAX_CHUNKS = 128;

struct GNGeom { int TX, R, threads, nchunk, rows_per_chunk; };

GNGeom gn_geom(int HW, int C) {
    GNGeom g;
    g.TX = C / 8;
    g.R = 256 / g.TX; if (g.R < 1) g.R = 1;
    g.threads = g.TX * g.R;
    long elems = (long)HW * C;
    long n = elems / 16384; if (n < 1) n = 1; if (n > GN_MAX_CHUNKS) n = GN_MAX_CHUNKS;
    if (n > HW) n = HW;
    g.rows_per_chunk = (int)((HW + n - 1) / n);
    g.nchunk = (HW + g.rows_per_chunk - 1) / g.rows_per_chunk;
    return g;
}

// ws layout: [B][nchunk][G][2]  (fwd: mean, M2 of the chunk; bwd: S1, S2)
HCP_KERNEL(1024) gn_fwd_partial(const hcp_bf16* x, float* ws, int HW, int C, int G, int TX, int R, int rows_per_chunk) {
    HCP_DYN_SMEM(smem);
    float* s_sum = (float*)smem;            // [R][C]
    float* s_sq = s_sum + R * C;            // [R][C]
    const int tid = threadIdx.x;
    const int cx = tid % TX, ry = tid / TX;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
    float s[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s[i] = 0.f; q[i] = 0.f; }
    const hcp_bf16* xb = x + (size_t)b * HW * C + cx * 8;
    for (int r = r0 + ry; r < r1; r += R) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(xb + (size_t)r * C);
#pragma unroll
        for (int i = 0; i < 8; ++i) { float f = hcp_bf2f((unsigned short)v[i]); s[i] += f; q[i] += f * f; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s_sum[ry * C + cx * 8 + i] = s[i]; s_sq[ry * C + cx * 8 + i] = q[i]; }
    HCP_SYNC();
    const int Cg = C / G;
    if (tid < G) {
        float a = 0.f, bq = 0.f;
        for (int rr = 0; rr < R; ++rr)
            for (int c = 0; c < Cg; ++c) { a += s_sum[rr * C + tid * Cg + c]; bq += s_sq[rr * C + tid * Cg + c]; }
        float n = (float)(r1 - r0) * Cg;
        float mean = a / n;
        float m2 = bq - a * mean; if (m2 < 0.f) m2 = 0.f;
        float* o = ws + (((size_t)b * gridDim.x + chunk) * G + tid) * 2;
        o[0] = mean; o[1] = m2;
    }
}

HCP_KERNEL(1024) gn_fwd_apply(const hcp_bf16* x, const float* gamma, const float* beta, const float* ws, hcp_bf16* y,
                              float* stats, int HW, int C, int G, int TX, int R, int rows_per_chunk, float eps, int silu) {
    HCP_DYN_SMEM(smem);
    float* s_a = (float*)smem;      // [C] scale
    float* s_b = s_a + C;           // [C] shift
    float* s_g = s_b + C;           // [G][2] mean, rstd
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int nchunk = gridDim.x;
    const int Cg = C / G;
    if (tid < G) {
        float n = 0.f, mean = 0.f, m2 = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            int r0 = c * rows_per_chunk; int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
            float nb = (float)(r1 - r0) * Cg;
            const float* o = ws + (((size_t)b * nchunk + c) * G + tid) * 2;
            float d = o[0] - mean; float nn = n + nb;
            mean += d * nb / nn; m2 += o[1] + d * d * n * nb / nn; n = nn;
        }
        float rstd = 1.0f / sqrtf(m2 / n + eps);
        s_g[tid * 2] = mean; s_g[tid * 2 + 1] = rstd;
        if (chunk == 0) { stats[((size_t)b * G + tid) * 2] = mean; stats[((size_t)b * G + tid) * 2 + 1] = rstd; }
    }
    HCP_SYNC();
    for (int c = tid; c < C; c += blockDim.x) {
        int g = c / Cg; float a = s_g[g * 2 + 1] * gamma[c];
        s_a[c] = a; s_b[c] = beta[c] - s_g[g * 2] * a;
    }
    HCP_SYNC();
    const int cx = tid % TX, ry = tid / TX;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
    float a8[8], b8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a8[i] = s_a[cx * 8 + i]; b8[i] = s_b[cx * 8 + i]; }
    const size_t base = (size_t)b * HW * C + cx * 8;
    for (int r = r0 + ry; r < r1; r += R) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(x + base + (size_t)r * C);
        hcp_bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float z = hcp_bf2f((unsigned short)v[i]) * a8[i] + b8[i];
            if (silu) z = hcp_silu(z);
            o[i] = (short)hcp_f2bf(z);
        }
        *(hcp_bf16x8*)(y + base + (size_t)r * C) = o;
    }
}

// dz = dy * silu'(z) ; dxhat = dz * gamma ; S1 = sum dxhat ; S2 = sum dxhat * xhat   (per batch, group)
HCP_KERNEL(1024) gn_bwd_partial(const hcp_bf16* x, const hcp_bf16* dy, const float* gamma, const float* beta,
                                const float* stats, float* ws, int HW, int C, int G, int TX, int R, int rows_per_chunk, int silu) {
    HCP_DYN_SMEM(smem);
    float* s_1 = (float*)smem;     // [R][C]
    float* s_2 = s_1 + R * C;      // [R][C]
    const int tid = threadIdx.x;
    const int cx = tid % TX, ry = tid / TX;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int Cg = C / G;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
    float mean8[8], rstd8[8], g8[8], be8[8], s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = cx * 8 + i; int g = c / Cg;
        mean8[i] = stats[((size_t)b * G + g) * 2]; rstd8[i] = stats[((size_t)b * G + g) * 2 + 1];
        g8[i] = gamma[c]; be8[i] = beta[c]; s1[i] = 0.f; s2[i] = 0.f;
    }
    const size_t base = (size_t)b * HW * C + cx * 8;
    for (int r = r0 + ry; r < r1; r += R) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(x + base + (size_t)r * C);
        hcp_bf16x8 d = *(const hcp_bf16x8*)(dy + base + (size_t)r * C);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (hcp_bf2f((unsigned short)v[i]) - mean8[i]) * rstd8[i];
            float dz = hcp_bf2f((unsigned short)d[i]);
            if (silu) { float z = xh * g8[i] + be8[i]; float sg = hcp_sigmoid(z); dz *= sg * (1.f + z * (1.f - sg)); }
            float dxh = dz * g8[i];
            s1[i] += dxh; s2[i] += dxh * xh;
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s_1[ry * C + cx * 8 + i] = s1[i]; s_2[ry * C + cx * 8 + i] = s2[i]; }
    HCP_SYNC();
    if (tid < G) {
        float a = 0.f, bq = 0.f;
        for (int rr = 0; rr < R; ++rr)
            for (int c = 0; c < Cg; ++c) { a += s_1[rr * C + tid * Cg + c]; bq += s_2[rr * C + tid * Cg + c]; }
        float* o = ws + (((size_t)b * gridDim.x + chunk) * G + tid) * 2;
        o[0] = a; o[1] = bq;
    }
}

HCP_KERNEL(1024) gn_bwd_apply(const hcp_bf16* x, const hcp_bf16* dy, const float* gamma, const float* beta,
                              const float* stats, const float* ws, hcp_bf16* dx, int HW, int C, int G, int TX, int R,
                              int rows_per_chunk, int silu) {
    HCP_DYN_SMEM(smem);
    float* s_c = (float*)smem;      // [G][2] c1, c2
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int nchunk = gridDim.x;
    const int Cg = C / G;
    if (tid < G) {
        float a = 0.f, bq = 0.f;
        for (int c = 0; c < nchunk; ++c) {
            const float* o = ws + (((size_t)b * nchunk + c) * G + tid) * 2;
            a += o[0]; bq += o[1];
        }
        float n = (float)HW * Cg;
        s_c[tid * 2] = a / n; s_c[tid * 2 + 1] = bq / n;
    }
    HCP_SYNC();
    const int cx = tid % TX, ry = tid / TX;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
    float mean8[8], rstd8[8], g8[8], be8[8], c1[8], c2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = cx * 8 + i; int g = c / Cg;
        mean8[i] = stats[((size_t)b * G + g) * 2]; rstd8[i] = stats[((size_t)b * G + g) * 2 + 1];
        g8[i] = gamma[c]; be8[i] = beta[c]; c1[i] = s_c[g * 2]; c2[i] = s_c[g * 2 + 1];
    }
    const size_t base = (size_t)b * HW * C + cx * 8;
    for (int r = r0 + ry; r < r1; r += R) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(x + base + (size_t)r * C);
        hcp_bf16x8 d = *(const hcp_bf16x8*)(dy + base + (size_t)r * C);
        hcp_bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (hcp_bf2f((unsigned short)v[i]) - mean8[i]) * rstd8[i];
            float dz = hcp_bf2f((unsigned short)d[i]);
            if (silu) { float z = xh * g8[i] + be8[i]; float sg = hcp_sigmoid(z); dz *= sg * (1.f + z * (1.f - sg)); }
            float dxh = dz * g8[i];
            o[i] = (short)hcp_f2bf(rstd8[i] * (dxh - c1[i] - xh * c2[i]));
        }
        *(hcp_bf16x8*)(dx + base + (size_t)r * C) = o;
    }
}

// ------------------------------------------------------------------ LayerNorm: one wave per row
HCP_KERNEL(256) ln_fwd_kernel(const hcp_bf16* x, const float* gamma, const float* beta, hcp_bf16* y, float* stats,
                              int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < M;
    const hcp_bf16* xr = x + (size_t)(live ? row : 0) * C;
    const int nch = C / 8;
    float s = 0.f;
    for (int j = lane; j < nch; j += 64) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(xr + j * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) s += hcp_bf2f((unsigned short)v[i]);
    }
    const float mean = hcp_wave_sum(s) / C;
    float q = 0.f;
    for (int j = lane; j < nch; j += 64) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(xr + j * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) { float d = hcp_bf2f((unsigned short)v[i]) - mean; q += d * d; }
    }
    const float rstd = 1.0f / sqrtf(hcp_wave_sum(q) / C + eps);
    if (!live) return;
    if (lane == 0) { stats[(size_t)row * 2] = mean; stats[(size_t)row * 2 + 1] = rstd; }
    hcp_bf16* yr = y + (size_t)row * C;
    for (int j = lane; j < nch; j += 64) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(xr + j * 8);
        hcp_f32x4 g0 = *(const hcp_f32x4*)(gamma + j * 8), g1 = *(const hcp_f32x4*)(gamma + j * 8 + 4);
        hcp_f32x4 b0 = *(const hcp_f32x4*)(beta + j * 8), b1 = *(const hcp_f32x4*)(beta + j * 8 + 4);
        hcp_bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float g = i < 4 ? g0[i] : g1[i - 4], bb = i < 4 ? b0[i] : b1[i - 4];
            o[i] = (short)hcp_f2bf((hcp_bf2f((unsigned short)v[i]) - mean) * rstd * g + bb);
        }
        *(hcp_bf16x8*)(yr + j * 8) = o;
    }
}

HCP_KERNEL(256) ln_bwd_kernel(const hcp_bf16* x, const hcp_bf16* dy, const float* gamma, const float* stats,
                              hcp_bf16* dx, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < M;
    const size_t off = (size_t)(live ? row : 0) * C;
    const float mean = stats[(size_t)(live ? row : 0) * 2], rstd = stats[(size_t)(live ? row : 0) * 2 + 1];
    const int nch = C / 8;
    float s1 = 0.f, s2 = 0.f;
    for (int j = lane; j < nch; j += 64) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(x + off + j * 8);
        hcp_bf16x8 d = *(const hcp_bf16x8*)(dy + off + j * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (hcp_bf2f((unsigned short)v[i]) - mean) * rstd;
            float dxh = hcp_bf2f((unsigned short)d[i]) * gamma[j * 8 + i];
            s1 += dxh; s2 += dxh * xh;
        }
    }
    const float c1 = hcp_wave_sum(s1) / C, c2 = hcp_wave_sum(s2) / C;
    if (!live) return;
    for (int j = lane; j < nch; j += 64) {
        hcp_bf16x8 v = *(const hcp_bf16x8*)(x + off + j * 8);
        hcp_bf16x8 d = *(const hcp_bf16x8*)(dy + off + j * 8);
        hcp_bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (hcp_bf2f((unsigned short)v[i]) - mean) * rstd;
            float dxh = hcp_bf2f((unsigned short)d[i]) * gamma[j * 8 + i];
            o[i] = (short)hcp_f2bf(rstd * (dxh - c1 - xh * c2));
        }
        *(hcp_bf16x8*)(dx + off + j * 8) = o;
    }
}

int gn_check(int B, int HW, int C, int G) {
    HCP_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0, "groupnorm: empty problem");
    HCP_REQUIRE(C % 8 == 0 && C % G == 0 && C <= 8192 && G <= 64, "groupnorm: C=%d G=%d unsupported", C, G);
    return 0;
}

}  // namespace

// Workspace (bytes) both GroupNorm entry points need: [B][nchunk][G][2] fp32 partials.
HCP_API size_t hcp_groupnorm_workspace_bytes(int B, int HW, int C, int G) {
    if (B <= 0 || HW <= 0 || C <= 0 || C % 8) return 0;
    GNGeom g = gn_geom(HW, C);
    return (size_t)B * g.nchunk * G * 2 * sizeof(float);
}

// y = [silu](group_norm(x; gamma, beta, eps)); stats[B,G,2] = (mean, rstd) saved for backward.
HCP_API int hcp_groupnorm_silu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                   void* workspace, int B, int HW, int C, int G, float eps, int silu, hipStream_t stream) {
    if (int e = gn_check(B, HW, C, G)) return e;
    HCP_REQUIRE(x && gamma && beta && y && stats && workspace, "hcp_groupnorm_silu_fwd: null pointer");
    GNGeom g = gn_geom(HW, C);
    size_t sm1 = (size_t)2 * g.R * C * sizeof(float);
    HCP_LAUNCH(gn_fwd_partial, dim3(g.nchunk, B), dim3(g.threads), sm1, stream, (const hcp_bf16*)x, (float*)workspace, HW, C,
               G, g.TX, g.R, g.rows_per_chunk);
    size_t sm2 = (size_t)(2 * C + 2 * G) * sizeof(float);
    HCP_LAUNCH(gn_fwd_apply, dim3(g.nchunk, B), dim3(g.threads), sm2, stream, (const hcp_bf16*)x, gamma, beta,
               (const float*)workspace, (hcp_bf16*)y, stats, HW, C, G, g.TX, g.R, g.rows_per_chunk, eps, silu);
    HCP_LAUNCH_CHECK("groupnorm_fwd");
}

// dx for y = [silu](group_norm(x)); gamma/beta frozen (their gradients are not produced here).
HCP_API int hcp_groupnorm_silu_bwd(const void* x, const void* dy, const float* gamma, const float* beta,
                                   const float* stats, void* dx, void* workspace, int B, int HW, int C, int G, int silu,
                                   hipStream_t stream) {
    if (int e = gn_check(B, HW, C, G)) return e;
    HCP_REQUIRE(x && dy && gamma && beta && stats && dx && workspace, "hcp_groupnorm_silu_bwd: null pointer");
    GNGeom g = gn_geom(HW, C);
    size_t sm1 = (size_t)2 * g.R * C * sizeof(float);
    HCP_LAUNCH(gn_bwd_partial, dim3(g.nchunk, B), dim3(g.threads), sm1, stream, (const hcp_bf16*)x, (const hcp_bf16*)dy,
               gamma, beta, stats, (float*)workspace, HW, C, G, g.TX, g.R, g.rows_per_chunk, silu);
    size_t sm2 = (size_t)2 * G * sizeof(float);
    HCP_LAUNCH(gn_bwd_apply, dim3(g.nchunk, B), dim3(g.threads), sm2, stream, (const hcp_bf16*)x, (const hcp_bf16*)dy,
               gamma, beta, stats, (const float*)workspace, (hcp_bf16*)dx, HW, C, G, g.TX, g.R, g.rows_per_chunk, silu);
    HCP_LAUNCH_CHECK("groupnorm_bwd");
}

HCP_API int hcp_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int M, int C,
                              float eps, hipStream_t stream) {
    HCP_REQUIRE(M > 0 && C > 0 && C % 8 == 0, "hcp_layernorm_fwd: bad shape M=%d C=%d", M, C);
    HCP_REQUIRE(x && gamma && beta && y && stats, "hcp_layernorm_fwd: null pointer");
    HCP_LAUNCH(ln_fwd_kernel, dim3(hcp_cdiv(M, 4)), dim3(256), 0, stream, (const hcp_bf16*)x, gamma, beta, (hcp_bf16*)y,
               stats, M, C, eps);
    HCP_LAUNCH_CHECK("layernorm_fwd");
}

HCP_API int hcp_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* stats, void* dx, int M,
                              int C, hipStream_t stream) {
    HCP_REQUIRE(M > 0 && C > 0 && C % 8 == 0, "hcp_layernorm_bwd: bad shape M=%d C=%d", M, C);
    HCP_REQUIRE(x && dy && gamma && stats && dx, "hcp_layernorm_bwd: null pointer");
    HCP_LAUNCH(ln_bwd_kernel, dim3(hcp_cdiv(M, 4)), dim3(256), 0, stream, (const hcp_bf16*)x, (const hcp_bf16*)dy, gamma,
               stats, (hcp_bf16*)dx, M, C);
    HCP_LAUNCH_CHECK("layernorm_bwd");
}
