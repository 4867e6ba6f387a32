// norm.hip — GroupNorm(+SiLU) and LayerNorm, forward and input-gradient, NHWC / token-major bf16.
//
// Replaces torch.nn.functional.group_norm + SiLU (diffusers ResnetBlock2D.norm1/norm2,
// conv_norm_out; Transformer2DModel.norm without SiLU — reference cfgs/unet_struct.txt:13,93,97,929)
// and F.layer_norm (BasicTransformerBlock.norm1-3, unet_struct.txt:44-46).
// HBM-bound: every pass streams the tensor with 16-byte loads; statistics are fp32,
// combined across row-chunks with Chan's parallel-variance formula (deterministic, no atomics).
#include "hcp_common.h"

namespace {

struct GNGeom { int TX, R, threads, nchunk, rows_per_chunk; };

// Thread (cx, ry) owns channels [8cx, 8cx+8) and rows ry, ry+R, ...; a chunk is >= 4R rows (every thread has >= 4 independent
// 16-byte loads in flight) and is sized so that the launch has about g_gn_target_wgs workgroups (2 per CU): enough to fill the
// chip, few enough that the per-sample partial table every apply workgroup merges stays ~100 entries per group.
HCP_TUNABLE(int, g_gn_target_wgs, 512);
HCP_TUNABLE(int, g_gn_slab, 1);      // tools: 0 = always the two-launch row-chunk path (A/B measurements; hcp_debug_set_gn_target(-1 / -2))
GNGeom gn_geom(int B, int HW, int C) {
    GNGeom g;
    g.TX = C / 8;
    g.R = 256 / g.TX; if (g.R < 1) g.R = 1;
    g.threads = g.TX * g.R;
    long rows = ((long)HW * B + g_gn_target_wgs - 1) / g_gn_target_wgs;
    rows = (rows + g.R - 1) / g.R * g.R;
    if (rows < 4 * g.R) rows = 4 * g.R;
    g.rows_per_chunk = (int)rows;
    g.nchunk = (HW + g.rows_per_chunk - 1) / g.rows_per_chunk;
    return g;
}

inline size_t gn_merge_smem(int threads, int G) { return ((size_t)2 * G + (size_t)(threads / G) * G * 3) * sizeof(float); }

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
    auto row = [&](const hcp_bf16x8& v) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { float f = hcp_bf2f((unsigned short)v[i]); s[i] += f; q[i] += f * f; }
    };
    // four rows requested before the first is used (the rolled loop was one load -> s_waitcnt vmcnt(0) -> adds per row: with two
    // waves per SIMD the pass ran at the latency of a memory round trip per row); same accumulation order
    int r = r0 + ry;
    for (; r + 3 * R < r1; r += 4 * R) {
        hcp_bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const hcp_bf16x8*)(xb + (size_t)(r + u * R) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) row(v[u]);
    }
    for (; r < r1; r += R) row(*(const hcp_bf16x8*)(xb + (size_t)r * C));
#pragma unroll
    for (int i = 0; i < 8; ++i) { s_sum[ry * C + cx * 8 + i] = s[i]; s_sq[ry * C + cx * 8 + i] = q[i]; }
    HCP_SYNC();
    const int Cg = C / G;
    // 4 lanes per group sum the R x Cg per-channel partials, then combine with two xor-shuffles
    float a = 0.f, bq = 0.f;
    {
        const int g = tid >> 2, sub = tid & 3;
        if (g < G)
            for (int j = sub; j < R * Cg; j += 4) {
                int rr = j / Cg, c = j - rr * Cg;
                a += s_sum[rr * C + g * Cg + c]; bq += s_sq[rr * C + g * Cg + c];
            }
        a += hcp_shfl_xor(a, 1); a += hcp_shfl_xor(a, 2);
        bq += hcp_shfl_xor(bq, 1); bq += hcp_shfl_xor(bq, 2);
    }
    if ((tid & 3) == 0 && (tid >> 2) < G) {
        const int gidx = tid >> 2;
        float n = (float)(r1 - r0) * Cg;
        float mean = a / n;
        float m2 = bq - a * mean; if (m2 < 0.f) m2 = 0.f;
        float* o = ws + (((size_t)b * gridDim.x + chunk) * G + gidx) * 2;
        o[0] = mean; o[1] = m2;
    }
}

// In-block merge of one sample's chunk partials (ws_b = [nchunk][G][2]) into s_out [G][2].  Thread (j, g) = (tid / G, tid % G)
// takes chunks j, j+J, ... (coalesced float2 loads, eight in flight); everything it accumulates is a PLAIN SUM, so there is no
// dependent chain of divisions (the first version Chan-merged sequentially: ~100 cycles per chunk per thread, which made every
// apply workgroup spend longer merging than streaming its rows).  The J per-group partial sums meet in LDS.
//   mode 0: (mean_c, M2_c) -> (mean, rstd).  With K = mean of chunk 0 as a shift and n_c the chunk's element count:
//           S1 = sum n_c (mean_c - K), S2 = sum n_c (mean_c - K)^2, SM = sum M2_c;
//           mean = K + S1/N, M2 = SM + S2 - S1^2/N   (the shifted form of Chan's pairwise formula: exact up to rounding, and
//           the subtraction only cancels the small between-chunk term because K is already within a chunk's spread of the mean)
//   mode 1: sums -> (S1/n, S2/n).
// Every apply workgroup does this itself (L2-resident partials) instead of a separate finalize launch.
HCP_DEVICE void gn_merge(const float* ws_b, float* s_out, int G, int nchunk, int rows_per_chunk, int HW, int Cg, float eps,
                         int mode, int tid, int nthreads) {
    float* s_tmp = s_out + 2 * G;                                // [J][G][3]
    const int J = nthreads / G;
    const int g = tid % G, j = tid / G;
    if (j < J) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        const float K = mode == 0 ? ws_b[g * 2] : 0.f;
        for (int c0 = j; c0 < nchunk; c0 += 8 * J) {
            float a[8], bq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * J;
                const float* o = ws_b + ((size_t)(c < nchunk ? c : 0) * G + g) * 2;
                a[u] = o[0]; bq[u] = o[1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * J;
                if (c >= nchunk) continue;
                if (mode == 0) {
                    int r0 = c * rows_per_chunk; int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
                    const float nb = (float)(r1 - r0) * Cg, d = a[u] - K;
                    t0 += nb * d; t1 += nb * d * d; t2 += bq[u];
                } else { t0 += a[u]; t1 += bq[u]; }
            }
        }
        float* t = s_tmp + ((size_t)j * G + g) * 3;
        t[0] = t0; t[1] = t1; t[2] = t2;
    }
    HCP_SYNC();
    if (tid < G) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int jj = 0; jj < J; ++jj) {
            const float* t = s_tmp + ((size_t)jj * G + tid) * 3;
            t0 += t[0]; t1 += t[1]; t2 += t[2];
        }
        const float cnt = (float)HW * Cg;
        if (mode == 0) {
            float m2 = t2 + t1 - t0 * t0 / cnt; if (m2 < 0.f) m2 = 0.f;
            s_out[tid * 2] = ws_b[tid * 2] + t0 / cnt; s_out[tid * 2 + 1] = 1.0f / sqrtf(m2 / cnt + eps);
        } else { s_out[tid * 2] = t0 / cnt; s_out[tid * 2 + 1] = t1 / cnt; }
    }
    HCP_SYNC();
}

HCP_KERNEL(1024) gn_fwd_apply(const hcp_bf16* x, const float* gamma, const float* beta, const float* ws, float* stats, hcp_bf16* y,
                              int HW, int C, int G, int TX, int R, int rows_per_chunk, int silu, float eps) {
    HCP_DYN_SMEM(smem);
    float* s_st = (float*)smem;                      // [G][2] (mean, rstd) of this sample
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int Cg = C / G;
    gn_merge(ws + (size_t)b * gridDim.x * G * 2, s_st, G, gridDim.x, rows_per_chunk, HW, Cg, eps, 0, tid, blockDim.x);
    if (chunk == 0 && tid < 2 * G) stats[(size_t)b * G * 2 + tid] = s_st[tid];       // saved for backward
    const int cx = tid % TX, ry = tid / TX;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
    float a8[8], b8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = cx * 8 + i; int g = c / Cg;
        float a = s_st[g * 2 + 1] * gamma[c];
        a8[i] = a; b8[i] = beta[c] - s_st[g * 2] * a;
    }
    const size_t base = (size_t)b * HW * C + cx * 8;
    auto row = [&](const hcp_bf16x8& v, int r) {
        hcp_bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float z = hcp_bf2f((unsigned short)v[i]) * a8[i] + b8[i];
            if (silu) z = hcp_silu(z);
            o[i] = (short)hcp_f2bf(z);
        }
        *(hcp_bf16x8*)(y + base + (size_t)r * C) = o;
    };
    int r = r0 + ry;
    for (; r + 3 * R < r1; r += 4 * R) {                 // four rows in flight (see gn_fwd_partial)
        hcp_bf16x8 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const hcp_bf16x8*)(x + base + (size_t)(r + u * R) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) row(v[u], r + u * R);
    }
    for (; r < r1; r += R) row(*(const hcp_bf16x8*)(x + base + (size_t)r * C), r);
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
    auto row = [&](const hcp_bf16x8& v, const hcp_bf16x8& d) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (hcp_bf2f((unsigned short)v[i]) - mean8[i]) * rstd8[i];
            float dz = hcp_bf2f((unsigned short)d[i]);
            if (silu) { float z = xh * g8[i] + be8[i]; float sg = hcp_sigmoid(z); dz *= sg * (1.f + z * (1.f - sg)); }
            float dxh = dz * g8[i];
            s1[i] += dxh; s2[i] += dxh * xh;
        }
    };
    int r = r0 + ry;
    for (; r + 1 * R < r1; r += 2 * R) {                 // two rows of both tensors in flight (see gn_fwd_partial; four rows spill); same accumulation order
        hcp_bf16x8 v[2], d[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v[u] = *(const hcp_bf16x8*)(x + base + (size_t)(r + u * R) * C);
            d[u] = *(const hcp_bf16x8*)(dy + base + (size_t)(r + u * R) * C);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) row(v[u], d[u]);
    }
    for (; r < r1; r += R) row(*(const hcp_bf16x8*)(x + base + (size_t)r * C), *(const hcp_bf16x8*)(dy + base + (size_t)r * C));
#pragma unroll
    for (int i = 0; i < 8; ++i) { s_1[ry * C + cx * 8 + i] = s1[i]; s_2[ry * C + cx * 8 + i] = s2[i]; }
    HCP_SYNC();
    float a = 0.f, bq = 0.f;
    {
        const int g = tid >> 2, sub = tid & 3;
        if (g < G)
            for (int j = sub; j < R * Cg; j += 4) {
                int rr = j / Cg, c = j - rr * Cg;
                a += s_1[rr * C + g * Cg + c]; bq += s_2[rr * C + g * Cg + c];
            }
        a += hcp_shfl_xor(a, 1); a += hcp_shfl_xor(a, 2);
        bq += hcp_shfl_xor(bq, 1); bq += hcp_shfl_xor(bq, 2);
    }
    if ((tid & 3) == 0 && (tid >> 2) < G) {
        float* o = ws + (((size_t)b * gridDim.x + chunk) * G + (tid >> 2)) * 2;
        o[0] = a; o[1] = bq;
    }
}

HCP_KERNEL(1024) gn_bwd_apply(const hcp_bf16* x, const hcp_bf16* dy, const float* gamma, const float* beta,
                              const float* stats, const float* ws, const hcp_bf16* addend, hcp_bf16* dx, int HW, int C, int G,
                              int TX, int R, int rows_per_chunk, int silu) {
    HCP_DYN_SMEM(smem);
    float* s_c12 = (float*)smem;                     // [G][2] (mean of dxhat, mean of dxhat * xhat) of this sample
    const int tid = threadIdx.x;
    const int chunk = blockIdx.x, b = blockIdx.y;
    const int Cg = C / G;
    gn_merge(ws + (size_t)b * gridDim.x * G * 2, s_c12, G, gridDim.x, rows_per_chunk, HW, Cg, 0.f, 1, tid, blockDim.x);
    const int cx = tid % TX, ry = tid / TX;
    const int r0 = chunk * rows_per_chunk;
    int r1 = r0 + rows_per_chunk; if (r1 > HW) r1 = HW;
    float mean8[8], rstd8[8], g8[8], be8[8], c1[8], c2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int c = cx * 8 + i; int g = c / Cg;
        mean8[i] = stats[((size_t)b * G + g) * 2]; rstd8[i] = stats[((size_t)b * G + g) * 2 + 1];
        g8[i] = gamma[c]; be8[i] = beta[c]; c1[i] = s_c12[g * 2]; c2[i] = s_c12[g * 2 + 1];
    }
    const size_t base = (size_t)b * HW * C + cx * 8;
    auto row = [&](const hcp_bf16x8& v, const hcp_bf16x8& d, const hcp_bf16x8& ad, int r) {
        hcp_bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (hcp_bf2f((unsigned short)v[i]) - mean8[i]) * rstd8[i];
            float dz = hcp_bf2f((unsigned short)d[i]);
            if (silu) { float z = xh * g8[i] + be8[i]; float sg = hcp_sigmoid(z); dz *= sg * (1.f + z * (1.f - sg)); }
            float dxh = dz * g8[i];
            o[i] = (short)hcp_f2bf(rstd8[i] * (dxh - c1[i] - xh * c2[i]) + (addend ? hcp_bf2f((unsigned short)ad[i]) : 0.f));
        }
        *(hcp_bf16x8*)(dx + base + (size_t)r * C) = o;
    };
    // the skip gradient is read from dy's rows where there is none (uniform select of the base pointer, no branch) and ignored
    const hcp_bf16* const adp = addend ? addend : dy;
    int r = r0 + ry;
    for (; r + 1 * R < r1; r += 2 * R) {                 // two rows of the three tensors in flight (see gn_fwd_partial)
        hcp_bf16x8 v[2], d[2], ad[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v[u] = *(const hcp_bf16x8*)(x + base + (size_t)(r + u * R) * C);
            d[u] = *(const hcp_bf16x8*)(dy + base + (size_t)(r + u * R) * C);
            ad[u] = *(const hcp_bf16x8*)(adp + base + (size_t)(r + u * R) * C);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) row(v[u], d[u], ad[u], r + u * R);
    }
    for (; r < r1; r += R)
        row(*(const hcp_bf16x8*)(x + base + (size_t)r * C), *(const hcp_bf16x8*)(dy + base + (size_t)r * C),
            *(const hcp_bf16x8*)(adp + base + (size_t)r * C), r);
}

// ------------------------------------------------------------------ GroupNorm, one launch: a (sample, group) slab per workgroup
// Wherever a group's slab — HW rows x Cg channels — is at most 128 KB (everything below the 64x64 level, and C = 320 at 64x64), ONE
// workgroup loads it into registers once (8-byte pieces of 4 channels where Cg % 4 == 0, else 4-byte pieces of 2), reduces in
// the block, and writes the result from the same registers.  One read of x instead of two, exact two-pass variance for free,
// and above all ONE launch instead of two at the ~5 us floor each.  The slab's rows are only Cg*2 = 20..160 contiguous bytes,
// but neighbouring groups' workgroups consume the rest of every line at the same time, so HBM traffic stays 1x.
HCP_DEVICE float gn_block_sum(float v, float* s_red, int tid, int nthreads) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += hcp_shfl_xor(v, off);
    HCP_SYNC();                                          // s_red may still be read from the previous reduction
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    HCP_SYNC();
    float t = 0.f;
    for (int w = 0; w < nthreads / 64; ++w) t += s_red[w];
    return t;
}
// chunk c of the slab -> element offset inside the sample (row = c / cpr through a host-computed reciprocal: exact for c < 2^16)
template <int CE>
HCP_DEVICE size_t gn_slab_off(int c, int cpr, unsigned cpr_magic, int C, int* ch_in_group) {
    const int row = cpr_magic ? (int)(((unsigned long long)(unsigned)c * cpr_magic) >> 32) : c;      // (magic 0: one chunk per row)
    *ch_in_group = (c - row * cpr) * CE;
    return (size_t)row * C + (size_t)*ch_in_group;
}
// a chunk = CE channels: 4 (8 bytes) where Cg % 4 == 0, else 2 (4 bytes: Cg = 10 / 30, i.e. C = 320 / 960)
template <int CE> struct GNChunk;
typedef float hcp_f32x2 __attribute__((ext_vector_type(2)));
template <> struct GNChunk<4> { typedef hcp_bf16x4 T; typedef hcp_f32x4 F; };
template <> struct GNChunk<2> { typedef hcp_bf16x2 T; typedef hcp_f32x2 F; };

// Both kernels are latency-bound, not bandwidth-bound (128 workgroups of one slab each: nothing else runs on the CU), so what counts is the
// number of DEPENDENT memory round trips.  The first form loaded a chunk under `if (c < total)` and used it at once — the unrolled loop
// became NCH blocks of load -> s_waitcnt vmcnt(0) -> arithmetic, and the passes that need gamma / beta fetched those per chunk as well.
// Now every pass requests all its chunks first (dead chunks read chunk 0 and are masked afterwards, so the loads are unconditional and the
// compiler keeps them together), and the group's gamma / beta sit in LDS (Cg floats each, staged once, behind s_red).
template <int NCH, int CE>
HCP_KERNEL(1024) gn_slab_fwd(const hcp_bf16* x, const float* gamma, const float* beta, float* stats, hcp_bf16* y, int HW, int C,
                             int G, int cpr, unsigned cpr_magic, int silu, float eps) {
    typedef typename GNChunk<CE>::T V;
    typedef typename GNChunk<CE>::F FV;
    HCP_DYN_SMEM(smem);
    float* s_red = (float*)smem;                          // [32 waves at most]
    const int tid = threadIdx.x, NT = blockDim.x;
    const int g = blockIdx.x, b = blockIdx.y;
    const int Cg = C / G, total = HW * cpr;
    float* s_ga = s_red + 32;                             // [Cg] gamma, [Cg] beta of this group (visible behind the first block sum's barrier)
    float* s_be = s_ga + Cg;
    for (int j = tid; j < Cg; j += NT) { s_ga[j] = gamma[g * Cg + j]; s_be[j] = beta[g * Cg + j]; }
    const size_t base = (size_t)b * HW * C + (size_t)g * Cg;
    V v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        int cg;
        v[i] = *(const V*)(x + base + gn_slab_off<CE>(c < total ? c : 0, cpr, cpr_magic, C, &cg));
        if (NCH > 8 && i % 8 == 7) hcp_sched_fence();      // (addresses of at most 8 chunks at a time)
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const bool live = tid + i * NT < total;
#pragma unroll
        for (int e = 0; e < CE; ++e) { v[i][e] = live ? v[i][e] : (short)0; s += hcp_bf2f((unsigned short)v[i][e]); }
    }
    const float n = (float)HW * Cg;
    const float mean = gn_block_sum(s, s_red, tid, NT) / n;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
        if (tid + i * NT < total) {
#pragma unroll
            for (int e = 0; e < CE; ++e) { const float d = hcp_bf2f((unsigned short)v[i][e]) - mean; q += d * d; }
        }
    const float rstd = 1.0f / sqrtf(gn_block_sum(q, s_red, tid, NT) / n + eps);
    if (tid == 0) { stats[((size_t)b * G + g) * 2] = mean; stats[((size_t)b * G + g) * 2 + 1] = rstd; }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        if (c >= total) continue;
        int cg;
        const size_t off = gn_slab_off<CE>(c, cpr, cpr_magic, C, &cg);
        const FV ga = *(const FV*)(s_ga + cg), be = *(const FV*)(s_be + cg);
        V o;
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            const float a = rstd * ga[e];
            float z = hcp_bf2f((unsigned short)v[i][e]) * a + (be[e] - mean * a);
            if (silu) z = hcp_silu(z);
            o[e] = (short)hcp_f2bf(z);
        }
        *(V*)(y + base + off) = o;
    }
}

template <int NCH, int CE>
HCP_KERNEL(1024) gn_slab_bwd(const hcp_bf16* x, const hcp_bf16* dy, const float* gamma, const float* beta, const float* stats,
                             const hcp_bf16* addend, hcp_bf16* dx, int HW, int C, int G, int cpr, unsigned cpr_magic, int silu) {
    typedef typename GNChunk<CE>::T V;
    typedef typename GNChunk<CE>::F FV;
    HCP_DYN_SMEM(smem);
    float* s_red = (float*)smem;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int g = blockIdx.x, b = blockIdx.y;
    const int Cg = C / G, total = HW * cpr;
    float* s_ga = s_red + 32;
    float* s_be = s_ga + Cg;
    for (int j = tid; j < Cg; j += NT) { s_ga[j] = gamma[g * Cg + j]; s_be[j] = beta[g * Cg + j]; }
    const size_t base = (size_t)b * HW * C + (size_t)g * Cg;
    const float mean = stats[((size_t)b * G + g) * 2], rstd = stats[((size_t)b * G + g) * 2 + 1];
    // the lane's chunks of x and dy stay in registers as loaded (bf16); xhat / dxhat are recomputed in the second pass.  The skip
    // gradient (addend) is requested with them where the registers allow (<= 32 elements per lane and tensor): it arrives while the
    // block sums run; the two largest shapes request it in the second pass, eight chunks at a time (anything more spills).
    constexpr bool EARLY_AD = NCH * CE <= 32;
    constexpr int GB = NCH < 8 ? NCH : 8;
    V v[NCH], d[NCH], ad[EARLY_AD ? NCH : 1];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + i * NT;
        int cg;
        const size_t off = gn_slab_off<CE>(c < total ? c : 0, cpr, cpr_magic, C, &cg);
        v[i] = *(const V*)(x + base + off); d[i] = *(const V*)(dy + base + off);
        if (NCH > 8 && i % 8 == 7) hcp_sched_fence();      // (addresses of at most 8 chunks at a time: 32 chunks' worth spills)
    }
    if constexpr (EARLY_AD) {
        if (addend) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + i * NT;
                int cg;
                ad[i] = *(const V*)(addend + base + gn_slab_off<CE>(c < total ? c : 0, cpr, cpr_magic, C, &cg));
            }
        }
    }
    HCP_SYNC();                                           // gamma / beta staged
    auto terms = [&](int i, int cg, float (&h)[CE], float (&dh)[CE]) {
        const FV ga = *(const FV*)(s_ga + cg), be = *(const FV*)(s_be + cg);
#pragma unroll
        for (int e = 0; e < CE; ++e) {
            h[e] = (hcp_bf2f((unsigned short)v[i][e]) - mean) * rstd;
            float dz = hcp_bf2f((unsigned short)d[i][e]);
            if (silu) { const float z = h[e] * ga[e] + be[e]; const float sg = hcp_sigmoid(z); dz *= sg * (1.f + z * (1.f - sg)); }
            dh[e] = dz * ga[e];
        }
    };
    // dxhat of the lane's elements is KEPT for the second pass where the registers allow (<= 32 elements per lane; xhat is two
    // instructions from the bf16 x that stays anyway): the kernel is VALU-bound (one workgroup per CU on half the chip, the SiLU
    // derivative per element in both passes), not memory-bound; the larger shapes recompute it.
    constexpr bool KEEP = NCH * CE <= 32;
    float dhk[KEEP ? NCH : 1][CE];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        int c = tid + i * NT;
        if (NCH > 16) hcp_force_ready(c);               // (32 chunks: recompute the chunk's position instead of keeping 32 of them live)
        if (c < total) {
            int cg;
            gn_slab_off<CE>(c, cpr, cpr_magic, C, &cg);
            float h[CE], dh[CE];
            terms(i, cg, h, dh);
#pragma unroll
            for (int e = 0; e < CE; ++e) {
                s1 += dh[e]; s2 += dh[e] * h[e];
                if constexpr (KEEP) dhk[i][e] = dh[e];
            }
        }
        if (NCH > 16 && i % 8 == 7) hcp_sched_fence();     // (32 chunks: their gamma / beta reads hoisted together spill)
    }
    const float n = (float)HW * Cg;
    const float c1 = gn_block_sum(s1, s_red, tid, NT) / n;
    const float c2 = gn_block_sum(s2, s_red, tid, NT) / n;
#pragma unroll
    for (int i0 = 0; i0 < NCH; i0 += GB) {
        V adg[GB];
        if (addend) {
#pragma unroll
            for (int j = 0; j < GB; ++j) {
                if constexpr (EARLY_AD) adg[j] = ad[i0 + j];
                else {
                    int c = tid + (i0 + j) * NT;
                    if (NCH > 16) hcp_force_ready(c);
                    int cg;
                    adg[j] = *(const V*)(addend + base + gn_slab_off<CE>(c < total ? c : 0, cpr, cpr_magic, C, &cg));
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GB; ++j) {
            const int i = i0 + j;
            int c = tid + i * NT;
            if (NCH > 16) hcp_force_ready(c);
            if (c >= total) continue;
            int cg;
            const size_t off = gn_slab_off<CE>(c, cpr, cpr_magic, C, &cg);
            float h[CE], dh[CE];
            if constexpr (KEEP) {
#pragma unroll
                for (int e = 0; e < CE; ++e) { h[e] = (hcp_bf2f((unsigned short)v[i][e]) - mean) * rstd; dh[e] = dhk[i][e]; }
            } else terms(i, cg, h, dh);
            V o;
#pragma unroll
            for (int e = 0; e < CE; ++e)
                o[e] = (short)hcp_f2bf(rstd * (dh[e] - c1 - h[e] * c2) + (addend ? hcp_bf2f((unsigned short)adg[j][e]) : 0.f));
            *(V*)(dx + base + off) = o;
        }
        if (NCH > 16) hcp_sched_fence();
    }
}

// slab path geometry: threads per workgroup and chunks per thread (0 = use the two-launch path)
struct GNSlab { int nt, nch, ce, cpr; unsigned magic; };
GNSlab gn_slab_geom(int HW, int C, int G) {
    GNSlab r = {0, 0, 0, 0, 0};
    const int Cg = C / G;
    if (Cg % 2) return r;
    const int ce = Cg % 4 == 0 ? 4 : 2;
    if ((long)HW * Cg > 65536 || Cg > 2048) return r;         // 128 KB of bf16 per slab: <= 64 elements per thread and tensor; gamma | beta of the group in LDS
    if (ce == 2 && HW > 1024) return r;                       // C = 320 at 64x64: rows of 20 bytes, measured 0.95 ms/step SLOWER than two launches
    const long total = (long)HW * (Cg / ce);
    r.ce = ce; r.cpr = Cg / ce;
    r.magic = r.cpr == 1 ? 0u : (unsigned)(((1ull << 32) + r.cpr - 1) / r.cpr);
    r.nt = total <= 256 * 8 ? 256 : 1024;
    int nch = 1;
    while ((long)nch * r.nt < total) nch *= 2;
    r.nch = nch;
    return r;
}

// instantiated (chunks per thread, channels per chunk): 8-byte chunks up to 16 per thread, 4-byte chunks up to 32
#define HCP_GN_SLAB_SWITCH(KERNEL, ...)                                                                              \
    do {                                                                                                             \
        const int key_ = sl.nch * 8 + sl.ce;                                                                         \
        switch (key_) {                                                                                              \
            case 1 * 8 + 4: HCP_LAUNCH((KERNEL<1, 4>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 2 * 8 + 4: HCP_LAUNCH((KERNEL<2, 4>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 4 * 8 + 4: HCP_LAUNCH((KERNEL<4, 4>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 8 * 8 + 4: HCP_LAUNCH((KERNEL<8, 4>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 16 * 8 + 4: HCP_LAUNCH((KERNEL<16, 4>), grid, blk, sm, stream, __VA_ARGS__); break;                 \
            case 1 * 8 + 2: HCP_LAUNCH((KERNEL<1, 2>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 2 * 8 + 2: HCP_LAUNCH((KERNEL<2, 2>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 4 * 8 + 2: HCP_LAUNCH((KERNEL<4, 2>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 8 * 8 + 2: HCP_LAUNCH((KERNEL<8, 2>), grid, blk, sm, stream, __VA_ARGS__); break;                   \
            case 16 * 8 + 2: HCP_LAUNCH((KERNEL<16, 2>), grid, blk, sm, stream, __VA_ARGS__); break;                 \
            default: HCP_LAUNCH((KERNEL<32, 2>), grid, blk, sm, stream, __VA_ARGS__); break;                         \
        }                                                                                                            \
    } while (0)

// ------------------------------------------------------------------ LayerNorm: one wave per row
// NV = 16-byte vectors per lane (C <= 512 NV): the row is loaded ONCE into registers — mean, variance and the normalised output
// (backward: both reductions and dx) come from there; the first version re-read the row from L1/L2 for every pass, three
// dependent memory round trips per wave for a 640-byte row.
template <int NV, bool HL>
HCP_KERNEL(256) ln_fwd_kernel(const hcp_bf16* x, const hcp_bf16* x_lo, const float* gamma, const float* beta, hcp_bf16* y, float* stats,
                              int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < M;
    const hcp_bf16* xr = x + (size_t)(live ? row : 0) * C;
    const int nch = C / 8;
    hcp_bf16x8 v[NV], vl[HL ? NV : 1];                  // vl (HL): the lo image of a (hi | lo) residual stream
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = lane + 64 * u;
        v[u] = j < nch ? *(const hcp_bf16x8*)(xr + j * 8) : hcp_zero8();
        if (HL) vl[u] = j < nch ? *(const hcp_bf16x8*)(x_lo + (size_t)(live ? row : 0) * C + j * 8) : hcp_zero8();
    }
#define HCP_LN_X(u_, i_) (HL ? hcp_bf2f((unsigned short)v[u_][i_]) + hcp_bf2f((unsigned short)vl[HL ? u_ : 0][i_]) : hcp_bf2f((unsigned short)v[u_][i_]))
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += HCP_LN_X(u, i);
    const float mean = hcp_wave_sum(s) / C;
    float q = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u)
        if (lane + 64 * u < nch) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { float d = HCP_LN_X(u, i) - mean; q += d * d; }
        }
    const float rstd = 1.0f / sqrtf(hcp_wave_sum(q) / C + eps);
    if (!live) return;
    if (lane == 0) { stats[(size_t)row * 2] = mean; stats[(size_t)row * 2 + 1] = rstd; }
    hcp_bf16* yr = y + (size_t)row * C;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = lane + 64 * u;
        if (j >= nch) continue;
        hcp_f32x4 g0 = *(const hcp_f32x4*)(gamma + j * 8), g1 = *(const hcp_f32x4*)(gamma + j * 8 + 4);
        hcp_f32x4 b0 = *(const hcp_f32x4*)(beta + j * 8), b1 = *(const hcp_f32x4*)(beta + j * 8 + 4);
        hcp_bf16x8 o;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float g = i < 4 ? g0[i] : g1[i - 4], bb = i < 4 ? b0[i] : b1[i - 4];
            o[i] = (short)hcp_f2bf((HCP_LN_X(u, i) - mean) * rstd * g + bb);
        }
        *(hcp_bf16x8*)(yr + j * 8) = o;
    }
}

template <int NV, bool HL>
HCP_KERNEL(256) ln_bwd_kernel(const hcp_bf16* x, const hcp_bf16* x_lo, const hcp_bf16* dy, const float* gamma, const float* stats,
                              const hcp_bf16* addend, const hcp_bf16* addend_lo, hcp_bf16* dx, hcp_bf16* dx_lo, int M, int C) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = row < M;
    const size_t off = (size_t)(live ? row : 0) * C;
    const float mean = stats[(size_t)(live ? row : 0) * 2], rstd = stats[(size_t)(live ? row : 0) * 2 + 1];
    const int nch = C / 8;
    hcp_bf16x8 v[NV], d[NV], ad[NV], vl[HL ? NV : 1], adl[HL ? NV : 1];      // HL: lo images of a (hi | lo) residual stream (each optional)
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = lane + 64 * u;
        const bool ok = j < nch;
        v[u] = ok ? *(const hcp_bf16x8*)(x + off + j * 8) : hcp_zero8();
        d[u] = ok ? *(const hcp_bf16x8*)(dy + off + j * 8) : hcp_zero8();                   // dy = 0: padding lanes add nothing
        ad[u] = (ok && addend) ? *(const hcp_bf16x8*)(addend + off + j * 8) : hcp_zero8();
        if (HL) {
            vl[u] = (ok && x_lo) ? *(const hcp_bf16x8*)(x_lo + off + j * 8) : hcp_zero8();
            adl[u] = (ok && addend_lo) ? *(const hcp_bf16x8*)(addend_lo + off + j * 8) : hcp_zero8();
        }
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = lane + 64 * u;
        if (j >= nch) continue;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (HCP_LN_X(u, i) - mean) * rstd;
            float dxh = hcp_bf2f((unsigned short)d[u][i]) * gamma[j * 8 + i];
            s1 += dxh; s2 += dxh * xh;
        }
    }
    const float c1 = hcp_wave_sum(s1) / C, c2 = hcp_wave_sum(s2) / C;
    if (!live) return;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int j = lane + 64 * u;
        if (j >= nch) continue;
        hcp_bf16x8 o, ol;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float xh = (HCP_LN_X(u, i) - mean) * rstd;
            float dxh = hcp_bf2f((unsigned short)d[u][i]) * gamma[j * 8 + i];
            float g = rstd * (dxh - c1 - xh * c2) + hcp_bf2f((unsigned short)ad[u][i]);
            if (HL) g += hcp_bf2f((unsigned short)adl[u][i]);
            o[i] = (short)hcp_f2bf(g);
            if (HL) ol[i] = (short)hcp_f2bf(g - hcp_bf2f((unsigned short)o[i]));
        }
        *(hcp_bf16x8*)(dx + off + j * 8) = o;
        if (HL && dx_lo) *(hcp_bf16x8*)(dx_lo + off + j * 8) = ol;
    }
}
#undef HCP_LN_X

// Affine-parameter gradients (full fine-tuning: every norm's weight/bias is trainable — DreamBooth.yaml:6-10):
//   dgamma[c] += sum_rows dz * xhat ; dbeta[c] += sum_rows dz ;  dz = dy [* silu'(xhat gamma + beta)]
// Thread (ch, rl) owns 8 channels x every 32nd row of its block's row range; LDS reduce over the 32 row lanes,
// fp32 atomics into the gradient bucket.  GROUP=true: stats per (sample, group), grid.z = sample; false: per row.
template <bool GROUP>
HCP_KERNEL(256) norm_affine_grad_kernel(const hcp_bf16* x, const hcp_bf16* dy, const float* gamma, const float* beta,
                                        const float* stats, float* dgamma, float* dbeta, int rows, int C, int G, int silu,
                                        int rows_per_block) {
    HCP_DYN_SMEM(smem);
    float (*red)[129] = (float (*)[129])smem;              // [32][64 dgamma | 64 dbeta]
    const int tid = threadIdx.x;
    const int ch = tid & 7, rl = tid >> 3;
    const int c0 = blockIdx.x * 64 + ch * 8;
    const int b = blockIdx.z;
    const int mb = blockIdx.y * rows_per_block;
    int me = mb + rows_per_block; if (me > rows) me = rows;
    float sg[8], sb[8], mean8[8], rstd8[8], g8[8], be8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sg[i] = 0.f; sb[i] = 0.f; mean8[i] = 0.f; rstd8[i] = 1.f; g8[i] = 1.f; be8[i] = 0.f; }
    if (c0 < C) {
        if (GROUP) {
            const int Cg = C / G;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int g = (c0 + i) / Cg;
                mean8[i] = stats[((size_t)b * G + g) * 2]; rstd8[i] = stats[((size_t)b * G + g) * 2 + 1];
                g8[i] = gamma[c0 + i]; be8[i] = beta[c0 + i];
            }
        }
        const size_t base = (size_t)b * rows * C + c0;
        for (int r = mb + rl; r < me; r += 32) {
            hcp_bf16x8 v = *(const hcp_bf16x8*)(x + base + (size_t)r * C);
            hcp_bf16x8 d = *(const hcp_bf16x8*)(dy + base + (size_t)r * C);
            float mean = 0.f, rstd = 1.f;
            if (!GROUP) { mean = stats[(size_t)r * 2]; rstd = stats[(size_t)r * 2 + 1]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (hcp_bf2f((unsigned short)v[i]) - (GROUP ? mean8[i] : mean)) * (GROUP ? rstd8[i] : rstd);
                float dz = hcp_bf2f((unsigned short)d[i]);
                if (GROUP && silu) { float z = xh * g8[i] + be8[i]; float s = hcp_sigmoid(z); dz *= s * (1.f + z * (1.f - s)); }
                sg[i] += dz * xh; sb[i] += dz;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[rl][ch * 8 + i] = sg[i]; red[rl][64 + ch * 8 + i] = sb[i]; }
    HCP_SYNC();
    if (tid < 128) {
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += red[r][tid];
        const int c = blockIdx.x * 64 + (tid & 63);
        if (c < C && mb < me) hcp_atomic_add((tid < 64 ? dgamma : dbeta) + c, t);
    }
}

int affine_grad_blocks(int rows, int col_tiles, int groups, int* rows_per_block) {
    int splits = hcp_cdiv(1024, col_tiles * groups);
    const int max_splits = hcp_cdiv(rows, 128);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    *rows_per_block = hcp_cdiv(hcp_cdiv(rows, splits), 32) * 32;
    return hcp_cdiv(rows, *rows_per_block);
}

int gn_check(int B, int HW, int C, int G) {
    HCP_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0, "groupnorm: empty problem");
    HCP_REQUIRE(C % 8 == 0 && C % G == 0 && C <= 8192 && G <= 32, "groupnorm: C=%d G=%d unsupported", C, G);
    HCP_REQUIRE(gn_geom(1, HW, C).threads >= 4 * G, "groupnorm: C=%d too small for G=%d", C, G);
    return 0;
}

}  // namespace

#if defined(HCP_TOOLS)
// TOOLS ONLY (tools/bench_norm.py): workgroups a GroupNorm launch aims for (default 512).  Changes the workspace size.
HCP_API int hcp_debug_set_gn_target(int workgroups) {
    if (workgroups == -1 || workgroups == -2) { g_gn_slab = workgroups == -1 ? 0 : 1; return 0; }   // -1: slab path off, -2: on
    HCP_REQUIRE(workgroups >= 1 && workgroups <= 65536, "hcp_debug_set_gn_target: bad arguments");
    g_gn_target_wgs = workgroups;
    return 0;
}
#endif

// Workspace (bytes) both GroupNorm entry points need: [B][nchunk][G][2] fp32 partials + [B][G][2] merged sums.
HCP_API size_t hcp_groupnorm_workspace_bytes(int B, int HW, int C, int G) {
    if (B <= 0 || HW <= 0 || C <= 0 || C % 8) return 0;
    GNGeom g = gn_geom(B, HW, C);
    return ((size_t)B * g.nchunk * G * 2 + (size_t)B * G * 2) * sizeof(float);
}

// y = [silu](group_norm(x; gamma, beta, eps)); stats[B,G,2] = (mean, rstd) saved for backward.
// Two launches: per-chunk (mean, M2) partials -> apply (every workgroup merges its sample's partials itself).
HCP_API int hcp_groupnorm_silu_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats,
                                   void* workspace, int B, int HW, int C, int G, float eps, int silu, hipStream_t stream) {
    if (int e = gn_check(B, HW, C, G)) return e;
    HCP_REQUIRE(x && gamma && beta && y && stats && workspace, "hcp_groupnorm_silu_fwd: null pointer");
    const GNSlab sl = gn_slab_geom(HW, C, G);
    if (sl.nch && g_gn_slab) {
        const hcp_bf16* xp = (const hcp_bf16*)x; hcp_bf16* yp = (hcp_bf16*)y;
        const dim3 grid(G, B), blk(sl.nt);
        const size_t sm = (32 + 2 * (size_t)(C / G)) * sizeof(float);     // block-sum scratch + the group's gamma | beta
        HCP_GN_SLAB_SWITCH(gn_slab_fwd, xp, gamma, beta, stats, yp, HW, C, G, sl.cpr, sl.magic, silu, eps);
        HCP_LAUNCH_CHECK("groupnorm_fwd (slab)");
    }
    GNGeom g = gn_geom(B, HW, C);
    size_t sm1 = (size_t)2 * g.R * C * sizeof(float);
    HCP_LAUNCH(gn_fwd_partial, dim3(g.nchunk, B), dim3(g.threads), sm1, stream, (const hcp_bf16*)x, (float*)workspace, HW, C,
               G, g.TX, g.R, g.rows_per_chunk);
    HCP_LAUNCH(gn_fwd_apply, dim3(g.nchunk, B), dim3(g.threads), gn_merge_smem(g.threads, G), stream, (const hcp_bf16*)x, gamma,
               beta, (const float*)workspace, stats, (hcp_bf16*)y, HW, C, G, g.TX, g.R, g.rows_per_chunk, silu, eps);
    HCP_LAUNCH_CHECK("groupnorm_fwd");
}

// dx for y = [silu](group_norm(x)); gamma/beta frozen (their gradients are not produced here).
// (optional addend: gradient arriving on the residual path that forks off before the norm; added in the same pass)
HCP_API int hcp_groupnorm_silu_bwd(const void* x, const void* dy, const float* gamma, const float* beta,
                                   const float* stats, const void* addend, void* dx, void* workspace, int B, int HW, int C,
                                   int G, int silu, hipStream_t stream) {
    if (int e = gn_check(B, HW, C, G)) return e;
    HCP_REQUIRE(x && dy && gamma && beta && stats && dx && workspace, "hcp_groupnorm_silu_bwd: null pointer");
    const GNSlab sl = gn_slab_geom(HW, C, G);
    if (sl.nch && g_gn_slab) {
        const hcp_bf16 *xp = (const hcp_bf16*)x, *dp = (const hcp_bf16*)dy, *ap = (const hcp_bf16*)addend; hcp_bf16* op = (hcp_bf16*)dx;
        const dim3 grid(G, B), blk(sl.nt);
        const size_t sm = (32 + 2 * (size_t)(C / G)) * sizeof(float);     // block-sum scratch + the group's gamma | beta
        HCP_GN_SLAB_SWITCH(gn_slab_bwd, xp, dp, gamma, beta, stats, ap, op, HW, C, G, sl.cpr, sl.magic, silu);
        HCP_LAUNCH_CHECK("groupnorm_bwd (slab)");
    }
    GNGeom g = gn_geom(B, HW, C);
    size_t sm1 = (size_t)2 * g.R * C * sizeof(float);
    HCP_LAUNCH(gn_bwd_partial, dim3(g.nchunk, B), dim3(g.threads), sm1, stream, (const hcp_bf16*)x, (const hcp_bf16*)dy,
               gamma, beta, stats, (float*)workspace, HW, C, G, g.TX, g.R, g.rows_per_chunk, silu);
    HCP_LAUNCH(gn_bwd_apply, dim3(g.nchunk, B), dim3(g.threads), gn_merge_smem(g.threads, G), stream, (const hcp_bf16*)x,
               (const hcp_bf16*)dy, gamma, beta, stats, (const float*)workspace, (const hcp_bf16*)addend, (hcp_bf16*)dx, HW, C, G,
               g.TX, g.R, g.rows_per_chunk, silu);
    HCP_LAUNCH_CHECK("groupnorm_bwd");
}

HCP_API int hcp_layernorm_fwd(const void* x, const void* x_lo, const float* gamma, const float* beta, void* y, float* stats, int M, int C,
                              float eps, hipStream_t stream) {
    HCP_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 4096, "hcp_layernorm_fwd: bad shape M=%d C=%d (C: multiple of 8 up to 4096)", M, C);
    HCP_REQUIRE(x && gamma && beta && y && stats, "hcp_layernorm_fwd: null pointer");
#define HCP_LN_FWD(NV_, HL_) HCP_LAUNCH((ln_fwd_kernel<NV_, HL_>), dim3(hcp_cdiv(M, 4)), dim3(256), 0, stream, (const hcp_bf16*)x, \
                                        (const hcp_bf16*)x_lo, gamma, beta, (hcp_bf16*)y, stats, M, C, eps)
    if (x_lo) { if (C <= 512) HCP_LN_FWD(1, true); else if (C <= 1024) HCP_LN_FWD(2, true); else if (C <= 2048) HCP_LN_FWD(4, true); else HCP_LN_FWD(8, true); }
    else { if (C <= 512) HCP_LN_FWD(1, false); else if (C <= 1024) HCP_LN_FWD(2, false); else if (C <= 2048) HCP_LN_FWD(4, false); else HCP_LN_FWD(8, false); }
#undef HCP_LN_FWD
    HCP_LAUNCH_CHECK("layernorm_fwd");
}

// dx = layer_norm_backward(dy) [+ addend]   (addend: gradient arriving on the residual path of a pre-norm block)
// (hi | lo) residual stream: x_lo, addend_lo, dx_lo each optional — the row is x + x_lo, the skip gradient addend + addend_lo, and
// the result g leaves as dx = bf16(g), dx_lo = bf16(g - dx).
HCP_API int hcp_layernorm_bwd(const void* x, const void* x_lo, const void* dy, const float* gamma, const float* stats, const void* addend,
                              const void* addend_lo, void* dx, void* dx_lo, int M, int C, hipStream_t stream) {
    HCP_REQUIRE(M > 0 && C > 0 && C % 8 == 0 && C <= 4096, "hcp_layernorm_bwd: bad shape M=%d C=%d (C: multiple of 8 up to 4096)", M, C);
    HCP_REQUIRE(x && dy && gamma && stats && dx, "hcp_layernorm_bwd: null pointer");
    HCP_REQUIRE(!addend_lo || addend, "hcp_layernorm_bwd: addend_lo needs addend");
#define HCP_LN_BWD(NV_, HL_) HCP_LAUNCH((ln_bwd_kernel<NV_, HL_>), dim3(hcp_cdiv(M, 4)), dim3(256), 0, stream, (const hcp_bf16*)x, \
                                        (const hcp_bf16*)x_lo, (const hcp_bf16*)dy, gamma, stats, (const hcp_bf16*)addend,       \
                                        (const hcp_bf16*)addend_lo, (hcp_bf16*)dx, (hcp_bf16*)dx_lo, M, C)
    if (x_lo || addend_lo || dx_lo) { if (C <= 512) HCP_LN_BWD(1, true); else if (C <= 1024) HCP_LN_BWD(2, true); else if (C <= 2048) HCP_LN_BWD(4, true); else HCP_LN_BWD(8, true); }
    else { if (C <= 512) HCP_LN_BWD(1, false); else if (C <= 1024) HCP_LN_BWD(2, false); else if (C <= 2048) HCP_LN_BWD(4, false); else HCP_LN_BWD(8, false); }
#undef HCP_LN_BWD
    HCP_LAUNCH_CHECK("layernorm_bwd");
}

// dgamma[C] += sum_{b,hw} dz * xhat, dbeta[C] += sum dz for y = [silu](group_norm(x)); stats from the forward.
HCP_API int hcp_groupnorm_affine_grad(const void* x, const void* dy, const float* gamma, const float* beta, const float* stats,
                                      float* dgamma, float* dbeta, int B, int HW, int C, int G, int silu, hipStream_t stream) {
    HCP_REQUIRE(x && dy && gamma && beta && stats && dgamma && dbeta, "hcp_groupnorm_affine_grad: null argument");
    HCP_REQUIRE(B > 0 && HW > 0 && C > 0 && G > 0 && C % 8 == 0 && C % G == 0, "hcp_groupnorm_affine_grad: C=%d G=%d unsupported", C, G);
    const int ct = hcp_cdiv(C, 64);
    int rpb; const int splits = affine_grad_blocks(HW, ct, B, &rpb);
    HCP_LAUNCH((norm_affine_grad_kernel<true>), dim3(ct, splits, B), dim3(256), 32 * 129 * sizeof(float), stream,
               (const hcp_bf16*)x, (const hcp_bf16*)dy, gamma, beta, stats, dgamma, dbeta, HW, C, G, silu, rpb);
    HCP_LAUNCH_CHECK("groupnorm_affine_grad");
}

// dgamma[C] += sum_m dy * xhat, dbeta[C] += sum_m dy for y = layer_norm(x); stats[M,2] = (mean, rstd) from the forward.
HCP_API int hcp_layernorm_affine_grad(const void* x, const void* dy, const float* stats, float* dgamma, float* dbeta, int M, int C,
                                      hipStream_t stream) {
    HCP_REQUIRE(x && dy && stats && dgamma && dbeta && M > 0 && C > 0 && C % 8 == 0, "hcp_layernorm_affine_grad: bad arguments");
    const int ct = hcp_cdiv(C, 64);
    int rpb; const int splits = affine_grad_blocks(M, ct, 1, &rpb);
    HCP_LAUNCH((norm_affine_grad_kernel<false>), dim3(ct, splits, 1), dim3(256), 32 * 129 * sizeof(float), stream,
               (const hcp_bf16*)x, (const hcp_bf16*)dy, (const float*)nullptr, (const float*)nullptr, stats, dgamma, dbeta, M, C, 1, 0, rpb);
    HCP_LAUNCH_CHECK("layernorm_affine_grad");
}
