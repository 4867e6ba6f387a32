// attention.hip — flash-style fused attention forward/backward for the SD UNet (gfx950, bf16, fp32 accum).
//
// Replaces diffusers' CrossAttention core (softmax(Q K^T / sqrt(d)) V; AttnProcessor2_0 -> SDPA, or xformers
// when `enable_xformers` — reference train_ac.py:258-260; call sites unet_struct.txt:17-43) for
// self-attention (Nk = Nq = H*W tokens) and cross-attention (Nk = 77*r text tokens), head_dim 40/64/80/160.
// Tensors stay in the token-major [B, N, heads*d] layout the QKV GEMMs write — no head permutes.
//
// Structure (per 256-thread workgroup = 4 waves):
//  * scores are computed TRANSPOSED, S^T = K Q^T, with mfma_f32_16x16x32_bf16, so each lane owns ONE query
//    column (lane&15) and 16 keys of the 64-key tile: row max / sum are in-lane + two cross-lane xor steps;
//  * P^T never leaves registers: MFMA sums over its 32 k-slots in any order as long as A and B agree, so the
//    lane's own 8 exponentiated scores ARE its B fragment for O^T = V^T P^T, and the matching A fragment is
//    two 8-byte reads of a transposed V copy in LDS;
//  * backward = delta pre-pass + a dQ kernel (same walk as forward) + a dK/dV kernel (walks query tiles with
//    S = Q K^T un-transposed so each lane owns one KEY column) — no atomics, deterministic.
#include "hcp_common.h"

namespace {

struct AttnParams {
    const hcp_bf16 *Q, *K, *V, *O, *dO;
    hcp_bf16 *Out, *dQ, *dK, *dV;
    float* lse;          // [B, H, Nq]
    float* delta;        // [B, H, Nq]
    long q_bs, k_bs, v_bs, o_bs;   // batch strides (elements)
    int q_rs, k_rs, v_rs, o_rs;    // token-row strides (elements); head h starts at column h*D
    int H, Nq, Nk;
    float scale;
};

constexpr int KVT = 64;            // keys per tile
constexpr int TS = KVT + 8;        // row stride (bf16) of transposed [d][64] LDS tiles

HCP_DEVICE hcp_bf16x8 pack8(const hcp_f32x4& a, const hcp_f32x4& b) {
    hcp_bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = (short)hcp_f2bf(a[i]); r[4 + i] = (short)hcp_f2bf(b[i]); }
    return r;
}
HCP_DEVICE hcp_bf16x8 join8(hcp_bf16x4 a, hcp_bf16x4 b) {
    hcp_bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
    return r;
}

// Stage `rows` x D of a token-major tensor into LDS: row-major [64][RS] zero-padded to DP columns (if rm)
// and/or transposed [DV][TS] (if tr).  Rows >= nvalid are zero.
template <int D, int DP, int DV>
HCP_DEVICE void stage_tile(const hcp_bf16* src, int rs, int nvalid, hcp_bf16* rm, hcp_bf16* tr, int tid) {
    constexpr int RS = DP + 8;
    constexpr int NC = (DP > DV ? DP : DV) / 8;
    for (int c = tid; c < KVT * NC; c += 256) {
        const int row = c / NC, dc = c - row * NC;
        hcp_bf16x8 v = hcp_zero8();
        if (row < nvalid && dc * 8 < D) v = *(const hcp_bf16x8*)(src + (size_t)row * rs + dc * 8);
        if (rm && dc * 8 < DP) *(hcp_bf16x8*)(rm + row * RS + dc * 8) = v;
        if (tr && dc * 8 < DV) {
#pragma unroll
            for (int i = 0; i < 8; ++i) tr[(dc * 8 + i) * TS + row] = (hcp_bf16)v[i];
        }
    }
}

template <int D> struct AttnGeom {
    static constexpr int DP = (D + 31) / 32 * 32;   // reduction length of the score MFMAs (zero padded)
    static constexpr int NQK = DP / 32;
    static constexpr int DV = (D + 15) / 16 * 16;   // output columns (zero padded)
    static constexpr int NDV = DV / 16;
    static constexpr int RS = DP + 8;
};

// ------------------------------------------------------------------------------------------ forward
template <int D, int QT>
HCP_KERNEL(256) attn_fwd_kernel(AttnParams p) {
    using G = AttnGeom<D>;
    HCP_DYN_SMEM(smem);
    hcp_bf16* sK = (hcp_bf16*)smem;                 // [64][RS]
    hcp_bf16* sVt = sK + KVT * G::RS;               // [DV][TS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q_base = blockIdx.x * (64 * QT) + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;

    hcp_bf16x8 qf[QT][G::NQK];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) {
            const int row = q_base + t * 16 + fr, dc = s * 32 + fg * 8;
            qf[t][s] = (row < p.Nq && dc < D) ? *(const hcp_bf16x8*)(Qb + (size_t)row * p.q_rs + dc) : hcp_zero8();
        }
    float m_i[QT], l_i[QT];
    hcp_f32x4 o[QT][G::NDV];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_i[t] = -INFINITY; l_i[t] = 0.f;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; o[t][d] = z; }
    }

    for (int kv0 = 0; kv0 < p.Nk; kv0 += KVT) {
        const int nvalid = p.Nk - kv0 < KVT ? p.Nk - kv0 : KVT;
        stage_tile<D, G::DP, G::DV>(Kb + (size_t)kv0 * p.k_rs, p.k_rs, nvalid, sK, nullptr, tid);
        stage_tile<D, G::DP, G::DV>(Vb + (size_t)kv0 * p.v_rs, p.v_rs, nvalid, nullptr, sVt, tid);
        HCP_SYNC();
        hcp_f32x4 sc[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; sc[t][kt] = z; }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int s = 0; s < G::NQK; ++s) {
                hcp_bf16x8 kf = *(const hcp_bf16x8*)(sK + (kt * 16 + fr) * G::RS + s * 32 + fg * 8);
#pragma unroll
                for (int t = 0; t < QT; ++t) sc[t][kt] = hcp_mfma16(kf, qf[t][s], sc[t][kt]);
            }
        hcp_bf16x8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * fg + r;
                    float v = key < nvalid ? sc[t][kt][r] * p.scale : -INFINITY;
                    sc[t][kt][r] = v; mx = v > mx ? v : mx;
                }
            float o1 = hcp_shfl_xor(mx, 16); mx = o1 > mx ? o1 : mx;
            o1 = hcp_shfl_xor(mx, 32); mx = o1 > mx ? o1 : mx;
            const float m_new = mx > m_i[t] ? mx : m_i[t];
            const float alpha = expf(m_i[t] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { float e = expf(sc[t][kt][r] - m_new); sc[t][kt][r] = e; rs += e; }
            rs += hcp_shfl_xor(rs, 16); rs += hcp_shfl_xor(rs, 32);
            l_i[t] = l_i[t] * alpha + rs; m_i[t] = m_new;
#pragma unroll
            for (int d = 0; d < G::NDV; ++d) o[t][d] *= alpha;
            pf[t][0] = pack8(sc[t][0], sc[t][1]);
            pf[t][1] = pack8(sc[t][2], sc[t][3]);
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const hcp_bf16* row = sVt + (d * 16 + fr) * TS + 4 * fg;
                hcp_bf16x8 vf = join8(*(const hcp_bf16x4*)(row + (2 * s2) * 16), *(const hcp_bf16x4*)(row + (2 * s2 + 1) * 16));
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][d] = hcp_mfma16(vf, pf[t][s2], o[t][d]);
            }
        HCP_SYNC();
    }
    // epilogue: lane holds O[q = q_base + t*16 + fr][d*16 + 4*fg + r]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q_base + t * 16 + fr;
        if (row >= p.Nq) continue;
        const float inv = 1.0f / l_i[t];
        hcp_bf16* orow = p.Out + (size_t)b * p.o_bs + (size_t)row * p.o_rs + h * D;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) {
            const int col = d * 16 + 4 * fg;
            if (col < D) {
                hcp_bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (short)hcp_f2bf(o[t][d][r] * inv);
                *(hcp_bf16x4*)(orow + col) = w;
            }
        }
        if (fg == 0) p.lse[((size_t)b * p.H + h) * p.Nq + row] = m_i[t] + logf(l_i[t]);
    }
}

// ------------------------------------------------------------------------------------------ delta = rowsum(dO * O)
template <int D>
HCP_KERNEL(256) attn_delta_kernel(AttnParams p, int B) {
    const long total = (long)B * p.Nq * p.H;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int h = (int)(i % p.H); long r = i / p.H; const int q = (int)(r % p.Nq); const int b = (int)(r / p.Nq);
        const hcp_bf16* o = p.O + (size_t)b * p.o_bs + (size_t)q * p.o_rs + h * D;
        const hcp_bf16* d = p.dO + (size_t)b * p.o_bs + (size_t)q * p.o_rs + h * D;
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < D / 8; ++c) {
            hcp_bf16x8 a = *(const hcp_bf16x8*)(o + c * 8), g = *(const hcp_bf16x8*)(d + c * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += hcp_bf2f((unsigned short)a[e]) * hcp_bf2f((unsigned short)g[e]);
        }
        p.delta[((size_t)b * p.H + h) * p.Nq + q] = acc;
    }
}

// ------------------------------------------------------------------------------------------ dQ
template <int D, int QT>
HCP_KERNEL(256) attn_bwd_dq_kernel(AttnParams p) {
    using G = AttnGeom<D>;
    HCP_DYN_SMEM(smem);
    hcp_bf16* sK = (hcp_bf16*)smem;                 // [64][RS]
    hcp_bf16* sV = sK + KVT * G::RS;                // [64][RS]
    hcp_bf16* sKt = sV + KVT * G::RS;               // [DV][TS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q_base = blockIdx.x * (64 * QT) + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const hcp_bf16* dOb = p.dO + (size_t)b * p.o_bs + h * D;

    hcp_bf16x8 qf[QT][G::NQK], gf[QT][G::NQK];
    float lse_i[QT], del_i[QT];
    hcp_f32x4 dq[QT][G::NDV];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q_base + t * 16 + fr;
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) {
            const int dc = s * 32 + fg * 8;
            const bool ok = row < p.Nq && dc < D;
            qf[t][s] = ok ? *(const hcp_bf16x8*)(Qb + (size_t)row * p.q_rs + dc) : hcp_zero8();
            gf[t][s] = ok ? *(const hcp_bf16x8*)(dOb + (size_t)row * p.o_rs + dc) : hcp_zero8();
        }
        lse_i[t] = row < p.Nq ? p.lse[((size_t)b * p.H + h) * p.Nq + row] : INFINITY;
        del_i[t] = row < p.Nq ? p.delta[((size_t)b * p.H + h) * p.Nq + row] : 0.f;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; dq[t][d] = z; }
    }
    for (int kv0 = 0; kv0 < p.Nk; kv0 += KVT) {
        const int nvalid = p.Nk - kv0 < KVT ? p.Nk - kv0 : KVT;
        stage_tile<D, G::DP, G::DV>(Kb + (size_t)kv0 * p.k_rs, p.k_rs, nvalid, sK, sKt, tid);
        stage_tile<D, G::DP, G::DV>(Vb + (size_t)kv0 * p.v_rs, p.v_rs, nvalid, sV, nullptr, tid);
        HCP_SYNC();
        hcp_f32x4 sc[QT][4], dp[QT][4];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; sc[t][kt] = z; dp[t][kt] = z; }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int s = 0; s < G::NQK; ++s) {
                hcp_bf16x8 kf = *(const hcp_bf16x8*)(sK + (kt * 16 + fr) * G::RS + s * 32 + fg * 8);
                hcp_bf16x8 vf = *(const hcp_bf16x8*)(sV + (kt * 16 + fr) * G::RS + s * 32 + fg * 8);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    sc[t][kt] = hcp_mfma16(kf, qf[t][s], sc[t][kt]);
                    dp[t][kt] = hcp_mfma16(vf, gf[t][s], dp[t][kt]);
                }
            }
        hcp_bf16x8 df[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt * 16 + 4 * fg + r;
                    float pr = key < nvalid ? expf(sc[t][kt][r] * p.scale - lse_i[t]) : 0.f;
                    sc[t][kt][r] = pr * (dp[t][kt][r] - del_i[t]) * p.scale;
                }
            df[t][0] = pack8(sc[t][0], sc[t][1]);
            df[t][1] = pack8(sc[t][2], sc[t][3]);
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const hcp_bf16* row = sKt + (d * 16 + fr) * TS + 4 * fg;
                hcp_bf16x8 kf = join8(*(const hcp_bf16x4*)(row + (2 * s2) * 16), *(const hcp_bf16x4*)(row + (2 * s2 + 1) * 16));
#pragma unroll
                for (int t = 0; t < QT; ++t) dq[t][d] = hcp_mfma16(kf, df[t][s2], dq[t][d]);
            }
        HCP_SYNC();
    }
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q_base + t * 16 + fr;
        if (row >= p.Nq) continue;
        hcp_bf16* orow = p.dQ + (size_t)b * p.q_bs + (size_t)row * p.q_rs + h * D;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) {
            const int col = d * 16 + 4 * fg;
            if (col < D) {
                hcp_bf16x4 w;
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = (short)hcp_f2bf(dq[t][d][r]);
                *(hcp_bf16x4*)(orow + col) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
template <int D, int KT>
HCP_KERNEL(256) attn_bwd_dkv_kernel(AttnParams p) {
    using G = AttnGeom<D>;
    HCP_DYN_SMEM(smem);
    hcp_bf16* sQ = (hcp_bf16*)smem;                 // [64][RS]
    hcp_bf16* sG = sQ + KVT * G::RS;                // [64][RS]   dO
    hcp_bf16* sQt = sG + KVT * G::RS;               // [DV][TS]
    hcp_bf16* sGt = sQt + G::DV * TS;               // [DV][TS]
    float* sL = (float*)(sGt + G::DV * TS);         // [64] lse, [64] delta
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int k_base = blockIdx.x * (64 * KT) + wave * (16 * KT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const hcp_bf16* dOb = p.dO + (size_t)b * p.o_bs + h * D;

    hcp_bf16x8 kf[KT][G::NQK], vf[KT][G::NQK];
    hcp_f32x4 dk[KT][G::NDV], dv[KT][G::NDV];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int row = k_base + t * 16 + fr;
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) {
            const int dc = s * 32 + fg * 8;
            const bool ok = row < p.Nk && dc < D;
            kf[t][s] = ok ? *(const hcp_bf16x8*)(Kb + (size_t)row * p.k_rs + dc) : hcp_zero8();
            vf[t][s] = ok ? *(const hcp_bf16x8*)(Vb + (size_t)row * p.v_rs + dc) : hcp_zero8();
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; dk[t][d] = z; dv[t][d] = z; }
    }
    for (int q0 = 0; q0 < p.Nq; q0 += KVT) {
        const int nvalid = p.Nq - q0 < KVT ? p.Nq - q0 : KVT;
        stage_tile<D, G::DP, G::DV>(Qb + (size_t)q0 * p.q_rs, p.q_rs, nvalid, sQ, sQt, tid);
        stage_tile<D, G::DP, G::DV>(dOb + (size_t)q0 * p.o_rs, p.o_rs, nvalid, sG, sGt, tid);
        if (tid < KVT) {
            const bool ok = tid < nvalid;
            sL[tid] = ok ? p.lse[((size_t)b * p.H + h) * p.Nq + q0 + tid] : INFINITY;
            sL[KVT + tid] = ok ? p.delta[((size_t)b * p.H + h) * p.Nq + q0 + tid] : 0.f;
        }
        HCP_SYNC();
        hcp_f32x4 sc[KT][4], dp[KT][4];
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; sc[t][qt] = z; dp[t][qt] = z; }
#pragma unroll
        for (int qt = 0; qt < 4; ++qt)
#pragma unroll
            for (int s = 0; s < G::NQK; ++s) {
                hcp_bf16x8 qa = *(const hcp_bf16x8*)(sQ + (qt * 16 + fr) * G::RS + s * 32 + fg * 8);
                hcp_bf16x8 ga = *(const hcp_bf16x8*)(sG + (qt * 16 + fr) * G::RS + s * 32 + fg * 8);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    sc[t][qt] = hcp_mfma16(qa, kf[t][s], sc[t][qt]);   // S[q = qt*16 + 4fg + r][key = fr]
                    dp[t][qt] = hcp_mfma16(ga, vf[t][s], dp[t][qt]);
                }
            }
        hcp_bf16x8 pf[KT][2], df[KT][2];
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const bool kok = k_base + t * 16 + fr < p.Nk;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qt * 16 + 4 * fg + r;
                    float pr = kok ? expf(sc[t][qt][r] * p.scale - sL[q]) : 0.f;
                    sc[t][qt][r] = pr;
                    dp[t][qt][r] = pr * (dp[t][qt][r] - sL[KVT + q]) * p.scale;
                }
            pf[t][0] = pack8(sc[t][0], sc[t][1]); pf[t][1] = pack8(sc[t][2], sc[t][3]);
            df[t][0] = pack8(dp[t][0], dp[t][1]); df[t][1] = pack8(dp[t][2], dp[t][3]);
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const hcp_bf16* rq = sQt + (d * 16 + fr) * TS + 4 * fg;
                const hcp_bf16* rg = sGt + (d * 16 + fr) * TS + 4 * fg;
                hcp_bf16x8 qa = join8(*(const hcp_bf16x4*)(rq + (2 * s2) * 16), *(const hcp_bf16x4*)(rq + (2 * s2 + 1) * 16));
                hcp_bf16x8 ga = join8(*(const hcp_bf16x4*)(rg + (2 * s2) * 16), *(const hcp_bf16x4*)(rg + (2 * s2 + 1) * 16));
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    dv[t][d] = hcp_mfma16(ga, pf[t][s2], dv[t][d]);   // dV^T[dcol][key] += dO^T P
                    dk[t][d] = hcp_mfma16(qa, df[t][s2], dk[t][d]);   // dK^T[dcol][key] += Q^T dS
                }
            }
        HCP_SYNC();
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int row = k_base + t * 16 + fr;
        if (row >= p.Nk) continue;
        hcp_bf16* krow = p.dK + (size_t)b * p.k_bs + (size_t)row * p.k_rs + h * D;
        hcp_bf16* vrow = p.dV + (size_t)b * p.v_bs + (size_t)row * p.v_rs + h * D;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) {
            const int col = d * 16 + 4 * fg;
            if (col < D) {
                hcp_bf16x4 wk, wv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { wk[r] = (short)hcp_f2bf(dk[t][d][r]); wv[r] = (short)hcp_f2bf(dv[t][d][r]); }
                *(hcp_bf16x4*)(krow + col) = wk;
                *(hcp_bf16x4*)(vrow + col) = wv;
            }
        }
    }
}

template <int D, int QT>
int launch_fwd(AttnParams& p, int B, hipStream_t stream) {
    using G = AttnGeom<D>;
    size_t smem = (size_t)(KVT * G::RS + G::DV * TS) * sizeof(hcp_bf16);
    HCP_LAUNCH((attn_fwd_kernel<D, QT>), dim3(hcp_cdiv(p.Nq, 64 * QT), p.H, B), dim3(256), smem, stream, p);
    HCP_LAUNCH_CHECK("attn_fwd");
}
template <int D>
int launch_delta(AttnParams& p, int B, hipStream_t stream) {
    long tot = (long)B * p.Nq * p.H;
    int g = (int)((tot + 255) / 256); if (g > 4096) g = 4096;
    HCP_LAUNCH((attn_delta_kernel<D>), dim3(g), dim3(256), 0, stream, p, B);
    HCP_LAUNCH_CHECK("attn_delta");
}
template <int D, int QT>
int launch_dq(AttnParams& p, int B, hipStream_t stream) {
    using G = AttnGeom<D>;
    size_t s1 = (size_t)(2 * KVT * G::RS + G::DV * TS) * sizeof(hcp_bf16);
    HCP_LAUNCH((attn_bwd_dq_kernel<D, QT>), dim3(hcp_cdiv(p.Nq, 64 * QT), p.H, B), dim3(256), s1, stream, p);
    HCP_LAUNCH_CHECK("attn_bwd_dq");
}
template <int D, int KT>
int launch_dkv(AttnParams& p, int B, hipStream_t stream) {
    using G = AttnGeom<D>;
    size_t s2 = (size_t)(2 * KVT * G::RS + 2 * G::DV * TS) * sizeof(hcp_bf16) + 2 * KVT * sizeof(float);
    HCP_LAUNCH((attn_bwd_dkv_kernel<D, KT>), dim3(hcp_cdiv(p.Nk, 64 * KT), p.H, B), dim3(256), s2, stream, p);
    HCP_LAUNCH_CHECK("attn_bwd_dkv");
}
template <int D, int QTMAX, int KTMAX>
int launch_bwd(AttnParams& p, int B, hipStream_t stream) {
    if (int e = launch_delta<D>(p, B, stream)) return e;
    const bool small_q = QTMAX == 1 || (long)B * p.H * hcp_cdiv(p.Nq, 128) < 256;
    const bool small_k = KTMAX == 1 || (long)B * p.H * hcp_cdiv(p.Nk, 128) < 256;
    if (int e = small_q ? launch_dq<D, 1>(p, B, stream) : launch_dq<D, QTMAX>(p, B, stream)) return e;
    return small_k ? launch_dkv<D, 1>(p, B, stream) : launch_dkv<D, KTMAX>(p, B, stream);
}

int attn_check(const AttnParams& p, int B, int D) {
    HCP_REQUIRE(B > 0 && p.H > 0 && p.Nq > 0 && p.Nk > 0, "attention: empty problem");
    HCP_REQUIRE(D == 40 || D == 64 || D == 80 || D == 160, "attention: head_dim %d unsupported (40/64/80/160)", D);
    HCP_REQUIRE(p.q_rs % 8 == 0 && p.k_rs % 8 == 0 && p.v_rs % 8 == 0 && p.o_rs % 8 == 0, "attention: row strides must be multiples of 8");
    HCP_REQUIRE(p.q_bs % 8 == 0 && p.k_bs % 8 == 0 && p.v_bs % 8 == 0 && p.o_bs % 8 == 0, "attention: batch strides must be multiples of 8");
    return 0;
}

}  // namespace

// O[b,q,h,:] = softmax_k(scale * Q[b,q,h,:].K[b,k,h,:]) V[b,k,h,:];  lse[b,h,q] = logsumexp of the scaled scores.
// All tensors bf16, token-major: element (b, n, h, c) at  base + b*bs + n*rs + h*D + c.
HCP_API int hcp_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H, int Nq, int Nk,
                              int D, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                              float scale, hipStream_t stream) {
    AttnParams p = {};
    p.Q = (const hcp_bf16*)Q; p.K = (const hcp_bf16*)K; p.V = (const hcp_bf16*)V; p.Out = (hcp_bf16*)O; p.lse = lse;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
    HCP_REQUIRE(Q && K && V && O && lse, "hcp_attention_fwd: null pointer");
    if (int e = attn_check(p, B, D)) return e;
    const bool small = (long)B * H * hcp_cdiv(Nq, 128) < 256;   // too few workgroups: use 64-row query blocks
    switch (D) {
        case 40: return small ? launch_fwd<40, 1>(p, B, stream) : launch_fwd<40, 2>(p, B, stream);
        case 64: return small ? launch_fwd<64, 1>(p, B, stream) : launch_fwd<64, 2>(p, B, stream);
        case 80: return small ? launch_fwd<80, 1>(p, B, stream) : launch_fwd<80, 2>(p, B, stream);
        default: return launch_fwd<160, 1>(p, B, stream);
    }
}

// Gradients of hcp_attention_fwd.  delta is a [B,H,Nq] fp32 scratch buffer (caller-provided workspace).
HCP_API int hcp_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                              float* delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int D, long q_bs,
                              int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                              hipStream_t stream) {
    AttnParams p = {};
    p.Q = (const hcp_bf16*)Q; p.K = (const hcp_bf16*)K; p.V = (const hcp_bf16*)V; p.O = (const hcp_bf16*)O;
    p.dO = (const hcp_bf16*)dO; p.lse = (float*)lse; p.delta = delta;
    p.dQ = (hcp_bf16*)dQ; p.dK = (hcp_bf16*)dK; p.dV = (hcp_bf16*)dV;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale;
    HCP_REQUIRE(Q && K && V && O && dO && lse && delta && dQ && dK && dV, "hcp_attention_bwd: null pointer");
    if (int e = attn_check(p, B, D)) return e;
    switch (D) {
        case 40: return launch_bwd<40, 2, 2>(p, B, stream);
        case 64: return launch_bwd<64, 2, 2>(p, B, stream);
        case 80: return launch_bwd<80, 2, 1>(p, B, stream);
        default: return launch_bwd<160, 1, 1>(p, B, stream);
    }
}
