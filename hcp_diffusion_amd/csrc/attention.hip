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
//    two ds_read_b64_tr_b16 (LDS transpose reads) of the ROW-major V tile — no transposed copy is ever staged;
//  * K/V (or Q/dO) tiles are double-buffered in LDS; the next tile's global loads are issued into registers
//    before the current tile's MFMAs and written to the other buffer after them: one barrier per tile;
//  * softmax runs in the exp2 domain with scale*log2(e) folded into one multiply; fully valid tiles skip masking;
//  * backward = delta pre-pass + a dQ kernel (same walk as forward) + a dK/dV kernel (walks query tiles with
//    S = Q K^T un-transposed so each lane owns one KEY column) — no atomics, deterministic.
#include "hcp_common.h"

namespace {

struct AttnParams {
    const hcp_bf16 *Q, *K, *V, *O, *dO;
    hcp_bf16 *Out, *dQ, *dK, *dV;
    float* lse;          // [B, H, Nq]  natural-log logsumexp of the scaled scores
    float* delta;        // [B, H, Nq]
    long q_bs, k_bs, v_bs, o_bs;   // batch strides (elements)
    int q_rs, k_rs, v_rs, o_rs;    // token-row strides (elements); head h starts at column h*D
    int H, Nq, Nk;
    float scale;
    // optional additive key bias [B, Nk] fp32 (diffusers' encoder_attention_mask -> (1 - mask) * -10000, added to the SCALED
    // scores of every head and query; reference models/wrapper.py:22-23,29): applied as bias/scale on the raw scores
    const float* kbias; long kb_bs;
    int causal;                     // 1: key k is visible to query q only if k <= q (CLIP text encoder); handled by the KB instantiations
    int dbg;              // tools/ablate_attn.py only (wrong results): 1 = no global loads inside the tile loop, 2 = no barriers in the loop
    // dK/dV kernel: the query loop may be split over gridDim.x / nkv workgroups that accumulate into fp32 buffers
    int qsplit;           // number of query-range splits (1 = none)
    float* dk32; float* dv32;   // [B, Nk, H*D] fp32 accumulators when qsplit > 1
};

constexpr int KVT = 64;            // keys (or queries, in the dK/dV kernel) per tile
constexpr int TS = KVT + 8;        // row stride (bf16) of transposed [d][64] LDS tiles
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

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

// A-operand fragment for O^T / dQ^T / dK^T / dV^T products: row-major tile [64][RS] (rows = keys or queries),
// output columns c0..c0+15, k-slots = rows {(2 s2) 16 + 4 fg + j} U {(2 s2 + 1) 16 + 4 fg + j}  (j = 0..3) — the same
// row set the lane's score registers hold, so P / dS never leave registers.  Two LDS transpose reads.
HCP_DEVICE hcp_bf16x8 tr_frag(const hcp_bf16* tile, int RSv, int c0, int s2, int fr, int fg) {
    const hcp_bf16* a = tile + ((2 * s2) * 16 + 4 * fg + (fr >> 2)) * RSv + c0 + 4 * (fr & 3);
    return join8(hcp_lds_read_tr4(a), hcp_lds_read_tr4(a + 16 * RSv));
}

template <int D> struct AttnGeom {
    static constexpr int DP = (D + 31) / 32 * 32;   // reduction length of the score MFMAs (zero padded)
    static constexpr int NQK = DP / 32;
    static constexpr int DV = (D + 15) / 16 * 16;   // output columns (zero padded)
    static constexpr int NDV = DV / 16;
    static constexpr int RS = DP + 8;               // row stride (bf16) of row-major [64][DP] LDS tiles
    static constexpr int RM_ELEMS = KVT * RS;
    static constexpr int TR_ELEMS = DV * TS;
};

// Per-thread staging registers for one [64][D] tile (only the D/8 real 16-byte chunks of each row move;
// the zero padding of the LDS images is written once at kernel start).
template <int D>
struct TileStage {
    static constexpr int NC = D / 8;
    static constexpr int IT = (KVT * NC + 255) / 256;
    hcp_bf16x8 r[IT];
    // buffer-addressed: the resource covers exactly the `nvalid` live rows of this tile, so rows past the end (ragged last
    // tile) and the unused lanes of the last pass read zeros in hardware — no exec-mask branches, 32-bit offsets only
    HCP_MEMBER void load(const hcp_bf16* src, int rs, int nvalid, int tid) {
        const hcp_rsrc rsrc = hcp_make_rsrc_n(src, (unsigned)(nvalid > 0 ? ((nvalid - 1) * rs + D) * 2 : 0));
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = tid + 256 * i;
            const int row = c / NC, dc = c - row * NC;
            r[i] = hcp_buf_load16(rsrc, c < KVT * NC ? (unsigned)((row * rs + dc * 8) * 2) : HCP_BUF_OOB);
        }
    }
    HCP_MEMBER void store_rm(hcp_bf16* rm, int RSv, int tid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = tid + 256 * i;
            const int row = c / NC, dc = c - row * NC;
            if (c < KVT * NC) *(hcp_bf16x8*)(rm + row * RSv + dc * 8) = r[i];
        }
    }
    HCP_MEMBER void store_tr(hcp_bf16* tr, int tid) const {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int c = tid + 256 * i;
            const int row = c / NC, dc = c - row * NC;
            if (c < KVT * NC) {
#pragma unroll
                for (int e = 0; e < 8; ++e) tr[(dc * 8 + e) * TS + row] = (hcp_bf16)r[i][e];
            }
        }
    }
};

HCP_DEVICE void zero_lds(hcp_bf16* p, int elems, int tid) {
    for (int i = tid * 8; i < elems; i += 256 * 8) *(hcp_bf16x8*)(p + i) = hcp_zero8();
}

// ------------------------------------------------------------------------------------------ forward
// KB: compile-time "has additive key bias" — the unmasked instantiation is exactly the bias-free code (the run-time
// branch cost 10-50 VGPRs and a wave of occupancy in the dQ kernel).
// Occupancy targets (amdgpu_waves_per_eu): left alone the compiler parks MFMA accumulators in AGPRs "because there is room"
// and lands at VGPR+AGPR > 256 = ONE wave per SIMD for several variants (all d=160 kernels, dK/dV wide); with an explicit
// target it fits the same code into the VGPR budget of 2-4 waves/SIMD with (almost) no scratch.
#if defined(HCP_EMU)
#define HCP_WAVES_PER_SIMD(n)
#else
#define HCP_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n, n)))
#endif
template <int D, int QT, bool KB = false>
HCP_WAVES_PER_SIMD(D > 80 ? 2 : QT == 2 ? 3 : 4) HCP_KERNEL(256) attn_fwd_kernel(AttnParams p) {
    using G = AttnGeom<D>;
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;                // 2 x { K [64][RS] | V [64][RS] }   (both row-major)
    constexpr int BUF = 2 * G::RM_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q_base = blockIdx.x * (64 * QT) + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const float c2 = p.scale * LOG2E;

    // Row sums ride on the PV MFMA when V's column padding has room: column D of the V image is 1.0, so output row D
    // of O^T accumulates sum_k P[k][q] (and is rescaled with O) — no VALU adds / cross-lane sums per tile.
    constexpr bool ONES_COL = G::DV > D;
    constexpr float RESCALE_LOG2 = 6.0f;
    zero_lds(lds, 2 * BUF, tid);
    TileStage<D> sk, sv;
    const int nt = (p.Nk + KVT - 1) / KVT;
    sk.load(Kb, p.k_rs, p.Nk < KVT ? p.Nk : KVT, tid);
    sv.load(Vb, p.v_rs, p.Nk < KVT ? p.Nk : KVT, tid);

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
    HCP_SYNC();                                      // zero fill complete
    if (ONES_COL && tid < 2 * KVT) lds[(tid >> 6) * BUF + G::RM_ELEMS + (tid & 63) * G::RS + D] = 0x3F80;   // bf16 1.0
    sk.store_rm(lds, G::RS, tid); sv.store_rm(lds + G::RM_ELEMS, G::RS, tid);
    HCP_SYNC();

#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) hcp_force_ready(qf[t][s]);      // retire the Q loads here, not inside the loop
    for (int it = 0; it < nt; ++it) {
        const int kv0 = it * KVT;
        const int nvalid = p.Nk - kv0 < KVT ? p.Nk - kv0 : KVT;
        const hcp_bf16* sK = lds + (it & 1) * BUF;
        const hcp_bf16* sV = sK + G::RM_ELEMS;
        if (it + 1 < nt && !(p.dbg & 1)) {
            const int nv = p.Nk - kv0 - KVT < KVT ? p.Nk - kv0 - KVT : KVT;
            sk.load(Kb + (size_t)(kv0 + KVT) * p.k_rs, p.k_rs, nv, tid);
            sv.load(Vb + (size_t)(kv0 + KVT) * p.v_rs, p.v_rs, nv, tid);
        }
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
        if (KB) {                                   // wave-uniform
            const float* kb = p.kbias ? p.kbias + (size_t)b * p.kb_bs + kv0 : nullptr;
            const float inv = 1.0f / p.scale;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = kt * 16 + 4 * fg + r;
                    const float bv = (kb && kk < nvalid) ? kb[kk] * inv : 0.f;
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        sc[t][kt][r] += bv;
                        if (p.causal && kv0 + kk > q_base + t * 16 + fr) sc[t][kt][r] = -INFINITY;     // future key: p = exp2(-inf) = 0
                    }
                }
        }
        hcp_bf16x8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            // running max m_i is kept on the RAW scores; p = exp2(s*c2 - m*c2) is one FMA + v_exp_f32
            float mx = -INFINITY;
            if (nvalid != KVT) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if ((kt * 16 + 4 * fg + r) >= nvalid) sc[t][kt][r] = -INFINITY;
            }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, sc[t][kt][r]);
            mx = fmaxf(mx, hcp_shfl_xor(mx, 16));
            mx = fmaxf(mx, hcp_shfl_xor(mx, 32));
            // lazy rescale: keep the old reference max while the new one is < 2^RESCALE_LOG2 above it (p stays bounded);
            // the accumulator / row-sum rescale then runs on few tiles only.  Wave-uniform decision.
            if (!hcp_all((mx - m_i[t]) * c2 <= RESCALE_LOG2)) {
                const float m_new = fmaxf(mx, m_i[t]);
                const float alpha = hcp_exp2((m_i[t] - m_new) * c2);      // m_i = -inf on the first tile -> 0
                m_i[t] = m_new;
                l_i[t] *= alpha;
#pragma unroll
                for (int d = 0; d < G::NDV; ++d) o[t][d] *= alpha;
            }
            const float mc = m_i[t] * c2;
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float e = hcp_exp2(fmaf(sc[t][kt][r], c2, -mc));
                    sc[t][kt][r] = e;
                    if (!ONES_COL) rs += e;
                }
            if (!ONES_COL) { rs += hcp_shfl_xor(rs, 16); rs += hcp_shfl_xor(rs, 32); l_i[t] += rs; }
            pf[t][0] = pack8(sc[t][0], sc[t][1]);
            pf[t][1] = pack8(sc[t][2], sc[t][3]);
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                hcp_bf16x8 vf = tr_frag(sV, G::RS, d * 16, s2, fr, fg);
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][d] = hcp_mfma16(vf, pf[t][s2], o[t][d]);
            }
        if (it + 1 < nt) {
            hcp_bf16* nK = lds + ((it + 1) & 1) * BUF;
            sk.store_rm(nK, G::RS, tid); sv.store_rm(nK + G::RM_ELEMS, G::RS, tid);
        }
        if (!(p.dbg & 2)) HCP_SYNC();
    }
    // epilogue: lane holds O[q = q_base + t*16 + fr][d*16 + 4*fg + r]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q_base + t * 16 + fr;
        float lsum = l_i[t];
        if (ONES_COL) lsum = hcp_shfl(o[t][D / 16][D % 16 % 4], ((D % 16) / 4) * 16 + fr);   // O^T row D lives in lane group (D%16)/4
        if (row >= p.Nq) continue;
        const float inv = 1.0f / lsum;
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
        if (fg == 0) p.lse[((size_t)b * p.H + h) * p.Nq + row] = (m_i[t] * c2 + log2f(lsum)) * LN2;
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
template <int D, int QT, bool KB = false>
HCP_WAVES_PER_SIMD(D > 80 ? 2 : 3) HCP_KERNEL(256) attn_bwd_dq_kernel(AttnParams p) {
    using G = AttnGeom<D>;
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;                // 2 x { K [64][RS] | V [64][RS] }
    constexpr int BUF = 2 * G::RM_ELEMS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q_base = blockIdx.x * (64 * QT) + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const hcp_bf16* dOb = p.dO + (size_t)b * p.o_bs + h * D;
    const float c2 = p.scale * LOG2E;

    zero_lds(lds, 2 * BUF, tid);
    TileStage<D> sk, sv;
    const int nt = (p.Nk + KVT - 1) / KVT;
    sk.load(Kb, p.k_rs, p.Nk < KVT ? p.Nk : KVT, tid);
    sv.load(Vb, p.v_rs, p.Nk < KVT ? p.Nk : KVT, tid);

    hcp_bf16x8 qf[QT][G::NQK], gf[QT][G::NQK];
    float lse2[QT], del_i[QT];
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
        lse2[t] = row < p.Nq ? p.lse[((size_t)b * p.H + h) * p.Nq + row] * LOG2E : INFINITY;
        del_i[t] = row < p.Nq ? p.delta[((size_t)b * p.H + h) * p.Nq + row] : 0.f;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; dq[t][d] = z; }
    }
    HCP_SYNC();
    sk.store_rm(lds, G::RS, tid); sv.store_rm(lds + G::RM_ELEMS, G::RS, tid);
    HCP_SYNC();

#pragma unroll
    for (int t = 0; t < QT; ++t) {
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) { hcp_force_ready(qf[t][s]); hcp_force_ready(gf[t][s]); }
        hcp_force_ready(lse2[t]); hcp_force_ready(del_i[t]);
    }
    for (int it = 0; it < nt; ++it) {
        const int kv0 = it * KVT;
        const int nvalid = p.Nk - kv0 < KVT ? p.Nk - kv0 : KVT;
        const hcp_bf16* sK = lds + (it & 1) * BUF;
        const hcp_bf16* sV = sK + G::RM_ELEMS;
        if (it + 1 < nt && !(p.dbg & 1)) {
            const int nv = p.Nk - kv0 - KVT < KVT ? p.Nk - kv0 - KVT : KVT;
            sk.load(Kb + (size_t)(kv0 + KVT) * p.k_rs, p.k_rs, nv, tid);
            sv.load(Vb + (size_t)(kv0 + KVT) * p.v_rs, p.v_rs, nv, tid);
        }
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
        if (KB) {
            const float* kb = p.kbias ? p.kbias + (size_t)b * p.kb_bs + kv0 : nullptr;
            const float inv = 1.0f / p.scale;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = kt * 16 + 4 * fg + r;
                    const float bv = (kb && kk < nvalid) ? kb[kk] * inv : 0.f;
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        sc[t][kt][r] += bv;
                        if (p.causal && kv0 + kk > q_base + t * 16 + fr) sc[t][kt][r] = -INFINITY;     // future key: p = exp2(-inf) = 0
                    }
                }
        }
        hcp_bf16x8 df[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = hcp_exp2(fmaf(sc[t][kt][r], c2, -lse2[t]));
                    if (nvalid != KVT && (kt * 16 + 4 * fg + r) >= nvalid) pr = 0.f;
                    sc[t][kt][r] = pr * (dp[t][kt][r] - del_i[t]);          // softmax scale applied once, at the store
                }
            df[t][0] = pack8(sc[t][0], sc[t][1]);
            df[t][1] = pack8(sc[t][2], sc[t][3]);
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                hcp_bf16x8 kf = tr_frag(sK, G::RS, d * 16, s2, fr, fg);
#pragma unroll
                for (int t = 0; t < QT; ++t) dq[t][d] = hcp_mfma16(kf, df[t][s2], dq[t][d]);
            }
        if (it + 1 < nt) {
            hcp_bf16* nb = lds + ((it + 1) & 1) * BUF;
            sk.store_rm(nb, G::RS, tid); sv.store_rm(nb + G::RM_ELEMS, G::RS, tid);
        }
        if (!(p.dbg & 2)) HCP_SYNC();
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
                for (int r = 0; r < 4; ++r) w[r] = (short)hcp_f2bf(dq[t][d][r] * p.scale);
                *(hcp_bf16x4*)(orow + col) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
template <int D, int KT, bool KB = false>
HCP_WAVES_PER_SIMD((D > 80 || KT == 2) ? 2 : 3) HCP_KERNEL(256) attn_bwd_dkv_kernel(AttnParams p) {
    using G = AttnGeom<D>;
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;     // 2 x { Q [64][RS] | dO [64][RS] | lse2[64], delta[64] (fp32) }
    constexpr int BUF = 2 * G::RM_ELEMS + 4 * KVT;   // 2*64 floats = 4*64 bf16 slots
    constexpr int NB = (2 * BUF * 2 <= 160 * 1024) ? 2 : 1;            // head_dim 160: one buffer, two barriers per tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int kblk = blockIdx.x / p.qsplit, qs = blockIdx.x - kblk * p.qsplit;
    const int k_base = kblk * (64 * KT) + wave * (16 * KT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const hcp_bf16* dOb = p.dO + (size_t)b * p.o_bs + h * D;
    const float* lse_b = p.lse + ((size_t)b * p.H + h) * p.Nq;
    const float* del_b = p.delta + ((size_t)b * p.H + h) * p.Nq;
    const float c2 = p.scale * LOG2E;

    zero_lds(lds, NB * BUF, tid);
    TileStage<D> sq, sg;
    float rl = 0.f;                                  // staged lse2 (tid < 64) / delta (64 <= tid < 128)
    const int nt_all = (p.Nq + KVT - 1) / KVT;
    const int per = (nt_all + p.qsplit - 1) / p.qsplit;
    const int it0 = qs * per;
    const int nt = it0 + per < nt_all ? it0 + per : nt_all;      // this workgroup walks query tiles [it0, nt)
    // raw value only: the LOG2E scaling / out-of-range defaults are applied when it is stored to LDS, so that nothing waits on
    // this load (a use right here makes the compiler drain vmcnt — including the Q/dO tile prefetch issued just before)
    const float* stat_ptr = tid < KVT ? lse_b : del_b - KVT;           // lse for tid < 64, delta for 64 <= tid < 128
    bool rl_ok = false;
    auto load_stats = [&](int q0) {
        rl_ok = tid < 2 * KVT && q0 + (tid & (KVT - 1)) < p.Nq;
        rl = rl_ok ? stat_ptr[q0 + tid] : 0.f;
    };
    {
        const int q0 = it0 * KVT;
        const int nv0 = p.Nq - q0 < KVT ? (p.Nq - q0 > 0 ? p.Nq - q0 : 0) : KVT;
        sq.load(Qb + (size_t)q0 * p.q_rs, p.q_rs, nv0, tid);
        sg.load(dOb + (size_t)q0 * p.o_rs, p.o_rs, nv0, tid);
        load_stats(q0);
    }

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
    auto store_all = [&](hcp_bf16* base) {
        sq.store_rm(base, G::RS, tid); sg.store_rm(base + G::RM_ELEMS, G::RS, tid);
        float* sl = (float*)(base + 2 * G::RM_ELEMS);
        if (tid < 2 * KVT) sl[tid] = tid < KVT ? (rl_ok ? rl * LOG2E : INFINITY) : (rl_ok ? rl : 0.f);
    };
    HCP_SYNC();
    store_all(lds);
    HCP_SYNC();

#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) { hcp_force_ready(kf[t][s]); hcp_force_ready(vf[t][s]); }
    for (int it = it0; it < nt; ++it) {
        const int q0 = it * KVT;
        const hcp_bf16* sQ = lds + (NB == 2 ? ((it - it0) & 1) : 0) * BUF;
        const hcp_bf16* sG = sQ + G::RM_ELEMS;
        const float* sL = (const float*)(sG + G::RM_ELEMS);
        if (it + 1 < nt && !(p.dbg & 1)) {
            const int nv = p.Nq - q0 - KVT < KVT ? p.Nq - q0 - KVT : KVT;
            sq.load(Qb + (size_t)(q0 + KVT) * p.q_rs, p.q_rs, nv, tid);
            sg.load(dOb + (size_t)(q0 + KVT) * p.o_rs, p.o_rs, nv, tid);
            load_stats(q0 + KVT);
        }
        // two halves of the 64-query tile, one after the other: only 2 x KT score / dP blocks are live at a time
        // (all four at once put this kernel at 266 VGPR+AGPR = ONE wave per SIMD)
        hcp_bf16x8 pf[KT][2], df[KT][2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            hcp_f32x4 sc[KT][2], dp[KT][2];
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; sc[t][q2] = z; dp[t][q2] = z; }
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int s = 0; s < G::NQK; ++s) {
                    const int qt = 2 * hf + q2;
                    hcp_bf16x8 qa = *(const hcp_bf16x8*)(sQ + (qt * 16 + fr) * G::RS + s * 32 + fg * 8);
                    hcp_bf16x8 ga = *(const hcp_bf16x8*)(sG + (qt * 16 + fr) * G::RS + s * 32 + fg * 8);
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        sc[t][q2] = hcp_mfma16(qa, kf[t][s], sc[t][q2]);   // S[q = qt*16 + 4fg + r][key = fr]
                        dp[t][q2] = hcp_mfma16(ga, vf[t][s], dp[t][q2]);
                    }
                }
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const bool kok = k_base + t * 16 + fr < p.Nk;
                const float kb2 = (KB && kok && p.kbias) ? p.kbias[(size_t)b * p.kb_bs + k_base + t * 16 + fr] * LOG2E : 0.f;   // this lane's key
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int qt = 2 * hf + q2;
                    const hcp_f32x4 l4 = *(const hcp_f32x4*)(sL + qt * 16 + 4 * fg);
                    const hcp_f32x4 d4 = *(const hcp_f32x4*)(sL + KVT + qt * 16 + 4 * fg);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pr = kok ? hcp_exp2(fmaf(sc[t][q2][r], c2, kb2 - l4[r])) : 0.f;  // lse2 = +inf for q >= Nq -> 0
                        if (KB && p.causal && k_base + t * 16 + fr > q0 + qt * 16 + 4 * fg + r) pr = 0.f;   // future key
                        sc[t][q2][r] = pr;
                        dp[t][q2][r] = pr * (dp[t][q2][r] - d4[r]);          // softmax scale applied once, at the store
                    }
                }
                pf[t][hf] = pack8(sc[t][0], sc[t][1]);
                df[t][hf] = pack8(dp[t][0], dp[t][1]);
            }
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                hcp_bf16x8 qa = tr_frag(sQ, G::RS, d * 16, s2, fr, fg);
                hcp_bf16x8 ga = tr_frag(sG, G::RS, d * 16, s2, fr, fg);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    dv[t][d] = hcp_mfma16(ga, pf[t][s2], dv[t][d]);   // dV^T[dcol][key] += dO^T P
                    dk[t][d] = hcp_mfma16(qa, df[t][s2], dk[t][d]);   // dK^T[dcol][key] += Q^T dS
                }
            }
        if (NB == 2) {
            if (it + 1 < nt) store_all(lds + ((it + 1 - it0) & 1) * BUF);
            if (!(p.dbg & 2)) HCP_SYNC();
        } else {
            HCP_SYNC();
            if (it + 1 < nt) store_all(lds);
            HCP_SYNC();
        }
    }
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int row = k_base + t * 16 + fr;
        if (row >= p.Nk) continue;
        if (p.qsplit > 1) {
            float* k32 = p.dk32 + ((size_t)b * p.Nk + row) * (p.H * D) + h * D;
            float* v32 = p.dv32 + ((size_t)b * p.Nk + row) * (p.H * D) + h * D;
#pragma unroll
            for (int d = 0; d < G::NDV; ++d) {
                const int col = d * 16 + 4 * fg;
                if (col < D) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { hcp_atomic_add(k32 + col + r, dk[t][d][r] * p.scale); hcp_atomic_add(v32 + col + r, dv[t][d][r]); }
                }
            }
            continue;
        }
        hcp_bf16* krow = p.dK + (size_t)b * p.k_bs + (size_t)row * p.k_rs + h * D;
        hcp_bf16* vrow = p.dV + (size_t)b * p.v_bs + (size_t)row * p.v_rs + h * D;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) {
            const int col = d * 16 + 4 * fg;
            if (col < D) {
                hcp_bf16x4 wk, wv;
#pragma unroll
                for (int r = 0; r < 4; ++r) { wk[r] = (short)hcp_f2bf(dk[t][d][r] * p.scale); wv[r] = (short)hcp_f2bf(dv[t][d][r]); }
                *(hcp_bf16x4*)(krow + col) = wk;
                *(hcp_bf16x4*)(vrow + col) = wv;
            }
        }
    }
}

// fp32 accumulators of the query-split dK/dV pass -> bf16 outputs (token-major, strided)
HCP_KERNEL(256) attn_dkv_convert_kernel(AttnParams p, int B, int C) {
    const int cv = C / 4;
    const long total = (long)B * p.Nk * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 4; long r = i / cv; const int n = (int)(r % p.Nk); const int b = (int)(r / p.Nk);
        const hcp_f32x4 k4 = *(const hcp_f32x4*)(p.dk32 + ((size_t)b * p.Nk + n) * C + c);
        const hcp_f32x4 v4 = *(const hcp_f32x4*)(p.dv32 + ((size_t)b * p.Nk + n) * C + c);
        hcp_bf16x4 wk, wv;
#pragma unroll
        for (int q = 0; q < 4; ++q) { wk[q] = (short)hcp_f2bf(k4[q]); wv[q] = (short)hcp_f2bf(v4[q]); }
        *(hcp_bf16x4*)(p.dK + (size_t)b * p.k_bs + (size_t)n * p.k_rs + c) = wk;
        *(hcp_bf16x4*)(p.dV + (size_t)b * p.v_bs + (size_t)n * p.v_rs + c) = wv;
    }
}

int g_attn_dbg = 0;    // tools only, see AttnParams::dbg
int g_attn_cfg = -1;   // tools: bit0 fwd rows/wave 32 (else 16), bit1 dQ 32, bit2 dK/dV 32; -1 = heuristic

template <int D, int QT>
int launch_fwd(AttnParams& p, int B, hipStream_t stream) {
    using G = AttnGeom<D>;
    size_t smem = (size_t)2 * (2 * G::RM_ELEMS) * sizeof(hcp_bf16);
    if (p.kbias || p.causal) HCP_LAUNCH((attn_fwd_kernel<D, QT, true>), dim3(hcp_cdiv(p.Nq, 64 * QT), p.H, B), dim3(256), smem, stream, p);
    else HCP_LAUNCH((attn_fwd_kernel<D, QT>), dim3(hcp_cdiv(p.Nq, 64 * QT), p.H, B), dim3(256), smem, stream, p);
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
    size_t s1 = (size_t)2 * (2 * G::RM_ELEMS) * sizeof(hcp_bf16);
    if (p.kbias || p.causal) HCP_LAUNCH((attn_bwd_dq_kernel<D, QT, true>), dim3(hcp_cdiv(p.Nq, 64 * QT), p.H, B), dim3(256), s1, stream, p);
    else HCP_LAUNCH((attn_bwd_dq_kernel<D, QT>), dim3(hcp_cdiv(p.Nq, 64 * QT), p.H, B), dim3(256), s1, stream, p);
    HCP_LAUNCH_CHECK("attn_bwd_dq");
}
template <int D, int KT>
int launch_dkv(AttnParams& p, int B, float* ws, size_t ws_bytes, hipStream_t stream) {
    using G = AttnGeom<D>;
    size_t s2 = (size_t)(2 * G::RM_ELEMS + 4 * KVT) * sizeof(hcp_bf16);
    if (2 * s2 <= 160 * 1024) s2 *= 2;
    const int nkv = hcp_cdiv(p.Nk, 64 * KT);
    // few key tiles (cross-attention: 77 keys): split the query loop so the grid still fills the chip
    const int nqt = hcp_cdiv(p.Nq, KVT);
    int qsplit = 1;
    const long base = (long)nkv * p.H * B;
    const size_t need = (size_t)2 * B * p.Nk * p.H * D * sizeof(float);
    if (base < 256 && nqt >= 8 && ws && ws_bytes >= need) {
        qsplit = (int)((512 + base - 1) / base);
        if (qsplit > nqt / 8) qsplit = nqt / 8;       // >= 8 query tiles per workgroup, else memset + convert dominate
        if (qsplit < 2) qsplit = 1;
    }
    p.qsplit = qsplit;
    if (qsplit > 1) {
        p.dk32 = ws; p.dv32 = ws + (size_t)B * p.Nk * p.H * D;
        if (hcp_memset_async(ws, 0, need, stream)) return hcp_set_error("attention_bwd: memset failed");
    }
    if (p.kbias || p.causal) HCP_LAUNCH((attn_bwd_dkv_kernel<D, KT, true>), dim3(nkv * qsplit, p.H, B), dim3(256), s2, stream, p);
    else HCP_LAUNCH((attn_bwd_dkv_kernel<D, KT>), dim3(nkv * qsplit, p.H, B), dim3(256), s2, stream, p);
    if (qsplit > 1) {
        long tot = (long)B * p.Nk * (p.H * D / 4);
        int g = (int)((tot + 255) / 256); if (g > 2048) g = 2048;
        HCP_LAUNCH(attn_dkv_convert_kernel, dim3(g), dim3(256), 0, stream, p, B, p.H * D);
    }
    HCP_LAUNCH_CHECK("attn_bwd_dkv");
}

// rows per wave: 32 only where the grid still fills the chip twice over AND registers allow >= 2 waves/SIMD
template <int D> constexpr bool kWide = D <= 64;

template <int D>
int run_fwd(AttnParams& p, int B, hipStream_t stream) {
    bool wide = kWide<D> && (long)B * p.H * hcp_cdiv(p.Nq, 128) >= 512;
    if (g_attn_cfg >= 0) wide = kWide<D> && (g_attn_cfg & 1);
    if constexpr (kWide<D>) { if (wide) return launch_fwd<D, 2>(p, B, stream); }
    return launch_fwd<D, 1>(p, B, stream);
}
template <int D>
int run_bwd(AttnParams& p, int B, float* ws, size_t ws_bytes, hipStream_t stream) {
    if (int e = launch_delta<D>(p, B, stream)) return e;
    // measured on MI355X: 32 rows per wave pay off once the grid has >= 512 such workgroups
    bool wq = kWide<D> && (long)B * p.H * hcp_cdiv(p.Nq, 128) >= 512, wk = kWide<D> && (long)B * p.H * hcp_cdiv(p.Nk, 128) >= 512;
    if (g_attn_cfg >= 0) { wq = kWide<D> && (g_attn_cfg & 2); wk = kWide<D> && (g_attn_cfg & 4); }
    int e;
    if constexpr (kWide<D>) { e = wq ? launch_dq<D, 2>(p, B, stream) : launch_dq<D, 1>(p, B, stream); }
    else e = launch_dq<D, 1>(p, B, stream);
    if (e) return e;
    if constexpr (kWide<D>) { return wk ? launch_dkv<D, 2>(p, B, ws, ws_bytes, stream) : launch_dkv<D, 1>(p, B, ws, ws_bytes, stream); }
    return launch_dkv<D, 1>(p, B, ws, ws_bytes, stream);
}

int attn_check(const AttnParams& p, int B, int D) {
    HCP_REQUIRE(B > 0 && p.H > 0 && p.Nq > 0 && p.Nk > 0, "attention: empty problem");
    HCP_REQUIRE(D == 40 || D == 64 || D == 80 || D == 160, "attention: head_dim %d unsupported (40/64/80/160)", D);
    HCP_REQUIRE(p.q_rs % 8 == 0 && p.k_rs % 8 == 0 && p.v_rs % 8 == 0 && p.o_rs % 8 == 0, "attention: row strides must be multiples of 8");
    HCP_REQUIRE(p.q_bs % 8 == 0 && p.k_bs % 8 == 0 && p.v_bs % 8 == 0 && p.o_bs % 8 == 0, "attention: batch strides must be multiples of 8");
    return 0;
}

}  // namespace

// TOOLS ONLY: bit0 forward / bit1 dQ / bit2 dK,dV use 32 rows per wave; -1 restores the heuristic.
HCP_API int hcp_debug_set_attention_config(int cfg) { g_attn_cfg = cfg; return 0; }
// TOOLS ONLY (results are wrong when != 0): 1 = skip the global loads inside the tile loops (latency ablation).
HCP_API int hcp_debug_set_attention_ablation(int flags) { g_attn_dbg = flags; return 0; }

// O[b,q,h,:] = softmax_k(scale * Q[b,q,h,:].K[b,k,h,:]) V[b,k,h,:];  lse[b,h,q] = logsumexp of the scaled scores.
// All tensors bf16, token-major: element (b, n, h, c) at  base + b*bs + n*rs + h*D + c.
HCP_API int hcp_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H, int Nq, int Nk,
                              int D, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                              float scale, const float* key_bias, long key_bias_bs, int causal, hipStream_t stream) {
    AttnParams p = {};
    p.kbias = key_bias; p.kb_bs = key_bias_bs; p.causal = causal ? 1 : 0; p.dbg = g_attn_dbg;
    p.Q = (const hcp_bf16*)Q; p.K = (const hcp_bf16*)K; p.V = (const hcp_bf16*)V; p.Out = (hcp_bf16*)O; p.lse = lse;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.qsplit = 1;
    HCP_REQUIRE(Q && K && V && O && lse, "hcp_attention_fwd: null pointer");
    HCP_REQUIRE(!causal || Nq == Nk, "hcp_attention_fwd: causal masking is defined for self-attention (Nq == Nk)");
    if (int e = attn_check(p, B, D)) return e;
    switch (D) {
        case 40: return run_fwd<40>(p, B, stream);
        case 64: return run_fwd<64>(p, B, stream);
        case 80: return run_fwd<80>(p, B, stream);
        default: return run_fwd<160>(p, B, stream);
    }
}

// Gradients of hcp_attention_fwd.  delta is a [B,H,Nq] fp32 scratch buffer; workspace (optional, 2*B*Nk*H*D floats)
// lets short-key problems (cross-attention) split the query loop of the dK/dV pass across workgroups.
HCP_API int hcp_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                              float* delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int D, long q_bs,
                              int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                              const float* key_bias, long key_bias_bs, int causal, void* workspace, size_t workspace_bytes,
                              hipStream_t stream) {
    AttnParams p = {};
    p.kbias = key_bias; p.kb_bs = key_bias_bs; p.causal = causal ? 1 : 0; p.dbg = g_attn_dbg;
    p.Q = (const hcp_bf16*)Q; p.K = (const hcp_bf16*)K; p.V = (const hcp_bf16*)V; p.O = (const hcp_bf16*)O;
    p.dO = (const hcp_bf16*)dO; p.lse = (float*)lse; p.delta = delta;
    p.dQ = (hcp_bf16*)dQ; p.dK = (hcp_bf16*)dK; p.dV = (hcp_bf16*)dV;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.qsplit = 1;
    HCP_REQUIRE(Q && K && V && O && dO && lse && delta && dQ && dK && dV, "hcp_attention_bwd: null pointer");
    HCP_REQUIRE(!causal || Nq == Nk, "hcp_attention_bwd: causal masking is defined for self-attention (Nq == Nk)");
    if (int e = attn_check(p, B, D)) return e;
    float* ws = (float*)workspace; const size_t wb = workspace ? workspace_bytes : 0;
    switch (D) {
        case 40: return run_bwd<40>(p, B, ws, wb, stream);
        case 64: return run_bwd<64>(p, B, ws, wb, stream);
        case 80: return run_bwd<80>(p, B, ws, wb, stream);
        default: return run_bwd<160>(p, B, ws, wb, stream);
    }
}
