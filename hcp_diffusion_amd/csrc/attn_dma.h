// attn_dma.h — second-generation attention kernels for gfx950: K/V (or Q/dO) tiles travel global -> LDS by DMA
// (buffer_load_dwordx4 ... lds), LDS rows are UNPADDED to the MFMA reduction length, softmax costs one VALU op per score.
//
// Same math and tensor contract as the first-generation kernels in attention.hip (diffusers CrossAttention core:
// softmax(Q K^T / sqrt(d)) V — reference call sites train_ac.py:258-260, unet_struct.txt:17-43); what changed is where
// the cycles went in the round-1 profile (profiles/r1_pmc_attention_conv_final.md: MFMA busy 0.29, 5.7x K/V over-fetch):
//
//  * XCD-aware work order: the launch is one-dimensional and workgroup `id` runs work item (id % 8) * (n / 8) + id / 8,
//    so all query tiles of one (batch, head) run on ONE XCD and its K/V (655 KB at N=4096, d=40) stays in that XCD's
//    4 MB L2 instead of being re-fetched by all eight (guide §5.5 T1; placement only changes speed, never results).
//  * LDS image: row stride = D*2 bytes + one or more 16-byte pad granules, chosen bank-conflict free for both the
//    ds_read_b128 K fragments and the ds_read_b64_tr_b16 V fragments (tools/attn_lab/bank_model.py): 96 B rows at d=40
//    instead of 144 B.  Lanes whose k-slots lie beyond the head dim read the row's ZERO pad granule (same address for
//    the three lane groups: a broadcast) — no zero-padded columns are stored or moved.
//  * The tile fill is 2*RG LDS-DMA instructions per 64-row tile for the whole workgroup (RG = granules per row), issued
//    one tile ahead into the other buffer: no staging VGPRs, no ds_write traffic on the LDS port that the fragment
//    reads need (round 1: LDS-array cycles per tile ~ MFMA cycles per tile).  Pad granules are never written by the
//    DMA (those lanes are masked off), so the V image's pad keeps the 1.0 column that makes the PV MFMA produce the
//    softmax row sums (d=40: output rows 40..47 of O^T) and the K image's pad keeps its zeros.
//  * Softmax: Q is pre-multiplied by scale*log2(e) once (registers), the score MFMA chain starts from the accumulator
//    {-m,-m,-m,-m} (m = the row's running reference max, one VGPR quad per 16 query rows, rebuilt only on a rescale), so
//    P = exp2(acc) is ONE v_exp_f32 per score; the rescale test is a wave vote on the lane-local maximum — the
//    cross-lane row maximum is only formed on the (rare) rescale path.
#pragma once
#include "hcp_common.h"

namespace hcp_attn {

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
    // scores of every head and query; reference models/wrapper.py:22-23,29)
    const float* kbias; long kb_bs;
    int causal;                     // 1: key k is visible to query q only if k <= q (CLIP text encoder); KB instantiations only
    int pre;                        // 1: Q holds Q * scale*log2(e) (its projection's weights carry the factor); `scale` then only names the factor
    int B;
    // dK/dV kernel: the query loop may be split over `qsplit` workgroups per key block.  Round 6: every split STORES its fp32 partial into
    // its own slab — dk32 / dv32 [qsplit][B, Nk, H*D], each element written exactly once — and the convert kernel behind adds the slabs
    // in split order: no atomics, nothing to clear, the same bits on every run.  (Rounds 2-5 added the partials with fp32 atomics into
    // one accumulator: cleared by a hipMemsetAsync that, as a hipGraph memset NODE, intermittently let the adding kernel see stale
    // workspace contents — absurd to_k / to_v LoRA gradients on the 64x64 cross-attention layers, tools/diag/nan_hunt.py — then by the dQ
    // kernel in front.)
    int qsplit;
    float* dk32; float* dv32;
};

constexpr int KVT = 64;            // keys (or queries, in the dK/dV kernel) per tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// Work item of workgroup `id` out of n (bijective for every n; guide §5 "XCD swizzle must be bijective").
HCP_DEVICE int xcd_work_item(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

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
HCP_DEVICE hcp_bf16x8 scale8(hcp_bf16x8 v, float c) {
    hcp_bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = (short)hcp_f2bf(hcp_bf2f((unsigned short)v[i]) * c);
    return r;
}

template <int D> struct Geom {
    static_assert(D == 40 || D == 64 || D == 80 || D == 160, "head_dim");
    static constexpr int RG = D == 40 ? 6 : D == 64 ? 10 : D == 80 ? 14 : 22;   // row stride in 16-byte granules
    static constexpr int RSB = RG * 16;               // row stride, bytes
    static constexpr int RS = RG * 8;                 // row stride, bf16 elements
    static constexpr int REALG = D / 8;               // granules of a row that carry data
    static constexpr int NQK = (D + 31) / 32;         // k-steps of a score MFMA chain
    static constexpr int NFULL = D / 32;              // ... of which complete
    static constexpr int NDV = (D + 15) / 16;         // 16-row output tiles of the O^T-shaped products
    static constexpr bool SPARE = NDV * 16 > D;       // spare output rows exist (d = 40): row sums ride on the MFMA
    static constexpr int IMG = KVT * RS;              // elements of one [64][RS] image
};

// One wave's share of the DMA fill of a {image0 | image1} tile pair (K|V, or Q|dO): chunk c = wave + 4*i covers the 64
// consecutive granules [64c', 64c'+64) of image c / RG.  Offsets are loop invariant; the descriptors are re-based per tile
// and sized to the tile's valid rows, so ragged tails read zeros in hardware.  Pad granules are masked off (never written).
template <int D, int NW = 4>
struct TileDma {
    using G = Geom<D>;
    static constexpr int NI = (2 * G::RG + NW - 1) / NW;      // instructions per wave per tile pair (the last may fall past the pair)
    unsigned voff[NI];          // per-lane source offset (bytes) or HCP_BUF_OOB for pad granules
    HCP_MEMBER void init(int wave, int lane, int rs0, int rs1) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = wave + NW * i;
            const int img = c >= G::RG ? 1 : 0;
            const int g = (c - img * G::RG) * 64 + lane;
            const int row = g / G::RG, cg = g - row * G::RG;
            voff[i] = (c < 2 * G::RG && cg < G::REALG) ? (unsigned)((row * (img ? rs1 : rs0) + cg * 8) * 2) : HCP_BUF_OOB;
        }
    }
    // src0/src1: first row of the tile in each tensor (column h*D applied); nvalid rows are live.
    template <bool MASKPAD = true>
    HCP_MEMBER void issue(const hcp_bf16* src0, int rs0, const hcp_bf16* src1, int rs1, int nvalid, hcp_bf16* dst, int wave) const {
        const unsigned n0 = nvalid > 0 ? (unsigned)(((nvalid - 1) * rs0 + D) * 2) : 0u;
        const unsigned n1 = nvalid > 0 ? (unsigned)(((nvalid - 1) * rs1 + D) * 2) : 0u;
        const hcp_desc4 d0 = hcp_make_desc(src0, n0), d1 = hcp_make_desc(src1, n1);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int c = wave + NW * i;                      // wave-uniform
            if (c >= 2 * G::RG) break;
            hcp_bf16* d = dst + c * 512;                      // chunk c of the pair = 1 KB = 512 elements (image 1 follows image 0)
            if (!MASKPAD || voff[i] != HCP_BUF_OOB) hcp_dma16(c >= G::RG ? d1 : d0, voff[i], d);   // !MASKPAD: pad lanes write zeros
        }
    }
};

// VAR bits (product builds use VAR_PRODUCT; tools/attn_lab instantiates others for A/B and ablation):
constexpr int VAR_XCD = 1;        // XCD-aware work order
constexpr int VAR_ONES = 2;       // d = 40: V pad granule = 1.0 -> softmax row sums come out of the PV MFMA; else VALU sums
constexpr int VAR_NOEXP = 4;      // ablation (wrong results): skip v_exp
constexpr int VAR_NODMA = 8;      // ablation (wrong results): no DMA inside the loop
constexpr int VAR_NOBAR = 16;     // ablation (wrong results): no barrier inside the loop
constexpr int VAR_RAW = 64;       // scores stay in raw q.k units: exp2(acc * scale*log2e) = 2 VALU ops per score, no bf16 rounding of Q*scale
constexpr int VAR_PRIO = 256;     // s_setprio(1) around the MFMA clusters (guide T5)
constexpr int VAR_DEEP = 128;     // forward: three LDS buffers, DMA two tiles ahead (counted vmcnt)
constexpr int VAR_PADZERO = 32;   // lab fallback: pad lanes stay in the DMA and write zeros (needs !VAR_ONES)
constexpr int VAR_PRE = 512;      // forward: Q arrives pre-multiplied by scale*log2(e) (its projection carries the factor): the accumulator IS the exp2 argument
constexpr int VAR_LSUM = 2048;    // forward: lazy rescale from the ROW SUMS instead of a per-score maximum (exact fallback on exp2 overflow)
constexpr int VAR_PRODUCT = VAR_XCD | VAR_ONES | VAR_RAW;


template <int D, int QT> constexpr int fwd_waves() { return D > 80 ? 2 : (QT == 2 ? (D == 40 ? 4 : 3) : 4); }   // per SIMD
template <bool V> struct BoolC { static constexpr bool value = V; };
template <int V> struct IntC { static constexpr int value = V; };

// (the per-lane maximum of 16 fresh MFMA results is hcp_max16, hcp_device.h: one asm statement, see there)

// ------------------------------------------------------------------------------------------ forward
template <int D, int QT, bool KB, int VAR, int NW = 4>
HCP_WAVES_PER_SIMD((fwd_waves<D, QT>())) HCP_KERNEL(64 * NW) attn2_fwd_kernel(AttnParams p) {
    using G = Geom<D>;
    constexpr bool MASKPAD = !(VAR & VAR_PADZERO);
    constexpr bool RAW = (VAR & VAR_RAW) != 0;
    constexpr int AHEAD = (VAR & VAR_DEEP) ? 2 : 1;   // tiles in flight ahead of the one being consumed
    constexpr int NBUF = AHEAD + 1;
    constexpr bool ONES = G::SPARE && (VAR & VAR_ONES) && MASKPAD;   // row sums from the PV MFMA
    constexpr bool PRE = (VAR & VAR_PRE) != 0;        // (implies the exp2-domain accumulator: no multiply in front of v_exp_f32)
    // Only with ONES (d = 40): there l is the sum of the bf16-ROUNDED probabilities the PV MFMA used, so a row dominated by one key
    // normalises exactly whatever its reference is.  With a VALU row sum of the unrounded exponentials a dominant P that is not
    // exactly 1 (any reference but the true maximum) leaves a 2^-9 relative error in O, which the backward's delta = rowsum(dO * O)
    // turns into a spurious gradient on exactly the saturated rows (measured: dQ error 0.98 % -> 2.4 % at d = 160).
    constexpr bool LSUM = (VAR & VAR_LSUM) != 0 && ONES;
    // LSUM: the reference maximum is the exact row maximum of the FIRST tile; afterwards no per-score maximum is taken (16 v_max3 +
    // hazard nops per 32 x 64 wave-tile).  A tile only looks at the running sum of probabilities — the PV MFMA keeps it in O's spare
    // row at d = 40, a VALU partial sum otherwise — and when a row's sum passes 2^20, O and l are divided by it and the reference moves
    // by its log2.  Magnitude costs no precision (P is bf16 with its own exponent, O / l are fp32); only exp2 overflow (a score more
    // than 2^127 above the reference) would be fatal: the epilogue checks the row sums and the workgroup then repeats its rows with
    // per-score maxima (the EXACT tile code below).
    constexpr float L_LAZY = 1048576.0f;
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;                  // NBUF x { K [64][RS] | V [64][RS] } (+ one flag word, LSUM)
    constexpr int BUF = 2 * G::IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = hcp_uniform(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int ROWS = 16 * QT * NW;               // query rows per workgroup
    const int nqt = (p.Nq + ROWS - 1) / ROWS;
    int item = blockIdx.x;
    if (VAR & VAR_XCD) item = xcd_work_item(item, gridDim.x);
    const int qtile = item % nqt, bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q_base = qtile * ROWS + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const float c2 = p.scale * LOG2E;
    const float cs = PRE ? 1.0f : RAW ? c2 : 1.0f;    // what one unit of the accumulator is worth in the exp2 domain
    const float rescale_thr = 6.0f / cs;              // lazy rescale: a score may exceed the reference max by 2^6 before O is rescaled

    // LDS init: zeros (the K pad granules stay zero for the kernel's life), then 1.0 in the V pad granules
    for (int i = tid * 8; i < NBUF * BUF + (LSUM ? 8 : 0); i += 64 * NW * 8) *(hcp_bf16x8*)(lds + i) = hcp_zero8();
    TileDma<D, NW> dma;
    dma.init(wave, lane, p.k_rs, p.v_rs);
    const int nt = (p.Nk + KVT - 1) / KVT;

    hcp_bf16x8 qf[QT][G::NQK];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) {
            const int row = q_base + t * 16 + fr, dc = s * 32 + fg * 8;
            qf[t][s] = (row < p.Nq && dc < D) ? *(const hcp_bf16x8*)(Qb + (size_t)row * p.q_rs + dc) : hcp_zero8();
        }
    HCP_SYNC();                                       // zero fill complete
    if (ONES && tid < NBUF * KVT) {
        hcp_bf16x8 one8;
#pragma unroll
        for (int e = 0; e < 8; ++e) one8[e] = 0x3F80;
        *(hcp_bf16x8*)(lds + (tid >> 6) * BUF + G::IMG + (tid & 63) * G::RS + D) = one8;
    }
    auto rows_of = [&](int t) { const int n = p.Nk - t * KVT; return n < KVT ? n : KVT; };
    auto fetch = [&](int t, int buf) {
        dma.template issue<MASKPAD>(Kb + (size_t)t * KVT * p.k_rs, p.k_rs, Vb + (size_t)t * KVT * p.v_rs, p.v_rs, rows_of(t), lds + buf * BUF, wave);
    };
    fetch(0, 0);
    if (AHEAD == 2 && nt > 1) fetch(1, 1);
    const int my_dmas = (2 * G::RG - wave + NW - 1) / NW;           // DMA instructions this wave issues per tile
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) if (!RAW && !PRE) qf[t][s] = scale8(qf[t][s], c2);     // scores come out in the exp2 domain

    float m_i[QT], l_i[QT];
    hcp_f32x4 o[QT][G::NDV], nm4[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_i[t] = 0.f; l_i[t] = 0.f;
        hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        nm4[t] = z;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) o[t][d] = z;
    }
    // per-lane fragment addresses (elements), loop invariant
    const int kfull = fr * G::RS + fg * 8;                                                    // + kt*16*RS + s*32
    const int ktail = fr * G::RS + ((G::NFULL * 32 + fg * 8) < D ? G::NFULL * 32 + fg * 8 : D);   // beyond the head dim: the zero pad granule
    const int vfrag = (4 * fg + (fr >> 2)) * G::RS + 4 * (fr & 3);                            // + (2*s2+j)*16*RS + dt*16
    hcp_dma_wait_all();
    HCP_SYNC();                                       // first tile landed, pads initialised

    int cur = 0;                                      // LDS buffer of the tile being consumed
    // One 64-key tile.  FIRST: the tile that sets the reference maximum (always rescales); RAGGED: fewer than 64 live keys
    // (the dead ones enter the score MFMA chain as -inf through the accumulator).
    auto tile = [&](auto first_c, auto ragged_c, auto exact_c, int it) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_c)::value, RAGGED = decltype(ragged_c)::value, EXACT = decltype(exact_c)::value || !LSUM;
        const int kv0 = it * KVT;
        const int nvalid = p.Nk - kv0 < KVT ? p.Nk - kv0 : KVT;
        const hcp_bf16* sK = lds + cur * BUF;
        const hcp_bf16* sV = sK + G::IMG;
        const bool prefetch = it + AHEAD < nt && !(VAR & VAR_NODMA);
        if (prefetch) fetch(it + AHEAD, cur + AHEAD >= NBUF ? cur + AHEAD - NBUF : cur + AHEAD);
        hcp_f32x4 sc[QT][4];
        if (VAR & VAR_PRIO) hcp_setprio<1>();
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int s = 0; s < G::NQK; ++s) {
                const hcp_bf16x8 kf = *(const hcp_bf16x8*)(sK + (s < G::NFULL ? kfull + s * 32 : ktail) + kt * 16 * G::RS);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    hcp_f32x4 init = nm4[t];
                    if (RAGGED && s == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (kt * 16 + 4 * fg + r >= nvalid) init[r] = -INFINITY;
                    }
                    sc[t][kt] = hcp_mfma16(kf, qf[t][s], s == 0 ? init : sc[t][kt]);
                }
            }
        if (VAR & VAR_PRIO) hcp_setprio<0>();
        if (KB) {                                     // additive key bias / causal mask (wave-uniform code path)
            const float* kb = p.kbias ? p.kbias + (size_t)b * p.kb_bs + kv0 : nullptr;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = kt * 16 + 4 * fg + r;
                    const float bv = (kb && kk < nvalid) ? kb[kk] * (RAW ? 1.0f / p.scale : LOG2E) : 0.f;
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        sc[t][kt][r] += bv;
                        if (p.causal && kv0 + kk > q_base + t * 16 + fr) sc[t][kt][r] = -INFINITY;     // future key: p = 0
                    }
                }
        }
        hcp_bf16x8 pf[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            if (!EXACT && !FIRST) {                                      // LSUM: one compare per 16 rows on the running row sum
                const float lane_sum = ONES ? o[t][D / 16][D % 16 % 4] : l_i[t];
                const bool ok = (ONES && fg != (D % 16) / 4) || lane_sum <= L_LAZY;
                if (!hcp_all(ok)) {
                    float rm;
                    if (ONES) rm = hcp_shfl(lane_sum, ((D % 16) / 4) * 16 + fr);
                    else { rm = lane_sum; rm += hcp_shfl_xor(rm, 16); rm += hcp_shfl_xor(rm, 32); }
                    const bool up = rm > 1.0f;
                    const float delta = up ? log2f(rm) / cs : 0.f, alpha = up ? 1.0f / rm : 1.0f;
                    m_i[t] += delta;
                    const hcp_f32x4 n4 = {-m_i[t], -m_i[t], -m_i[t], -m_i[t]};
                    nm4[t] = n4;
                    l_i[t] *= alpha;
#pragma unroll
                    for (int d = 0; d < G::NDV; ++d) o[t][d] *= alpha;
#pragma unroll
                    for (int kt = 0; kt < 4; ++kt) sc[t][kt] -= delta;
                }
            }
            const float mx = (EXACT || FIRST) ? hcp_max16(sc[t]) : 0.f;
            // Lazy rescale: the scores are already relative to the reference max m_i; while no lane of the wave sees one above
            // 2^6 nothing is rescaled.
            if (FIRST || (EXACT && !hcp_all(mx <= rescale_thr))) {
                float rm = fmaxf(mx, hcp_shfl_xor(mx, 16));
                rm = fmaxf(rm, hcp_shfl_xor(rm, 32));                   // row maximum of this tile, relative to m_i
                float delta = FIRST ? rm : fmaxf(rm, 0.f);
                delta = delta > -1e30f ? delta : 0.f;                    // a fully masked row keeps its reference
                m_i[t] += delta;
                const hcp_f32x4 n4 = {-m_i[t], -m_i[t], -m_i[t], -m_i[t]};
                nm4[t] = n4;
                if (!FIRST) {                                            // nothing accumulated yet on the first tile
                    const float alpha = hcp_exp2(-delta * cs);
                    l_i[t] *= alpha;
#pragma unroll
                    for (int d = 0; d < G::NDV; ++d) o[t][d] *= alpha;
                }
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) sc[t][kt] -= delta;
            }
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = (VAR & VAR_NOEXP) ? sc[t][kt][r] : hcp_exp2(PRE ? sc[t][kt][r] : sc[t][kt][r] * cs);
                    sc[t][kt][r] = e;
                    if (!ONES) rs += e;
                }
            if (!ONES) l_i[t] += rs;                                     // lane-local partial; lanes are combined in the epilogue
            pf[t][0] = pack8(sc[t][0], sc[t][1]);
            pf[t][1] = pack8(sc[t][2], sc[t][3]);
        }
        if (VAR & VAR_PRIO) hcp_setprio<1>();
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const hcp_bf16* a = sV + vfrag + (2 * s2) * 16 * G::RS + d * 16;
                const hcp_bf16x8 vf = join8(hcp_lds_read_tr4(a), hcp_lds_read_tr4(a + 16 * G::RS));
#pragma unroll
                for (int t = 0; t < QT; ++t) o[t][d] = hcp_mfma16(vf, pf[t][s2], o[t][d]);
            }
        if (VAR & VAR_PRIO) hcp_setprio<0>();
        // the next tile has landed (with two tiles ahead, the one fetched in this call stays in flight); this buffer is free again
        if (AHEAD == 2 && prefetch) hcp_wait_vmcnt(my_dmas); else hcp_dma_wait_all();
        if (!(VAR & VAR_NOBAR)) HCP_SYNC();
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    };
    const bool ragged = (p.Nk & (KVT - 1)) != 0;
    auto all_tiles = [&](auto exact_c) __attribute__((always_inline)) {   // (a call boundary would put the DMA descriptors into VGPRs)
        if (nt == 1) {
            if (ragged) tile(BoolC<true>{}, BoolC<true>{}, exact_c, 0); else tile(BoolC<true>{}, BoolC<false>{}, exact_c, 0);
        } else {
            tile(BoolC<true>{}, BoolC<false>{}, exact_c, 0);
            for (int it = 1; it < nt - 1; ++it) tile(BoolC<false>{}, BoolC<false>{}, exact_c, it);
            if (ragged) tile(BoolC<false>{}, BoolC<true>{}, exact_c, nt - 1); else tile(BoolC<false>{}, BoolC<false>{}, exact_c, nt - 1);
        }
    };
    auto row_sum = [&](int t) {
        float lsum;
        if (ONES) lsum = hcp_shfl(o[t][D / 16][D % 16 % 4], ((D % 16) / 4) * 16 + fr);   // O^T row D lives in lane group (D%16)/4
        else { lsum = l_i[t]; lsum += hcp_shfl_xor(lsum, 16); lsum += hcp_shfl_xor(lsum, 32); }
        return lsum;
    };
    all_tiles(BoolC<false>{});
    if (LSUM) {                                        // exp2 overflow check: any bad row sum sends the WORKGROUP through the exact pass
        int* redo_flag = (int*)(lds + NBUF * BUF);
        bool bad = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) { const float ls = row_sum(t); bad = bad || !(ls > 0.f && ls < 3.0e38f); }
        if (!hcp_all(!bad) && lane == 0) *redo_flag = 1;
        HCP_SYNC();
        if (hcp_uniform(*redo_flag) != 0) {                       // workgroup-uniform (SGPR: the DMA descriptors below need a scalar branch)
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                m_i[t] = 0.f; l_i[t] = 0.f;
                hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
                nm4[t] = z;
#pragma unroll
                for (int d = 0; d < G::NDV; ++d) o[t][d] = z;
            }
            cur = 0;
            fetch(0, 0);
            if (AHEAD == 2 && nt > 1) fetch(1, 1);
            hcp_dma_wait_all();
            HCP_SYNC();
            all_tiles(BoolC<true>{});
        }
    }
    // epilogue: lane holds O[q = q_base + t*16 + fr][d*16 + 4*fg + r]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q_base + t * 16 + fr;
        const float lsum = row_sum(t);
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
        if (fg == 0) p.lse[((size_t)b * p.H + h) * p.Nq + row] = (m_i[t] * cs + log2f(lsum)) * LN2;
    }
}

template <int D> constexpr size_t fwd_smem(int nbuf = 2) { return (size_t)nbuf * 2 * Geom<D>::IMG * sizeof(hcp_bf16) + 16; }

// ------------------------------------------------------------------------------------------ dQ
// Same walk as the forward: a workgroup owns 64*QT query rows and streams the K|V tiles.  Per score: P = exp2(acc) with the
// accumulator started at -lse (log2 domain), dP - delta with the accumulator started at -delta, dS = P * (dP - delta): one
// v_exp, one v_mul and half a convert.  dQ^T += K^T dS^T reads K with transpose reads from the same image the scores used.
template <int D, int QT> constexpr int dq_waves() { return D > 80 ? 2 : 3; }

template <int D, int QT, bool KB, int VAR>
HCP_WAVES_PER_SIMD((dq_waves<D, QT>())) HCP_KERNEL(256) attn2_bwd_dq_kernel(AttnParams p) {
    using G = Geom<D>;
    constexpr bool PRE = (VAR & VAR_PRE) != 0;        // Q arrives pre-multiplied by scale*log2(e): dQ is the gradient w.r.t. THAT tensor
    constexpr bool RAW = (VAR & VAR_RAW) != 0 && !PRE;
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;                  // 2 x { K [64][RS] | V [64][RS] }, pad granules zero
    constexpr int BUF = 2 * G::IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = hcp_uniform(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nqt = (p.Nq + 64 * QT - 1) / (64 * QT);
    int item = blockIdx.x;
    if (VAR & VAR_XCD) item = xcd_work_item(item, gridDim.x);
    const int qtile = item % nqt, bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q_base = qtile * (64 * QT) + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const hcp_bf16* dOb = p.dO + (size_t)b * p.o_bs + h * D;
    const float c2 = p.scale * LOG2E;
    const float cs = RAW ? c2 : 1.0f;

    for (int i = tid * 8; i < 2 * BUF; i += 256 * 8) *(hcp_bf16x8*)(lds + i) = hcp_zero8();
    TileDma<D> dma;
    dma.init(wave, lane, p.k_rs, p.v_rs);
    const int nt = (p.Nk + KVT - 1) / KVT;

    hcp_bf16x8 qf[QT][G::NQK], gf[QT][G::NQK];
    hcp_f32x4 nl4[QT], nd4[QT], dq[QT][G::NDV];
    const hcp_bf16* Ob = p.O + (size_t)b * p.o_bs + h * D;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q_base + t * 16 + fr;
        // delta = rowsum(dO * O) is computed HERE (it was a launch of its own in front of this kernel): the lane already holds its
        // 8-column pieces of the row's dO; the four lanes fr, fr+16, fr+32, fr+48 own the row between them.  Written out for the
        // dK/dV kernel that follows on the stream.
        float dpart = 0.f;
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) {
            const int dc = s * 32 + fg * 8;
            const bool ok = row < p.Nq && dc < D;
            qf[t][s] = ok ? *(const hcp_bf16x8*)(Qb + (size_t)row * p.q_rs + dc) : hcp_zero8();
            gf[t][s] = ok ? *(const hcp_bf16x8*)(dOb + (size_t)row * p.o_rs + dc) : hcp_zero8();
            const hcp_bf16x8 of = ok ? *(const hcp_bf16x8*)(Ob + (size_t)row * p.o_rs + dc) : hcp_zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) dpart += hcp_bf2f((unsigned short)of[e]) * hcp_bf2f((unsigned short)gf[t][s][e]);
        }
        dpart += hcp_shfl_xor(dpart, 16); dpart += hcp_shfl_xor(dpart, 32);
        if (fg == 0 && row < p.Nq) p.delta[((size_t)b * p.H + h) * p.Nq + row] = dpart;
        // rows past the end: P = exp2(0 - 0) = 1 (finite) against dP - delta = 0: dS = 0, and nothing of theirs is stored
        const float l2 = row < p.Nq ? p.lse[((size_t)b * p.H + h) * p.Nq + row] * LOG2E / cs : 0.f;
        const float dl = row < p.Nq ? dpart : 0.f;
        const hcp_f32x4 a = {-l2, -l2, -l2, -l2}, c = {-dl, -dl, -dl, -dl}, z = {0.f, 0.f, 0.f, 0.f};
        nl4[t] = a; nd4[t] = c;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) dq[t][d] = z;
    }
    HCP_SYNC();                                       // zero fill complete
    dma.issue(Kb, p.k_rs, Vb, p.v_rs, p.Nk < KVT ? p.Nk : KVT, lds, wave);
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) if (!RAW && !PRE) qf[t][s] = scale8(qf[t][s], c2);     // the forward's rounding of Q * scale*log2e
    const int kfull = fr * G::RS + fg * 8;
    const int ktail = fr * G::RS + ((G::NFULL * 32 + fg * 8) < D ? G::NFULL * 32 + fg * 8 : D);
    const int vfrag = (4 * fg + (fr >> 2)) * G::RS + 4 * (fr & 3);
    hcp_dma_wait_all();
    HCP_SYNC();

    auto tile = [&](auto ragged_c, int it) {
        constexpr bool RAGGED = decltype(ragged_c)::value;
        const int kv0 = it * KVT;
        const int nvalid = p.Nk - kv0 < KVT ? p.Nk - kv0 : KVT;
        const hcp_bf16* sK = lds + (it & 1) * BUF;
        const hcp_bf16* sV = sK + G::IMG;
        if (it + 1 < nt) {
            const int nv = p.Nk - kv0 - KVT < KVT ? p.Nk - kv0 - KVT : KVT;
            dma.issue(Kb + (size_t)(kv0 + KVT) * p.k_rs, p.k_rs, Vb + (size_t)(kv0 + KVT) * p.v_rs, p.v_rs, nv, lds + ((it + 1) & 1) * BUF, wave);
        }
        hcp_f32x4 sc[QT][4], dp[QT][4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int s = 0; s < G::NQK; ++s) {
                const int off = (s < G::NFULL ? kfull + s * 32 : ktail) + kt * 16 * G::RS;
                const hcp_bf16x8 kf = *(const hcp_bf16x8*)(sK + off);
                const hcp_bf16x8 vf = *(const hcp_bf16x8*)(sV + off);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    hcp_f32x4 init = nl4[t];
                    if (RAGGED && s == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (kt * 16 + 4 * fg + r >= nvalid) init[r] = -INFINITY;   // dead key: P = 0
                    }
                    sc[t][kt] = hcp_mfma16(kf, qf[t][s], s == 0 ? init : sc[t][kt]);
                    dp[t][kt] = hcp_mfma16(vf, gf[t][s], s == 0 ? nd4[t] : dp[t][kt]);
                }
            }
        if (KB) {
            const float* kb = p.kbias ? p.kbias + (size_t)b * p.kb_bs + kv0 : nullptr;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = kt * 16 + 4 * fg + r;
                    const float bv = (kb && kk < nvalid) ? kb[kk] * (RAW ? 1.0f / p.scale : LOG2E) : 0.f;
#pragma unroll
                    for (int t = 0; t < QT; ++t) {
                        sc[t][kt][r] += bv;
                        if (p.causal && kv0 + kk > q_base + t * 16 + fr) sc[t][kt][r] = -INFINITY;
                    }
                }
        }
        hcp_bf16x8 df[QT][2];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) sc[t][kt][r] = hcp_exp2(sc[t][kt][r] * cs) * dp[t][kt][r];   // dS; the softmax scale is applied once, at the store
            df[t][0] = pack8(sc[t][0], sc[t][1]);
            df[t][1] = pack8(sc[t][2], sc[t][3]);
        }
#pragma unroll
        for (int d = 0; d < G::NDV; ++d)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const hcp_bf16* a = sK + vfrag + (2 * s2) * 16 * G::RS + d * 16;
                const hcp_bf16x8 kf = join8(hcp_lds_read_tr4(a), hcp_lds_read_tr4(a + 16 * G::RS));
#pragma unroll
                for (int t = 0; t < QT; ++t) dq[t][d] = hcp_mfma16(kf, df[t][s2], dq[t][d]);
            }
        hcp_dma_wait_all();
        HCP_SYNC();
    };
    const bool ragged = (p.Nk & (KVT - 1)) != 0;
    for (int it = 0; it < nt - 1; ++it) tile(BoolC<false>{}, it);
    if (ragged) tile(BoolC<true>{}, nt - 1); else tile(BoolC<false>{}, nt - 1);
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
                for (int r = 0; r < 4; ++r) w[r] = (short)hcp_f2bf(dq[t][d][r] * (PRE ? LN2 : p.scale));   // PRE: d/dQ' = (scale / (scale log2e)) dS K
                *(hcp_bf16x4*)(orow + col) = w;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
// A workgroup owns 64*KT keys (K / V fragments in registers, K pre-multiplied by scale*log2e) and streams the Q|dO tiles plus
// the 64 (-lse2, -delta) pairs of each tile; S = Q K^T un-transposed, so each lane owns one KEY column and the accumulator
// initialisers are the per-query quads read straight from LDS.  The 64-query tile is processed in two halves to bound the
// live score registers.  dV^T += dO^T P and dK^T += Q^T dS read both streamed images with transpose reads.
template <int D, int KT> constexpr int dkv_waves() { return (D > 80 || KT == 2) ? 2 : 3; }

template <int D, int KT, bool KB, int VAR>
HCP_WAVES_PER_SIMD((dkv_waves<D, KT>())) HCP_KERNEL(256) attn2_bwd_dkv_kernel(AttnParams p) {
    using G = Geom<D>;
    constexpr bool PRE = (VAR & VAR_PRE) != 0;        // the streamed Q tiles are Q' = Q * scale*log2(e): dK = ln2 * dS^T Q'
    constexpr bool RAW = (VAR & VAR_RAW) != 0 && !PRE;
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;                  // 2 x { Q [64][RS] | dO [64][RS] | -lse2[64], -delta[64] (fp32) }
    constexpr int BUF = 2 * G::IMG + 4 * KVT;         // 2*64 floats = 4*64 bf16 slots
    const int tid = threadIdx.x, lane = tid & 63, wave = hcp_uniform(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    int item = blockIdx.x;
    if (VAR & VAR_XCD) item = xcd_work_item(item, gridDim.x);
    const int nkb = (p.Nk + 64 * KT - 1) / (64 * KT);
    const int per_bh = nkb * p.qsplit;
    const int bh = item / per_bh, rem = item - bh * per_bh, h = bh % p.H, b = bh / p.H;
    const int kblk = rem / p.qsplit, qs = rem - kblk * p.qsplit;
    const int k_base = kblk * (64 * KT) + wave * (16 * KT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const hcp_bf16* dOb = p.dO + (size_t)b * p.o_bs + h * D;
    const float* lse_b = p.lse + ((size_t)b * p.H + h) * p.Nq;
    const float* del_b = p.delta + ((size_t)b * p.H + h) * p.Nq;
    const float c2 = p.scale * LOG2E;
    const float cs = RAW ? c2 : 1.0f;

    for (int i = tid * 8; i < 2 * BUF; i += 256 * 8) *(hcp_bf16x8*)(lds + i) = hcp_zero8();
    TileDma<D> dma;
    dma.init(wave, lane, p.q_rs, p.o_rs);
    const int nt_all = (p.Nq + KVT - 1) / KVT;
    const int per = (nt_all + p.qsplit - 1) / p.qsplit;
    const int it0 = qs * per;
    const int nt = it0 + per < nt_all ? it0 + per : nt_all;      // this workgroup walks query tiles [it0, nt)
    // statistics of the next tile: raw values travel in one register (tid < 64: lse, 64 <= tid < 128: delta) and are negated /
    // scaled when they are stored to LDS, so nothing waits on the load inside the tile
    const float* stat_ptr = tid < KVT ? lse_b : del_b - KVT;
    float rl = 0.f;
    bool rl_ok = false;
    auto load_stats = [&](int q0) {
        rl_ok = tid < 2 * KVT && q0 + (tid & (KVT - 1)) < p.Nq;
        rl = rl_ok ? stat_ptr[q0 + tid] : 0.f;
    };
    auto store_stats = [&](hcp_bf16* base) {
        float* sl = (float*)(base + 2 * G::IMG);
        if (tid < 2 * KVT) sl[tid] = tid < KVT ? -rl * (LOG2E / cs) : -rl;       // rows past the end: 0 (finite P against dO = 0, Q = 0)
    };
    auto nvalid_q = [&](int q0) { const int n = p.Nq - q0; return n < KVT ? (n > 0 ? n : 0) : KVT; };

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
    HCP_SYNC();                                       // zero fill complete
    if (it0 < nt) {
        dma.issue(Qb + (size_t)it0 * KVT * p.q_rs, p.q_rs, dOb + (size_t)it0 * KVT * p.o_rs, p.o_rs, nvalid_q(it0 * KVT), lds, wave);
        load_stats(it0 * KVT);
        store_stats(lds);
    }
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) if (!RAW && !PRE) kf[t][s] = scale8(kf[t][s], c2);
    const int qfull = fr * G::RS + fg * 8;
    const int qtail = fr * G::RS + ((G::NFULL * 32 + fg * 8) < D ? G::NFULL * 32 + fg * 8 : D);
    const int tfrag = (4 * fg + (fr >> 2)) * G::RS + 4 * (fr & 3);
    // the K / V fragments (and the keys' additive bias: one value per lane, loop invariant) are retired HERE, not at their first use
    // inside the loop (where the compiler's vmcnt(0) would sit right behind each tile's DMA issue)
    float kb2_t[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int key = k_base + t * 16 + fr;
        kb2_t[t] = (KB && key < p.Nk && p.kbias) ? p.kbias[(size_t)b * p.kb_bs + key] * (RAW ? 1.0f / p.scale : LOG2E) : 0.f;
        if (KB) hcp_force_ready(kb2_t[t]);
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) { hcp_force_ready(kf[t][s]); hcp_force_ready(vf[t][s]); }
    }
    hcp_dma_wait_all();
    HCP_SYNC();

    for (int it = it0; it < nt; ++it) {
        const int q0 = it * KVT;
        const hcp_bf16* sQ = lds + ((it - it0) & 1) * BUF;
        const hcp_bf16* sG = sQ + G::IMG;
        const float* sL = (const float*)(sG + G::IMG);
        hcp_bf16* nxt = lds + ((it + 1 - it0) & 1) * BUF;
        // (statistics load FIRST: the compiler guards the reuse of its destination register with s_waitcnt vmcnt(0), and behind the
        //  DMA issue that wait would also drain the tile prefetch — the wave would sit out the whole L2 -> LDS latency every tile)
        if (it + 1 < nt) {
            load_stats(q0 + KVT);
            dma.issue(Qb + (size_t)(q0 + KVT) * p.q_rs, p.q_rs, dOb + (size_t)(q0 + KVT) * p.o_rs, p.o_rs, nvalid_q(q0 + KVT), nxt, wave);
        }
        hcp_bf16x8 pf[KT][2], df[KT][2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            hcp_f32x4 sc[KT][2], dp[KT][2];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int qt = 2 * hf + q2;
                const hcp_f32x4 l4 = *(const hcp_f32x4*)(sL + qt * 16 + 4 * fg);
                const hcp_f32x4 d4 = *(const hcp_f32x4*)(sL + KVT + qt * 16 + 4 * fg);
#pragma unroll
                for (int s = 0; s < G::NQK; ++s) {
                    const int off = (s < G::NFULL ? qfull + s * 32 : qtail) + qt * 16 * G::RS;
                    const hcp_bf16x8 qa = *(const hcp_bf16x8*)(sQ + off);
                    const hcp_bf16x8 ga = *(const hcp_bf16x8*)(sG + off);
#pragma unroll
                    for (int t = 0; t < KT; ++t) {
                        sc[t][q2] = hcp_mfma16(qa, kf[t][s], s == 0 ? l4 : sc[t][q2]);   // S[q = qt*16 + 4fg + r][key = fr] - lse
                        dp[t][q2] = hcp_mfma16(ga, vf[t][s], s == 0 ? d4 : dp[t][q2]);   // dP - delta
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < KT; ++t) {
                const bool kok = k_base + t * 16 + fr < p.Nk;
                const float kb2 = kb2_t[t];
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int qt = 2 * hf + q2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pr = hcp_exp2((KB ? sc[t][q2][r] + kb2 : sc[t][q2][r]) * cs);
                        // A key past the end owns its lane's COLUMN of P / dS and nothing else: its (finite: K = V = 0) values only reach the
                        // dK / dV rows of that key, which are never stored — no select per score on the unmasked path.
                        if (KB && !kok) pr = 0.f;
                        if (KB && p.causal && k_base + t * 16 + fr > q0 + qt * 16 + 4 * fg + r) pr = 0.f;   // future key
                        sc[t][q2][r] = pr;
                        dp[t][q2][r] = pr * dp[t][q2][r];
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
                const int off = tfrag + (2 * s2) * 16 * G::RS + d * 16;
                const hcp_bf16x8 qa = join8(hcp_lds_read_tr4(sQ + off), hcp_lds_read_tr4(sQ + off + 16 * G::RS));
                const hcp_bf16x8 ga = join8(hcp_lds_read_tr4(sG + off), hcp_lds_read_tr4(sG + off + 16 * G::RS));
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    dv[t][d] = hcp_mfma16(ga, pf[t][s2], dv[t][d]);   // dV^T[dcol][key] += dO^T P
                    dk[t][d] = hcp_mfma16(qa, df[t][s2], dk[t][d]);   // dK^T[dcol][key] += Q^T dS
                }
            }
        if (it + 1 < nt) store_stats(nxt);
        hcp_dma_wait_all();
        HCP_SYNC();
    }
    const float kscale = PRE ? LN2 : p.scale;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
        const int row = k_base + t * 16 + fr;
        if (row >= p.Nk) continue;
        if (p.qsplit > 1) {                               // this split's own slab: plain 16-byte stores, summed by attn_dkv_convert_kernel
            const size_t slab = (size_t)qs * p.B * p.Nk * (p.H * D);
            float* k32 = p.dk32 + slab + ((size_t)b * p.Nk + row) * (p.H * D) + h * D;
            float* v32 = p.dv32 + slab + ((size_t)b * p.Nk + row) * (p.H * D) + h * D;
#pragma unroll
            for (int d = 0; d < G::NDV; ++d) {
                const int col = d * 16 + 4 * fg;
                if (col < D) {
                    *(hcp_f32x4*)(k32 + col) = dk[t][d] * kscale;
                    *(hcp_f32x4*)(v32 + col) = dv[t][d];
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
                for (int r = 0; r < 4; ++r) { wk[r] = (short)hcp_f2bf(dk[t][d][r] * kscale); wv[r] = (short)hcp_f2bf(dv[t][d][r]); }
                *(hcp_bf16x4*)(krow + col) = wk;
                *(hcp_bf16x4*)(vrow + col) = wv;
            }
        }
    }
}

template <int D> constexpr size_t dkv_smem() { return (size_t)2 * (2 * Geom<D>::IMG + 4 * KVT) * sizeof(hcp_bf16); }

}  // namespace hcp_attn
