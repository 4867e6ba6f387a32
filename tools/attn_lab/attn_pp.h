// attn_pp.h — LAB ONLY (tools/attn_lab; not part of the product library): the "ping-pong" forward, TWO WAVE GROUPS HALF A TILE APART.
// Kept as the measured negative result that led to attn_sw.h: on gfx950 a wave's MFMAs do not overlap with the VALU work of ANOTHER wave
// of the same SIMD (tools/probes/mfma_valu_overlap.hip), so pairing a matrix segment with the partner's VALU segment buys nothing
// (121-128 us against 129 us for generation 2; matrix-only ablation 81 us, VALU-only 55 us: additive).
//
// Same math, tensor contract, LDS images and fragment code as attn_dma.h (diffusers CrossAttention core: softmax(Q K^T / sqrt(d)) V —
// reference call sites train_ac.py:258-260, unet_struct.txt:17-43).  What changes is WHEN each wave does what.
//
// Round 2 measured the second-generation forward at 1165 cycles per (32 query rows x 64 keys) wave-tile per SIMD against 28 MFMAs
// (~480-540 cycles of matrix pipe) and ~96 VALU ops (~450-550 cycles, the 32 quarter-rate v_exp_f32 first): close to the SUM of the
// two, not their maximum (MFMA busy 0.41).  Every wave ran  QK^T (matrix) -> softmax (VALU) -> PV (matrix) -> barrier  and the
// one-barrier-per-tile rendezvous kept the four waves of a SIMD in the SAME phase: they competed for the matrix pipe, then for the VALU.
//
// Here a workgroup is 8 waves = two groups of four (waves w and w+4 share a SIMD: a workgroup's waves are dealt to SIMDs cyclically).
// Per 64-key tile each wave runs two SEGMENTS separated by workgroup barriers:
//
//     M(k): PV(k-1) and QK^T(k)   — matrix pipe + LDS fragment reads (P of tile k-1 is packed bf16 in registers, scores of tile k
//                                    come out in fp32 accumulators)
//     V(k): softmax(k), LDS-DMA issue for tile k+2 — VALU / transcendental + a few scalar instructions
//
// and group 1 runs ONE SEGMENT BEHIND group 0, so in every slot one wave of a SIMD is in its matrix segment while its partner is in its
// VALU segment (MI355X guide, "Two waves per SIMD": matrix beside VALU is the complementary pairing; matrix beside matrix is not).
//
//     slot        0      1      2      3      4     ...
//     group 0   M(0)   V(0)   M(1)   V(1)   M(2)
//     group 1    -     M(0)   V(0)   M(1)   V(1)
//
// K/V tiles travel global -> LDS by DMA into a ring of FOUR {K | V} image pairs: tile j is read in M(j) (K) and M(j+1) (V), i.e. in
// slots 2j .. 2j+3 across both groups; tile j+2 is issued from V(j) (slots 2j+1 / 2j+2) into the pair that tile j-2 left in slot 2j-1,
// and every wave retires its own DMA (vmcnt(0)) at the end of the next M segment, one barrier before the first reader.  (Three pairs
// are one too few: group 0 would overwrite tile j-1's V while group 1 still reads it.)
#pragma once
#include "attn_sw.h"

namespace hcp_attn {

constexpr int VAR_PRIO1 = 1024;   // static s_setprio(1) for the second-dispatched half (guide: "static priority for the younger half")
constexpr int VAR_NOSM = 2048;    // ablation (wrong results): the VALU segment only packs the raw scores
constexpr int VAR_PREF = 16384;   // fragments of the NEXT matrix segment are requested at the end of the VALU segment, across the barrier
constexpr int PP_NBUF = 4;

// scheduling fence + workgroup barrier: the segments must not leak into each other (MFMAs are pure register ops the scheduler would
// otherwise sink below the barrier, putting matrix work of both groups into the same slot again)
HCP_DEVICE void pp_barrier() {
    hcp_sched_fence();
    hcp_barrier_only();          // no lgkmcnt(0): fragment reads in flight across the barrier target registers, and every LDS image
    hcp_sched_fence();           // outlives its last read by >= 2 slots (ring of four); the compiler waits on each read at its use
}

template <int D, int VAR, int WPS = 2, int NW = 8>
HCP_WAVES_PER_SIMD(WPS) HCP_KERNEL(64 * NW) attn3_fwd_kernel(AttnParams p) {
    using G = Geom<D>;
    constexpr int QT = 2, NBUF = PP_NBUF;
    static_assert(NW == 8 || NW == 16, "two groups of 4 or 8 waves");
    constexpr bool PRE = (VAR & VAR_PRE) != 0;
    constexpr bool ONES = G::SPARE && (VAR & VAR_ONES);                 // row sums from the PV MFMA (d = 40)
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;                                    // NBUF x { K [64][RS] | V [64][RS] }
    constexpr int BUF = 2 * G::IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = hcp_uniform(tid >> 6);
    const int grp = wave / (NW / 2);                                    // 0 leads, 1 runs one segment behind (waves w, w+4, w+8, ... share a SIMD)
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int ROWS = 16 * QT * NW;                                  // 256 / 512 query rows per workgroup
    const int nqt = (p.Nq + ROWS - 1) / ROWS;
    int item = blockIdx.x;
    if (VAR & VAR_XCD) item = xcd_work_item(item, gridDim.x);
    const int qtile = item % nqt, bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q_base = qtile * ROWS + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const float cs = PRE ? 1.0f : p.scale * LOG2E;                      // what one unit of the accumulator is worth in the exp2 domain
    const float rescale_thr = 6.0f / cs;

    for (int i = tid * 8; i < NBUF * BUF; i += 64 * NW * 8) *(hcp_bf16x8*)(lds + i) = hcp_zero8();
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
    HCP_SYNC();                                                         // zero fill complete
    if (ONES && tid < NBUF * KVT) {
        hcp_bf16x8 one8;
#pragma unroll
        for (int e = 0; e < 8; ++e) one8[e] = 0x3F80;
        *(hcp_bf16x8*)(lds + (tid >> 6) * BUF + G::IMG + (tid & 63) * G::RS + D) = one8;
    }
    auto rows_of = [&](int t) { const int n = p.Nk - t * KVT; return n < KVT ? n : KVT; };
    auto fetch = [&](int t) {
        dma.template issue<true>(Kb + (size_t)t * KVT * p.k_rs, p.k_rs, Vb + (size_t)t * KVT * p.v_rs, p.v_rs, rows_of(t), lds + (t & (NBUF - 1)) * BUF, wave);
    };
    fetch(0);
    if (nt > 1) fetch(1);
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) hcp_force_ready(qf[t][s]);

    float m_i[QT], l_i[QT];
    hcp_f32x4 o[QT][G::NDV], nm4[QT], sc[QT][4];
    hcp_bf16x8 pf[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_i[t] = 0.f; l_i[t] = 0.f;
        hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
        nm4[t] = z;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) o[t][d] = z;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) sc[t][kt] = z;
        pf[t][0] = hcp_zero8(); pf[t][1] = hcp_zero8();
    }
    const int kfull = fr * G::RS + fg * 8;                                                    // + kt*16*RS + s*32
    const int ktail = fr * G::RS + ((G::NFULL * 32 + fg * 8) < D ? G::NFULL * 32 + fg * 8 : D);   // beyond the head dim: the zero pad granule
    const int vfrag = (4 * fg + (fr >> 2)) * G::RS + 4 * (fr & 3);                            // + (2*s2+j)*16*RS + dt*16
    hcp_dma_wait_all();
    HCP_SYNC();                                                         // tiles 0 (and 1) landed, pads initialised
    if ((VAR & VAR_PRIO1) && grp) hcp_setprio<1>();

    // ---- fragments requested one segment early (VAR_PREF): k-step 0 of the next QK^T and key half 0 of the next PV
    constexpr bool PREF = (VAR & VAR_PREF) != 0;
    hcp_bf16x8 kfA[4], vfA[G::NDV];
    auto load_k0 = [&](int it) {
        const hcp_bf16* sK = lds + (it & (NBUF - 1)) * BUF;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) kfA[kt] = *(const hcp_bf16x8*)(sK + (0 < G::NFULL ? kfull : ktail) + kt * 16 * G::RS);
    };
    auto load_v0 = [&](int it) {
        const hcp_bf16* sV = lds + (it & (NBUF - 1)) * BUF + G::IMG;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) {
            const hcp_bf16* a = sV + vfrag + d * 16;
            vfA[d] = join8(hcp_lds_read_tr4(a), hcp_lds_read_tr4(a + 16 * G::RS));
        }
    };
    // ---- matrix segment pieces
    auto qk = [&](auto ragged_c, int it) {                              // scores of tile `it`, relative to the running reference max
        constexpr bool RAGGED = decltype(ragged_c)::value;
        const hcp_bf16* sK = lds + (it & (NBUF - 1)) * BUF;
        const int nvalid = p.Nk - it * KVT;
        // k-step outermost: the MFMAs that chain on one accumulator are 2*4 instructions apart (a dependent 16x16x32 issued right behind
        // its producer stalls ~8 cycles: measured 23.7 instead of 16 cycles per MFMA with the chains two apart)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const hcp_bf16x8 kf = (VAR & VAR_NOLDS) ? qf[0][s] : (PREF && s == 0) ? kfA[kt]
                                      : *(const hcp_bf16x8*)(sK + (s < G::NFULL ? kfull + s * 32 : ktail) + kt * 16 * G::RS);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    hcp_f32x4 init = nm4[t];
                    if (RAGGED && s == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) if (kt * 16 + 4 * fg + r >= nvalid) init[r] = -INFINITY;   // dead key: p = 0
                    }
                    if (VAR & VAR_NOMM) { if (s == 0) sc[t][kt] = init; sc[t][kt][0] += hcp_bf2f((unsigned short)kf[s]); }
                    else sc[t][kt] = hcp_mfma16(kf, qf[t][s], s == 0 ? init : sc[t][kt]);
                }
            }
    };
    auto pv = [&](int it) {                                             // O^T += V^T P^T of tile `it` (P packed by softmax(it))
        const hcp_bf16* sV = lds + (it & (NBUF - 1)) * BUF + G::IMG;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int d = 0; d < G::NDV; ++d) {
                const hcp_bf16* a = sV + vfrag + (2 * s2) * 16 * G::RS + d * 16;
                const hcp_bf16x8 vf = (VAR & VAR_NOLDS) ? qf[1][0] : (PREF && s2 == 0) ? vfA[d] : join8(hcp_lds_read_tr4(a), hcp_lds_read_tr4(a + 16 * G::RS));
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    if (VAR & VAR_NOMM) o[t][d][0] += hcp_bf2f((unsigned short)vf[s2]);
                    else o[t][d] = hcp_mfma16(vf, pf[t][s2], o[t][d]);
                }
            }
    };
    // ---- VALU segment: issue the DMA of tile it+2, then scores -> packed probabilities (lazy rescale by wave vote)
    auto softmax = [&](auto first_c, int it) {
        constexpr bool FIRST = decltype(first_c)::value;
        if (it + 2 < nt) fetch(it + 2);
        if (VAR & VAR_NOSM) {
#pragma unroll
            for (int t = 0; t < QT; ++t) { pf[t][0] = pack8(sc[t][0], sc[t][1]); pf[t][1] = pack8(sc[t][2], sc[t][3]); }
            return;
        }
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float mx = hcp_max16(sc[t]);
            if (FIRST || !hcp_all(mx <= rescale_thr)) {
                float rm = fmaxf(mx, hcp_shfl_xor(mx, 16));
                rm = fmaxf(rm, hcp_shfl_xor(rm, 32));                   // row maximum of this tile, relative to m_i
                float delta = FIRST ? rm : fmaxf(rm, 0.f);
                delta = delta > -1e30f ? delta : 0.f;                    // a fully masked row keeps its reference
                m_i[t] += delta;
                const hcp_f32x4 n4 = {-m_i[t], -m_i[t], -m_i[t], -m_i[t]};
                nm4[t] = n4;
                if (!FIRST) {                                            // everything accumulated so far sits at the old reference
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
                    const float e = (VAR & VAR_NOEXP) ? sc[t][kt][r] * cs : hcp_exp2(PRE ? sc[t][kt][r] : sc[t][kt][r] * cs);
                    sc[t][kt][r] = e;
                    if (!ONES) rs += e;
                }
            if (!ONES) l_i[t] += rs;
            pf[t][0] = pack8(sc[t][0], sc[t][1]);
            pf[t][1] = pack8(sc[t][2], sc[t][3]);
        }
    };

    const bool ragged = (p.Nk & (KVT - 1)) != 0;
    if (grp) pp_barrier();                                              // group 1 starts one slot late
    // M(0) / V(0)
    if (PREF) load_k0(0);
    if (nt == 1 && ragged) qk(BoolC<true>{}, 0); else qk(BoolC<false>{}, 0);
    pp_barrier();
    softmax(BoolC<true>{}, 0);
    if (PREF) { load_v0(0); if (nt > 1) load_k0(1); }                   // what M(1) consumes first: in flight across the barrier
    pp_barrier();
    for (int k = 1; k < nt - 1; ++k) {
        pv(k - 1);
        qk(BoolC<false>{}, k);
        hcp_dma_wait_all();                                             // own share of tile k+1 (issued in V(k-1)) has landed
        pp_barrier();
        softmax(BoolC<false>{}, k);
        if (PREF) { load_v0(k); load_k0(k + 1); }
        pp_barrier();
    }
    if (nt > 1) {
        const int k = nt - 1;
        pv(k - 1);
        if (ragged) qk(BoolC<true>{}, k); else qk(BoolC<false>{}, k);
        hcp_dma_wait_all();
        pp_barrier();
        softmax(BoolC<false>{}, k);
        if (PREF) load_v0(k);
        pp_barrier();
    }
    pv(nt - 1);                                                         // M(nt): the last tile's PV
    pp_barrier();
    if (!grp) pp_barrier();                                             // (same barrier count in both groups)

    // epilogue: lane holds O[q = q_base + t*16 + fr][d*16 + 4*fg + r]
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = q_base + t * 16 + fr;
        float lsum;
        if (ONES) lsum = hcp_shfl(o[t][D / 16][D % 16 % 4], ((D % 16) / 4) * 16 + fr);   // O^T row D lives in lane group (D%16)/4
        else { lsum = l_i[t]; lsum += hcp_shfl_xor(lsum, 16); lsum += hcp_shfl_xor(lsum, 32); }
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

template <int D> constexpr int pp_fwd_rows(int nw) { return 32 * nw; }
template <int D> constexpr size_t pp_fwd_smem() { return (size_t)PP_NBUF * 2 * Geom<D>::IMG * sizeof(hcp_bf16); }

}  // namespace hcp_attn
