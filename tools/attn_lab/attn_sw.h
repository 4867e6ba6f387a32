// attn_sw.h — third-generation attention forward for gfx950: the softmax of tile k, the QK^T of tile k+1 and the PV of tile k-1 are
// ONE instruction stream per wave (software pipeline inside the wave), because that is the only place where this chip overlaps
// matrix and vector work.
//
// Same math, tensor contract, LDS images and fragment code as attn_dma.h (diffusers CrossAttention core: softmax(Q K^T / sqrt(d)) V —
// reference call sites train_ac.py:258-260, unet_struct.txt:17-43).
//
// What round 3 measured first (tools/probes/mfma_valu_overlap.hip, MI355X): a wave issuing MFMAs back to back and a second wave ON THE
// SAME SIMD issuing v_exp_f32 / v_mul_f32 take the SUM of their times (168 vs 67 + 106 ns; 89 vs 67 + 32 ns) — the matrix pipe does
// not overlap with another wave's VALU work.  The same MFMAs and exps INTERLEAVED IN ONE WAVE take 94 ns (max, not sum): an MFMA is
// covered by the independent VALU instructions that follow it in its own wave.  The second-generation kernel ran QK^T -> softmax -> PV
// in strictly dependent phases (each needs the previous one's result), so nothing followed an MFMA but another MFMA: 1165 cycles per
// (32 rows x 64 keys) wave-tile = MFMA time + VALU time.  A two-group "ping-pong" variant (matrix segment of one wave beside the VALU
// segment of its SIMD partner, tools/attn_lab/attn_pp.h) confirmed the additive law: 121 us against 81 us matrix-only and 55 us
// VALU-only ablations.
//
// Here every wave holds TWO score tiles: while the VALU turns sc[cur] (tile k, finished in the previous iteration) into packed
// probabilities, the matrix pipe fills sc[nxt] with tile k+1 and adds P(k-1) V(k-1) into O.  The 28 MFMAs of an iteration depend on
// nothing the iteration's ~100 VALU instructions produce, and sched_group_barrier pins the interleave (guide T15 / T19).  The rare
// work that breaks the pattern — the first tile (sets the reference maximum), a tile whose scores outgrow the reference by more than
// 2^6 (lazy rescale, wave vote), a ragged last tile — runs on an un-pipelined slow path.
#pragma once
#include "attn_dma.h"

namespace hcp_attn {

constexpr int VAR_NOSGB = 1024;   // lab: leave the interleave to the compiler's scheduler (no sched_group_barrier)
constexpr int VAR_NOMAX = 2048;   // ablation (wrong on adversarial data): no per-step maximum / vote (the reference maximum is the first half tile's)
constexpr int VAR_NOMM = 4096;    // ablation (wrong results): fragments are read, no MFMA is issued
constexpr int VAR_NOLDS = 8192;   // ablation (wrong results): MFMAs on stale fragments (no LDS reads)
constexpr int SW_NBUF = 4;        // {K | V} image pairs in the LDS ring: tiles k-1 (V), k+1 (K) are read while k+2 lands

// scheduling pipeline of one fast iteration: N x { 1 MFMA, NT transcendental, NV plain VALU, ND LDS reads }
template <int N, int NT, int NV, int ND>
HCP_DEVICE void sw_interleave() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        hcp_sched_group<0x008, 1>();                      // MFMA
        if (NT) hcp_sched_group<0x400, NT ? NT : 1>();    // TRANS (v_exp_f32)
        if (NV) hcp_sched_group<0x002, NV ? NV : 1>();    // VALU
        if (ND) hcp_sched_group<0x100, ND ? ND : 1>();    // DS read
    }
}

// f(IntC<I>{}) for I = 0 .. N-1 (the chunk index must be a constant expression: sched_group_barrier takes immediates)
template <int I, int N, class F>
HCP_DEVICE void sw_static_for(F&& f) {
    if constexpr (I < N) { f(IntC<I>{}); sw_static_for<I + 1, N>(f); }
}

// (the per-lane maximum of 8 fresh MFMA results is hcp_max8, hcp_device.h)

// The pipeline's unit is a HALF tile (32 keys = one k-step of the PV MFMA): two score half-tiles + two packed-P half-tiles per wave are
// 48 registers (a whole 64-key tile in flight twice needed 96 and spilled at the 256-register budget of two waves per SIMD).
//
// Every step is the SAME straight-line block — QK^T(hh+1), exp(hh), PV(hh-1) — so the loop has no merge points to copy registers at
// (a first version with separate slow / fast / last paths spent 30+ v_mov / v_cndmask per tile at the joins and 33 us of 182 on the
// running-maximum check: an asm block reading fresh MFMA results, a vote and a three-way branch per step).  What used to be paths are
// rare in-place FIXUPS in front of / behind the block:
//   * lazy rescale from the ROW SUMS: no per-score maximum is taken at all.  The reference is the exact row maximum of the first half
//     tile; after that a step only looks at the running sum of probabilities the PV MFMA keeps in O's spare row (d = 40; a VALU partial
//     sum otherwise): one compare per 16 rows.  When a sum passes 2^20 the next step first adds the pending P(hh-1) V into O (it sits at
//     the old reference), then divides O and l by the row's sum and moves the reference by its log2.  Magnitude costs no precision (P is
//     bf16 with its own exponent, O and l are fp32); only exp2 overflow (a score more than 2^127 above the reference) would be fatal:
//     the epilogue checks the row sums and the workgroup then repeats its rows on the exact, un-pipelined path (per-score maxima).
//   * dead keys of a ragged last tile: their scores are overwritten with -inf behind the block (K rows past the end are zero-filled
//     by the DMA, so the MFMAs themselves need no mask).
template <int D, int VAR, int NW = 8, int WPS = 2>
HCP_WAVES_PER_SIMD(WPS) HCP_KERNEL(64 * NW) attn4_fwd_kernel(AttnParams p) {
    using G = Geom<D>;
    constexpr int QT = 2, NBUF = SW_NBUF, HT = 32;
    constexpr bool PRE = (VAR & VAR_PRE) != 0;
    constexpr bool ONES = G::SPARE && (VAR & VAR_ONES);                 // row sums from the PV MFMA (d = 40)
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;                                    // NBUF x { K [64][RS] | V [64][RS] } + one flag word
    constexpr int BUF = 2 * G::IMG;
    const int tid = threadIdx.x, lane = tid & 63, wave = hcp_uniform(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    constexpr int ROWS = 16 * QT * NW;                                  // query rows per workgroup
    const int nqt = (p.Nq + ROWS - 1) / ROWS;
    int item = blockIdx.x;
    if (VAR & VAR_XCD) item = xcd_work_item(item, gridDim.x);
    const int qtile = item % nqt, bh = item / nqt, h = bh % p.H, b = bh / p.H;
    const int q_base = qtile * ROWS + wave * (16 * QT);
    const hcp_bf16* Qb = p.Q + (size_t)b * p.q_bs + h * D;
    const hcp_bf16* Kb = p.K + (size_t)b * p.k_bs + h * D;
    const hcp_bf16* Vb = p.V + (size_t)b * p.v_bs + h * D;
    const float cs = PRE ? 1.0f : p.scale * LOG2E;                      // what one unit of the accumulator is worth in the exp2 domain
    const float rcs = 1.0f / cs;
    constexpr float L_LAZY = 1048576.0f;                                // a row's sum of probabilities may reach 2^20 before O is rescaled

    for (int i = tid * 8; i < NBUF * BUF + 8; i += 64 * NW * 8) *(hcp_bf16x8*)(lds + i) = hcp_zero8();
    int* redo_flag = (int*)(lds + NBUF * BUF);
    TileDma<D, NW> dma;
    dma.init(wave, lane, p.k_rs, p.v_rs);
    const int nt = (p.Nk + KVT - 1) / KVT;
    const int nh = 2 * nt;                                              // half tiles (the last one or two may be partly / wholly dead)
    const int first_dead = (p.Nk & (HT - 1)) ? p.Nk / HT : (p.Nk / HT < nh ? p.Nk / HT : nh);   // first half tile that contains a dead key

    hcp_bf16x8 qf[QT][G::NQK];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < G::NQK; ++s) {
            const int row = q_base + t * 16 + fr, dc = s * 32 + fg * 8;
            qf[t][s] = (row < p.Nq && dc < D) ? *(const hcp_bf16x8*)(Qb + (size_t)row * p.q_rs + dc) : hcp_zero8();
        }
    HCP_SYNC();                                                         // zero fill complete
    if (ONES) {
        for (int i = tid; i < NBUF * KVT; i += 64 * NW) {
            hcp_bf16x8 one8;
#pragma unroll
            for (int e = 0; e < 8; ++e) one8[e] = 0x3F80;
            *(hcp_bf16x8*)(lds + (i >> 6) * BUF + G::IMG + (i & 63) * G::RS + D) = one8;
        }
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
    hcp_f32x4 o[QT][G::NDV], nm4[QT], sc[2][QT][2];                     // sc[hh & 1]: scores of half tile hh
    hcp_bf16x8 pf[2][QT];                                               // pf[hh & 1]: packed probabilities of half tile hh
    hcp_bf16x8 kfA[2];                                                  // k-step-0 fragments of the NEXT step's QK^T, requested one step early
    const int kfull = fr * G::RS + fg * 8;                                                    // + kt*16*RS + s*32
    const int ktail = fr * G::RS + ((G::NFULL * 32 + fg * 8) < D ? G::NFULL * 32 + fg * 8 : D);   // beyond the head dim: the zero pad granule
    const int vfrag = (4 * fg + (fr >> 2)) * G::RS + 4 * (fr & 3);                            // + (2*s2+j)*16*RS + dt*16
    hcp_dma_wait_all();
    HCP_SYNC();                                                         // tiles 0 (and 1) landed, pads initialised

    // ---- pieces (straight-line code).  Half tile hh = rows 32*(hh & 1) .. +31 of tile hh >> 1.
    auto k_half = [&](int hh) { return lds + ((hh >> 1) & (NBUF - 1)) * BUF + (hh & 1) * HT * G::RS; };
    auto ldk0 = [&](int hh) {                                            // k-step 0 of half tile hh's K fragments
        const hcp_bf16* sK = k_half(hh);
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) kfA[kt] = *(const hcp_bf16x8*)(sK + (0 < G::NFULL ? kfull : ktail) + kt * 16 * G::RS);
    };
    auto qk = [&](int hh, hcp_f32x4 (&s_out)[QT][2]) {                  // scores of half tile hh, relative to the running reference (kfA = its k-step 0)
        const hcp_bf16* sK = k_half(hh);
#pragma unroll
        for (int s = 0; s < G::NQK; ++s)                                // k-step outermost: MFMAs chained on one accumulator are 4 apart
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const hcp_bf16x8 kf = (VAR & VAR_NOLDS) ? qf[0][s] : s == 0 ? kfA[kt]
                                      : *(const hcp_bf16x8*)(sK + (s < G::NFULL ? kfull + s * 32 : ktail) + kt * 16 * G::RS);
#pragma unroll
                for (int t = 0; t < QT; ++t) {
                    if (VAR & VAR_NOMM) { if (s == 0) s_out[t][kt] = nm4[t]; s_out[t][kt][0] += hcp_bf2f((unsigned short)kf[s]); }
                    else s_out[t][kt] = hcp_mfma16(kf, qf[t][s], s == 0 ? nm4[t] : s_out[t][kt]);
                }
            }
    };
    auto mask_dead = [&](int hh, hcp_f32x4 (&s_out)[QT][2]) {           // ragged end: keys past Nk get probability 0
        const int nvalid = p.Nk - hh * HT;
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) if (kt * 16 + 4 * fg + r >= nvalid) s_out[t][kt][r] = -INFINITY;
    };
    auto pv = [&](int hh, const hcp_bf16x8 (&pin)[QT]) {                // O^T += V^T P^T of half tile hh (one 32-key k-step)
        const hcp_bf16* sV = lds + ((hh >> 1) & (NBUF - 1)) * BUF + G::IMG + (hh & 1) * HT * G::RS;
#pragma unroll
        for (int d = 0; d < G::NDV; ++d) {
            const hcp_bf16* a = sV + vfrag + d * 16;
            const hcp_bf16x8 vf = (VAR & VAR_NOLDS) ? qf[1][0] : join8(hcp_lds_read_tr4(a), hcp_lds_read_tr4(a + 16 * G::RS));
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                if (VAR & VAR_NOMM) o[t][d][0] += hcp_bf2f((unsigned short)vf[t]);
                else o[t][d] = hcp_mfma16(vf, pin[t], o[t][d]);
            }
        }
    };
    auto expo = [&](hcp_f32x4 (&s_in)[QT][2], hcp_bf16x8 (&pout)[QT]) {  // scores (relative to the reference) -> packed P; emax = the largest P
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float rs = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = (VAR & VAR_NOEXP) ? s_in[t][kt][r] * cs : hcp_exp2(PRE ? s_in[t][kt][r] : s_in[t][kt][r] * cs);
                    s_in[t][kt][r] = e;
                    if (!ONES) rs += e;
                }
            if (!ONES) l_i[t] += rs;                                     // lane-local partial; lanes are combined in the epilogue
            pout[t] = pack8(s_in[t][0], s_in[t][1]);
        }
    };
    // rare: the probabilities of half tile hh-1 outgrew the reference.  P(hh-1) (pending in p_prev) goes into O first, then O, l and
    // the scores of half tile hh (computed against the old reference) move to the new one.
    auto lsum_lane = [&](int t) { return ONES ? o[t][D / 16][D % 16 % 4] : l_i[t]; };   // (ONES: meaningful in lane group (D%16)/4 only)
    auto lsum_ok = [&](int t) { return (ONES && fg != (D % 16) / 4) || lsum_lane(t) <= L_LAZY; };
    auto fix = [&](int hh, hcp_f32x4 (&s_cur)[QT][2], hcp_bf16x8 (&p_prev)[QT]) {
        pv(hh - 1, p_prev);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            p_prev[t] = hcp_zero8();
            float rm;                                                   // the row's running sum of probabilities
            if (ONES) rm = hcp_shfl(lsum_lane(t), ((D % 16) / 4) * 16 + fr);
            else { rm = l_i[t]; rm += hcp_shfl_xor(rm, 16); rm += hcp_shfl_xor(rm, 32); }
            const bool up = rm > 1.0f;
            const float delta = up ? log2f(rm) * rcs : 0.f, alpha = up ? 1.0f / rm : 1.0f;
            m_i[t] += delta;
            const hcp_f32x4 n4 = {-m_i[t], -m_i[t], -m_i[t], -m_i[t]};
            nm4[t] = n4;
            l_i[t] *= alpha;
#pragma unroll
            for (int d = 0; d < G::NDV; ++d) o[t][d] *= alpha;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) s_cur[t][kt] -= delta;
        }
    };

    // One step: exp of half tile hh (scores in sc[PAR]), QK^T of half hh+1 (into sc[PAR ^ 1]), PV of half hh-1 (P in pf[PAR ^ 1]).
    auto step = [&](auto par_c, int hh) {
        constexpr int PAR = decltype(par_c)::value;
        if (!(VAR & VAR_NOMAX) && !hcp_all(lsum_ok(0) && lsum_ok(1))) fix(hh, sc[PAR], pf[PAR ^ 1]);
        hcp_sched_fence();
        qk(hh + 1, sc[PAR ^ 1]);                                        // (one half tile past the end on the last step: stale LDS, never used)
        expo(sc[PAR], pf[PAR]);
        pv(hh - 1, pf[PAR ^ 1]);
        if (!(VAR & VAR_NOLDS)) ldk0(hh + 2);                           // in flight across the step boundary
        if (!(VAR & VAR_NOSGB)) {
            sw_interleave<4 * G::NQK, (4 * G::NQK >= 16) ? 1 : 2, (4 * G::NQK >= 16) ? 2 : 3, 0>();   // QK^T MFMAs carry the exps (+ multiplies, maxima)
            sw_interleave<2 * G::NDV, 0, 2, 0>();                                                     // PV MFMAs the converts
        }
        hcp_sched_fence();
        if (hh + 1 >= first_dead && hh + 1 < nh) mask_dead(hh + 1, sc[PAR ^ 1]);
    };

    // exact, un-pipelined pass over half tiles [0, nh): the prologue uses it for half tile 0 (sets the first reference maximum), the
    // overflow fallback for everything
    auto exact_half = [&](int hh, bool first) {
        ldk0(hh);
        qk(hh, sc[0]);
        if (hh >= first_dead) mask_dead(hh, sc[0]);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float rm = hcp_max8(sc[0][t]);
            rm = fmaxf(rm, hcp_shfl_xor(rm, 16));
            rm = fmaxf(rm, hcp_shfl_xor(rm, 32));
            float delta = first ? rm : fmaxf(rm, 0.f);
            delta = delta > -1e30f ? delta : 0.f;                        // a fully masked row keeps its reference
            m_i[t] += delta;
            const hcp_f32x4 n4 = {-m_i[t], -m_i[t], -m_i[t], -m_i[t]};
            nm4[t] = n4;
            if (!first) {
                const float alpha = hcp_exp2(-delta * cs);
                l_i[t] *= alpha;
#pragma unroll
                for (int d = 0; d < G::NDV; ++d) o[t][d] *= alpha;
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) sc[0][t][kt] -= delta;
        }
    };
    auto reset = [&]() {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            m_i[t] = 0.f; l_i[t] = 0.f;
            hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
            nm4[t] = z;
#pragma unroll
            for (int d = 0; d < G::NDV; ++d) o[t][d] = z;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) { sc[0][t][kt] = z; sc[1][t][kt] = z; }
            pf[0][t] = hcp_zero8(); pf[1][t] = hcp_zero8();
        }
    };

    reset();
    exact_half(0, true);                                                // scores of half tile 0 against their own row maximum
    ldk0(1);
    for (int k = 0; k < nt; ++k) {
        if (k + 2 < nt) fetch(k + 2);
        step(IntC<0>{}, 2 * k);
        step(IntC<1>{}, 2 * k + 1);
        hcp_dma_wait_all();                                             // own share of tile k+2 has landed
        HCP_SYNC();                                                     // ... everyone's has; tile k-1's pair is free
    }
    pv(nh - 1, pf[1]);                                                  // the last half tile's PV

    auto row_sum = [&](int t) {
        float lsum;
        if (ONES) lsum = hcp_shfl(o[t][D / 16][D % 16 % 4], ((D % 16) / 4) * 16 + fr);   // O^T row D lives in lane group (D%16)/4
        else { lsum = l_i[t]; lsum += hcp_shfl_xor(lsum, 16); lsum += hcp_shfl_xor(lsum, 32); }
        return lsum;
    };
    // overflow check (see the header comment): a non-finite or zero row sum in ANY wave sends the whole workgroup through the exact pass
    {
        bool bad = false;
#pragma unroll
        for (int t = 0; t < QT; ++t) { const float ls = row_sum(t); bad = bad || !(ls > 0.f && ls < 3.0e38f); }
        if (!hcp_all(!bad) && lane == 0) *redo_flag = 1;
        HCP_SYNC();
        const bool redo = hcp_uniform(*redo_flag) != 0;
        if (redo) {                                                     // workgroup-uniform
            reset();
            HCP_SYNC();
            for (int k = 0; k < nt; ++k) {                              // one tile at a time through pair 0: correctness, not speed
                fetch(k);                                               // (lands in pair k & 3; nobody reads while it is in flight)
                hcp_dma_wait_all();
                HCP_SYNC();
                for (int j = 0; j < 2; ++j) {
                    exact_half(2 * k + j, k == 0 && j == 0);
                    expo(sc[0], pf[0]);
                    pv(2 * k + j, pf[0]);
                }
                HCP_SYNC();
            }
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

template <int D> constexpr size_t sw_fwd_smem() { return (size_t)SW_NBUF * 2 * Geom<D>::IMG * sizeof(hcp_bf16) + 16; }

}  // namespace hcp_attn
