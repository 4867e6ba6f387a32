// gemm_pp.hip — the "ping-pong" main loop of the bf16 GEMM / fused-LoRA GEMM / 3x3 implicit-GEMM convolution (gfx950).
//
// Same problem statement, operands, LDS tile image, fused-LoRA tail and epilogue as gemm_v2_kernel (gemm.hip); what changes is
// WHO does what WHEN inside a workgroup.  The v2 loop runs its eight compute waves in lock step: between two workgroup barriers
// every wave first reads its fragments from LDS and then issues its MFMAs, so the two compute waves that share a SIMD wait for
// LDS together and then queue on the matrix pipe together (ablation of the C320 64x64 convolution, tools/probes/gemm_pp_probe:
// multiply alone 37 us, feed alone 34 us of a 39 us kernel).  Here:
//   * 12 waves: 4 loader waves (all LDS-DMA, ring of NST K tiles, counted vmcnt) + 8 compute waves in two GROUPS of 2 x 2;
//     waves w and w+4 (one of each group) share a SIMD.
//   * Both groups own the SAME BM x BN tile and split every 64-deep K tile: group g multiplies k-step g (k = 32g .. 32g+31).
//     A wave's tile is (BM/2) x (BN/2) — twice the rows of v2's 8-wave layout for the same workgroup tile, i.e. 36 % fewer LDS
//     fragment bytes per MFMA — and the two partial sums meet once, through LDS, after the loop (each wave keeps the half of
//     the rows it will finish: LoRA tail, epilogue).
//   * The groups run HALF A PHASE apart (group 1 executes one extra barrier up front): in every barrier interval one group
//     issues its 20 MFMAs while the other group's fragment reads for ITS next unit are in flight, so on each SIMD the matrix
//     pipe always has a wave whose operands are already in registers (multiply alone: 30 us, i.e. 0.33 us per K tile against
//     0.27-0.32 us of MFMA issue time; v2: 0.49).
//   * The loaders issue a K tile in PARTS spread over the barrier intervals of a tile period instead of all at once (see
//     issue_part): feed alone 27 us.
// What bounds it now (same probe): feed and multiply together take 36 us, not max(27, 30) — the LDS array is the shared resource.
// Per K tile the DMA writes 36.8 KB at the ~56 B/clk the LDS-DMA path sustains (tools/probes/lds_dma_feed: 120-140 GB/s per CU)
// and the fragment reads take 72 KB at 256 B/clk: 660 + 290 = 950 LDS cycles against 640 MFMA cycles per SIMD; v2 reads 114 KB:
// 660 + 450 = 1110.  Both kernels sit on that line (0.47 / 0.53 us per K tile), so at a 128 x 160 tile the MFMA pipe cannot be
// more than ~2/3 busy inside the loop; what is left is fewer LDS bytes per FLOP (larger tiles do not fill 256 CUs at batch 4).
// Barrier bookkeeping (nk K tiles; barriers are numbered after the prologue barrier P):
//     group 0:            [R(t) | #2t | M(t) | #2t+1] for t < nk, then #2nk
//     group 1:  #0, then  [R(t) | #2t+1 | M(t) | #2t+2] for t < nk
//     loaders:  per K tile t: #2t | first part of tile t+NST-1 into the slot of tile t-1 | wait(tile t+1) | #2t+1 | second part; then #2nk
//   RAW: tile t is read first by group 0 after barrier #2t-1, which every loader reaches only after its counted wait.
//   WAR: the last reads of tile t-1 (group 1) are consumed by MFMAs that precede its barrier #2t.
#include "gemm_params.h"
#include <type_traits>

namespace hcp_gemm {
namespace {

template <int BM, int BN, int MODE, bool LORA, int NST>
HCP_KERNEL(768) gemm_pp_kernel(GemmParams p) {
    constexpr int NC = 8, NLD = 4;
    constexpr int NTC = 64 * NC, NTL = 64 * NLD;
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int UPT = 1;                                // units per K tile and group (one k-step each)
    constexpr int TMF = TM / 2;                           // 16-row blocks a wave finishes (tail + epilogue)
    constexpr int RPP = NTL / 8;                          // rows one DMA pass of the loaders covers (32)
    constexpr int A_IT = BM / RPP, B_IT = BN / RPP;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % RPP == 0 && BN % RPP == 0 && TM % 2 == 0, "tile shape");
    static_assert(NST >= 2 && NST <= 4, "ring depth");
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, L_ELEMS = LORA ? 32 * BK : 0, BUF_ELEMS = A_ELEMS + B_ELEMS + L_ELEMS;
    constexpr int E_ELEMS = LORA ? BN * 32 : 0;           // E rows of this N tile (loaders, before the ring)
    HCP_DYN_SMEM(smem);
    hcp_bf16* const lds_e = (hcp_bf16*)smem;
    hcp_bf16* const ring = lds_e + E_ELEMS;

    const int tid_all = threadIdx.x;
    const int lane = tid_all & 63;
    const int wave_all = hcp_uniform(tid_all >> 6);
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;

    const int nk1 = p.K / BK;
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split; if (kt_end > nk1) kt_end = nk1;
    const int nprim = kt_end - kt_begin;
    const bool has_ext = (split == p.nsplit - 1) && p.K2 > 0;
    const int nk = nprim + (has_ext ? 1 : 0);

    // ======================================================================================================== loader waves
    if (wave_all >= NC) {
        const int wave = wave_all - NC, tid = tid_all - NTC;
        const int kc = tid & 7, lrow = tid >> 3;
#if defined(HCP_TOOLS)
        if (p.dbg & 0x2000) hcp_setprio<3>();             // A/B: loaders above the MFMA blocks in the issue arbitration
#endif
        // loop-invariant per-lane byte offsets; HCP_BUF_OOB = this lane contributes zeros (masked row / conv tap)
        const int Ctot = p.cv.C1 + p.cv.C2;
        unsigned va[A_IT], vb[B_IT];                      // MODE 0: byte offset of the row; conv: pixel index of the row
        unsigned a_msk[(A_IT + 2) / 3];                   // conv: word i/3, bit 9*(i%3) + tap = tap valid for row i
#pragma unroll
        for (int i = 0; i < (A_IT + 2) / 3; ++i) a_msk[i] = 0;
        const unsigned a_chunk = (unsigned)(((kc ^ ((lrow >> 1) & 7)) << 3) * 2);     // rows RPP apart share their swizzle term
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = lrow + RPP * i, nl = n0 + r, n = geglu_col(p, BN, nl);
            vb[i] = nl < p.N ? (unsigned)(((size_t)n * p.ldb + ((kc ^ ((r >> 1) & 7)) << 3)) * 2) : HCP_BUF_OOB;
        }
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            const int r = lrow + RPP * i, m = m0 + r;
            va[i] = HCP_BUF_OOB;
            if (m < p.M) {
                if (MODE == 0) {
                    va[i] = (unsigned)(((size_t)m * p.lda + ((kc ^ ((r >> 1) & 7)) << 3)) * 2);
                } else {
                    const int hw = p.cv.Ho * p.cv.Wo;
                    const int b = m / hw; const int rem = m - b * hw;
                    const int py = rem / p.cv.Wo, px = rem - py * p.cv.Wo;
                    const int s = MODE == 1 ? p.cv.stride : 1;
                    va[i] = (unsigned)((b * p.cv.Hs + py * s) * p.cv.Ws + px * s);
                    unsigned msk = 0;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int sy = MODE == 1 ? py * s + ky - p.cv.pad : py + 1 - ky;
                            const int sx = MODE == 1 ? px * s + kx - p.cv.pad : px + 1 - kx;
                            if (sy >= 0 && sy < p.cv.Hs && sx >= 0 && sx < p.cv.Ws) msk |= 1u << (ky * 3 + kx);
                        }
                    a_msk[i / 3] |= msk << (9 * (i % 3));
                }
            }
        }
        int tap = 0, cb = 0;                              // conv: (tap, channel cursor) of the NEXT tile to issue
        if (MODE != 0) { const int k0 = kt_begin * BK; tap = k0 / Ctot; cb = k0 - tap * Ctot; }
        // conv: the per-tile offset of row i is (tap valid ? pixel * 2 C + chunk : out of range).  Formed per tile that was a bit test, a
        // compare, a 32-bit multiply (quarter rate), an add and a select per row — ~25 VALU issues per K tile in every loader wave, and on gfx950
        // VALU work of ANY wave on a SIMD holds up the MFMAs of the compute waves it shares the SIMD with (measured with conv_patch.hip,
        // LAB_NOTEBOOK round 6).  Now: the products are loop invariants (one set per source tensor of a concat) and an invalid tap ORs bit 31
        // into the offset (>= num_records = out of range): one v_bfe + one v_lshl_or per row.
        unsigned vo1[MODE != 0 ? A_IT : 1], vo2[MODE != 0 ? A_IT : 1], a_nmsk[(A_IT + 2) / 3];
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const bool live = va[i] != HCP_BUF_OOB;
                vo1[i] = live ? va[i] * (unsigned)(2 * p.cv.C1) + a_chunk : HCP_BUF_OOB;
                vo2[i] = live ? va[i] * (unsigned)(2 * p.cv.C2) + a_chunk : HCP_BUF_OOB;
            }
#pragma unroll
            for (int i = 0; i < (A_IT + 2) / 3; ++i) a_nmsk[i] = ~a_msk[i];
        }
        const hcp_bf16* Ab = p.A + (size_t)kt_begin * BK;
        const hcp_bf16* Bb = p.B + (size_t)kt_begin * BK;
        const hcp_bf16* Lb = LORA ? p.L + (size_t)kt_begin * BK : nullptr;
        const unsigned vl = (unsigned)(((size_t)lrow * p.K + ((kc ^ ((lrow >> 1) & 7)) << 3)) * 2);

        // Instruction q of a K tile (q < IPT): B row group q, then (LORA) the 32 rows of L, then the A row groups.  A tile is issued in
        // PARTS (compile-time instruction ranges) spread over the barrier intervals of one K tile: an LDS-DMA instruction blocks its
        // wave until the CU's vector-memory path accepts it, and that path holds only a few instructions — a loader that issues a
        // whole tile and then sits at the barriers leaves the path idle for the rest of the tile period (measured on the C320 64x64
        // convolution: feed alone 30 us, multiply alone 30 us, together 41 us with whole-tile issues).
        constexpr int IPT = A_IT + B_IT + (LORA ? 1 : 0);
        static_assert(IPT * (NST - 1) < 64, "vmcnt immediate");
        auto issue_part = [&](auto Q0, auto Q1, int buf, bool ext) {
            constexpr int q0 = decltype(Q0)::value, q1 = decltype(Q1)::value;
            hcp_bf16* la = ring + buf * BUF_ELEMS;
            hcp_bf16* lb = la + A_ELEMS;
            if (!ext) {
                const hcp_rsrc rb = hcp_make_rsrc(Bb);
#pragma unroll
                for (int i = 0; i < B_IT; ++i)
                    if (i >= q0 && i < q1) hcp_buf_glds16(rb, vb[i], lb + (wave * 8 + RPP * i) * BK);
                if (LORA && B_IT >= q0 && B_IT < q1) {
                    const hcp_rsrc rl = hcp_make_rsrc(Lb);
                    hcp_buf_glds16(rl, vl, lb + B_ELEMS + (wave * 8) * BK);
                }
                constexpr int qa = B_IT + (LORA ? 1 : 0);
                if (qa + A_IT > q0 && qa < q1) {
                    if (MODE == 0) {
                        const hcp_rsrc ra = hcp_make_rsrc(Ab);
#pragma unroll
                        for (int i = 0; i < A_IT; ++i)
                            if (qa + i >= q0 && qa + i < q1) hcp_buf_glds16(ra, va[i], la + (wave * 8 + RPP * i) * BK);
                    } else {
                        const int ky = tap / 3, kx = tap - ky * 3;
                        const int doff = MODE == 1 ? (ky - p.cv.pad) * p.cv.Ws + (kx - p.cv.pad) : (1 - ky) * p.cv.Ws + (1 - kx);
                        const bool first = cb < p.cv.C1;
                        const hcp_bf16* base = first ? p.cv.X1 + (long)doff * p.cv.C1 + cb : p.cv.X2 + (long)doff * p.cv.C2 + (cb - p.cv.C1);
                        const hcp_rsrc ra = hcp_make_rsrc(base);
                        // (two copies of the loop under a wave-uniform branch: `first ? vo1[i] : vo2[i]` made hipcc keep both arrays in SCRATCH —
                        //  a scratch load per row inside the loop, whose vmcnt wait drains the DMA queue: +25 % on every convolution)
                        if (first) {
#pragma unroll
                            for (int i = 0; i < A_IT; ++i)
                                if (qa + i >= q0 && qa + i < q1)
                                    hcp_buf_glds16(ra, vo1[i] | (((a_nmsk[i / 3] >> (9 * (i % 3) + tap)) & 1u) << 31), la + (wave * 8 + RPP * i) * BK);
                        } else {
#pragma unroll
                            for (int i = 0; i < A_IT; ++i)
                                if (qa + i >= q0 && qa + i < q1)
                                    hcp_buf_glds16(ra, vo2[i] | (((a_nmsk[i / 3] >> (9 * (i % 3) + tap)) & 1u) << 31), la + (wave * 8 + RPP * i) * BK);
                        }
                    }
                }
            } else {                                      // the rank-32 K-extension tile: plain rows of A2 / B2, k < K2 only
                const hcp_rsrc rb = hcp_make_rsrc(p.B2), ra = hcp_make_rsrc(p.A2);
#pragma unroll
                for (int i = 0; i < B_IT; ++i)
                    if (i >= q0 && i < q1) {
                        const int r = lrow + RPP * i, nl = n0 + r, n = geglu_col(p, BN, nl), k = (kc ^ ((r >> 1) & 7)) << 3;
                        hcp_buf_glds16(rb, (nl < p.N && k < p.K2) ? (unsigned)(((size_t)n * p.ldb2 + k) * 2) : HCP_BUF_OOB, lb + (wave * 8 + RPP * i) * BK);
                    }
                if (LORA && B_IT >= q0 && B_IT < q1) hcp_buf_glds16(rb, HCP_BUF_OOB, lb + B_ELEMS + (wave * 8) * BK);      // keeps IPT uniform (LORA has no K2)
                constexpr int qa = B_IT + (LORA ? 1 : 0);
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    if (qa + i >= q0 && qa + i < q1) {
                        const int r = lrow + RPP * i, m = m0 + r, k = (kc ^ ((r >> 1) & 7)) << 3;
                        hcp_buf_glds16(ra, (m < p.M && k < p.K2) ? (unsigned)(((size_t)m * p.lda2 + k) * 2) : HCP_BUF_OOB, la + (wave * 8 + RPP * i) * BK);
                    }
            }
        };
        int issued = 0, wbuf = 0;                         // tiles completely issued; ring slot of the tile being issued
        auto tile_done = [&]() {                          // after the last part of a tile: advance the cursors
            if (issued < nprim) {
                Bb += BK;
                if (LORA) Lb += BK;
                if (MODE == 0) Ab += BK;
                else { cb += BK; if (cb >= Ctot) { cb -= Ctot; ++tap; } }
            }
            ++issued; wbuf = wbuf + 1 == NST ? 0 : wbuf + 1;
        };
        using I0 = std::integral_constant<int, 0>;
        using IALL = std::integral_constant<int, IPT>;
        if (LORA) {                                       // E rows n0 .. n0+BN, 64 bytes each: 16 rows per DMA instruction
            const hcp_rsrc re = hcp_make_rsrc(p.E);
            for (int gq = wave; gq < BN / 16; gq += NLD) {
                const int nl = n0 + gq * 16 + (lane >> 2), n = geglu_col(p, BN, nl);
                hcp_buf_glds16(re, nl < p.N ? (unsigned)(((size_t)n * 32 + (lane & 3) * 8) * 2) : HCP_BUF_OOB, lds_e + gq * 16 * 32);
            }
        }
        // every loader issues the same IPT instructions per tile and loads return in order: "tile x has landed" = at most
        // (instructions issued after x's last) of this wave's loads are still in flight
        auto wait_tiles = [&](int rem) {                  // rem = whole tiles that may stay in flight
            if (NST >= 4 && rem >= 3) hcp_wait_vmcnt_c<(NST >= 4 ? 3 : 0) * IPT>();
            else if (NST >= 4 && rem == 2) hcp_wait_vmcnt_c<(NST >= 4 ? 2 : 0) * IPT>();
            else if (NST >= 3 && rem >= 1) hcp_wait_vmcnt_c<(NST >= 3 ? 1 : 0) * IPT>();
            else hcp_wait_vmcnt_c<0>();
        };
        auto quiet_iter = [&](int t) {                    // an iteration with nothing left to issue: tile t+1 must have landed at its end
            hcp_barrier_only();
#pragma unroll
            for (int e = 0; e < 2 * UPT - 2; ++e) hcp_barrier_only();
            wait_tiles(issued - (t + 2));
            hcp_barrier_only();
        };
        for (int i = 0; i < NST && issued < nk; ++i) { issue_part(I0{}, IALL{}, wbuf, issued >= nprim); tile_done(); }
        wait_tiles(issued - 1);
        hcp_barrier_only();                               // P: tile 0 is in LDS
        quiet_iter(0);                                    // every ring slot was filled in the prologue
        int t = 1;
        constexpr int NPART = NST >= 3 ? 2 * UPT : 1;     // a 2-slot ring has no interval to spare: tile t+1 is the tile being issued
        constexpr int LASTQ = IPT * (NPART - 1) / NPART;  // first instruction of the final part
        for (; t + NST - 1 < nk; ++t) {                   // issues tile t+NST-1 into the slot of tile t-1 (free after barrier #2*UPT*t)
            const bool ext = issued >= nprim;
            hcp_barrier_only();                           // #2*UPT*t
#if defined(HCP_TOOLS)
            if (p.dbg & 0x300) {                          // ablation (results are wrong): 0x100 skip the A rows, 0x200 skip the B rows
                const hcp_rsrc rz = hcp_make_rsrc(p.B);
                hcp_bf16* la = ring + wbuf * BUF_ELEMS;
                if (!(p.dbg & 0x100)) for (int i = 0; i < A_IT; ++i) hcp_buf_glds16(rz, MODE == 0 ? (va[i] & 0xfffff) : (unsigned)(lane * 16 + i * 1024 + wave * 8192), la + (wave * 8 + RPP * i) * BK);
                if (!(p.dbg & 0x200)) for (int i = 0; i < B_IT; ++i) hcp_buf_glds16(rz, vb[i], la + A_ELEMS + (wave * 8 + RPP * i) * BK);
                ++issued; wbuf = wbuf + 1 == NST ? 0 : wbuf + 1;
                for (int e = 0; e < 2 * UPT - 2; ++e) hcp_barrier_only();
                hcp_wait_vmcnt_c<0>();
                hcp_barrier_only();
                continue;
            }
#endif
            if constexpr (NPART == 1) {
                issue_part(I0{}, IALL{}, wbuf, ext); tile_done();
#pragma unroll
                for (int e = 0; e < 2 * UPT - 2; ++e) hcp_barrier_only();
                hcp_wait_vmcnt_c<0>();
                hcp_barrier_only();                       // #2*UPT*(t+1)-1: tile t+1 is in LDS
            } else {
                if constexpr (NPART == 2) {
                    issue_part(I0{}, std::integral_constant<int, LASTQ>{}, wbuf, ext);
                } else {
                    issue_part(I0{}, std::integral_constant<int, IPT / 4>{}, wbuf, ext);
                    hcp_barrier_only();
                    issue_part(std::integral_constant<int, IPT / 4>{}, std::integral_constant<int, IPT / 2>{}, wbuf, ext);
                    hcp_barrier_only();
                    issue_part(std::integral_constant<int, IPT / 2>{}, std::integral_constant<int, LASTQ>{}, wbuf, ext);
                }
                hcp_wait_vmcnt_c<(NST - 3) * IPT + LASTQ>();      // whole tiles t+2 .. t+NST-2 and the parts above may stay in flight
                hcp_barrier_only();                       // #2*UPT*(t+1)-1: tile t+1 is in LDS
                issue_part(std::integral_constant<int, LASTQ>{}, IALL{}, wbuf, ext); tile_done();
            }
        }
        for (; t < nk; ++t) quiet_iter(t);
        hcp_barrier_only();                               // #2U
        hcp_barrier_only(); hcp_barrier_only();          // the exchange of the compute groups
        if (LORA) HCP_SYNC();                             // the compute waves' tail barrier
        if ((p.geglu_hg || p.geglu_out || p.epi_tile) && p.nsplit == 1) { if (LORA) HCP_SYNC(); HCP_SYNC(); }       // ... and those of the GEGLU / epilogue tiles
        return;
    }

    // ======================================================================================================= compute waves
    const int g = wave_all >> 2, gm = (wave_all >> 1) & 1, gn = wave_all & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int row0 = gm * WTM;                            // first row of this wave's tile in the main loop
    const int frow0 = gm * WTM + g * (WTM / 2);           // first row of the TMF blocks this wave finishes
    const int col0 = gn * WTN;
    // fragment addresses (elements) inside a ring slot: row R, 16-byte slot (ks*4 + fg) ^ ((R >> 1) & 7); the swizzle term only
    // depends on fr because every 16-row block starts at a multiple of 16; k-step 1 = k-step 0 XOR 32 elements
    const int sw0 = ((fg ^ ((fr >> 1) & 7)) << 3) ^ (g * 32);
    const int a_rd0 = (row0 + fr) * BK + sw0;
    const int b_rd0 = A_ELEMS + (col0 + fr) * BK + sw0;
    const int l_rd0 = A_ELEMS + B_ELEMS + (gn * 16 + fr) * BK + sw0;

    hcp_f32x4 acc[TM][TN];
    hcp_f32x4 tacc[LORA ? TM : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
#pragma unroll
    for (int i = 0; i < (LORA ? TM : 1); ++i) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; tacc[i] = z; }

    // residual fragments of the blocks this wave finishes.  The 128 x 160 fused-LoRA kernel has no registers to park them in
    // through the loop (80 accumulators + 16 for T + 40 fragment registers of the 168 a wave gets at three waves per SIMD) and
    // requests them in its tail.  Bounded buffer loads: rows past M read zeros, so there is no per-element branch (and no
    // per-element s_waitcnt) in front of the stores.
    constexpr bool EARLY = !(LORA && TM * TN >= 20);      // request them BEFORE the loop wherever the registers exist: a late request
                                                          // leaves ~2 us of round trip exposed (measured: conv C320 64x64 36.3 -> 38.2 us)
    hcp_f32x4 bias_v[TN];
    hcp_bf16x4 res_v[TMF][TN];
    auto load_bias = [&]() {                              // 640 bytes shared by every workgroup of the N tile: an L2 hit, requested late
        const hcp_rsrc rbias = hcp_make_rsrc_n(p.bias, p.bias ? (unsigned)p.N * 4u : 0u);
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_v[j] = hcp_buf_load16f(rbias, (unsigned)geglu_col(p, BN, n0 + col0 + j * 16 + 4 * fg) * 4u);
    };
    // Rows past M need no select: their offset is >= the resource's num_records (ldr >= N), so the hardware range check returns zeros —
    // and a `m < M ? offset : OOB` select here compiled to divergent branches with a WAW `s_waitcnt vmcnt(0)` between the loads: TMF
    // serialised round trips in front of the main loop.  32-bit offsets: gemm_pp_launch rejects residuals of 2^31 bytes or more.
    auto load_residual = [&]() {
        const hcp_rsrc rres = hcp_make_rsrc_n(p.residual, p.residual ? (unsigned)(((size_t)(p.M - 1) * p.ldr + p.N) * 2) : 0u);
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int m = m0 + frow0 + i * 16 + fr;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                res_v[i][j] = hcp_buf_load8(rres, ((unsigned)m * (unsigned)p.ldr + (unsigned)(n0 + col0 + j * 16 + 4 * fg)) * 2u);
        }
    };

    if (EARLY && p.nsplit == 1 && !p.epi_tile) load_residual();
    hcp_barrier_only();                                   // P
    if (g == 1) hcp_barrier_only();                       // #0: group 1 runs half a phase behind group 0
    for (int t = 0, st = 0; t < nk; ++t) {
        const hcp_bf16* sp = ring + st * BUF_ELEMS;
#pragma unroll
        for (int u = 0; u < UPT; ++u) {
#if defined(HCP_TOOLS)
            if (p.dbg & 0x800) { hcp_barrier_only(); hcp_barrier_only(); continue; }     // ablation: barriers only
#endif
            hcp_bf16x8 fa[TM], fb[TN], fl;
            constexpr int x = 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const hcp_bf16x8*)(sp + (a_rd0 ^ x) + i * 16 * BK);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const hcp_bf16x8*)(sp + (b_rd0 ^ x) + j * 16 * BK);
            if (LORA) fl = *(const hcp_bf16x8*)(sp + (l_rd0 ^ x));
            hcp_barrier_only();                           // the other group's MFMA block ends here; this wave's fragments are on their way
#if defined(HCP_TOOLS)
            if (p.dbg & 0x400) {                          // ablation: fragment reads only, no MFMAs
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
                hcp_barrier_only();
                continue;
            }
#endif
#if defined(HCP_TOOLS)
            if (!(p.dbg & 0x1000))
#endif
            hcp_setprio<1>();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fb[j], fa[i], acc[i][j]);
            if (LORA) {
#pragma unroll
                for (int i = 0; i < TM; ++i) tacc[i] = hcp_mfma16(fl, fa[i], tacc[i]);
            }
            hcp_setprio<0>();
            hcp_barrier_only();
        }
        st = st + 1 == NST ? 0 : st + 1;
    }
    if (g == 0) hcp_barrier_only();                       // #2nk: every MFMA of the loop has been issued, the ring is free

    // ---- the two groups' partial sums meet: a wave stores the row blocks its partner will finish and adds the partner's partial
    // of its own; [wave][block][column block][lane] in 16-byte pieces = conflict-free, 16 KB per (wave, block row).  Group 1
    // first swaps its halves so that both groups keep blocks 0 .. TMF-1 of the arrays (static register indices).
    if (g == 1) {
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) { const hcp_f32x4 tmp = acc[i][j]; acc[i][j] = acc[TMF + i][j]; acc[TMF + i][j] = tmp; }
            if (LORA) { const hcp_f32x4 tmp = tacc[i]; tacc[i] = tacc[TMF + i]; tacc[TMF + i] = tmp; }
        }
    }
    {
        hcp_f32x4* const xb = (hcp_f32x4*)ring;
        hcp_f32x4* const xt = xb + NC * TMF * TN * 64;
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) xb[((wave_all * TMF + i) * TN + j) * 64 + lane] = acc[TMF + i][j];
            if (LORA) xt[(wave_all * TMF + i) * 64 + lane] = tacc[TMF + i];
        }
        if (p.nsplit == 1) load_bias();                   // the round trip hides under the exchange
        hcp_barrier_keep_dma();
        const int pw = wave_all ^ 4;
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += xb[((pw * TMF + i) * TN + j) * 64 + lane];
            if (LORA) tacc[i] += xt[(pw * TMF + i) * 64 + lane];
        }
        hcp_barrier_keep_dma();                           // the exchange area is the ring: the LoRA tail re-uses it
    }

    if (LORA) {
        // T (bf16-rounded) and E = alpha * W_up rows of this N tile meet in LDS; one extra k-step adds T E^T
        constexpr int TS2 = 40;
        hcp_bf16* lt = ring;
        hcp_bf16* lt2 = ring + BM * TS2;                    // split T (p.ldt == 64): the residual image T_lo
        const bool split = p.ldt == 64;
        const int ldt = split ? 64 : 32;
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            hcp_bf16x4 o, o2;
            lora_t_split(tacc[i], o, o2);
            const int ml = frow0 + i * 16 + fr;
            *(hcp_bf16x4*)(lt + ml * TS2 + gn * 16 + 4 * fg) = o;
            if (split) *(hcp_bf16x4*)(lt2 + ml * TS2 + gn * 16 + 4 * fg) = o2;
            if (tile_n == 0 && p.Tout && m0 + ml < p.M) {
                *(hcp_bf16x4*)(p.Tout + (size_t)(m0 + ml) * ldt + gn * 16 + 4 * fg) = o;
                if (split) *(hcp_bf16x4*)(p.Tout + (size_t)(m0 + ml) * ldt + 32 + gn * 16 + 4 * fg) = o2;
            }
        }
        if (!EARLY && p.nsplit == 1 && !p.epi_tile) load_residual();
        HCP_SYNC();
        hcp_bf16x8 ft[TMF], fe[TN];
#pragma unroll
        for (int i = 0; i < TMF; ++i) ft[i] = *(const hcp_bf16x8*)(lt + (frow0 + i * 16 + fr) * TS2 + fg * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) fe[j] = *(const hcp_bf16x8*)(lds_e + (col0 + j * 16 + fr) * 32 + fg * 8);
#pragma unroll
        for (int i = 0; i < TMF; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fe[j], ft[i], acc[i][j]);
        if (split) {
#pragma unroll
            for (int i = 0; i < TMF; ++i) ft[i] = *(const hcp_bf16x8*)(lt2 + (frow0 + i * 16 + fr) * TS2 + fg * 8);
#pragma unroll
            for (int i = 0; i < TMF; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fe[j], ft[i], acc[i][j]);
        }
    }

    if (p.nsplit > 1) {
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int m = m0 + frow0 + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + col0 + j * 16 + 4 * fg;
                if (n < p.N) *(hcp_f32x4*)(p.slabs + ((size_t)split * p.M + m) * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    if (p.geglu_hg) {                                       // GEGLU-backward epilogue through LDS (gemm_params.h: geglu_tile_*)
        if (LORA) HCP_SYNC();                               // the LoRA tail's T image lives in the ring
#pragma unroll
        for (int i = 0; i < TMF; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) geglu_tile_put(ring, geglu_tile_ld(BN), frow0 + i * 16 + fr, col0 + j * 16 + 4 * fg, acc[i][j], p.alpha);
        HCP_SYNC();
        geglu_tile_apply<BM, BN, NTC>(p, ring, m0, n0, tid_all);
        return;
    }
    if (p.geglu_out) {                                      // GEGLU-forward epilogue (gemm_params.h: geglu_out): D = bf16(h | g), geglu_out = bf16(h gelu(g))
        static_assert((size_t)BM * BN * 2 <= (size_t)NST * BUF_ELEMS * sizeof(hcp_bf16), "the gelu(g) tile fits the ring");
        if (LORA) HCP_SYNC();                               // the LoRA tail's T image lives in the ring
        hcp_f32x4 v[TMF][TN];
        int rows[TMF], cols[TN];
        const bool is_g = gn == 1;                          // wave column 0 holds h, wave column 1 the matching g (WTN = BN / 2)
#pragma unroll
        for (int j = 0; j < TN; ++j) cols[j] = tile_n * (BN / 2) + j * 16 + 4 * fg;
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int m = m0 + frow0 + i * 16 + fr;
            rows[i] = m;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                v[i][j] = acc[i][j] * p.alpha + bias_v[j];
                if (m < p.M) store_hi_lo(p, m, cols[j] + (is_g ? (p.N >> 1) : 0), v[i][j]);
            }
        }
        geglu_fwd_pair<TMF, TN>(v, is_g, wave_all >> 1, lane, (hcp_f32x4*)ring, [] { HCP_SYNC(); }, p.geglu_out, p.N >> 1, rows, p.M, cols);
        return;
    }
    if (p.epi_tile) {                                       // tile epilogue (gemm_params.h: epi_tile_store): 16-byte row pieces
        if (LORA) HCP_SYNC();                               // the LoRA tail's T image lives in the ring
        float* const tile = (float*)ring;
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int ml = frow0 + i * 16 + fr, m = m0 + ml;
            const float* rbp = (p.rowbias && m < p.M) ? p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = col0 + j * 16 + 4 * fg;
                hcp_f32x4 v = acc[i][j] * p.alpha + bias_v[j];
                if (rbp && n0 + nl < p.N) v += *(const hcp_f32x4*)(rbp + n0 + nl);
                *(hcp_f32x4*)(tile + ml * epi_tile_ld(BN) + nl) = v;
            }
        }
        HCP_SYNC();
        epi_tile_store<BM, BN, NTC>(p, tile, m0, n0, tid_all);
        return;
    }
#pragma unroll
    for (int i = 0; i < TMF; ++i) {
        const int m = m0 + frow0 + i * 16 + fr;
        if (m >= p.M) continue;
        hcp_f32x4 rb_v[TN];
        if (p.rowbias) {
            const float* rbp = p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + col0 + j * 16 + 4 * fg;
                hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
                rb_v[j] = n < p.N ? *(const hcp_f32x4*)(rbp + n) : z;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + col0 + j * 16 + 4 * fg;
            if (n >= p.N) continue;
            hcp_f32x4 v = acc[i][j] * p.alpha + bias_v[j];
            if (p.rowbias) v += rb_v[j];
            if (p.residual) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += hcp_bf2f((unsigned short)res_v[i][j][q]);
                add_residual_lo(p, m, n, v);
            }
            if (p.out_f32) *(hcp_f32x4*)((float*)p.D + (size_t)m * p.ldd + n) = v;
            else store_hi_lo(p, m, n, v);
        }
    }
}

template <int BM, int BN, int MODE, bool LORA>
int launch_pp(GemmParams& p, int ring, hipStream_t stream) {
    constexpr size_t stage = (size_t)(BM + BN + (LORA ? 32 : 0)) * BK * sizeof(hcp_bf16);
    constexpr size_t eimg = LORA ? (size_t)BN * 32 * sizeof(hcp_bf16) : 0;
    constexpr int TMF = BM / 64;
    constexpr size_t xchg = (size_t)8 * TMF * (BN / 32) * 64 * 16 + (LORA ? (size_t)8 * TMF * 64 * 16 : 0);   // the groups' exchange area
    constexpr size_t tail = LORA ? (size_t)2 * BM * 40 * sizeof(hcp_bf16) : 0;   // T_hi and T_lo images
    constexpr size_t tile_bytes = (size_t)BM * epi_tile_ld(BN) * sizeof(float);     // tile epilogue (lives in the ring like the exchange area)
    if (p.epi_tile && tile_bytes + eimg > 160 * 1024) p.epi_tile = 0;
    const size_t floor_ = (xchg > tail ? xchg : tail) > (p.epi_tile ? tile_bytes : 0) ? (xchg > tail ? xchg : tail) : tile_bytes;
    constexpr size_t cap = 160 * 1024;
    const dim3 grid(p.tiles_m * hcp_cdiv(p.N, BN), p.nsplit);
    if (ring >= 4 && 4 * stage + eimg <= cap) {
        constexpr int R = 4 * stage + eimg <= cap ? 4 : 2;
        const size_t sm = (R * stage > floor_ ? R * stage : floor_) + eimg;
        HCP_LAUNCH((gemm_pp_kernel<BM, BN, MODE, LORA, R>), grid, dim3(768), sm, stream, p);
    } else if (ring >= 3 && 3 * stage + eimg <= cap) {
        constexpr int R = 3 * stage + eimg <= cap ? 3 : 2;
        const size_t sm = (R * stage > floor_ ? R * stage : floor_) + eimg;
        HCP_LAUNCH((gemm_pp_kernel<BM, BN, MODE, LORA, R>), grid, dim3(768), sm, stream, p);
    } else {
        const size_t sm = (2 * stage > floor_ ? 2 * stage : floor_) + eimg;
        HCP_LAUNCH((gemm_pp_kernel<BM, BN, MODE, LORA, 2>), grid, dim3(768), sm, stream, p);
    }
    HCP_LAUNCH_CHECK("gemm_pp_kernel");
}

template <int BM, int BN>
int launch_pp_mode(GemmParams& p, int mode, bool lora, int ring, hipStream_t stream) {
    if (lora) return launch_pp<BM, BN, 0, true>(p, ring, stream);
    if (mode == 0) return launch_pp<BM, BN, 0, false>(p, ring, stream);
    if (mode == 1) return launch_pp<BM, BN, 1, false>(p, ring, stream);
    return launch_pp<BM, BN, 2, false>(p, ring, stream);
}

}  // namespace

int gemm_pp_launch(GemmParams& p, int bm, int bn, int mode, bool lora, int ring, hipStream_t stream) {
    if (p.residual && (size_t)p.M * p.ldr * 2 >= (1ul << 31)) return -2;       // 32-bit buffer offsets
    if (bm == 128 && bn == 160) return launch_pp_mode<128, 160>(p, mode, lora, ring, stream);
    if (bm == 64 && bn == 160) return launch_pp_mode<64, 160>(p, mode, lora, ring, stream);
    if (bm == 128 && bn == 128) return launch_pp_mode<128, 128>(p, mode, lora, ring, stream);
    return -2;
}

}  // namespace hcp_gemm
