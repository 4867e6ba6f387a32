// conv_patch.hip — 3x3 / stride 1 / pad 1 implicit-GEMM convolution (forward and data gradient) with an LDS-RESIDENT INPUT PATCH (gfx950).
//
// The ping-pong kernel (gemm_pp.hip) treats the convolution as a GEMM with K = 9 C: for every 64-channel chunk of the input it DMAs the
// workgroup's [BM pixels][64 channels] A tile NINE times — once per tap, shifted by one pixel — and sits on the LDS-array line (DESIGN 3a:
// per K tile 36.8 KB of LDS-DMA writes at ~56 B/clk + 72 KB of fragment reads against 640 MFMA cycles per SIMD).  Here the A operand of
// a chunk is written to LDS ONCE, as the (R + 2) x (W + 2) pixel patch the tile's BM = R W output pixels read through all nine taps; a
// tap is an address offset of the fragment reads.  LDS-DMA bytes per chunk: 33.8 KB (patch) + 9 x 20 KB (weights) instead of
// 9 x (16 + 20) KB: -34 %.
//   * K tiles run chunk-major: tile T = (chunk T / 9, tap T % 9); weights stay [Cout][ky][kx][C] (forward) / [Cin][ky][kx][Cout] (data
//     gradient, whose taps are mirrored: source pixel = output pixel + (1 - ky, 1 - kx)).
//   * THREE patch buffers: while chunk c is multiplied the loaders fill the patch of chunk c + 2 into the buffer chunk c - 1 left, one
//     1 KB piece (8 pixels x 64 channels) per loader wave per K tile — 4 waves x 9 taps = 36 slots for the 33 pieces of a 264-pixel
//     patch (slots 33..35 re-load pieces 0..2: every loader issues the same number of DMA instructions per tile, which is what the
//     counted vmcnt waits count).  A patch is complete a whole chunk before its first reader, so no wait beyond the ring's is needed.
//   * patch image: pixel p at p * 128 bytes, 16-byte chunk c of the 64 channels at slot c ^ ((p >> 1) & 7) — the swizzle of the GEMM
//     tiles with the patch pixel in the place of the tile row: 16 consecutive pixels read conflict-free from any start.
//   * everything else is the ping-pong kernel: 4 loader waves + two compute groups of 2 x 2 waves half a phase apart that split every
//     K tile by k-step, partial sums exchanged through LDS once, epilogue (bias, row bias, residual).
// Eligible launches (conv_patch_launch returns -2 otherwise and the caller keeps its kernels): stride 1, pad 1, no upsample, Ho x Wo =
// Hs x Ws with Ho Wo % BM == 0 and BM % Wo == 0, (BM / Wo + 2)(Wo + 2) <= 264, C1 and C2 multiples of 64, no K-extension.
#include "gemm_params.h"
#include <type_traits>

namespace hcp_gemm {
namespace {

constexpr int PATCH_PX = 264;                      // pixels a patch buffer holds (64 x 64 level: 4 x 66)
constexpr int PATCH_PIECES = PATCH_PX / 8;         // 1 KB DMA pieces of a patch
constexpr int PATCH_ELEMS = PATCH_PX * BK;

template <int BM, int BN, int MODE, int NST>
HCP_KERNEL(768) conv_patch_kernel(GemmParams p) {
    static_assert(MODE == 1 || MODE == 2, "forward / data gradient");
    constexpr int NC = 8, NLD = 4;
    constexpr int NTC = 64 * NC, NTL = 64 * NLD;
    constexpr int WTM = BM / 2, WTN = BN / 2;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int TMF = TM / 2;
    constexpr int RPP = NTL / 8;                          // rows one DMA pass of the loaders covers (32)
    constexpr int B_IT = BN / RPP;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BN % RPP == 0 && TM % 2 == 0, "tile shape");
    static_assert(NST >= 2 && NST <= 4, "ring depth");
    constexpr int B_ELEMS = BN * BK;
    HCP_DYN_SMEM(smem);
    hcp_bf16* const patch = (hcp_bf16*)smem;              // 3 x [PATCH_PX][64]
    hcp_bf16* const ring = patch + 3 * PATCH_ELEMS;       // NST x [BN][64]

    const int tid_all = threadIdx.x;
    const int lane = tid_all & 63;
    const int wave_all = hcp_uniform(tid_all >> 6);
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;

    const int Ctot = p.cv.C1 + p.cv.C2;
    const int W = p.cv.Wo, H = p.cv.Ho;
    const int W2 = W + 2;
    const int nk1 = p.K / BK;                             // = 9 * (Ctot / 64)
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split; if (kt_end > nk1) kt_end = nk1;
    const int nk = kt_end - kt_begin;
    const int nch = Ctot / BK;
    const int c_begin = kt_begin / 9, tap_begin = kt_begin - 9 * c_begin;
    // the tile's pixels: sample b, rows y0 .. y0 + BM / W - 1 (all of the sample's columns)
    const int hw = H * W;
    const int bimg = m0 / hw, y0 = (m0 - bimg * hw) / W;

    // ======================================================================================================== loader waves
    if (wave_all >= NC) {
        const int wave = wave_all - NC, tid = tid_all - NTC;
        const int kc = tid & 7, lrow = tid >> 3;
        unsigned vb[B_IT];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) {
            const int r = lrow + RPP * i, n = n0 + r;
            vb[i] = n < p.N ? (unsigned)(((size_t)n * p.ldb + ((kc ^ ((r >> 1) & 7)) << 3)) * 2) : HCP_BUF_OOB;
        }
        // patch pieces of this wave: slot s (one per tap of a chunk) -> piece q = (4 s + wave) % 33, pixels 8 q .. 8 q + 7; lane = (pixel, chunk)
        unsigned vp[9];                                   // source pixel index of the lane's patch pixel, or ~0u (outside the image / the patch)
        unsigned pch[9];                                  // byte offset of the lane's swizzled 16-byte channel chunk inside a 64-channel chunk
        const int NP = (BM / W + 2) * W2;
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int q = (4 * s + wave) % PATCH_PIECES;
            const int pp = q * 8 + (lane >> 3);
            const int ry = pp / W2, cx = pp - ry * W2;
            const int sy = y0 - 1 + ry, sx = cx - 1;
            vp[s] = (pp < NP && sy >= 0 && sy < H && sx >= 0 && sx < W) ? (unsigned)((bimg * H + sy) * W + sx) : ~0u;
            pch[s] = (unsigned)((((lane & 7) ^ ((cx >> 1) & 7)) << 3) * 2);          // swizzle by the patch COLUMN: a tap's row shift leaves it alone
        }
        constexpr int IPT = B_IT + 1;                     // B row groups, then the patch piece
        static_assert(IPT * (NST - 1) < 64, "vmcnt immediate");

        // piece of slot S of the patch of channel chunk `cchunk`; past the last chunk every lane is out of range: zeros into a buffer nobody
        // reads, no memory traffic.  The slot is a compile-time index (wave-uniform switch below): a run-time index into vp / pch compiled to a
        // chain of v_cndmask per tile — VALU work that holds up the compute waves' MFMAs on the same SIMD.
        auto piece_at = [&](auto S, int cchunk) {
            constexpr int sidx = decltype(S)::value;
            const bool live = cchunk < nch;
            const int cb = (live ? cchunk : 0) * BK;
            const bool first = cb < p.cv.C1;
            const hcp_bf16* base = first ? p.cv.X1 + cb : p.cv.X2 + (cb - p.cv.C1);
            const hcp_rsrc ra = hcp_make_rsrc(base);
            const unsigned cs2 = (unsigned)(2 * (first ? p.cv.C1 : p.cv.C2));
            const int q = (4 * sidx + wave) % PATCH_PIECES;
            hcp_bf16* dst = patch + (cchunk % 3) * PATCH_ELEMS + q * 8 * BK;
            hcp_buf_glds16(ra, (live && vp[sidx] != ~0u) ? vp[sidx] * cs2 + pch[sidx] : HCP_BUF_OOB, dst);
        };
        auto patch_piece = [&](int s, int cchunk) {
            switch (s) {
                case 0: piece_at(std::integral_constant<int, 0>{}, cchunk); break;
                case 1: piece_at(std::integral_constant<int, 1>{}, cchunk); break;
                case 2: piece_at(std::integral_constant<int, 2>{}, cchunk); break;
                case 3: piece_at(std::integral_constant<int, 3>{}, cchunk); break;
                case 4: piece_at(std::integral_constant<int, 4>{}, cchunk); break;
                case 5: piece_at(std::integral_constant<int, 5>{}, cchunk); break;
                case 6: piece_at(std::integral_constant<int, 6>{}, cchunk); break;
                case 7: piece_at(std::integral_constant<int, 7>{}, cchunk); break;
                default: piece_at(std::integral_constant<int, 8>{}, cchunk); break;
            }
        };
        // K tile cursor of the NEXT tile to issue: (chunk ci, tap ti); B rows of tile (c, tap) start at B + tap * Ctot + 64 c
        int ci = c_begin, ti = tap_begin;
        auto issue_part = [&](auto Q0, auto Q1, int buf, int s_cons, int c_cons) {
            constexpr int q0 = decltype(Q0)::value, q1 = decltype(Q1)::value;
            hcp_bf16* lb = ring + buf * B_ELEMS;
            const hcp_rsrc rb = hcp_make_rsrc(p.B + (size_t)ti * Ctot + (size_t)ci * BK);
#pragma unroll
            for (int i = 0; i < B_IT; ++i)
                if (i >= q0 && i < q1) hcp_buf_glds16(rb, vb[i], lb + (wave * 8 + RPP * i) * BK);
            if (B_IT >= q0 && B_IT < q1) patch_piece(s_cons, c_cons + 2);      // the consumer is at (c_cons, tap s_cons): fill the patch two chunks ahead
        };
        int issued = 0, wbuf = 0;
        auto tile_done = [&]() {
            ++ti; if (ti == 9) { ti = 0; ++ci; }
            ++issued; wbuf = wbuf + 1 == NST ? 0 : wbuf + 1;
        };
        using I0 = std::integral_constant<int, 0>;
        using IALL = std::integral_constant<int, IPT>;
        auto wait_tiles = [&](int rem) {                  // rem = whole tile groups that may stay in flight
            if (NST >= 4 && rem >= 3) hcp_wait_vmcnt_c<(NST >= 4 ? 3 : 0) * IPT>();
            else if (NST >= 4 && rem == 2) hcp_wait_vmcnt_c<(NST >= 4 ? 2 : 0) * IPT>();
            else if (NST >= 3 && rem >= 1) hcp_wait_vmcnt_c<(NST >= 3 ? 1 : 0) * IPT>();
            else hcp_wait_vmcnt_c<0>();
        };
        // consumer cursor (the tile being multiplied in iteration t)
        int cc = c_begin, ct = tap_begin;
        auto cons_next = [&]() { ++ct; if (ct == 9) { ct = 0; ++cc; } };
        auto quiet_iter = [&](int t) {                    // nothing left to issue for the ring: the patch pieces keep the instruction count uniform
            hcp_barrier_only();
            wait_tiles(issued - (t + 2));
            hcp_barrier_only();
        };
        // prologue: the patches of the first two chunks in full (nine pieces per wave each), then the ring
#pragma unroll
        for (int s = 0; s < 9; ++s) patch_piece(s, c_begin);
#pragma unroll
        for (int s = 0; s < 9; ++s) patch_piece(s, c_begin + 1);
        // the ring; every group carries a patch instruction too (uniform instruction count): group 0 the piece consumer iteration 0 owes
        // the patch of chunk c_begin + 2 (that iteration issues nothing: the ring is full), the others write zeros into pieces of that same (idle) buffer which iterations 1 .. NST - 1 fill for real later (DMA of one wave lands in order)
        for (int i = 0; i < NST && issued < nk; ++i) {
            issue_part(I0{}, IALL{}, wbuf, i == 0 ? tap_begin : i, i == 0 ? c_begin : c_begin + 3 * 4096); tile_done();
        }
        wait_tiles(issued - 1);
        hcp_barrier_only();                               // P: tile 0 and both patches are in LDS
        quiet_iter(0);
        cons_next();
        int t = 1;
        constexpr int NPART = NST >= 3 ? 2 : 1;
        constexpr int LASTQ = IPT * (NPART - 1) / NPART;
        for (; t + NST - 1 < nk; ++t) {
            hcp_barrier_only();                           // #2t
            if constexpr (NPART == 1) {
                issue_part(I0{}, IALL{}, wbuf, ct, cc); tile_done();
                hcp_wait_vmcnt_c<0>();
                hcp_barrier_only();
            } else {
                issue_part(I0{}, std::integral_constant<int, LASTQ>{}, wbuf, ct, cc);
                hcp_wait_vmcnt_c<(NST - 3) * IPT + LASTQ>();
                hcp_barrier_only();                       // #2t+1: tile t+1 is in LDS
                issue_part(std::integral_constant<int, LASTQ>{}, IALL{}, wbuf, ct, cc); tile_done();
            }
            cons_next();
        }
        for (; t < nk; ++t) { quiet_iter(t); cons_next(); }
        hcp_barrier_only();                               // #2nk
        hcp_barrier_only(); hcp_barrier_only();          // the exchange of the compute groups
        return;
    }

    // ======================================================================================================= compute waves
    const int g = wave_all >> 2, gm = (wave_all >> 1) & 1, gn = wave_all & 1;
    const int fr = lane & 15, fg = lane >> 4;
    // Row of the 16-row block this lane's A fragment (and so its accumulator column) stands for.  A ds_read_b128 is served in groups of 16
    // lanes {fr 0-3, 12-15 of one fg; fr 4-11 of the next fg}; the two fg of a group read different 16-byte slots of the pixel, so with
    // pixel = fr the XOR swizzle is conflict-free only for block starts that are multiples of 16 (the GEMM tiles) — a patch read starts at
    // any pixel (measured: 22 % of the LDS read cycles were bank conflicts).  With the even pixels on fr 0-3 / 12-15 and the odd ones
    // on fr 4-11, every group's two fg fall into different bank halves (pixel parity = address bit 7) and inside a half the eight pixels
    // have eight distinct (p >> 1) & 7: conflict-free from any start.  Only the row index of the epilogue changes with it.
    const int frp = fr < 4 ? 2 * fr : (fr >= 12 ? 2 * (fr - 8) : 2 * (fr - 4) + 1);
    const int row0 = gm * WTM;
    const int frow0 = gm * WTM + g * (WTM / 2);
    const int col0 = gn * WTN;
    // Element offsets of this lane's A fragments inside a patch buffer for the three COLUMN shifts a tap can have (row shifts and the buffer
    // are a wave-uniform addend): the first version derived every address from the tap inside the loop — ~25 VALU instructions per K tile
    // and wave, which on gfx950 do not overlap with the other wave's MFMAs on the same SIMD: +30 % on the whole kernel.
    const int kslot = g * 4 + fg;                         // 16-byte slot of this lane's k-step inside the 64-channel chunk (group g = k-step g)
    int a_rd[3][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int ml = row0 + i * 16 + frp;
        const int ly = ml / W, lx = ml - ly * W;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) a_rd[dx][i] = (ly * W2 + lx + dx) * BK + ((kslot ^ (((lx + dx) >> 1) & 7)) << 3);
    }
    const int b_rd0 = (col0 + fr) * BK + (((fg ^ ((fr >> 1) & 7)) << 3) ^ (g * 32));

    hcp_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }

    hcp_f32x4 bias_v[TN];
    hcp_bf16x4 res_v[TMF][TN];
    auto load_bias = [&]() {
        const hcp_rsrc rbias = hcp_make_rsrc_n(p.bias, p.bias ? (unsigned)p.N * 4u : 0u);
#pragma unroll
        for (int j = 0; j < TN; ++j) bias_v[j] = hcp_buf_load16f(rbias, (unsigned)(n0 + col0 + j * 16 + 4 * fg) * 4u);
    };
    auto load_residual = [&]() {
        const hcp_rsrc rres = hcp_make_rsrc_n(p.residual, p.residual ? (unsigned)(((size_t)(p.M - 1) * p.ldr + p.N) * 2) : 0u);
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int m = m0 + frow0 + i * 16 + frp;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                res_v[i][j] = hcp_buf_load8(rres, ((unsigned)m * (unsigned)p.ldr + (unsigned)(n0 + col0 + j * 16 + 4 * fg)) * 2u);
        }
    };
    if (p.nsplit == 1) load_residual();
    hcp_barrier_only();                                   // P
    if (g == 1) hcp_barrier_only();                       // #0: group 1 runs half a phase behind group 0
    int cc = c_begin, ky = 0;                             // (splits start on chunk boundaries: tap 0)
    for (int t = 0, st = 0; t < nk; t += 3) {             // three taps (one kernel row) per trip: the column shift is a compile-time index
        const int dy = MODE == 1 ? ky : 2 - ky;
        const hcp_bf16* pa = patch + (cc % 3) * PATCH_ELEMS + dy * W2 * BK;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            constexpr int unused = 0; (void)unused;
            const int dx = MODE == 1 ? kx : 2 - kx;
            const hcp_bf16* sp = ring + st * B_ELEMS;
            hcp_bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *(const hcp_bf16x8*)(pa + a_rd[dx][i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *(const hcp_bf16x8*)(sp + b_rd0 + j * 16 * BK);
            hcp_barrier_only();
            hcp_setprio<1>();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fb[j], fa[i], acc[i][j]);
            hcp_setprio<0>();
            hcp_barrier_only();
            st = st + 1 == NST ? 0 : st + 1;
        }
        ++ky; if (ky == 3) { ky = 0; ++cc; }
    }
    if (g == 0) hcp_barrier_only();                       // #2nk

    // ---- the two groups' partial sums meet (gemm_pp.hip): exchange area = the patch buffers
    if (g == 1) {
#pragma unroll
        for (int i = 0; i < TMF; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) { const hcp_f32x4 tmp = acc[i][j]; acc[i][j] = acc[TMF + i][j]; acc[TMF + i][j] = tmp; }
    }
    {
        hcp_f32x4* const xb = (hcp_f32x4*)smem;
#pragma unroll
        for (int i = 0; i < TMF; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) xb[((wave_all * TMF + i) * TN + j) * 64 + lane] = acc[TMF + i][j];
        if (p.nsplit == 1) load_bias();
        hcp_barrier_keep_dma();
        const int pw = wave_all ^ 4;
#pragma unroll
        for (int i = 0; i < TMF; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += xb[((pw * TMF + i) * TN + j) * 64 + lane];
        hcp_barrier_keep_dma();
    }

    if (p.nsplit > 1) {
#pragma unroll
        for (int i = 0; i < TMF; ++i) {
            const int m = m0 + frow0 + i * 16 + frp;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + col0 + j * 16 + 4 * fg;
                if (n < p.N) *(hcp_f32x4*)(p.slabs + ((size_t)split * p.M + m) * p.N + n) = acc[i][j];
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TMF; ++i) {
        const int m = m0 + frow0 + i * 16 + frp;
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

template <int BM, int BN, int MODE>
int launch_patch(GemmParams& p, int ring, hipStream_t stream) {
    constexpr size_t stage = (size_t)BN * BK * sizeof(hcp_bf16);
    constexpr size_t pbytes = (size_t)3 * PATCH_ELEMS * sizeof(hcp_bf16);
    constexpr size_t xchg = (size_t)8 * (BM / 64) * (BN / 32) * 64 * 16;
    static_assert(xchg <= pbytes, "the exchange area fits the patch buffers");
    constexpr size_t cap = 160 * 1024;
    const dim3 grid(p.tiles_m * hcp_cdiv(p.N, BN), p.nsplit);
    if (ring >= 3 && pbytes + 3 * stage <= cap) {
        constexpr int R = pbytes + 3 * stage <= cap ? 3 : 2;
        HCP_LAUNCH((conv_patch_kernel<BM, BN, MODE, R>), grid, dim3(768), pbytes + R * stage, stream, p);
    } else {
        static_assert(pbytes + 2 * stage <= cap, "LDS budget");
        HCP_LAUNCH((conv_patch_kernel<BM, BN, MODE, 2>), grid, dim3(768), pbytes + 2 * stage, stream, p);
    }
    HCP_LAUNCH_CHECK("conv_patch_kernel");
}

}  // namespace

// -2: not eligible / not instantiated (the caller keeps its own kernels).  `p` arrives as for gemm_pp_launch.
int conv_patch_launch(GemmParams& p, int bm, int bn, int mode, int ring, hipStream_t stream) {
    const ConvDesc& cv = p.cv;
    if (mode != 1 && mode != 2) return -2;
    if (cv.stride != 1 || cv.up || cv.pad != 1 || cv.Ho != cv.Hs || cv.Wo != cv.Ws) return -2;
    if (cv.C1 % BK || cv.C2 % BK || p.K2 || p.K != 9 * (cv.C1 + cv.C2) || p.geglu_hg || p.geglu_out) return -2;
    const int hw = cv.Ho * cv.Wo;
    if (hw % bm || bm % cv.Wo || (bm / cv.Wo + 2) * (cv.Wo + 2) > PATCH_PX) return -2;
    if ((size_t)p.N * p.ldb * 2 >= (1ul << 31) || (size_t)p.M * (cv.C1 > cv.C2 ? cv.C1 : cv.C2) * 2 >= (1ul << 31)) return -2;
    if (p.residual && (size_t)p.M * p.ldr * 2 >= (1ul << 31)) return -2;
    if (p.nsplit > 1) {                                   // splits on chunk boundaries (a patch is filled tap by tap of the chunk two ahead)
        const int nk1 = p.K / BK;
        const int kt = hcp_cdiv(hcp_cdiv(nk1, p.nsplit), 9) * 9;
        p.kt_per_split = kt; p.nsplit = hcp_cdiv(nk1, kt);
        if (p.nsplit == 1) p.kt_per_split = nk1;
    }
    p.epi_tile = 0;
    if (bm == 128 && bn == 160) return mode == 1 ? launch_patch<128, 160, 1>(p, ring, stream) : launch_patch<128, 160, 2>(p, ring, stream);
    if (bm == 128 && bn == 128) return mode == 1 ? launch_patch<128, 128, 1>(p, ring, stream) : launch_patch<128, 128, 2>(p, ring, stream);
    return -2;
}

}  // namespace hcp_gemm
