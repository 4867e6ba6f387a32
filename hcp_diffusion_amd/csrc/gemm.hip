// gemm.hip — bf16 MFMA GEMM (NT) and 3x3 implicit-GEMM convolution for gfx950.
//
//   D[M,N] = alpha * ( A[M,K] * B[N,K]^T  +  A2[M,K2] * B2[N,K2]^T ) + bias[N]
//            + rowbias[m / rows_per_group, N] + residual[M,N]
//
// * A, B are K-contiguous bf16.  The optional (A2,B2) pair extends the reduction: it is the
//   LoRA low-rank side path  y = x W^T + (x A^T)(alpha B)^T  (replaces the reference's
//   per-forward weight merge W + alpha*B@A, lora_base_patch.py:61-74).
// * MODE 1/2 replace the A operand by an on-the-fly im2col gather of an NHWC tensor:
//   3x3 forward (stride 1/2, fused nearest-2x upsample, two-pointer channel concat) and
//   3x3 data-gradient (transposed gather, stride 1/2).  Replaces F.conv2d reached through
//   diffusers ResnetBlock2D/Downsample2D/Upsample2D.
//   FAST variants (channel count a multiple of 64, no upsample / strided dgrad) hoist all
//   per-row address math out of the K loop: one base offset + a 9-bit tap-validity mask per row,
//   a wave-uniform (tap, channel) cursor advanced per K tile.
// * 64*WGM*WGN threads, BK=64, LDS double buffer (row stride 72 bf16 = conflict-free
//   ds_read_b128), register-staged prefetch of the next K tile, one barrier per K tile.
//   MFMA is issued with swapped operands so each lane owns 4 consecutive N of one row:
//   8-byte bf16x4 stores and float4 bias loads in the epilogue.
// * Tile shapes are chosen per problem so the grid fills 256 CUs: BN=160 divides every
//   Cout of the SD UNet (320/640/1280/2560/5120/10240); small-M (16x16, 8x8 latent) layers
//   use split-K into fp32 slabs + a fused reduce/epilogue kernel.
#include "hcp_common.h"
#include "gemm_params.h"

extern "C" int hcp_geglu_fwd(const void* h, void* y, long M, int F, hipStream_t stream);      // pointwise.hip

namespace {

using namespace hcp_gemm;


HCP_DEVICE void epilogue_store(const GemmParams& p, int m, int n, hcp_f32x4 v) {
    if (p.geglu_hg) { epilogue_geglu_bwd(p, m, n, v); return; }
    v = v * p.alpha;
    if (p.bias) v += *(const hcp_f32x4*)(p.bias + n);
    if (p.rowbias) v += *(const hcp_f32x4*)(p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld + n);
    if (p.residual) {
        hcp_bf16x4 r = *(const hcp_bf16x4*)(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += hcp_bf2f((unsigned short)r[q]);
        add_residual_lo(p, m, n, v);
    }
    if (p.out_f32) *(hcp_f32x4*)((float*)p.D + (size_t)m * p.ldd + n) = v;
    else store_hi_lo(p, m, n, v);
}

// MODE: 0 plain A, 1 conv forward gather, 2 conv data-gradient gather.  FAST: hoisted im2col addressing.
// (The first, register-staged main loop — global -> VGPR -> ds_write_b128 — was removed after the LDS-DMA loop replaced it.)

// 16 zero bytes in device memory: the source of every masked lane of an LDS-DMA load (out-of-range rows / columns,
// the zero padding of the convolution, K tails).
HCP_DEVICE_GLOBAL __attribute__((aligned(16))) unsigned char g_zero_page[16];

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA main loop.  Same tiling, fragment mapping, epilogues and fused-LoRA tail as gemm_kernel, but the K tiles
// travel global -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write: the register-staged loop is bound
// by the ~79 B/clk ds_write_b128 port, not by MFMA).  DMA destinations are lane-linear, so the LDS image is unpadded
// [rows][64] and bank conflicts are removed by an XOR swizzle applied to the SOURCE chunk index and to the reads:
// position (r, c) holds global chunk c ^ ((r >> 1) & 7), which makes the 16-lane ds_read_b128 groups of this
// fragment mapping conflict-free (even/odd rows fall in different bank halves of the 256-byte bank row).
template <int BM, int BN, int WGM, int WGN, int MODE, bool FAST, bool LORA = false, int NSTAGE = 2>
HCP_KERNEL(64 * WGM * WGN) gemm_glds_kernel(GemmParams p) {
    static_assert(!LORA || (MODE == 0 && WGN == 2), "fused LoRA: plain GEMM, two waves across N");
    static_assert(NSTAGE == 2 || NSTAGE == 3, "2-stage (barrier drains the DMA) or 3-stage (counted vmcnt) pipeline");
    constexpr int NT = 64 * WGM * WGN;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int RPP = NT / 8;
    constexpr int A_IT = (BM + RPP - 1) / RPP, B_IT = (BN + RPP - 1) / RPP;
    static_assert(BM % 8 == 0 && BN % 8 == 0, "a wave stages 8 whole rows per DMA instruction");
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, L_ELEMS = LORA ? 32 * BK : 0;
    constexpr int BUF_ELEMS = A_ELEMS + B_ELEMS + L_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kc = tid & 7, lrow = tid >> 3;

    const int nk1 = (p.K + BK - 1) / BK;
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split; if (kt_end > nk1) kt_end = nk1;
    const bool last_split = split == p.nsplit - 1;
    const int nk2 = last_split ? (p.K2 + BK - 1) / BK : 0;
    const int nprim = kt_end - kt_begin;
    const int nk = nprim + nk2;

    int a_pix[A_IT], a_msk[A_IT];
    const int Ctot = p.cv.C1 + p.cv.C2;
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int m = m0 + lrow + RPP * i;
            a_pix[i] = -1; a_msk[i] = 0;
            if (lrow + RPP * i < BM && m < p.M) {
                int hw = p.cv.Ho * p.cv.Wo;
                int b = m / hw; int rem = m - b * hw;
                int py = rem / p.cv.Wo; int px = rem - py * p.cv.Wo;
                if (!FAST) {
                    a_pix[i] = (b << 20) | (py << 10) | px;
                } else {
                    const int s = MODE == 1 ? p.cv.stride : 1;
                    a_pix[i] = (b * p.cv.Hs + py * s) * p.cv.Ws + px * s;
                    int msk = 0;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            int sy = MODE == 1 ? py * s + ky - p.cv.pad : py + 1 - ky;
                            int sx = MODE == 1 ? px * s + kx - p.cv.pad : px + 1 - kx;
                            if (sy >= 0 && sy < p.cv.Hs && sx >= 0 && sx < p.cv.Ws) msk |= 1 << (ky * 3 + kx);
                        }
                    a_msk[i] = msk;
                }
            }
        }
    }
    const hcp_bf16* zero = (const hcp_bf16*)g_zero_page;

    auto issue_tile = [&](int t, int buf) {
        hcp_bf16* la = lds + buf * BUF_ELEMS;
        hcp_bf16* lb = la + A_ELEMS;
        const bool ext = t >= nprim;
        const int kt = ext ? t - nprim : kt_begin + t;
        const int klim = ext ? p.K2 : p.K;
        // ---- B rows
        {
            const hcp_bf16* Bp = ext ? p.B2 : p.B; const int ld = ext ? p.ldb2 : p.ldb;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                const int r = lrow + RPP * i;
                if (wave * 8 + RPP * i < BN) {                       // wave-uniform: this wave's 8 rows exist in the tile
                    const int k = kt * BK + ((kc ^ ((r >> 1) & 7)) << 3);
                    const int n = n0 + r;
                    const hcp_bf16* src = (n < p.N && k < klim) ? Bp + (size_t)n * ld + k : zero;
                    hcp_glds16(src, lb + (wave * 8 + RPP * i) * BK);
                }
            }
        }
        if (LORA && wave < 4) {
            const int r = lrow;                                      // 32 rows of L, one DMA instruction per wave 0..3
            const int k = kt * BK + ((kc ^ ((r >> 1) & 7)) << 3);
            const hcp_bf16* src = k < klim ? p.L + (size_t)r * p.K + k : zero;
            hcp_glds16(src, lb + B_ELEMS + (wave * 8) * BK);
        }
        // ---- A rows
        if (MODE == 0 || ext) {
            const hcp_bf16* Ap = ext ? p.A2 : p.A; const int ld = ext ? p.lda2 : p.lda;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int r = lrow + RPP * i;
                if (wave * 8 + RPP * i < BM) {
                    const int k = kt * BK + ((kc ^ ((r >> 1) & 7)) << 3);
                    const int m = m0 + r;
                    const hcp_bf16* src = (m < p.M && k < klim) ? Ap + (size_t)m * ld + k : zero;
                    hcp_glds16(src, la + (wave * 8 + RPP * i) * BK);
                }
            }
        } else if (FAST) {
            const int k0 = kt * BK;
            const int tap = k0 / Ctot; const int cb = k0 - tap * Ctot;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int doff = MODE == 1 ? (ky - p.cv.pad) * p.cv.Ws + (kx - p.cv.pad) : (1 - ky) * p.cv.Ws + (1 - kx);
            const hcp_bf16* srcb; int cs, cbase;
            if (cb < p.cv.C1) { srcb = p.cv.X1; cs = p.cv.C1; cbase = cb; }
            else { srcb = p.cv.X2; cs = p.cv.C2; cbase = cb - p.cv.C1; }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int r = lrow + RPP * i;
                if (wave * 8 + RPP * i < BM) {
                    const int c = cbase + ((kc ^ ((r >> 1) & 7)) << 3);
                    const hcp_bf16* src = ((a_msk[i] >> tap) & 1) ? srcb + (size_t)(a_pix[i] + doff) * cs + c : zero;
                    hcp_glds16(src, la + (wave * 8 + RPP * i) * BK);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                const int r = lrow + RPP * i;
                if (wave * 8 + RPP * i < BM) {
                    const int k = kt * BK + ((kc ^ ((r >> 1) & 7)) << 3);
                    int tap = k / Ctot; int ci = k - tap * Ctot;
                    int ky = tap / 3, kx = tap - ky * 3;
                    const hcp_bf16* srcb; int cs, c;
                    if (ci < p.cv.C1) { srcb = p.cv.X1; cs = p.cv.C1; c = ci; }
                    else { srcb = p.cv.X2; cs = p.cv.C2; c = ci - p.cv.C1; }
                    const hcp_bf16* src = zero;
                    int pix = a_pix[i];
                    if (pix >= 0 && k < klim) {
                        int b = pix >> 20, py = (pix >> 10) & 1023, px = pix & 1023;
                        int sy, sx; bool ok;
                        if (MODE == 1) {
                            sy = py * p.cv.stride + ky - p.cv.pad; sx = px * p.cv.stride + kx - p.cv.pad;
                            int He = p.cv.Hs << p.cv.up, We = p.cv.Ws << p.cv.up;
                            ok = sy >= 0 && sy < He && sx >= 0 && sx < We;
                            sy >>= p.cv.up; sx >>= p.cv.up;
                        } else {
                            int ty = py + 1 - ky, tx = px + 1 - kx;
                            ok = ty >= 0 && tx >= 0;
                            if (p.cv.stride == 2) { ok = ok && ((ty | tx) & 1) == 0; ty >>= 1; tx >>= 1; }
                            ok = ok && ty < p.cv.Hs && tx < p.cv.Ws;
                            sy = ty; sx = tx;
                        }
                        if (ok) src = srcb + ((size_t)(b * p.cv.Hs + sy) * p.cv.Ws + sx) * cs + c;
                    }
                    hcp_glds16(src, la + (wave * 8 + RPP * i) * BK);
                }
            }
        }
    };

    hcp_f32x4 acc[TM][TN];
    hcp_f32x4 tacc[LORA ? TM : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
#pragma unroll
    for (int i = 0; i < (LORA ? TM : 1); ++i) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; tacc[i] = z; }

    // DMA instructions this wave issues per K tile (wave-uniform): the count the 3-stage pipeline leaves in flight
    int dma_per_tile = 0;
#pragma unroll
    for (int i = 0; i < B_IT; ++i) dma_per_tile += (wave * 8 + RPP * i < BN) ? 1 : 0;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) dma_per_tile += (wave * 8 + RPP * i < BM) ? 1 : 0;
    if (LORA && wave < 4) dma_per_tile += 1;

    const int fr = lane & 15, fg = lane >> 4;
    auto compute_tile = [&](int stage) {
        const hcp_bf16* la = lds + stage * BUF_ELEMS;
        const hcp_bf16* lb = la + A_ELEMS;
        // (issuing both k-steps' LDS reads ahead of the MFMAs was measured: no gain, +40 VGPRs)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int q = ks * 4 + fg;
            hcp_bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int R = wm * WTM + i * 16 + fr;
                fa[i] = *(const hcp_bf16x8*)(la + R * BK + ((q ^ ((R >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int R = wn * WTN + j * 16 + fr;
                fb[j] = *(const hcp_bf16x8*)(lb + R * BK + ((q ^ ((R >> 1) & 7)) << 3));
            }
            if (p.dbg & 2) {                              // ablation: keep the LDS reads alive, drop the matrix work
#pragma unroll
                for (int i = 0; i < TM; ++i) acc[i][0][0] += (float)fa[i][0];
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[0][j][1] += (float)fb[j][0];
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fb[j], fa[i], acc[i][j]);
            }
            if (LORA) {
                const int R = wn * 16 + fr;
                hcp_bf16x8 fl = *(const hcp_bf16x8*)(lb + B_ELEMS + R * BK + ((q ^ ((R >> 1) & 7)) << 3));
#pragma unroll
                for (int i = 0; i < TM; ++i) tacc[i] = hcp_mfma16(fl, fa[i], tacc[i]);
            }
        }
    };

    if (NSTAGE == 2) {
        if (nk > 0) issue_tile(0, 0);
        HCP_SYNC();
        for (int t = 0; t < nk; ++t) {
            const int cur = t & 1;
            if (t + 1 < nk && !(p.dbg & 1)) issue_tile(t + 1, cur ^ 1);
            if (!(p.dbg & 4)) compute_tile(cur);
            HCP_SYNC();                                  // drains the DMA of tile t+1 (vmcnt(0)) and fences the LDS reads
        }
    } else {
        // tile t+2 is in flight while tile t is multiplied; a barrier never waits for more than tile t+1
        if (nk > 0) issue_tile(0, 0);
        if (nk > 1) issue_tile(1, 1);
        hcp_wait_vmcnt(nk > 1 ? dma_per_tile : 0);
        hcp_barrier_keep_dma();
        int st = 0;
        for (int t = 0; t < nk; ++t) {
            const bool more = t + 2 < nk;
            if (more) issue_tile(t + 2, st == 0 ? 2 : st - 1);     // stage (t+2) % 3
            compute_tile(st);
            hcp_wait_vmcnt(more ? dma_per_tile : 0);                // own share of tile t+1 has landed
            hcp_barrier_keep_dma();                                 // ... everyone's has; stage t%3 is free again
            st = st == 2 ? 0 : st + 1;
        }
    }

    if (LORA) {
        constexpr int TS2 = 40;
        hcp_bf16* lt = lds;
        hcp_bf16* le = lds + BM * TS2;
        hcp_bf16* lt2 = le + BN * TS2;                      // split T (p.ldt == 64): the residual image T_lo
        const bool split = p.ldt == 64;
        const int ldt = split ? 64 : 32;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            hcp_bf16x4 o, o2;
            lora_t_split(tacc[i], o, o2);
            const int ml = wm * WTM + i * 16 + fr;
            *(hcp_bf16x4*)(lt + ml * TS2 + wn * 16 + 4 * fg) = o;
            if (split) *(hcp_bf16x4*)(lt2 + ml * TS2 + wn * 16 + 4 * fg) = o2;
            if (tile_n == 0 && p.Tout && m0 + ml < p.M) {
                *(hcp_bf16x4*)(p.Tout + (size_t)(m0 + ml) * ldt + wn * 16 + 4 * fg) = o;
                if (split) *(hcp_bf16x4*)(p.Tout + (size_t)(m0 + ml) * ldt + 32 + wn * 16 + 4 * fg) = o2;
            }
        }
        for (int c = tid; c < BN * 4; c += NT) {
            const int r = c >> 2, q = c & 3;
            hcp_bf16x8 v = hcp_zero8();
            if (n0 + r < p.N) v = *(const hcp_bf16x8*)(p.E + (size_t)(n0 + r) * 32 + q * 8);
            *(hcp_bf16x8*)(le + r * TS2 + q * 8) = v;
        }
        HCP_SYNC();
        hcp_bf16x8 ft[TM], fe[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) ft[i] = *(const hcp_bf16x8*)(lt + (wm * WTM + i * 16 + fr) * TS2 + fg * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) fe[j] = *(const hcp_bf16x8*)(le + (wn * WTN + j * 16 + fr) * TS2 + fg * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fe[j], ft[i], acc[i][j]);
        if (split) {
#pragma unroll
            for (int i = 0; i < TM; ++i) ft[i] = *(const hcp_bf16x8*)(lt2 + (wm * WTM + i * 16 + fr) * TS2 + fg * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fe[j], ft[i], acc[i][j]);
        }
    }

#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + 4 * fg;
            if (n >= p.N) continue;
            if (p.nsplit > 1) *(hcp_f32x4*)(p.slabs + ((size_t)split * p.M + m) * p.N + n) = acc[i][j];
            else epilogue_store(p, m, n, acc[i][j]);
        }
    }
}

// ---- v2 main loop ------------------------------------------------------------------------------------------------------
// Same tiles, LDS image and MFMA schedule as gemm_glds_kernel, with the per-K-tile overhead stripped out (the ISA of the loop
// above spends ~250 scalar/vector instructions, three in-loop kernarg loads and nine s_waitcnt per 20 MFMAs on address
// arithmetic and masking):
//   * sources are addressed through BUFFER resources (buffer_load_dwordx4 ... offen lds): the wave-uniform resource base is
//     re-pointed at each K tile (a few SALU ops), every lane's byte offset is LOOP-INVARIANT, and a masked row / conv tap is
//     an out-of-range offset that the hardware turns into zeros — no zero page, no exec-mask branches, no 64-bit VALU math;
//   * LDS fragment addresses are two precomputed VGPRs per operand (the XOR swizzle does not depend on the 16-row block);
//   * the rank-32 K-extension tile and the epilogue are outside the loop; no ablation hooks.
// Requirements (else the dispatcher keeps the kernel above): K % 64 == 0; conv gathers in their FAST form.
// NLD > 0: wave specialisation.  NLD extra "loader" waves issue ALL of the tile's LDS-DMA instructions; the WGM x WGN compute waves
// only read fragments and issue MFMAs.  An LDS-DMA instruction costs its issuing wave ~100+ cycles of in-order issue time (the
// per-CU global->LDS path moves ~40-60 B/clk), so with every wave loading AND computing, each K tile serialises
// [5 DMA issues] -> [LDS reads + 20 MFMAs] inside every wave (measured: DMA ~14 us + LDS ~10 us + MFMA ~12 us of a 44 us
// convolution, back to back).  With loaders the two phases belong to different waves and overlap inside ONE workgroup —
// which is what shapes with a single workgroup per CU (grid <= 256, the common case at 64x64 resolution) need.
// NST (loader-wave variant only): depth of the LDS ring.  The loaders keep NST-1 K tiles in flight (counted vmcnt waits, barriers
// that do not drain the DMA queue), so a short-K problem is no longer a chain of one memory latency per K tile: the bytes in
// flight per CU are what hides the latency, and with one workgroup per CU only a deeper ring raises them.
template <int BM, int BN, int WGM, int WGN, int MODE, bool LORA = false, int NLD = 0, int NST = 2>
HCP_KERNEL(64 * (WGM * WGN + NLD)) gemm_v2_kernel(GemmParams p) {
    static_assert(!LORA || (MODE == 0 && WGN == 2), "fused LoRA: plain GEMM, two waves across N");
    static_assert(NST == 2 || NLD > 0, "deeper rings need the loader waves (nobody else may hold DMA across a barrier)");
    constexpr int NC = WGM * WGN;                       // compute waves
    constexpr int NT = 64 * (NLD ? NLD : NC);           // threads that issue DMA: the loaders, or everybody
    constexpr int NTC = 64 * NC;                        // compute threads
    constexpr int WTM = BM / WGM, WTN = BN / WGN;
    constexpr int TM = WTM / 16, TN = WTN / 16;
    constexpr int RPP = NT / 8;
    constexpr int A_IT = (BM + RPP - 1) / RPP, B_IT = (BN + RPP - 1) / RPP;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0 && BM % 8 == 0 && BN % 8 == 0, "tile shape");
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, L_ELEMS = LORA ? 32 * BK : 0, BUF_ELEMS = A_ELEMS + B_ELEMS + L_ELEMS;

    const int tid_all = threadIdx.x;
    const int lane = tid_all & 63;
    const int wave_all = hcp_uniform(tid_all >> 6);
    const bool is_loader = NLD > 0 && wave_all >= NC;
    // DMA geometry is indexed by the issuing thread: loader-local when loaders exist
    const int tid = NLD ? (is_loader ? tid_all - NTC : tid_all % NT) : tid_all;
    const int wave = NLD ? (is_loader ? wave_all - NC : 0) : wave_all;
    const int cwave = is_loader ? 0 : wave_all;         // compute-wave index (tile position)
    const int wm = cwave / WGN, wn = cwave % WGN;
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int split = blockIdx.y;
    const int kc = tid & 7, lrow = tid >> 3;
    const int fr = lane & 15, fg = lane >> 4;

    const int nk1 = p.K / BK;
    const int kt_begin = split * p.kt_per_split;
    int kt_end = kt_begin + p.kt_per_split; if (kt_end > nk1) kt_end = nk1;
    const int nprim = kt_end - kt_begin;
    const bool has_ext = (split == p.nsplit - 1) && p.K2 > 0;
    const int nk = nprim + (has_ext ? 1 : 0);

    // ---- loop-invariant per-lane byte offsets (HCP_BUF_OOB = this lane contributes zeros)
    // Register diet of the conv gathers (every VGPR over the occupancy step became a scratch reload INSIDE the loop, and an
    // ordinary load behind the tile's LDS-DMA makes its s_waitcnt drain the DMA queue too — guide §5 trap (b): the phases
    // then run back to back): one pixel index per row instead of two byte offsets, all 3x3 validity masks in one register
    // (9 bits per row), one shared 16-byte chunk term (the swizzle term (r >> 1) & 7 is the same for rows RPP apart).
    const int Ctot = p.cv.C1 + p.cv.C2;
    static_assert(MODE == 0 || RPP % 16 == 0, "conv gather: rows RPP apart share their swizzle term");
    unsigned va[A_IT], vb[B_IT];                          // MODE 0: byte offset of the row; conv: pixel index of the row (or ~0u)
    unsigned a_msk[(A_IT + 2) / 3];                       // conv: word i/3, bit 9*(i%3) + tap = tap valid for row i
#pragma unroll
    for (int i = 0; i < (A_IT + 2) / 3; ++i) a_msk[i] = 0;
    const unsigned a_chunk = (unsigned)(((kc ^ ((lrow >> 1) & 7)) << 3) * 2);
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
        const int r = lrow + RPP * i, nl = n0 + r, n = geglu_col(p, BN, nl);
        vb[i] = (r < BN && nl < p.N) ? (unsigned)(((size_t)n * p.ldb + ((kc ^ ((r >> 1) & 7)) << 3)) * 2) : HCP_BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
        const int r = lrow + RPP * i, m = m0 + r;
        const int chunk = (kc ^ ((r >> 1) & 7)) << 3;
        va[i] = HCP_BUF_OOB;
        if (r < BM && m < p.M) {
            if (MODE == 0) {
                va[i] = (unsigned)(((size_t)m * p.lda + chunk) * 2);
            } else {
                const int hw = p.cv.Ho * p.cv.Wo;
                const int b = m / hw; const int rem = m - b * hw;
                const int py = rem / p.cv.Wo, px = rem - py * p.cv.Wo;
                const int s = MODE == 1 ? p.cv.stride : 1;
                const int pix = (b * p.cv.Hs + py * s) * p.cv.Ws + px * s;
                va[i] = (unsigned)pix;
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
    // conv: (tap, channel cursor) of the NEXT tile to issue; advanced by one K tile per issue
    int tap = 0, cb = 0;
    if (MODE != 0) { const int k0 = kt_begin * BK; tap = k0 / Ctot; cb = k0 - tap * Ctot; }
    // conv with ONE source tensor (every convolution but the up blocks' concat inputs): the row's byte offset pixel * 2 C + chunk is a loop
    // invariant — kept in va[] itself — and an invalid tap ORs bit 31 into it (>= num_records: zeros); the per-tile form below (bit test,
    // compare, 32-bit multiply, add, select per row) is VALU work in front of this wave's own MFMAs (gemm_pp.hip, LAB_NOTEBOOK round 6).
    // (no second array for the concat case: a select between two register arrays sent both to scratch)
    const bool pre_off = MODE != 0 && p.cv.C2 == 0;
    if (MODE != 0 && pre_off) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) va[i] = va[i] != HCP_BUF_OOB ? va[i] * (unsigned)(2 * p.cv.C1) + a_chunk : HCP_BUF_OOB;
#pragma unroll
        for (int i = 0; i < (A_IT + 2) / 3; ++i) a_msk[i] = ~a_msk[i];
    }
    const hcp_bf16* Ab = p.A + (size_t)kt_begin * BK;      // MODE 0: first element of this split's first A tile column block
    const hcp_bf16* Bb = p.B + (size_t)kt_begin * BK;
    const hcp_bf16* Lb = LORA ? p.L + (size_t)kt_begin * BK : nullptr;      // fused LoRA: 32 rows of L [32, K], staged by waves 0..3
    const unsigned vl = (unsigned)(((size_t)lrow * p.K + ((kc ^ ((lrow >> 1) & 7)) << 3)) * 2);

    auto issue = [&](int buf) {                             // issues the next primary tile (called in tile order)
        hcp_bf16* la = lds + buf * BUF_ELEMS;
        hcp_bf16* lb = la + A_ELEMS;
        const hcp_rsrc rb = hcp_make_rsrc(Bb);
        Bb += BK;
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if ((i + 1) * RPP <= BN || wave * 8 + RPP * i < BN) hcp_buf_glds16(rb, vb[i], lb + (wave * 8 + RPP * i) * BK);
        if (LORA) {
            const hcp_rsrc rl = hcp_make_rsrc(Lb);
            Lb += BK;
            if (wave < 4) hcp_buf_glds16(rl, vl, lb + B_ELEMS + (wave * 8) * BK);
        }
        if (MODE == 0) {
            const hcp_rsrc ra = hcp_make_rsrc(Ab);
            Ab += BK;
#pragma unroll
            for (int i = 0; i < A_IT; ++i)
                if ((i + 1) * RPP <= BM || wave * 8 + RPP * i < BM) hcp_buf_glds16(ra, va[i], la + (wave * 8 + RPP * i) * BK);
        } else {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int doff = MODE == 1 ? (ky - p.cv.pad) * p.cv.Ws + (kx - p.cv.pad) : (1 - ky) * p.cv.Ws + (1 - kx);
            const bool first = cb < p.cv.C1;
            const hcp_bf16* base = first ? p.cv.X1 + (long)doff * p.cv.C1 + cb : p.cv.X2 + (long)doff * p.cv.C2 + (cb - p.cv.C1);
            const hcp_rsrc ra = hcp_make_rsrc(base);
            if (pre_off) {
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    if ((i + 1) * RPP <= BM || wave * 8 + RPP * i < BM)
                        hcp_buf_glds16(ra, va[i] | (((a_msk[i / 3] >> (9 * (i % 3) + tap)) & 1u) << 31), la + (wave * 8 + RPP * i) * BK);
            } else {
#pragma unroll
                for (int i = 0; i < A_IT; ++i)
                    if ((i + 1) * RPP <= BM || wave * 8 + RPP * i < BM) {
                        const unsigned v = ((a_msk[i / 3] >> (9 * (i % 3) + tap)) & 1) ? va[i] * (unsigned)(2 * (first ? p.cv.C1 : p.cv.C2)) + a_chunk : HCP_BUF_OOB;
                        hcp_buf_glds16(ra, v, la + (wave * 8 + RPP * i) * BK);
                    }
            }
            cb += BK;
            if (cb >= Ctot) { cb -= Ctot; ++tap; }
        }
    };
    auto issue_ext = [&](int buf) {                         // the rank-32 K-extension tile: plain rows of A2 / B2, k < K2 only
        hcp_bf16* la = lds + buf * BUF_ELEMS;
        hcp_bf16* lb = la + A_ELEMS;
        const hcp_rsrc rb = hcp_make_rsrc(p.B2), ra = hcp_make_rsrc(p.A2);
#pragma unroll
        for (int i = 0; i < B_IT; ++i)
            if ((i + 1) * RPP <= BN || wave * 8 + RPP * i < BN) {
                const int r = lrow + RPP * i, nl = n0 + r, n = geglu_col(p, BN, nl), k = (kc ^ ((r >> 1) & 7)) << 3;
                hcp_buf_glds16(rb, (nl < p.N && k < p.K2) ? (unsigned)(((size_t)n * p.ldb2 + k) * 2) : HCP_BUF_OOB, lb + (wave * 8 + RPP * i) * BK);
            }
#pragma unroll
        for (int i = 0; i < A_IT; ++i)
            if ((i + 1) * RPP <= BM || wave * 8 + RPP * i < BM) {
                const int r = lrow + RPP * i, m = m0 + r, k = (kc ^ ((r >> 1) & 7)) << 3;
                hcp_buf_glds16(ra, (m < p.M && k < p.K2) ? (unsigned)(((size_t)m * p.lda2 + k) * 2) : HCP_BUF_OOB, la + (wave * 8 + RPP * i) * BK);
            }
    };

    hcp_f32x4 acc[TM][TN];
    hcp_f32x4 tacc[LORA ? TM : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }
#pragma unroll
    for (int i = 0; i < (LORA ? TM : 1); ++i) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; tacc[i] = z; }

    // fragment addresses (elements) inside a stage: row R = base + 16*i, slot (ks*4 + fg) ^ ((R >> 1) & 7) — the swizzle term only
    // depends on fr because every 16-row block starts at a multiple of 16
    // (slot of k-step 1 = slot of k-step 0 XOR 4, i.e. element offset XOR 32: one register per operand, the other is one v_xor away)
    const int sw0 = (fg ^ ((fr >> 1) & 7)) << 3;
    const int a_rd0 = (wm * WTM + fr) * BK + sw0;
    const int b_rd0 = A_ELEMS + (wn * WTN + fr) * BK + sw0;
    const int l_rd0 = A_ELEMS + B_ELEMS + (wn * 16 + fr) * BK + sw0;
    auto compute = [&](int stage) {
        const hcp_bf16* st = lds + stage * BUF_ELEMS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            hcp_bf16x8 fa[TM], fb[TN];
            const int a_rd = a_rd0 ^ (ks * 32), b_rd = b_rd0 ^ (ks * 32);
#if defined(HCP_TOOLS)
            if (p.dbg & 32) {                               // ablation: no LDS reads (operands = loop-invariant registers)
#pragma unroll
                for (int i = 0; i < TM; ++i) { hcp_f32x4 z = {(float)tid_all, (float)i, 1.f, 2.f}; fa[i] = __builtin_bit_cast(hcp_bf16x8, z); }
#pragma unroll
                for (int j = 0; j < TN; ++j) { hcp_f32x4 z = {(float)lane, (float)j, 3.f, 4.f}; fb[j] = __builtin_bit_cast(hcp_bf16x8, z); }
            } else
#endif
            {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *(const hcp_bf16x8*)(st + a_rd + i * 16 * BK);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *(const hcp_bf16x8*)(st + b_rd + j * 16 * BK);
            }
#if defined(HCP_TOOLS)
            if (p.dbg & 16) {                               // ablation: no MFMA (keep the fragment reads alive)
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(fa[i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(fb[j]));
                continue;
            }
#endif
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fb[j], fa[i], acc[i][j]);
            if (LORA) {                                    // this wave's 16 of the 32 rank slots: T += A L^T from the same A fragments
                const hcp_bf16x8 fl = *(const hcp_bf16x8*)(st + (l_rd0 ^ (ks * 32)));
#pragma unroll
                for (int i = 0; i < TM; ++i) tacc[i] = hcp_mfma16(fl, fa[i], tacc[i]);
            }
        }
    };

    // Loader-wave variant: everything the tail needs is requested BEFORE the main loop instead of after it (each was one more
    // exposed memory latency on a kernel that is a chain of them): the compute waves fetch their bias / residual fragments now
    // (their wait folds into the wait for K tile 0), the loaders DMA the tile's E rows into their own LDS area.
    constexpr bool EARLY = NLD > 0;
    constexpr int ES = EARLY ? 32 : 40;                     // row stride (elements) of the E image: DMA rows are unpadded
    hcp_bf16* const le_early = lds + NST * BUF_ELEMS;
    hcp_f32x4 bias_v[TN];
    hcp_bf16x4 res_v[TM][TN];
    auto load_epilogue_operands = [&]() {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nl = n0 + wn * WTN + j * 16 + 4 * fg, n = geglu_col(p, BN, nl);
            hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
            bias_v[j] = (p.bias && nl < p.N) ? *(const hcp_f32x4*)(p.bias + n) : z;
        }
        if (p.residual && !p.epi_tile) {                    // (the tile epilogue reads the residual in its own 16-byte layout)
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wm * WTM + i * 16 + fr;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int n = n0 + wn * WTN + j * 16 + 4 * fg;
                    hcp_bf16x4 z = {0, 0, 0, 0};
                    res_v[i][j] = (m < p.M && n < p.N) ? *(const hcp_bf16x4*)(p.residual + (size_t)m * p.ldr + n) : z;
                }
            }
        }
    };
    if (EARLY && !is_loader && p.nsplit == 1) load_epilogue_operands();

    if (NLD > 0 && is_loader) {
        if (LORA) {                                         // E rows n0 .. n0+BN, 64 bytes each: 16 rows per DMA instruction
            const hcp_rsrc re = hcp_make_rsrc(p.E);
            for (int g = wave; g < BN / 16; g += NLD) {
                const int nl = n0 + g * 16 + (lane >> 2), n = geglu_col(p, BN, nl);
                hcp_buf_glds16(re, nl < p.N ? (unsigned)(((size_t)n * 32 + (lane & 3) * 8) * 2) : HCP_BUF_OOB, le_early + g * 16 * 32);
            }
        }
        // every loader wave issues the same IPT instructions per tile (whole 8-row groups: static_assert below), so "tile t+1
        // has landed" = at most IPT * (tiles issued after it) of this wave's loads are still in flight (loads return in order)
        static_assert(NLD == 0 || (BM % RPP == 0 && BN % RPP == 0), "loader waves: whole row groups");
        constexpr int IPT = A_IT + B_IT + (LORA ? 1 : 0);
        int issued = 0, wbuf = 0;
        auto issue_next = [&]() {
            if (issued < nprim) issue(wbuf); else issue_ext(wbuf);
            ++issued; wbuf = wbuf + 1 == NST ? 0 : wbuf + 1;
        };
        auto wait_newer = [&](int newer) {                  // newer = tiles issued after the one that must be complete
            if (NST > 3 && newer >= 2) hcp_wait_vmcnt_c<2 * IPT>();
            else if (NST > 2 && newer >= 1) hcp_wait_vmcnt_c<IPT>();
            else hcp_wait_vmcnt_c<0>();
        };
        for (int i = 0; i < NST - 1 && issued < nk; ++i) issue_next();
        wait_newer(issued - 1);
        hcp_barrier_keep_dma();                             // tile 0 is in LDS
        for (int t = 0; t < nk; ++t) {
            // ring slot of tile t+NST-1 = slot of tile t-1: its readers passed the barrier that ended iteration t-1
#if defined(HCP_TOOLS)
            if (p.dbg & 8) issued = nk;                     // ablation: no DMA after the prologue (results are wrong)
#endif
            if (issued < nk) issue_next();
            wait_newer(issued - (t + 2));
            hcp_barrier_keep_dma();                         // tile t+1 is in LDS, tile t is consumed
        }
        if (LORA) HCP_SYNC();                               // the compute waves' epilogue barrier
        if ((p.geglu_hg || p.geglu_out || p.epi_tile) && p.nsplit == 1) { if (LORA) HCP_SYNC(); HCP_SYNC(); }       // ... and those of the GEGLU / epilogue tiles
        return;
    }
    if (NLD == 0) { if (nprim > 0) issue(0); else if (has_ext) issue_ext(0); }
    HCP_SYNC();
    for (int t = 0, cur = 0; t < nk; ++t) {
        if (NLD == 0) {
            if (t + 1 < nprim) issue(cur ^ 1);
            else if (t + 1 == nprim && has_ext) issue_ext(cur ^ 1);
        }
        compute(cur);
        HCP_SYNC();                                         // drains the DMA of tile t+1 and fences the LDS reads of tile t
        cur = cur + 1 == NST ? 0 : cur + 1;
    }

    if (LORA) {
        // T (bf16-rounded) and E = alpha * W_up rows of this N tile meet in LDS; one extra k-step adds T E^T
        constexpr int TS2 = 40;
        hcp_bf16* lt = lds;
        hcp_bf16* le = EARLY ? le_early : lds + BM * TS2;
        hcp_bf16* lt2 = lds + BM * TS2 + (EARLY ? 0 : BN * TS2);   // split T (p.ldt == 64): the residual image T_lo
        const bool split = p.ldt == 64;
        const int ldt = split ? 64 : 32;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            hcp_bf16x4 o, o2;
            lora_t_split(tacc[i], o, o2);
            const int ml = wm * WTM + i * 16 + fr;
            *(hcp_bf16x4*)(lt + ml * TS2 + wn * 16 + 4 * fg) = o;
            if (split) *(hcp_bf16x4*)(lt2 + ml * TS2 + wn * 16 + 4 * fg) = o2;
            if (tile_n == 0 && p.Tout && m0 + ml < p.M) {
                *(hcp_bf16x4*)(p.Tout + (size_t)(m0 + ml) * ldt + wn * 16 + 4 * fg) = o;
                if (split) *(hcp_bf16x4*)(p.Tout + (size_t)(m0 + ml) * ldt + 32 + wn * 16 + 4 * fg) = o2;
            }
        }
        if (!EARLY) {
            for (int c = tid_all; c < BN * 4; c += NTC) {
                const int r = c >> 2, q = c & 3;
                hcp_bf16x8 v = hcp_zero8();
                if (n0 + r < p.N) v = *(const hcp_bf16x8*)(p.E + (size_t)geglu_col(p, BN, n0 + r) * 32 + q * 8);
                *(hcp_bf16x8*)(le + r * TS2 + q * 8) = v;
            }
        }
        HCP_SYNC();
        hcp_bf16x8 ft[TM], fe[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) ft[i] = *(const hcp_bf16x8*)(lt + (wm * WTM + i * 16 + fr) * TS2 + fg * 8);
#pragma unroll
        for (int j = 0; j < TN; ++j) fe[j] = *(const hcp_bf16x8*)(le + (wn * WTN + j * 16 + fr) * ES + fg * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fe[j], ft[i], acc[i][j]);
        if (split) {
#pragma unroll
            for (int i = 0; i < TM; ++i) ft[i] = *(const hcp_bf16x8*)(lt2 + (wm * WTM + i * 16 + fr) * TS2 + fg * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = hcp_mfma16(fe[j], ft[i], acc[i][j]);
        }
    }

    if (p.nsplit > 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + 4 * fg;
                if (n < p.N) *(hcp_f32x4*)(p.slabs + ((size_t)split * p.M + m) * p.N + n) = acc[i][j];
            }
        }
        return;
    }
    if (p.geglu_hg) {                                       // GEGLU-backward epilogue through LDS (gemm_params.h: geglu_tile_*)
        static_assert((size_t)BM * geglu_tile_ld(BN) <= (size_t)2 * BUF_ELEMS, "the product tile fits the ring");
        if (LORA) HCP_SYNC();                               // the LoRA tail's T / E images live in the same LDS
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) geglu_tile_put(lds, geglu_tile_ld(BN), wm * WTM + i * 16 + fr, wn * WTN + j * 16 + 4 * fg, acc[i][j], p.alpha);
        HCP_SYNC();
        geglu_tile_apply<BM, BN, NTC>(p, lds, m0, n0, tid_all);
        return;
    }
    if (p.geglu_out) {                                      // GEGLU-forward epilogue (gemm_params.h: geglu_out): D = bf16(h | g), geglu_out = bf16(h gelu(g))
        static_assert((size_t)BM * BN * 2 <= (size_t)2 * BUF_ELEMS * sizeof(hcp_bf16), "the gelu(g) tile fits the ring");
        if (!EARLY) load_epilogue_operands();
        if (LORA) HCP_SYNC();                               // the LoRA tail's T / E images live in the same LDS
        hcp_f32x4 v[TM][TN];
        int rows[TM], cols[TN];
        constexpr int HW = WGN / 2;                         // wave columns per half: waves wn < HW hold h, the others the matching g
        const bool is_g = wn >= HW;
#pragma unroll
        for (int j = 0; j < TN; ++j) cols[j] = tile_n * (BN / 2) + (wn - (is_g ? HW : 0)) * WTN + j * 16 + 4 * fg;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int m = m0 + wm * WTM + i * 16 + fr;
            rows[i] = m;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                v[i][j] = acc[i][j] * p.alpha + bias_v[j];
                if (m < p.M) store_hi_lo(p, m, cols[j] + (is_g ? (p.N >> 1) : 0), v[i][j]);
            }
        }
        geglu_fwd_pair<TM, TN>(v, is_g, wm * HW + (wn - (is_g ? HW : 0)), lane, (hcp_f32x4*)lds, [] { HCP_SYNC(); }, p.geglu_out, p.N >> 1, rows, p.M, cols);
        return;
    }
    if (p.epi_tile) {                                       // tile epilogue (gemm_params.h: epi_tile_store): 16-byte row pieces
        if (!EARLY) load_epilogue_operands();
        if (LORA) HCP_SYNC();                               // the LoRA tail's T / E images live in the same LDS
        float* const tile = (float*)lds;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int ml = wm * WTM + i * 16 + fr, m = m0 + ml;
            const float* rbp = (p.rowbias && m < p.M) ? p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int nl = wn * WTN + j * 16 + 4 * fg;
                hcp_f32x4 v = acc[i][j] * p.alpha + bias_v[j];
                if (rbp && n0 + nl < p.N) v += *(const hcp_f32x4*)(rbp + n0 + nl);
                *(hcp_f32x4*)(tile + ml * epi_tile_ld(BN) + nl) = v;
            }
        }
        HCP_SYNC();
        epi_tile_store<BM, BN, NTC>(p, tile, m0, n0, tid_all);
        return;
    }
    // epilogue: ALL of the lane's bias / row-bias / residual loads are issued before the first store (one vmcnt wait instead
    // of one per 16x16 block: with K = 320 the serialized form cost as much as the whole main loop)
    if (!EARLY) load_epilogue_operands();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + fr;
        if (m >= p.M) continue;
        hcp_f32x4 rb_v[TN];
        if (p.rowbias) {
            const float* rbp = p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 16 + 4 * fg;
                hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f};
                rb_v[j] = n < p.N ? *(const hcp_f32x4*)(rbp + n) : z;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + 4 * fg;
            if (n >= p.N) continue;
            hcp_f32x4 v = acc[i][j] * p.alpha + bias_v[j];
            if (p.rowbias) v += rb_v[j];
            if (p.residual) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += hcp_bf2f((unsigned short)res_v[i][j][q]);
                add_residual_lo(p, m, n, v);
            }
#if defined(HCP_TOOLS)
            if ((p.dbg & 64) && v[0] != 12345.678f) continue;      // ablation: no output stores (the compare keeps the math alive)
#endif
            if (p.out_f32) *(hcp_f32x4*)((float*)p.D + (size_t)m * p.ldd + n) = v;
            else store_hi_lo(p, m, n, v);
        }
    }
}

// Sum the split-K slabs in split order and apply the epilogue.  One thread per output quad; the quad index is the slab offset / 4 (rows are
// N floats), 32-bit throughout (the slabs of one launch fit the workspace: M * N <= 2^26).  A thread's loads are independent of each
// other, so they are requested in batches — the epilogue's operands with the first one, four slabs at a time after that — instead of one
// round trip to memory per slab (the round-5 form: a load and an `s_waitcnt vmcnt(0)` per split, and a 64-bit division per quad).
HCP_KERNEL(256) splitk_reduce_kernel(GemmParams p) {
    const unsigned nv = (unsigned)p.N / 4;
    const unsigned total = (unsigned)p.M * nv;
    const size_t slab = (size_t)p.M * p.N;
    const int ns = p.nsplit;
    const bool plain = !p.geglu_hg;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const int m = (int)(i / nv), n = (int)(i - (unsigned)m * nv) * 4;
        const float* src = p.slabs + (size_t)i * 4;
        hcp_f32x4 v = *(const hcp_f32x4*)src;
        // (an operand the launch does not have is read from the zero page and not used: no branch, so no wait at a join)
        const void* const zp = g_zero_page;
        const hcp_f32x4 bias = *(const hcp_f32x4*)(plain && p.bias ? (const void*)(p.bias + n) : zp);
        const hcp_f32x4 rowb = *(const hcp_f32x4*)(plain && p.rowbias ? (const void*)(p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld + n) : zp);
        const hcp_bf16x4 res = *(const hcp_bf16x4*)(plain && p.residual ? (const void*)(p.residual + (size_t)m * p.ldr + n) : zp);
        const hcp_bf16x4 res_lo = *(const hcp_bf16x4*)(plain && p.residual && p.residual_lo ? (const void*)(p.residual_lo + (size_t)m * p.ldr + n) : zp);
        int s = 1;
        for (; s + 4 <= ns; s += 4) {
            const hcp_f32x4 a = *(const hcp_f32x4*)(src + (size_t)s * slab), b = *(const hcp_f32x4*)(src + (size_t)(s + 1) * slab);
            const hcp_f32x4 c = *(const hcp_f32x4*)(src + (size_t)(s + 2) * slab), d = *(const hcp_f32x4*)(src + (size_t)(s + 3) * slab);
            v += a; v += b; v += c; v += d;
        }
        if (s + 2 <= ns) {
            const hcp_f32x4 a = *(const hcp_f32x4*)(src + (size_t)s * slab), b = *(const hcp_f32x4*)(src + (size_t)(s + 1) * slab);
            v += a; v += b;
            s += 2;
        }
        if (s < ns) v += *(const hcp_f32x4*)(src + (size_t)s * slab);
        if (!plain) { epilogue_geglu_bwd(p, m, n, v); continue; }
        // epilogue_store's arithmetic, on the operands requested above
        v = v * p.alpha;
        if (p.bias) v += bias;
        if (p.rowbias) v += rowb;
        if (p.residual) {
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += hcp_bf2f((unsigned short)res[q]);
            if (p.residual_lo) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += hcp_bf2f((unsigned short)res_lo[q]);
            }
        }
        if (p.out_f32) *(hcp_f32x4*)((float*)p.D + (size_t)m * p.ldd + n) = v;
        else store_hi_lo(p, m, n, v);
    }
}

HCP_TUNABLE(int, g_force_cfg, -1);   // tools/tune: force a tile configuration (see hcp_debug_set_gemm_config)
HCP_TUNABLE(int, g_dbg_ablate, 0);   // tools only, see GemmParams::dbg
HCP_TUNABLE(int, g_use_glds, 1);     // 1: LDS-DMA main loop (default), 0: register-staged main loop (kept for A/B measurements)
HCP_TUNABLE(int, g_use_v2, 1);       // 1: buffer-addressed v2 main loop where its requirements hold (default), 0: gemm_glds_kernel everywhere

HCP_TUNABLE(int, g_conv_patch, 1);   // tools: 0 = the ping-pong kernel for every convolution, 1 = the rule in try_pp, 2 = conv_patch.hip wherever eligible
HCP_TUNABLE(int, g_epi_tile, -1);    // tools: -1 = the rule below, 0 = lane-layout epilogue everywhere, 1 = tile epilogue wherever it is possible
HCP_TUNABLE(int, g_force_loaders, -1);   // tools: -1 = as dispatched, 0 = no loader waves, 1 / 3 / 4 = loader-wave variant with a 2 / 3 / 4 tile ring

// Tile epilogue (gemm_params.h: epi_tile_store) for this launch?  Needs an unsplit bf16 output in 16-byte pieces.  The rule, from
// tools/lab/epilogue_ab.py on rotating operand sets (profiles/r6_ab_tile_epilogue.txt): it wins where the epilogue READS — a plain GEMM
// (MODE 0) with a residual: to_out / ff.net.2 at 64x64 -5 ... -7 %, neutral at the smaller levels, more with the two extra images of a
// (hi | lo) stream — because the lane layout fetches the residual in 8-byte pieces that a 5-20 tile main loop cannot hide; it LOSES on
// pure stores (ff.net.0.proj at 64x64 +20 %: the LDS round trip and its barrier buy nothing, the 8-byte stores were not the limit) and on
// the convolutions (+5 %: their long K loop already hides the early residual request).  Whole step, tile everywhere: -0.8 % (SD1.5),
// -1.2 % (SDXL).
bool want_epi_tile(const GemmParams& p, int mode) {
    if (p.nsplit != 1 || p.out_f32 || p.geglu_hg || p.geglu_out || p.N % 8 || p.ldd % 8 || (p.residual && p.ldr % 8)) return false;
    if ((((size_t)p.D) | ((size_t)p.D_lo) | ((size_t)p.residual) | ((size_t)p.residual_lo)) & 15) return false;
    if (g_epi_tile >= 0) return g_epi_tile == 1;
    return mode == 0 && p.residual != nullptr;
}

template <int BM, int BN, int WGM, int WGN, int MODE, bool FAST, bool LORA = false, int NSTAGE = 2, int NLD = 0>
int launch_cfg(GemmParams& p, hipStream_t stream) {
    p.tiles_m = hcp_cdiv(p.M, BM);
    const int tiles_n = hcp_cdiv(p.N, BN);
    p.dbg = g_dbg_ablate;
    if constexpr (NSTAGE == 2 && (MODE == 0 || FAST)) {
        if (g_use_v2 && g_use_glds && !(p.dbg & 7) && p.K % BK == 0 && (p.K2 == 0 || p.K2 == 32) &&
            (size_t)p.M * p.lda * 2 < (1ul << 31) && (size_t)p.N * p.ldb * 2 < (1ul << 31)) {
            constexpr size_t stage = (size_t)(BM + BN + (LORA ? 32 : 0)) * BK * sizeof(hcp_bf16);
            constexpr size_t eimg = (LORA && NLD > 0) ? (size_t)BN * 32 * sizeof(hcp_bf16) : 0;   // loader variant: the E rows, behind the ring
            if (p.geglu_out) {                               // GEGLU-forward epilogue: this tile must pair h and g columns (gemm_params.h)
                if (MODE == 0 && WGN % 2 == 0 && p.nsplit == 1 && (p.N / 2) % (BN / 2) == 0) p.geglu_fused = 1;
                else p.geglu_out = nullptr;
            }
            constexpr size_t tile_bytes = (size_t)BM * epi_tile_ld(BN) * sizeof(float);
            p.epi_tile = want_epi_tile(p, MODE) && tile_bytes <= 160 * 1024 - eimg ? 1 : 0;
            size_t smem = 2 * stage;
            const size_t tail = (size_t)(2 * BM + BN) * 40 * sizeof(hcp_bf16);   // fused-LoRA tail images: T_hi, E, T_lo
            if (LORA && smem < tail) smem = tail;
            if (p.epi_tile && smem < tile_bytes) smem = tile_bytes;
            [[maybe_unused]] const dim3 grid(p.tiles_m * tiles_n, p.nsplit);
            if constexpr (NLD > 0) {
                // p.loaders: 1 (or 2) = two-stage ring, 3 / 4 = deeper ring where it fits the 160 KB of LDS
                const size_t tb = p.epi_tile ? tile_bytes : 0;      // the epilogue tile re-uses the ring (the E image sits behind it)
                if (p.loaders == 4 && 4 * stage + eimg <= 160 * 1024) {
                    HCP_LAUNCH((gemm_v2_kernel<BM, BN, WGM, WGN, MODE, LORA, NLD, (4 * stage + eimg <= 160 * 1024 ? 4 : 2)>), grid, dim3(64 * (WGM * WGN + NLD)), (4 * stage > tb ? 4 * stage : tb) + eimg, stream, p);
                } else if (p.loaders >= 3 && 3 * stage + eimg <= 160 * 1024) {
                    HCP_LAUNCH((gemm_v2_kernel<BM, BN, WGM, WGN, MODE, LORA, NLD, (3 * stage + eimg <= 160 * 1024 ? 3 : 2)>), grid, dim3(64 * (WGM * WGN + NLD)), (3 * stage > tb ? 3 * stage : tb) + eimg, stream, p);
                } else if (p.loaders) {
                    HCP_LAUNCH((gemm_v2_kernel<BM, BN, WGM, WGN, MODE, LORA, NLD>), grid, dim3(64 * (WGM * WGN + NLD)), smem + eimg, stream, p);
                } else {
                    HCP_LAUNCH((gemm_v2_kernel<BM, BN, WGM, WGN, MODE, LORA>), grid, dim3(64 * WGM * WGN), smem, stream, p);
                }
            } else {
                HCP_LAUNCH((gemm_v2_kernel<BM, BN, WGM, WGN, MODE, LORA>), dim3(p.tiles_m * tiles_n, p.nsplit), dim3(64 * WGM * WGN), smem, stream, p);
            }
            if (p.nsplit > 1) {
                long nv = (long)p.M * (p.N / 4);
                int g = (int)((nv + 255) / 256); if (g > 2048) g = 2048;
                HCP_LAUNCH(splitk_reduce_kernel, dim3(g), dim3(256), 0, stream, p);
            }
            HCP_LAUNCH_CHECK("gemm_v2_kernel");
        }
    }
    {
        p.epi_tile = 0;
        p.geglu_out = nullptr;                               // (the first LDS-DMA loop has no pairing epilogue: the entry point runs hcp_geglu_fwd behind it)
        size_t smem = (size_t)NSTAGE * (BM + BN + (LORA ? 32 : 0)) * BK * sizeof(hcp_bf16);
        const size_t tail = (size_t)(2 * BM + BN) * 40 * sizeof(hcp_bf16);  // fused-LoRA tail images: T_hi, E, T_lo
        if (LORA && smem < tail) smem = tail;
        HCP_LAUNCH((gemm_glds_kernel<BM, BN, WGM, WGN, MODE, FAST, LORA, NSTAGE>), dim3(p.tiles_m * tiles_n, p.nsplit),
                   dim3(64 * WGM * WGN), smem, stream, p);
    }
    if (p.nsplit > 1) {
        long nv = (long)p.M * (p.N / 4);
        int g = (int)((nv + 255) / 256); if (g > 2048) g = 2048;
        HCP_LAUNCH(splitk_reduce_kernel, dim3(g), dim3(256), 0, stream, p);
    }
    HCP_LAUNCH_CHECK("gemm_kernel");
}

struct TileCfg { int bm, bn; };
constexpr TileCfg kCfgs[] = {{128, 128}, {128, 64}, {64, 64}, {128, 160}, {64, 160}, {256, 128}, {256, 160}, {128, 320},
                             {128, 160}, {128, 160}, {256, 160},    // 8: 8 waves x 3 stages, 9: 4 waves x 3 stages, 10: 8 waves x 3 stages
                             {128, 320}, {256, 160}, {128, 160}, {64, 160}, {128, 128}};   // 11: 16 waves (4x4), 12: 16 waves (8x2), 13-15: 8 waves as 4x2
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

// p.loaders >= 8: the ping-pong main loop (gemm_pp.hip) with an LDS ring of p.loaders - 8 tiles, where it is instantiated for
// this tile shape and its requirements hold (the v2 ones: K % 64 == 0, FAST conv gathers, 32-bit offsets).  -2 = not taken.
int try_pp(int id, int mode, bool fast_or_plain, bool lora, GemmParams& p, hipStream_t stream) {
    if (p.loaders < 8 || !fast_or_plain || !g_use_v2 || (g_dbg_ablate & 7)) return -2;
    if (p.K % BK != 0 || !(p.K2 == 0 || p.K2 == 32) || (size_t)p.M * p.lda * 2 >= (1ul << 31) || (size_t)p.N * p.ldb * 2 >= (1ul << 31)) return -2;
    const int bm = kCfgs[id].bm, bn = kCfgs[id].bn;
    p.tiles_m = hcp_cdiv(p.M, bm);
    p.dbg = g_dbg_ablate;
    p.epi_tile = want_epi_tile(p, mode) ? 1 : 0;            // (gemm_pp_launch clears it where the tile does not fit)
    hcp_bf16* const gout = p.geglu_out;                     // GEGLU-forward epilogue: see launch_cfg
    if (gout) {
        if (mode == 0 && p.nsplit == 1 && (p.N / 2) % (bn / 2) == 0) p.geglu_fused = 1;
        else p.geglu_out = nullptr;
    }
    int r = -2;
    // LDS-resident input patch (conv_patch.hip) where it wins (profiles/r6_ab_conv_patch.txt): data gradients (-4 ... -7 %) and split-K
    // launches (its chunk-aligned split needs fewer slabs: SDXL C640 @64x64 -25 %); unsplit forward convolutions are 1-10 % faster on the
    // ping-pong kernel.  tools: g_conv_patch 0 = never, 2 = wherever eligible.
    if (mode != 0 && !lora && (g_conv_patch == 2 || (g_conv_patch == 1 && (mode == 2 || p.nsplit > 1))))
        r = conv_patch_launch(p, bm, bn, mode, p.loaders - 8, stream);
    if (r == -2) r = gemm_pp_launch(p, bm, bn, mode, lora, p.loaders - 8, stream);
    if (r == -2) { p.geglu_out = gout; p.geglu_fused = 0; }     // not instantiated for this tile: the caller's own kernels decide again
    if (r != 0 || p.nsplit <= 1) return r;
    long nv = (long)p.M * (p.N / 4);
    int g = (int)((nv + 255) / 256); if (g > 2048) g = 2048;
    HCP_LAUNCH(splitk_reduce_kernel, dim3(g), dim3(256), 0, stream, p);
    HCP_LAUNCH_CHECK("splitk_reduce_kernel");
}

template <int MODE, bool FAST>
int launch_by_id(int id, GemmParams& p, hipStream_t stream) {
    if (id >= 0 && id < kNumCfgs) { const int r = try_pp(id, MODE, MODE == 0 || FAST, false, p, stream); if (r != -2) return r; }
    switch (id) {
        case 0: return launch_cfg<128, 128, 2, 2, MODE, FAST>(p, stream);
        case 1: return launch_cfg<128, 64, 2, 2, MODE, FAST>(p, stream);
        case 2: return launch_cfg<64, 64, 2, 2, MODE, FAST>(p, stream);
        case 3: return launch_cfg<128, 160, 2, 2, MODE, FAST>(p, stream);
        case 4: return launch_cfg<64, 160, 2, 2, MODE, FAST>(p, stream);
        case 5: return launch_cfg<256, 128, 4, 2, MODE, FAST>(p, stream);
        case 6: return launch_cfg<256, 160, 4, 2, MODE, FAST>(p, stream);
        case 8: return launch_cfg<128, 160, 4, 2, MODE, FAST, false, 3>(p, stream);
        case 9: return launch_cfg<128, 160, 2, 2, MODE, FAST, false, 3>(p, stream);
        case 10: return launch_cfg<256, 160, 4, 2, MODE, FAST, false, 3>(p, stream);
        case 11: return launch_cfg<128, 320, 4, 4, MODE, FAST>(p, stream);
        case 12: return launch_cfg<256, 160, 8, 2, MODE, FAST>(p, stream);            // 16 waves already: no room for loader waves
        case 13: return launch_cfg<128, 160, 4, 2, MODE, FAST, false, 2, 4>(p, stream);
        case 14: return launch_cfg<64, 160, 4, 2, MODE, FAST, false, 2, 4>(p, stream);
        case 15: return launch_cfg<128, 128, 4, 2, MODE, FAST, false, 2, 4>(p, stream);
        default: return launch_cfg<128, 320, 2, 4, MODE, FAST>(p, stream);
    }
}

int launch_lora_by_id(int id, GemmParams& p, hipStream_t stream) {
    if (id >= 0 && id < kNumCfgs) { const int r = try_pp(id, 0, true, true, p, stream); if (r != -2) return r; }
    switch (id) {
        case 0: return launch_cfg<128, 128, 2, 2, 0, false, true>(p, stream);
        case 1: return launch_cfg<128, 64, 2, 2, 0, false, true>(p, stream);
        case 2: return launch_cfg<64, 64, 2, 2, 0, false, true>(p, stream);
        case 3: return launch_cfg<128, 160, 2, 2, 0, false, true>(p, stream);
        case 5: return launch_cfg<256, 128, 4, 2, 0, false, true>(p, stream);
        case 6: return launch_cfg<256, 160, 4, 2, 0, false, true>(p, stream);
        case 8: return launch_cfg<128, 160, 4, 2, 0, false, true, 3>(p, stream);
        case 9: return launch_cfg<128, 160, 2, 2, 0, false, true, 3>(p, stream);
        case 12: return launch_cfg<256, 160, 8, 2, 0, false, true>(p, stream);
        case 13: return launch_cfg<128, 160, 4, 2, 0, false, true, 2, 4>(p, stream);
        case 14: return launch_cfg<64, 160, 4, 2, 0, false, true, 2, 4>(p, stream);
        case 15: return launch_cfg<128, 128, 4, 2, 0, false, true, 2, 4>(p, stream);
        default: return launch_cfg<64, 160, 2, 2, 0, false, true>(p, stream);
    }
}

// Fallback for shapes outside the measured table (gemm_tuned.inc).  The sweep shows the kernel is latency-bound per
// workgroup, so: narrow-M tiles (64x160, BN | every Cout) unless the problem is large, and split K until ~1024
// workgroups are in flight.
int choose_cfg(const GemmParams& p, int* nsplit_out) {
    const int nk1 = hcp_cdiv(p.K, BK);
    int id;
    if (p.N <= 64) id = 2;
    else if (p.N % 160 == 0) id = ((long)p.M * p.N >= (long)4096 * 2560) ? 6 : 4;
    else id = ((long)p.M * p.N >= (long)4096 * 2048) ? 0 : 2;
    const long tiles = (long)hcp_cdiv(p.M, kCfgs[id].bm) * hcp_cdiv(p.N, kCfgs[id].bn);
    int s = 1;
    while (s < 16 && tiles * s < 1024 && nk1 / (s * 2) >= 4) s *= 2;
    *nsplit_out = s;
    return id;
}

struct TunedEntry { int mode, M, N, K, has_k2, stride, up, cfg, split, loaders; };   // (loaders: 0 when the entry omits it)
const TunedEntry kTuned[] = {
#include "gemm_tuned_loaders.inc"
#include "gemm_tuned.inc"
};

#if defined(HCP_TOOLS)
long g_table_hits = 0, g_table_misses = 0;   // tools: how much of a workload the measured dispatch table covers
#endif

bool lookup_tuned(const GemmParams& p, int mode, int* cfg, int* split, int* loaders = nullptr) {
    bool hit = false;
    for (const TunedEntry& e : kTuned) {
        if (g_force_loaders == 0 && e.loaders) continue;     // tools: "no loader waves" = the dispatch before the loader table
        if (e.mode == mode && e.M == p.M && e.N == p.N && e.K == p.K && e.has_k2 == (p.K2 > 0) &&
            (mode == 0 || mode == 3 || (e.stride == p.cv.stride && e.up == p.cv.up))) {     // 0 / 3: plain GEMMs carry no conv geometry
            *cfg = e.cfg; *split = e.split; if (loaders) *loaders = e.loaders; hit = true; break;
        }
    }
#if defined(HCP_TOOLS)
    (hit ? g_table_hits : g_table_misses) += 1;
#endif
    return hit;
}

// The loader table was swept on warm loops, where LDS rings of 3 and 4 K tiles tie; in the step the operands are not in L2 (weights come
// from HBM, activations from the Infinity Cache) and the deeper ring is 2.5-7 % faster on every shape probed (tools/lab/cold_sweep_probe.py:
// conv C320@64^2 43.5 -> 42.5 us, fused-LoRA M1024 N1280 K1280 18.5 -> 17.2, M4096 N640 K2560 31.4 -> 29.9).  In the step itself the
// effect is within noise (same-box A/B of the headline: 19.60 -> 19.56 ms, SDXL unchanged) — kept because it never loses.  launch_cfg
// falls back to ring 3 where four stages do not fit the 160 KB of LDS.
inline int deepest_ring(int ld) { return ld == 3 ? 4 : ld; }

template <int MODE, bool FAST>
int dispatch_gemm(GemmParams& p, float* ws, size_t ws_bytes, hipStream_t stream) {
    int nsplit = 1;
    int id = 0, ld = 0;
    if (!lookup_tuned(p, MODE, &id, &nsplit, &ld)) id = choose_cfg(p, &nsplit);
    if (g_force_cfg >= 0) { id = g_force_cfg % 16; nsplit = g_force_cfg / 16 > 0 ? g_force_cfg / 16 : 1; }
    p.loaders = g_force_loaders >= 0 ? g_force_loaders : deepest_ring(ld);
    if (nsplit > 1 && (size_t)nsplit * p.M * p.N * sizeof(float) > ws_bytes) nsplit = 1;
    if (p.geglu_out) nsplit = 1;                             // the pairing epilogue lives in the GEMM kernels, not in the split-K reduce
    const int nk1 = hcp_cdiv(p.K, BK);
    p.nsplit = nsplit;
    p.kt_per_split = hcp_cdiv(nk1, nsplit);
    p.nsplit = hcp_cdiv(nk1, p.kt_per_split);
    p.slabs = ws;
    return launch_by_id<MODE, FAST>(id, p, stream);
}

int check_common(const GemmParams& p) {
    HCP_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0, "gemm: empty problem M=%d N=%d K=%d", p.M, p.N, p.K);
    HCP_REQUIRE(p.K % 8 == 0 && p.K2 % 8 == 0, "gemm: K (%d) and K2 (%d) must be multiples of 8", p.K, p.K2);
    HCP_REQUIRE(p.N % 4 == 0 && p.ldd % 4 == 0, "gemm: N (%d) and ldd (%d) must be multiples of 4", p.N, p.ldd);
    HCP_REQUIRE(p.ldb % 8 == 0, "gemm: ldb (%d) must be a multiple of 8", p.ldb);
    HCP_REQUIRE(p.K2 == 0 || (p.A2 && p.B2 && p.lda2 % 8 == 0 && p.ldb2 % 8 == 0), "gemm: bad K-extension operands");
    HCP_REQUIRE(!p.residual || p.ldr % 4 == 0, "gemm: ldr must be a multiple of 4");
    HCP_REQUIRE(!p.rowbias || p.rows_per_group > 0, "gemm: rows_per_group must be > 0");
    return 0;
}

// GEGLU-forward output (gact [M, N/2] = bf16(h gelu(g)) for D = (h | g) [M, N]): argument rules, and the second launch when the
// dispatched kernel could not pair the halves in its epilogue (GemmParams::geglu_out)
int check_gact(GemmParams& p, void* gact, const char* who) {
    if (!gact) return 0;
    HCP_REQUIRE(!p.out_f32 && !p.residual && !p.rowbias && !p.D_lo && p.alpha == 1.0f && p.N % 16 == 0 && p.ldd == p.N,
                "%s: the GEGLU output needs a contiguous bf16 (h | g) [M, N] (N %% 16 == 0) with no residual / row bias / alpha", who);
    p.geglu_out = (hcp_bf16*)gact; p.geglu_fused = 0;
    return 0;
}
int finish_gact(const GemmParams& p, void* gact, hipStream_t stream) {
    if (!gact || p.geglu_fused) return 0;
    return hcp_geglu_fwd(p.D, gact, p.M, p.N / 2, stream);
}

// tile choice + launch of a fused-LoRA problem (p.L / p.E / p.Tout set): measured table, else the fallback rule; deep-K / small-M
// shapes run as two launches (T GEMM, then split-K K-extension GEMM)
int launch_lora_dispatched(GemmParams& p, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    int id = 4, nsplit = 1, ld = 0;
    GemmParams q = p; q.K2 = 32;
    if (!lookup_tuned(q, 3, &id, &nsplit, &ld)) {
        // unseen shape: deep-K / small-M problems want split-K (two launches); otherwise fuse with a narrow-M tile
        if (p.K >= 4096 && p.M <= 4096) id = -1;
        else if ((long)p.M * p.N >= (long)4096 * 2560 && p.N % 160 == 0) id = 6;
        else if (p.N % 160 == 0 && (long)p.M * p.N >= (long)4096 * 640) id = 4;
        else id = 2;
    }
    if (g_force_cfg >= 0) id = g_force_cfg % 16;
    p.loaders = g_force_loaders >= 0 ? g_force_loaders : deepest_ring(ld);
    if (id == 7 || id == 10 || id == 11) id = 6;
    if (id < 0) {
        // two-launch form: T = A L^T, then D = A B^T + T E^T with the measured tile / split-K choice
        HCP_REQUIRE(p.Tout, "hcp_gemm_lora_bf16: this shape runs as two launches and needs the T buffer");
        // (split T, ldt = 64: this form keeps the bf16-rounded T — its T is a GEMM output of its own; the residual half of Tout is zeroed so
        //  that the weight-gradient kernel reads one format)
        const int ldt = p.ldt == 64 ? 64 : 32;
        if (ldt == 64 && hcp_memset_async(p.Tout, 0, (size_t)p.M * 64 * sizeof(hcp_bf16), stream)) return hcp_set_error("hcp_gemm_lora_bf16: memset failed");
        GemmParams t = {};
        t.A = p.A; t.lda = p.lda; t.B = p.L; t.ldb = p.K; t.M = p.M; t.N = 32; t.K = p.K; t.D = p.Tout; t.ldd = ldt; t.alpha = 1.0f;
        if (int e2 = dispatch_gemm<0, false>(t, (float*)workspace, workspace ? workspace_bytes : 0, stream)) return e2;
        p.A2 = (const hcp_bf16*)p.Tout; p.lda2 = ldt; p.B2 = p.E; p.ldb2 = 32; p.K2 = 32;
        p.L = nullptr; p.E = nullptr; p.Tout = nullptr; p.ldt = 0;
        return dispatch_gemm<0, false>(p, (float*)workspace, workspace ? workspace_bytes : 0, stream);
    }
    p.nsplit = 1; p.kt_per_split = hcp_cdiv(p.K, BK); p.slabs = nullptr;
    return launch_lora_by_id(id, p, stream);
}

}  // namespace

#if defined(HCP_TOOLS)
// TOOLS ONLY: dispatch-table lookups since the last call (hits, misses); resets the counters.
HCP_API int hcp_debug_gemm_table_stats(long* hits, long* misses) {
    if (hits) *hits = g_table_hits;
    if (misses) *misses = g_table_misses;
    g_table_hits = 0; g_table_misses = 0;
    return 0;
}
// TOOLS ONLY (tools/tune_gemm.py): cfg = tile id + 16 * nsplit; -1 restores the heuristic.
HCP_API int hcp_debug_set_gemm_config(int cfg) { g_force_cfg = cfg; return 0; }
// TOOLS ONLY: 1 = default (v2 main loop where its requirements hold), 0 / 2 = the first LDS-DMA loop (gemm_glds_kernel) everywhere.
HCP_API int hcp_debug_set_gemm_glds(int on) { g_use_glds = 1; g_use_v2 = on == 1; return 0; }
// TOOLS ONLY: ablation (results are wrong when != 0).  First LDS-DMA loop: 1 no DMA after tile 0, 2 no MFMA, 4 no LDS reads;
// v2 loop (tools build only): 8 no DMA after the ring prologue, 16 no MFMA, 32 no LDS fragment reads, 64 no output stores.
HCP_API int hcp_debug_set_gemm_ablation(int flags) { g_dbg_ablate = flags; return 0; }
// TOOLS ONLY: -1 = as the dispatch table says, 0 = never, 1 = the loader-wave variant wherever one is instantiated (tile ids 12-15).
HCP_API int hcp_debug_set_gemm_loaders(int mode) { g_force_loaders = mode; return 0; }
// TOOLS ONLY: -1 = the rule (want_epi_tile), 0 = lane-layout epilogue everywhere, 1 = tile epilogue (16-byte row pieces through LDS) wherever possible.
HCP_API int hcp_debug_set_gemm_epilogue(int mode) { g_epi_tile = mode; return 0; }
// TOOLS ONLY: 1 = default (conv_patch.hip for data gradients and split-K launches), 2 = for every eligible 3x3 convolution, 0 = ping-pong kernel only.
HCP_API int hcp_debug_set_conv_patch(int on) { g_conv_patch = on; return 0; }
#endif

// Bytes of fp32 split-K workspace that lets every launch of this shape use its preferred decomposition.
HCP_API size_t hcp_gemm_workspace_bytes(int M, int N) { return (size_t)16 * M * N * sizeof(float); }

// Replaces: torch.mm(x2d, (W_host + dW)^T) + bias  (reference lora_layers_patch.py:50-57),
// nn.Linear / 1x1 nn.Conv2d forward and their input-gradient (dX = dY W) in diffusers' UNet.
// workspace may be null (then no split-K is used).
HCP_API int hcp_gemm_bf16(const void* A, int lda, const void* B, int ldb, void* D, int ldd, int M, int N, int K,
                          const void* A2, int lda2, const void* B2, int ldb2, int K2, const float* bias,
                          const float* rowbias, int rowbias_ld, int rows_per_group, const void* residual, int ldr,
                          const void* residual_lo, void* D_lo, void* gact, float alpha, int out_f32, void* workspace,
                          size_t workspace_bytes, hipStream_t stream) {
    GemmParams p = {};
    p.A = (const hcp_bf16*)A; p.lda = lda; p.B = (const hcp_bf16*)B; p.ldb = ldb;
    p.A2 = (const hcp_bf16*)A2; p.lda2 = lda2; p.B2 = (const hcp_bf16*)B2; p.ldb2 = ldb2; p.K2 = K2;
    p.M = M; p.N = N; p.K = K; p.D = D; p.ldd = ldd; p.out_f32 = out_f32;
    p.bias = bias; p.rowbias = rowbias; p.rowbias_ld = rowbias_ld; p.rows_per_group = rows_per_group;
    p.residual = (const hcp_bf16*)residual; p.ldr = ldr; p.alpha = alpha;
    p.residual_lo = (const hcp_bf16*)residual_lo; p.D_lo = (hcp_bf16*)D_lo;
    HCP_REQUIRE(A && B && D, "hcp_gemm_bf16: null operand");
    HCP_REQUIRE((!residual_lo || residual) && (!D_lo || !out_f32), "hcp_gemm_bf16: residual_lo needs residual; D_lo needs a bf16 output");
    HCP_REQUIRE(lda % 8 == 0, "hcp_gemm_bf16: lda (%d) must be a multiple of 8", lda);
    if (int e = check_common(p)) return e;
    if (int e = check_gact(p, gact, "hcp_gemm_bf16")) return e;
    if (int e = dispatch_gemm<0, false>(p, (float*)workspace, workspace ? workspace_bytes : 0, stream)) return e;
    return finish_gact(p, gact, stream);
}

// Replaces F.conv2d(x, W[Cout,Cin,3,3], stride, padding=1) on NHWC bf16 activations:
// mode 0 = forward  (Wp packed [Cout][ky][kx][Cin1+Cin2]),   output [B,Ho,Wo,Cout]
// mode 1 = data gradient (Wp packed [Cin][ky][kx][Cout]), source = dY [B,Hs,Ws,Cout], output [B,Ho,Wo,Cin]
// (diffusers ResnetBlock2D.conv1/conv2, Downsample2D.conv (stride 2), Upsample2D = nearest-2x + conv,
//  skip-concat inputs of the up blocks; structure: reference cfgs/unet_struct.txt:92-114,390-393.)
HCP_API int hcp_conv3x3_bf16(const void* X1, int C1, const void* X2, int C2, int Bn, int Hs, int Ws, int Ho, int Wo,
                             int mode, int stride, int upsample, int pad, const void* Wp, int Cout, void* D, int ldd,
                             const float* bias, const float* rowbias, int rowbias_ld, const void* residual, int ldr,
                             int out_f32, const void* A2, const void* B2, void* workspace, size_t workspace_bytes,
                             hipStream_t stream) {
    GemmParams p = {};
    HCP_REQUIRE(X1 && Wp && D, "hcp_conv3x3_bf16: null operand");
    HCP_REQUIRE((A2 == nullptr) == (B2 == nullptr), "hcp_conv3x3_bf16: the rank-32 K-extension needs both A2 and B2");
    if (A2) { p.A2 = (const hcp_bf16*)A2; p.lda2 = 32; p.B2 = (const hcp_bf16*)B2; p.ldb2 = 32; p.K2 = 32; }
    HCP_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && (C2 == 0 || X2), "hcp_conv3x3_bf16: channels must be multiples of 8");
    HCP_REQUIRE(stride == 1 || stride == 2, "hcp_conv3x3_bf16: stride must be 1 or 2");
    HCP_REQUIRE(mode == 0 || (mode == 1 && upsample == 0 && C2 == 0), "hcp_conv3x3_bf16: bad mode/options");
    HCP_REQUIRE(pad == 1 || (pad == 0 && mode == 0 && !upsample), "hcp_conv3x3_bf16: pad must be 1 (or 0 for a forward conv without upsampling)");
    HCP_REQUIRE(Ho <= 1024 && Wo <= 1024 && Bn < 2048, "hcp_conv3x3_bf16: dims too large for packed pixel ids (10 bits per output axis)");
    HCP_REQUIRE((long)Bn * Hs * Ws * (C1 > C2 ? C1 : C2) < (1L << 31), "hcp_conv3x3_bf16: source tensor too large for 32-bit offsets");
    p.cv.X1 = (const hcp_bf16*)X1; p.cv.C1 = C1; p.cv.X2 = (const hcp_bf16*)X2; p.cv.C2 = C2;
    p.cv.Hs = Hs; p.cv.Ws = Ws; p.cv.Ho = Ho; p.cv.Wo = Wo; p.cv.stride = stride; p.cv.up = upsample ? 1 : 0; p.cv.pad = pad;
    p.M = Bn * Ho * Wo; p.N = Cout; p.K = 9 * (C1 + C2);
    p.B = (const hcp_bf16*)Wp; p.ldb = p.K;
    p.D = D; p.ldd = ldd; p.out_f32 = out_f32; p.bias = bias;
    p.rowbias = rowbias; p.rowbias_ld = rowbias_ld; p.rows_per_group = Ho * Wo;
    p.residual = (const hcp_bf16*)residual; p.ldr = ldr; p.alpha = 1.0f;
    if (int e = check_common(p)) return e;
    float* ws = (float*)workspace; size_t wb = workspace ? workspace_bytes : 0;
    // FAST needs: every 64-wide K tile inside one tap and one source tensor, linear source addressing
    const bool fast = (C1 % 64 == 0) && (C2 % 64 == 0) && !upsample && !(mode == 1 && stride == 2);
    if (mode == 0) return fast ? dispatch_gemm<1, true>(p, ws, wb, stream) : dispatch_gemm<1, false>(p, ws, wb, stream);
    return fast ? dispatch_gemm<2, true>(p, ws, wb, stream) : dispatch_gemm<2, false>(p, ws, wb, stream);
}

// FF-out input-gradient with the GEGLU backward in its epilogue:  dY_ff[M,F] = dY W (+ LoRA side path, as hcp_gemm_lora_bf16's
// backward form: L = W_up^T, E = alpha W_down^T, Tout = U = dY W_up) is never written; with (h | g) = HG[M, 2F] of the forward
//   DHG[m, n] = dY_ff * gelu(g),   DHG[m, F + n] = dY_ff * h * gelu'(g)
// Replaces the input-gradient GEMM of FeedForward.net[2] followed by the GEGLU backward pass (diffusers GEGLU under
// BasicTransformerBlock.ff, reference cfgs/unet_struct.txt:27-33; autograd of F.gelu / chunk).  L == NULL: plain host (no LoRA).
HCP_API int hcp_gemm_geglu_bwd_bf16(const void* A, int lda, const void* B, int ldb, const void* L, const void* E, void* Tout, int ldt,
                                    const void* HG, void* DHG, int M, int F, int K, void* workspace, size_t workspace_bytes,
                                    hipStream_t stream) {
    GemmParams p = {};
    p.A = (const hcp_bf16*)A; p.lda = lda; p.B = (const hcp_bf16*)B; p.ldb = ldb;
    p.M = M; p.N = F; p.K = K; p.D = DHG; p.ldd = 2 * F; p.out_f32 = 0; p.alpha = 1.0f;
    p.geglu_hg = (const hcp_bf16*)HG; p.geglu_ld = 2 * F;
    HCP_REQUIRE(A && B && HG && DHG, "hcp_gemm_geglu_bwd_bf16: null operand");
    HCP_REQUIRE((L == nullptr) == (E == nullptr), "hcp_gemm_geglu_bwd_bf16: L and E come together");
    HCP_REQUIRE(lda % 8 == 0 && F % 8 == 0, "hcp_gemm_geglu_bwd_bf16: lda (%d) and F (%d) must be multiples of 8", lda, F);
    if (int e = check_common(p)) return e;
    if (!L) return dispatch_gemm<0, false>(p, (float*)workspace, workspace ? workspace_bytes : 0, stream);
    HCP_REQUIRE(ldt == 32 || ldt == 64, "hcp_gemm_geglu_bwd_bf16: ldt (%d) is 32 (bf16 U) or 64 (split U: hi | lo)", ldt);
    p.L = (const hcp_bf16*)L; p.E = (const hcp_bf16*)E; p.Tout = (hcp_bf16*)Tout; p.ldt = ldt;
    return launch_lora_dispatched(p, workspace, workspace_bytes, stream);
}

// Fused LoRA linear (forward AND input-gradient use the same entry point):
//   T[M,32] = A[M,K] L[32,K]^T  (fp32 accumulator)
//   D[M,N]  = A B[N,K]^T + T E[N,32]^T + bias + residual
// ldt = 32: T enters the product (and Tout [M,32], if non-null) rounded to bf16.  ldt = 64 ("split"): T is carried as T_hi = bf16(T) and
// T_lo = bf16(T - T_hi) — 16 mantissa bits, the rank-r intermediate of the side path no longer sets the precision of y, dX or of the
// factor gradients — D gets T_hi E^T + T_lo E^T and Tout [M,64] = (T_hi | T_lo) for hcp_lora_wgrad*.
// forward : A = x,  B = W,   L = W_down (rank-padded), E = alpha*W_up   -> y,  T = x W_down^T   (for dW_up)
// backward: A = dY, B = W^T, L = W_up^T,               E = alpha*W_down^T -> dX, T = dY W_up    (for dW_down)
// One launch replaces LoraPatchContainer.forward's weight merge + mm (reference lora_base_patch.py:20-35,61-74).
HCP_API int hcp_gemm_lora_bf16(const void* A, int lda, const void* B, int ldb, const void* L, const void* E, void* Tout, int ldt, void* D,
                               int ldd, int M, int N, int K, const float* bias, const void* residual, int ldr, const void* residual_lo,
                               void* D_lo, void* gact, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    GemmParams p = {};
    p.A = (const hcp_bf16*)A; p.lda = lda; p.B = (const hcp_bf16*)B; p.ldb = ldb;
    p.M = M; p.N = N; p.K = K; p.D = D; p.ldd = ldd; p.out_f32 = 0;
    p.bias = bias; p.residual = (const hcp_bf16*)residual; p.ldr = ldr; p.alpha = 1.0f;
    p.L = (const hcp_bf16*)L; p.E = (const hcp_bf16*)E; p.Tout = (hcp_bf16*)Tout; p.ldt = ldt;
    p.residual_lo = (const hcp_bf16*)residual_lo; p.D_lo = (hcp_bf16*)D_lo;
    HCP_REQUIRE(A && B && D && L && E, "hcp_gemm_lora_bf16: null operand");
    HCP_REQUIRE(!residual_lo || residual, "hcp_gemm_lora_bf16: residual_lo needs residual");
    HCP_REQUIRE(ldt == 32 || ldt == 64, "hcp_gemm_lora_bf16: ldt (%d) is 32 (bf16 T) or 64 (split T: hi | lo)", ldt);
    HCP_REQUIRE(lda % 8 == 0, "hcp_gemm_lora_bf16: lda (%d) must be a multiple of 8", lda);
    if (int e = check_common(p)) return e;
    if (int e = check_gact(p, gact, "hcp_gemm_lora_bf16")) return e;
    if (int e = launch_lora_dispatched(p, workspace, workspace_bytes, stream)) return e;
    return finish_gact(p, gact, stream);
}
