// gemm_params.h — the launch descriptor shared by the GEMM / implicit-conv translation units (gemm.hip, gemm_pp.hip).
#pragma once
#include "hcp_common.h"

namespace hcp_gemm {

struct ConvDesc {
    const hcp_bf16* X1; int C1;   // first source tensor  [B, Hs, Ws, C1]
    const hcp_bf16* X2; int C2;   // optional second source (channel concat), else null/0
    int Hs, Ws;                   // source spatial dims (memory)
    int Ho, Wo;                   // output spatial dims (rows of the implicit A matrix)
    int stride;                   // 1 or 2
    int up;                       // 1: source is nearest-upsampled 2x before the conv (fwd only)
    int pad;                      // 1: taps -1..+1 (padding 1); 0: taps 0..+2 (F.pad(0,1,0,1) + padding 0: the VAE encoder's Downsample2D), fwd only
};

struct GemmParams {
    const hcp_bf16* A; int lda;
    const hcp_bf16* A2; int lda2; int K2;
    const hcp_bf16* B; int ldb;
    const hcp_bf16* B2; int ldb2;
    int M, N, K;
    void* D; int ldd; int out_f32;
    const float* bias;
    const float* rowbias; int rowbias_ld; int rows_per_group;
    const hcp_bf16* residual; int ldr;
    // (hi | lo) residual stream (round 6): the transformer blocks' residual stream carried as TWO bf16 tensors, x = hi + lo with
    // hi = bf16(x) and lo = bf16(x - hi) — 16 mantissa bits, what the reference's LoRA layers keep by promoting their output to fp32
    // (mm(...) [bf16] + bias [fp32], lora_layers_patch.py:50-57).  residual_lo (same ldr, needs residual) is added in fp32 with the
    // rest of the epilogue; D_lo (same ldd, bf16 output only) receives bf16(v - bf16(v)) next to D = bf16(v).  Null = plain bf16.
    const hcp_bf16* residual_lo; hcp_bf16* D_lo;
    float alpha;
    int tiles_m;
    int nsplit; int kt_per_split;   // split-K over the primary K tiles (grid.y)
    float* slabs;                   // [nsplit][M][N] fp32 partials when nsplit > 1
    // fused LoRA (LORA kernels): T = A L^T is accumulated next to the main tile from the same A tiles, rounded to
    // bf16, then D += T E^T as one extra k-step.  L [32,K] (ldl = K), E [N,32], Tout [M,ldt] (optional, for wgrad).
    // ldt = 64 ("split" T): the fp32 T leaves the accumulator as TWO bf16 images, T_hi = bf16(T) and T_lo = bf16(T - T_hi) (16 mantissa
    // bits between them); the K-extension adds T_hi E^T + T_lo E^T (one more MFMA per output block) and Tout gets T_hi in columns
    // 0..31, T_lo in columns 32..63 for the weight-gradient kernel.  ldt = 32 (or 0): the bf16-rounded T only.
    const hcp_bf16* L; const hcp_bf16* E; hcp_bf16* Tout; int ldt;
    // GEGLU-backward epilogue (hcp_gemm_geglu_bwd_bf16): the product is dY_ff = d(h * gelu(g)) [M, N = F]; hg [M, 2F] holds the forward's
    // (h | g); D is d(h | g) [M, 2F] (ldd = 2F): D[m, n] = v * gelu(g), D[m, F + n] = v * h * gelu'(g).  Null = ordinary epilogue.
    const hcp_bf16* geglu_hg; int geglu_ld;
    // GEGLU-FORWARD epilogue (round 6): the product is (h | g) [M, N = 2F] of diffusers' GEGLU projection; geglu_out [M, F] receives
    // bf16(h * gelu(g)) computed from the fp32 epilogue values (the reference's LoRA layer hands GEGLU an fp32 (h | g) under autocast,
    // lora_layers_patch.py:50-57: one rounding instead of two), next to D = bf16(h | g) which the backward needs anyway.  A workgroup's N
    // tile pairs the columns: tile j = h columns [j BN/2, (j+1) BN/2) then the matching g columns F + ..., so the wave that holds h and
    // the wave that holds g of one (row, column) meet through LDS (geglu_col below maps tile columns to actual ones everywhere a column
    // is used: B / E / bias rows, stores).  The launch site clears it when its kernel cannot pair (F % (BN / 2) != 0, split-K, first
    // LDS-DMA loop) and the entry point then runs hcp_geglu_fwd behind the GEMM; geglu_fused reports which happened.
    hcp_bf16* geglu_out; int geglu_fused;
    // "tile epilogue" (round 6): the finished fp32 values go through an LDS tile and leave in 16-byte pieces of full output rows (residual
    // read the same way) instead of 8-byte pieces in the MFMA lane layout — see epi_tile_store below; set by the dispatcher.
    int epi_tile;
    int loaders;                    // 1: launch the loader-wave variant of the v2 kernel where one is instantiated (dispatch table / tools)
    int dbg;                        // tools/ablate_gemm.py: 1 = skip the DMA after the first tile, 2 = skip the MFMAs, 4 = skip LDS reads + MFMAs
    ConvDesc cv;
};

constexpr int BK = 64;

// One accumulator quad of the fused-LoRA T tile -> its bf16 image(s): hi = bf16(t), lo = bf16(t - hi) (the rounding residual).
HCP_DEVICE void lora_t_split(const hcp_f32x4& t, hcp_bf16x4& hi, hcp_bf16x4& lo) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned short h = hcp_f2bf(t[q]);
        hi[q] = (short)h;
        lo[q] = (short)hcp_f2bf(t[q] - hcp_bf2f(h));
    }
}

// The output quad of an epilogue: D = bf16(v) and, when the launch carries a lo image, D_lo = bf16(v - D).
HCP_DEVICE void store_hi_lo(const GemmParams& p, int m, int n, const hcp_f32x4& v) {
    hcp_bf16x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (short)hcp_f2bf(v[q]);
    *(hcp_bf16x4*)((hcp_bf16*)p.D + (size_t)m * p.ldd + n) = o;
    if (p.D_lo) {
        hcp_bf16x4 l;
#pragma unroll
        for (int q = 0; q < 4; ++q) l[q] = (short)hcp_f2bf(v[q] - hcp_bf2f((unsigned short)o[q]));
        *(hcp_bf16x4*)(p.D_lo + (size_t)m * p.ldd + n) = l;
    }
}
// v += residual_lo[m, n .. n+3] (requested where it is used: the lo image is read once, behind the main loop)
HCP_DEVICE void add_residual_lo(const GemmParams& p, int m, int n, hcp_f32x4& v) {
    if (!p.residual_lo) return;
    const hcp_bf16x4 r = *(const hcp_bf16x4*)(p.residual_lo + (size_t)m * p.ldr + n);
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += hcp_bf2f((unsigned short)r[q]);
}

// Tile epilogue.  In the MFMA layout a lane owns 4 consecutive columns of one row: the epilogue's global stores (and residual loads) are
// 8-byte pieces, 16 rows x 32-byte runs per wave instruction — issue-bound long before HBM is (guide: the bf16 row-per-lane store tail).
// Here the waves first park v = alpha acc + bias (+ row bias) as fp32 in LDS ([BM][BN + 4]: the 16 rows a wave stores at once start on
// distinct bank quads), then every compute thread walks (row, 8 columns) pieces: one 16-byte residual load (+ one for a lo image), one
// 16-byte store (+ one for D_lo), consecutive threads on consecutive pieces of a row.  Same arithmetic as the lane-layout epilogue.
constexpr int epi_tile_ld(int BN) { return BN + 4; }
template <int BM, int BN, int NTHREADS>
HCP_DEVICE void epi_tile_store(const GemmParams& p, const float* tile, int m0, int n0, int tid) {
    constexpr int LD = epi_tile_ld(BN), PIECES = BN / 8;
    for (int idx = tid; idx < BM * PIECES; idx += NTHREADS) {
        const int row = idx / PIECES, c8 = (idx - row * PIECES) * 8;
        const int m = m0 + row, n = n0 + c8;
        if (m >= p.M || n >= p.N) continue;                                   // (N % 8 == 0: a piece is inside or outside)
        const hcp_f32x4 a = *(const hcp_f32x4*)(tile + row * LD + c8), b = *(const hcp_f32x4*)(tile + row * LD + c8 + 4);
        float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        if (p.residual) {
            const hcp_bf16x8 r = *(const hcp_bf16x8*)(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += hcp_bf2f((unsigned short)r[q]);
            if (p.residual_lo) {
                const hcp_bf16x8 rl = *(const hcp_bf16x8*)(p.residual_lo + (size_t)m * p.ldr + n);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += hcp_bf2f((unsigned short)rl[q]);
            }
        }
        hcp_bf16x8 o;
#pragma unroll
        for (int q = 0; q < 8; ++q) o[q] = (short)hcp_f2bf(v[q]);
        *(hcp_bf16x8*)((hcp_bf16*)p.D + (size_t)m * p.ldd + n) = o;
        if (p.D_lo) {
            hcp_bf16x8 l;
#pragma unroll
            for (int q = 0; q < 8; ++q) l[q] = (short)hcp_f2bf(v[q] - hcp_bf2f((unsigned short)o[q]));
            *(hcp_bf16x8*)(p.D_lo + (size_t)m * p.ldd + n) = l;
        }
    }
}

// GEGLU-forward launches: logical column nl of the tiled problem -> the column of B / E / bias / D it stands for (see geglu_out).
HCP_DEVICE int geglu_col(const GemmParams& p, int BN, int nl) {
    if (!p.geglu_out) return nl;
    const int half = BN >> 1, j = nl / BN, c = nl - j * BN;
    return c < half ? j * half + c : (p.N >> 1) + j * half + (c - half);
}
// The pairing step of the GEGLU-forward epilogue.  v[i][j]: this wave's finished fp32 quads (TI row blocks x TJ column blocks, the MFMA
// layout: lane (fr, fg) holds row fr, columns 4 fg .. 4 fg + 3 of a 16 x 16 block); is_g: the wave holds g columns; slot: index of the
// (h wave, g wave) pair among the workgroup's pairs; xg: LDS, pairs * TI * TJ * 64 float4.  The g wave publishes gelu(g), the h wave
// multiplies and stores bf16(h * gelu(g)) at out[m, nh .. nh + 3].  `sync`: the workgroup barrier of the calling kernel.
template <int TI, int TJ, typename SYNC>
HCP_DEVICE void geglu_fwd_pair(hcp_f32x4 (&v)[TI][TJ], bool is_g, int slot, int lane, hcp_f32x4* xg, SYNC&& sync,
                               hcp_bf16* out, int F, const int (&rows)[TI], int M, const int (&cols)[TJ]) {
    hcp_f32x4* mine = xg + (size_t)slot * TI * TJ * 64 + lane;
    if (is_g) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                hcp_f32x4 ge;
#pragma unroll
                for (int q = 0; q < 4; ++q) ge[q] = hcp_gelu_erf(v[i][j][q]);
                mine[(i * TJ + j) * 64] = ge;
            }
    }
    sync();
    if (!is_g) {
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            if (rows[i] >= M) continue;
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const hcp_f32x4 ge = mine[(i * TJ + j) * 64];
                hcp_bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (short)hcp_f2bf(v[i][j][q] * ge[q]);
                *(hcp_bf16x4*)(out + (size_t)rows[i] * F + cols[j]) = o;
            }
        }
    }
}

// One 4-column piece of the GEGLU-backward epilogue (replaces the stand-alone geglu_bwd pass over dY_ff, h|g and d(h|g)).
HCP_DEVICE void epilogue_geglu_bwd(const GemmParams& p, int m, int n, hcp_f32x4 v) {
    const hcp_bf16* hp = p.geglu_hg + (size_t)m * p.geglu_ld + n;
    const hcp_bf16x4 h = *(const hcp_bf16x4*)hp, g = *(const hcp_bf16x4*)(hp + p.N);
    hcp_bf16x4 dh, dg;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float hf = hcp_bf2f((unsigned short)h[q]), gf = hcp_bf2f((unsigned short)g[q]);
        const float d = hcp_bf2f(hcp_f2bf(v[q] * p.alpha));                 // the two-kernel form rounds dY_ff to bf16 first: same values
        dh[q] = (short)hcp_f2bf(d * hcp_gelu_erf(gf));
        dg[q] = (short)hcp_f2bf(d * hf * hcp_gelu_erf_grad(gf));
    }
    hcp_bf16* dp = (hcp_bf16*)p.D + (size_t)m * p.ldd + n;
    *(hcp_bf16x4*)dp = dh;
    *(hcp_bf16x4*)(dp + p.N) = dg;
}

// The same epilogue for a whole workgroup tile, in full 16-byte row pieces.  In the MFMA layout a lane owns 4 columns of one row, so the
// four streams of this epilogue (h, g in; dh, dg out) would move in 32-byte runs (measured: the FF-out input gradient at 64x64 went 40 ->
// 101 us that way, against 26 us for the stand-alone pass it replaces).  The waves first park the bf16 product tile in LDS (row stride
// BN + 8: the 16 rows a wave stores at once fall on distinct banks), then every thread walks (row, 8 columns) pieces of the tile.
constexpr int geglu_tile_ld(int BN) { return BN + 8; }
HCP_DEVICE void geglu_tile_put(hcp_bf16* tile, int ld, int row, int col, hcp_f32x4 v, float alpha) {
    hcp_bf16x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = (short)hcp_f2bf(v[q] * alpha);
    *(hcp_bf16x4*)(tile + row * ld + col) = o;
}
template <int BM, int BN, int NTHREADS>
HCP_DEVICE void geglu_tile_apply(const GemmParams& p, const hcp_bf16* tile, int m0, int n0, int tid) {
    constexpr int LD = geglu_tile_ld(BN), PIECES = BN / 8;
    for (int idx = tid; idx < BM * PIECES; idx += NTHREADS) {
        const int row = idx / PIECES, c8 = (idx - row * PIECES) * 8;
        const int m = m0 + row, n = n0 + c8;
        if (m >= p.M || n >= p.N) continue;                                   // (N % 8 == 0: a piece is inside or outside)
        const hcp_bf16x8 d = *(const hcp_bf16x8*)(tile + row * LD + c8);
        const hcp_bf16* hp = p.geglu_hg + (size_t)m * p.geglu_ld + n;
        const hcp_bf16x8 h = *(const hcp_bf16x8*)hp, g = *(const hcp_bf16x8*)(hp + p.N);
        hcp_bf16x8 dh, dg;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float df = hcp_bf2f((unsigned short)d[q]), hf = hcp_bf2f((unsigned short)h[q]), gf = hcp_bf2f((unsigned short)g[q]);
            dh[q] = (short)hcp_f2bf(df * hcp_gelu_erf(gf));
            dg[q] = (short)hcp_f2bf(df * hf * hcp_gelu_erf_grad(gf));
        }
        hcp_bf16* dp = (hcp_bf16*)p.D + (size_t)m * p.ldd + n;
        *(hcp_bf16x8*)dp = dh;
        *(hcp_bf16x8*)(dp + p.N) = dg;
    }
}

// gemm_pp.hip — the ping-pong main loop (two compute groups half a phase apart + 4 loader waves).  `p` arrives with tiles_m,
// nsplit, kt_per_split and slabs set by the dispatcher; ring = depth of the LDS ring (2..4, lowered to what fits 160 KB).
// Returns -2 when no kernel is instantiated for (bm, bn, mode, lora) — the caller then keeps its own kernels.
int gemm_pp_launch(GemmParams& p, int bm, int bn, int mode, bool lora, int ring, hipStream_t stream);
// conv_patch.hip — the same workgroup shape with the convolution's input held in LDS as a pixel patch (3x3, stride 1, pad 1, forward and
// data gradient).  Returns -2 when the launch is not eligible; may round p.kt_per_split up to whole 64-channel chunks (9 K tiles).
int conv_patch_launch(GemmParams& p, int bm, int bn, int mode, int ring, hipStream_t stream);

}  // namespace hcp_gemm
