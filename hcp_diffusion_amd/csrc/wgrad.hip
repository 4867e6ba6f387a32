// wgrad.hip — weight gradients of the host layers (full fine-tuning / DreamBooth / ControlNet-branch training:
// reference cfgs/train/examples/DreamBooth.yaml:6-10 `unet: [{lr, layers:['']}]`, where autograd computes
// dW = dY^T X for every nn.Linear / nn.Conv2d of the UNet).
//
//   D[n][k] += sum_m Y[m][n] * X[m][k]            "TN" GEMM: the reduction runs down the ROWS of both operands.
//
// Both operands arrive row-major ([tokens][channels], the activation layout of the whole path), so no operand is
// ever transposed in memory: the 64-row tiles are DMA'd to LDS as they are (global_load_lds_dwordx4, lane-linear
// destination, XOR-swizzled 32-byte segments) and the MFMA fragments whose k-slots run down the rows are gathered
// with ds_read_b64_tr_b16.  Accumulation is fp32.  The token dimension is split across workgroups to fill 256 CUs:
// each split writes its partial tile to a workspace slab and a reduce kernel adds the slabs into the flat gradient
// bucket (`+=`, which is also what gradient accumulation needs); a single split adds in place.  (Device-scope fp32
// atomics were measured first: ~100 G atomics/s made every small layer 5-10x slower than its data movement.)
//
// The conv3x3 variant gathers X rows on the fly exactly like the forward implicit GEMM (stride 2, nearest-2x upsample
// folded into the gather, two-source channel concat), column k = tap * (C1 + C2) + c, and writes the gradient in the
// weight's physical layout [Cout][ky][kx][Cin] (= torch channels_last of diffusers' [Cout,Cin,3,3]).
#include "hcp_common.h"

namespace {

constexpr int TM_ROWS = 64;    // token rows per LDS tile (two 32-deep MFMA k-steps)
constexpr int WY = 128;        // output rows (n) per workgroup = columns of the Y tile

struct WgradParams {
    const hcp_bf16* Y; int ldy;
    const hcp_bf16* X1; const hcp_bf16* X2; int C1, C2; int ldx;
    float* D; int ldw; int Cw;
    int M, N, K;
    int rows_per_split, tiles_n;
    float* slabs; int nsplit;       // nsplit > 1: partial sums go to slabs[split][N][K]
    int Hs, Ws, Ho, Wo, stride, up;
};

// 32-byte-segment swizzle of a row-major [64][W] bf16 LDS image (W = 64 or 128): chunk' = chunk ^ sw(row) keeps the
// two 16-byte chunks of a segment adjacent (the DMA writes 16-byte chunks, the transpose reads fetch 8 bytes per lane)
// and makes the 8 rows a 32-lane ds_read_b64_tr pass touches fall into 8 different segments of the 256-byte bank row.
template <int W> HCP_DEVICE int sw_chunk(int row) { return W == 128 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1); }

HCP_DEVICE hcp_bf16x8 join8w(hcp_bf16x4 a, hcp_bf16x4 b) {
    hcp_bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
    return r;
}

// fragment for output index c0 + fr, k-slots = tile rows {32 s2 + 4 fg + j} U {32 s2 + 16 + 4 fg + j}, j = 0..3
template <int W>
HCP_DEVICE hcp_bf16x8 tr_frag_sw(const hcp_bf16* tile, int c0, int s2, int fr, int fg) {
    const int r0 = 32 * s2 + 4 * fg + (fr >> 2), r1 = r0 + 16;
    const int col = c0 + 4 * (fr & 3);
    const hcp_bf16* a0 = tile + r0 * W + ((((col >> 3) ^ sw_chunk<W>(r0)) << 3) | (col & 7));
    const hcp_bf16* a1 = tile + r1 * W + ((((col >> 3) ^ sw_chunk<W>(r1)) << 3) | (col & 7));
    return join8w(hcp_lds_read_tr4(a0), hcp_lds_read_tr4(a1));
}

template <int WX, bool CONV>
HCP_KERNEL(256) wgrad_tn_kernel(WgradParams p) {
    constexpr int CPR_X = WX / 8, RPI_X = 64 / CPR_X, NI_X = TM_ROWS / RPI_X / 4;     // chunks/row, rows/DMA instr, instrs/wave
    constexpr int CPR_Y = WY / 8, RPI_Y = 64 / CPR_Y, NI_Y = TM_ROWS / RPI_Y / 4;
    constexpr int Y_ELEMS = TM_ROWS * WY, X_ELEMS = TM_ROWS * WX, BUF = Y_ELEMS + X_ELEMS;
    constexpr int TN_ = 4, TK_ = WX / 32;                                                // 16x16 blocks per wave (n, k)
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int wn = wave >> 1, wk = wave & 1;
    const int tile_n = blockIdx.x % p.tiles_n, tile_k = blockIdx.x / p.tiles_n;
    const int n0 = tile_n * WY, k0 = tile_k * WX;
    const int mb = blockIdx.y * p.rows_per_split;
    int me = mb + p.rows_per_split; if (me > p.M) me = p.M;
    const int Ctot = p.C1 + p.C2;

    // Buffer-addressed DMA (see gemm_v2_kernel): per-lane byte offsets are loop-invariant for Y and for a linear X — the
    // resource base advances by 64 rows per tile and its num_records ends at row `me`, so the ragged last tile reads zeros.
    // The conv X gather recomputes the pixel per tile (shift/mask when the output map is a power of two wide, as in SD).
    unsigned vy[NI_Y], vx[NI_X];
    int x_row[NI_X], x_tap[NI_X], x_cs[NI_X];
    bool x_second[NI_X];
#pragma unroll
    for (int i = 0; i < NI_Y; ++i) {
        const int r = (wave * NI_Y + i) * RPI_Y + lane / CPR_Y;
        const int col = n0 + (((lane % CPR_Y) ^ sw_chunk<WY>(r)) << 3);
        vy[i] = col < p.N ? (unsigned)(((size_t)r * p.ldy + col) * 2) : HCP_BUF_OOB;
    }
#pragma unroll
    for (int i = 0; i < NI_X; ++i) {
        const int r = (wave * NI_X + i) * RPI_X + lane / CPR_X;
        const int k = k0 + (((lane % CPR_X) ^ sw_chunk<WX>(r)) << 3);
        x_row[i] = r; x_tap[i] = -1; x_cs[i] = p.ldx; x_second[i] = false;
        vx[i] = k < p.K ? (unsigned)(((size_t)r * p.ldx + k) * 2) : HCP_BUF_OOB;
        if (CONV) {
            vx[i] = HCP_BUF_OOB;
            if (k < p.K) {
                const int tap = k / Ctot, c = k - tap * Ctot;
                x_tap[i] = tap;
                x_second[i] = c >= p.C1;
                x_cs[i] = x_second[i] ? p.C2 : p.C1;
                vx[i] = (unsigned)((x_second[i] ? c - p.C1 : c) * 2);          // channel byte offset inside the pixel
            }
        }
    }
    const hcp_rsrc rx1 = hcp_make_rsrc(p.X1), rx2 = hcp_make_rsrc(p.X2 ? p.X2 : p.X1);
    const int hw = p.Ho * p.Wo;
    const bool pow2 = (hw & (hw - 1)) == 0 && (p.Wo & (p.Wo - 1)) == 0;
    const int hw_shift = pow2 ? __builtin_ctz(hw) : 0, wo_shift = pow2 ? __builtin_ctz(p.Wo) : 0;

    auto issue_tile = [&](int m0, int buf) {
        hcp_bf16* ly = lds + buf * BUF;
        hcp_bf16* lx = ly + Y_ELEMS;
        const int live = me - m0;                                  // rows of this tile that exist (>= 1)
        const hcp_rsrc ry = hcp_make_rsrc_n(p.Y + (size_t)m0 * p.ldy, (unsigned)(((size_t)(live < TM_ROWS ? live : TM_ROWS) - 1) * p.ldy * 2 + p.ldy * 2));
#pragma unroll
        for (int i = 0; i < NI_Y; ++i) hcp_buf_glds16(ry, vy[i], ly + (wave * NI_Y + i) * RPI_Y * WY);
        if (!CONV) {
            const hcp_rsrc rx = hcp_make_rsrc_n(p.X1 + (size_t)m0 * p.ldx, (unsigned)((size_t)(live < TM_ROWS ? live : TM_ROWS) * p.ldx * 2));
#pragma unroll
            for (int i = 0; i < NI_X; ++i) hcp_buf_glds16(rx, vx[i], lx + (wave * NI_X + i) * RPI_X * WX);
        } else {
#pragma unroll
            for (int i = 0; i < NI_X; ++i) {
                const int m = m0 + x_row[i];
                unsigned v = HCP_BUF_OOB;
                if (m < me && x_tap[i] >= 0) {
                    int b, py, px;
                    if (pow2) { b = m >> hw_shift; const int rem = m & (hw - 1); py = rem >> wo_shift; px = rem & (p.Wo - 1); }
                    else { b = m / hw; const int rem = m - b * hw; py = rem / p.Wo; px = rem - py * p.Wo; }
                    const int ky = x_tap[i] / 3, kx = x_tap[i] - ky * 3;
                    int sy = py * p.stride + ky - 1, sx = px * p.stride + kx - 1;
                    const bool ok = sy >= 0 && sx >= 0 && sy < (p.Hs << p.up) && sx < (p.Ws << p.up);
                    sy >>= p.up; sx >>= p.up;
                    if (ok) v = (unsigned)(((b * p.Hs + sy) * p.Ws + sx) * x_cs[i] * 2) + vx[i];
                }
                if (x_second[i]) hcp_buf_glds16(rx2, v, lx + (wave * NI_X + i) * RPI_X * WX);
                else hcp_buf_glds16(rx1, v, lx + (wave * NI_X + i) * RPI_X * WX);
            }
        }
    };

    hcp_f32x4 acc[TN_][TK_];
#pragma unroll
    for (int i = 0; i < TN_; ++i)
#pragma unroll
        for (int j = 0; j < TK_; ++j) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }

    const int nt = (me - mb + TM_ROWS - 1) / TM_ROWS;
    if (nt > 0) issue_tile(mb, 0);
    HCP_SYNC();
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) issue_tile(mb + (t + 1) * TM_ROWS, cur ^ 1);
        const hcp_bf16* ly = lds + cur * BUF;
        const hcp_bf16* lx = ly + Y_ELEMS;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            hcp_bf16x8 fy[TN_], fx[TK_];
#pragma unroll
            for (int i = 0; i < TN_; ++i) fy[i] = tr_frag_sw<WY>(ly, wn * 64 + i * 16, s2, fr, fg);
#pragma unroll
            for (int j = 0; j < TK_; ++j) fx[j] = tr_frag_sw<WX>(lx, wk * (WX / 2) + j * 16, s2, fr, fg);
#pragma unroll
            for (int i = 0; i < TN_; ++i)
#pragma unroll
                for (int j = 0; j < TK_; ++j) acc[i][j] = hcp_mfma16(fx[j], fy[i], acc[i][j]);
        }
        HCP_SYNC();                                       // drains the DMA of tile t+1 and fences this tile's reads
    }

    // lane holds D[n = .. + fr][k = .. + 4 fg + r]   (K % 8 == 0: a lane's 4 columns are all in or all out)
#pragma unroll
    for (int i = 0; i < TN_; ++i) {
        const int n = n0 + wn * 64 + i * 16 + fr;
        if (n >= p.N) continue;
#pragma unroll
        for (int j = 0; j < TK_; ++j) {
            const int k = k0 + wk * (WX / 2) + j * 16 + 4 * fg;
            if (k >= p.K) continue;
            if (p.nsplit > 1) {
                *(hcp_f32x4*)(p.slabs + ((size_t)blockIdx.y * p.N + n) * p.K + k) = acc[i][j];
                continue;
            }
            float* dst; int lim;
            if (CONV) {
                const int tap = k / Ctot, c = k - tap * Ctot;
                dst = p.D + (size_t)n * p.ldw + tap * p.Cw + c; lim = p.Cw - c;
            } else {
                dst = p.D + (size_t)n * p.ldw + k; lim = 4;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < lim) dst[r] += acc[i][j][r];        // this workgroup is the only writer of its tile
        }
    }
}

// D (+)= sum over the splits of slabs[s][n][k], with the conv column -> [tap][Cw] remap
HCP_KERNEL(256) wgrad_reduce_kernel(WgradParams p, int conv) {
    const int kv = p.K / 4;
    const long total = (long)p.N * kv;
    const int Ctot = p.C1 + p.C2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int n = (int)(i / kv), k = (int)(i - (long)n * kv) * 4;
        hcp_f32x4 v = *(const hcp_f32x4*)(p.slabs + (size_t)n * p.K + k);
        for (int s = 1; s < p.nsplit; ++s) v += *(const hcp_f32x4*)(p.slabs + ((size_t)s * p.N + n) * p.K + k);
        float* dst; int lim = 4;
        if (conv) {
            const int tap = k / Ctot, c = k - tap * Ctot;
            dst = p.D + (size_t)n * p.ldw + tap * p.Cw + c; lim = p.Cw - c;
        } else {
            dst = p.D + (size_t)n * p.ldw + k;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < lim) dst[r] += v[r];
    }
}

// out[n] += sum_m Y[m][n]  (bias gradients; per-sample variant for the time-embedding row bias: group g = m / rows_per_group)
HCP_KERNEL(256) colsum_kernel(const hcp_bf16* Y, int ldy, float* out, int ldo, int M, int N, int rows_per_group, int rows_per_block) {
    HCP_DYN_SMEM(smem);
    float (*red)[65] = (float (*)[65])smem;              // [32][65]
    const int tid = threadIdx.x;
    const int ch = tid & 7, rl = tid >> 3;                 // 8 chunks of 8 columns x 32 row lanes
    const int c0 = blockIdx.x * 64 + ch * 8;
    const int g = blockIdx.z;
    const int gb = g * rows_per_group;
    int ge = gb + rows_per_group; if (ge > M) ge = M;
    const int mb = gb + blockIdx.y * rows_per_block;
    int me = mb + rows_per_block; if (me > ge) me = ge;
    float s[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] = 0.f;
    if (c0 < N)
        for (int m = mb + rl; m < me; m += 32) {
            hcp_bf16x8 v = *(const hcp_bf16x8*)(Y + (size_t)m * ldy + c0);
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += hcp_bf2f((hcp_bf16)v[i]);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[rl][ch * 8 + i] = s[i];
    HCP_SYNC();
    if (tid < 64) {
        float t = 0.f;
        for (int r = 0; r < 32; ++r) t += red[r][tid];
        const int c = blockIdx.x * 64 + tid;
        if (c < N && mb < me) hcp_atomic_add(out + (size_t)g * ldo + c, t);
    }
}

HCP_TUNABLE(int, g_force_wx, 0);     // tools/tests: 64 or 128 forces the X-tile width, 0 = heuristic
HCP_TUNABLE(int, g_force_split, 0);  // tools/tests: > 0 forces the number of token splits

template <bool CONV>
int launch_wgrad(WgradParams& p, float* ws, size_t ws_bytes, hipStream_t stream) {
    // fitted to a sweep over the SD1.5 layer shapes (tools/bench_wgrad.py, profiles/r1_wgrad_sweep_bs2.txt): the narrow X tile
    // with ~450 workgroups in flight is within 7 % of the per-shape optimum; the 128-wide tile only wins on a few deep layers
    const bool wide = g_force_wx == 128;
    const int wx = wide ? 128 : 64;
    p.tiles_n = hcp_cdiv(p.N, WY);
    const int tiles_k = hcp_cdiv(p.K, wx);
    const long tiles = (long)p.tiles_n * tiles_k;
    const int row_tiles = hcp_cdiv(p.M, TM_ROWS);
    int nsplit = (int)((450 + tiles / 2) / tiles);                // ~450 workgroups ...
    if (nsplit > row_tiles / 2) nsplit = row_tiles / 2;           // ... each with at least two row tiles to pipeline
    if (g_force_split > 0) nsplit = g_force_split;
    const size_t slab = (size_t)p.N * p.K * sizeof(float);
    if (!ws || (size_t)nsplit * slab > ws_bytes) nsplit = ws ? (int)(ws_bytes / slab) : 1;
    if (nsplit > row_tiles) nsplit = row_tiles;
    if (nsplit < 1) nsplit = 1;
    p.rows_per_split = hcp_cdiv(row_tiles, nsplit) * TM_ROWS;
    nsplit = hcp_cdiv(p.M, p.rows_per_split);
    p.nsplit = nsplit; p.slabs = ws;
    const size_t smem = (size_t)2 * TM_ROWS * (WY + wx) * sizeof(hcp_bf16);
    if (wide) HCP_LAUNCH((wgrad_tn_kernel<128, CONV>), dim3((unsigned)tiles, nsplit), dim3(256), smem, stream, p);
    else HCP_LAUNCH((wgrad_tn_kernel<64, CONV>), dim3((unsigned)tiles, nsplit), dim3(256), smem, stream, p);
    if (nsplit > 1) {
        long nv = (long)p.N * (p.K / 4);
        int g = (int)((nv + 255) / 256); if (g > 2048) g = 2048;
        HCP_LAUNCH(wgrad_reduce_kernel, dim3(g), dim3(256), 0, stream, p, CONV ? 1 : 0);
    }
    HCP_LAUNCH_CHECK("wgrad_tn_kernel");
}

}  // namespace

#if defined(HCP_TOOLS)
// TOOLS / TESTS ONLY: cfg = X-tile width (64 / 128 / 0 = heuristic) + 256 * forced token splits (0 = heuristic).
HCP_API int hcp_debug_set_wgrad_tile(int cfg) { g_force_wx = cfg & 255; g_force_split = cfg >> 8; return 0; }
#endif

// dW[N,K] (fp32, leading dim ldw) += dY[M,N]^T X[M,K]     (nn.Linear / 1x1 conv weight gradient)
// workspace: optional fp32 scratch for the token-split partial sums (any size; more allows more parallelism)
HCP_API int hcp_wgrad_linear_bf16(const void* dY, int ldy, const void* X, int ldx, float* dW, int ldw, int M, int N, int K,
                                  float* workspace, size_t workspace_bytes, hipStream_t stream) {
    HCP_REQUIRE(dY && X && dW && M > 0 && N > 0 && K > 0, "hcp_wgrad_linear_bf16: bad arguments");
    HCP_REQUIRE(K % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldy >= (N + 7) / 8 * 8 && ldx >= K && ldw >= K,
                "hcp_wgrad_linear_bf16: K (%d), ldx (%d), ldy (%d) must be multiples of 8 and cover the operands", K, ldx, ldy);
    WgradParams p = {};
    p.Y = (const hcp_bf16*)dY; p.ldy = ldy; p.X1 = (const hcp_bf16*)X; p.ldx = ldx; p.C1 = K;
    p.D = dW; p.ldw = ldw; p.Cw = K; p.M = M; p.N = N; p.K = K;
    return launch_wgrad<false>(p, workspace, workspace_bytes, stream);
}

// dW[Cout][3][3][Cw] (fp32) += sum over output pixels of dY[b,py,px,co] * gathered X (the forward gather of
// hcp_conv3x3_bf16: stride, fused nearest-2x upsample, channel concat X1|X2).  Cw <= C1 + C2 is the weight's true Cin
// (conv_in stages 4 channels padded to 8).  dY rows have ldy >= Cout (conv_out: 4 channels padded to 8).
HCP_API int hcp_wgrad_conv3x3_bf16(const void* dY, int ldy, const void* X1, int C1, const void* X2, int C2, float* dW, int Cw,
                                   int B, int Hs, int Ws, int Ho, int Wo, int Cout, int stride, int upsample, float* workspace,
                                   size_t workspace_bytes, hipStream_t stream) {
    HCP_REQUIRE(dY && X1 && dW && B > 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0 && Cout > 0, "hcp_wgrad_conv3x3_bf16: bad arguments");
    HCP_REQUIRE(C1 > 0 && C1 % 8 == 0 && C2 >= 0 && C2 % 8 == 0 && (C2 == 0 || X2), "hcp_wgrad_conv3x3_bf16: channel counts must be multiples of 8");
    HCP_REQUIRE(Cw > 0 && Cw <= C1 + C2 && ldy % 8 == 0 && ldy >= (Cout + 7) / 8 * 8, "hcp_wgrad_conv3x3_bf16: bad Cw / ldy");
    HCP_REQUIRE((stride == 1 || stride == 2) && (upsample == 0 || upsample == 1) && !(stride == 2 && upsample),
                "hcp_wgrad_conv3x3_bf16: stride must be 1 or 2, upsample 0 or 1");
    HCP_REQUIRE((long)B * Hs * Ws * (C1 > C2 ? C1 : C2) < (1L << 30), "hcp_wgrad_conv3x3_bf16: source tensor too large for 32-bit byte offsets");
    WgradParams p = {};
    p.Y = (const hcp_bf16*)dY; p.ldy = ldy; p.X1 = (const hcp_bf16*)X1; p.X2 = (const hcp_bf16*)X2; p.C1 = C1; p.C2 = C2;
    p.D = dW; p.ldw = 9 * Cw; p.Cw = Cw; p.M = B * Ho * Wo; p.N = Cout; p.K = 9 * (C1 + C2);
    p.Hs = Hs; p.Ws = Ws; p.Ho = Ho; p.Wo = Wo; p.stride = stride; p.up = upsample;
    return launch_wgrad<true>(p, workspace, workspace_bytes, stream);
}

// out[g][n] += sum over the rows of group g of Y[m][n]   (rows_per_group = M: one bias gradient; = Ho*Wo: the
// per-sample gradient of ResnetBlock2D's time-embedding row bias).  out: fp32 [M / rows_per_group][ldo].
HCP_API int hcp_colsum_bf16(const void* Y, int ldy, float* out, int ldo, int M, int N, int rows_per_group, hipStream_t stream) {
    HCP_REQUIRE(Y && out && M > 0 && N > 0 && rows_per_group > 0 && M % rows_per_group == 0, "hcp_colsum_bf16: bad arguments");
    HCP_REQUIRE(N % 8 == 0 && ldy % 8 == 0 && ldy >= N && ldo >= N, "hcp_colsum_bf16: N (%d) and ldy (%d) must be multiples of 8", N, ldy);
    const int groups = M / rows_per_group;
    const int ct = hcp_cdiv(N, 64);
    int splits = hcp_cdiv(1024, ct * groups);
    const int max_splits = hcp_cdiv(rows_per_group, 128);
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int rpb = hcp_cdiv(hcp_cdiv(rows_per_group, splits), 32) * 32;
    splits = hcp_cdiv(rows_per_group, rpb);
    HCP_LAUNCH(colsum_kernel, dim3(ct, splits, groups), dim3(256), 32 * 65 * sizeof(float), stream, (const hcp_bf16*)Y, ldy, out, ldo, M, N,
               rows_per_group, rpb);
    HCP_LAUNCH_CHECK("colsum_kernel");
}
