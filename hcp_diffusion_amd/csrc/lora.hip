// lora.hip — LoRA-specific kernels: operand packing and the rank-r weight gradients.
//
// Reference semantics (hcpdiff/models/lora_base_patch.py:59-74, lora_layers_patch.py:31-57):
//   y = x (W + alpha * W_up W_down)^T + b,  alpha = cfg_alpha / rank, W_down[r,in], W_up[out,r].
// Native formulation (side path):  T = x W_down^T ; y = x W^T + T (alpha W_up)^T
//   dW_down = alpha * (dY W_up)^T x = alpha * U^T x,   dW_up = alpha * dY^T T
// Both gradients are skinny reductions over the token dimension M (HBM-bound: x and dY are read once),
// computed with MFMA 16x16x32 (reduction dim = tokens) and accumulated with fp32 atomics straight
// into the flat gradient bucket the optimizer / all-reduce operate on.
#include "hcp_common.h"

namespace {

constexpr int WG_BQ = 128;        // columns of R per block
constexpr int WG_BM = 64;         // token rows per LDS tile
// Row strides (bf16 elements) of the two row-major LDS images: an odd number of 16-byte granules (17 / 5), so that the 16-byte staging
// stores stay aligned and the transposed fragment reads of four consecutive rows fall on distinct banks (the attention images' rule,
// tools/attn_lab/bank_model.py).
constexpr int WG_RS = WG_BQ + 8;
constexpr int WG_LS = 32 + 8;

HCP_DEVICE hcp_bf16x8 wg_join8(hcp_bf16x4 a, hcp_bf16x4 b) {
    hcp_bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { r[i] = a[i]; r[4 + i] = b[i]; }
    return r;
}
// MFMA operand whose k-slots run down 32 token ROWS of a row-major image, for the 16 columns starting at `col`: two ds_read_b64_tr_b16
// (hcp_lds_read_tr4: lane group fg receives rows 4 fg .. 4 fg + 3 of a 16-row block) — the same row -> k-slot order for the L and the R
// image, so the products pair the same tokens.  (Round 5: the fragments were gathered with 16 ds_read_u16 per operand and the tiles
// staged with 8 ds_write_b16 per 16 bytes; the LDS instruction stream, not HBM, set the kernel's time.)
HCP_DEVICE hcp_bf16x8 wg_frag(const hcp_bf16* img, int rs, int row0, int col, int fr, int fg) {
    const hcp_bf16* a = img + (row0 + 4 * fg + (fr >> 2)) * rs + col + 4 * (fr & 3);
    return wg_join8(hcp_lds_read_tr4(a), hcp_lds_read_tr4(a + 16 * rs));
}

// out[p, q] (+)= scale * sum_m L[m, p] * R[m, q]      p < P (<= 32), q < Q
// transpose_out: element (p,q) lives at out[q * ldo + p] instead of out[p * ldo + q]
// lo: 0, or the column offset (32) of the residual half of a split L = (L_hi | L_lo) (hcp_gemm_lora_bf16 with ldt = 64): the product then
// is (L_hi + L_lo)^T R — both halves ride the same staged R tile, two MFMAs per output block instead of one.
struct WgradProb { const hcp_bf16* L; int ldl; int lo; const hcp_bf16* R; int ldr; float* out; int ldo; int Q; int transpose_out; int pcol0; };

HCP_DEVICE void wgrad_block(const WgradProb& pr, int M, int P, float scale, int rows_per_split, int qtile, int split) {
    const hcp_bf16* L = pr.L; const int ldl = pr.ldl; const hcp_bf16* R = pr.R; const int ldr = pr.ldr;
    float* out = pr.out; const int ldo = pr.ldo; const int Q = pr.Q; const int transpose_out = pr.transpose_out;
    const int pcol0 = pr.pcol0;                // this layer's rank slots are columns [pcol0, pcol0 + P) of L
    if (qtile * WG_BQ >= Q) return;          // the pair shares one grid sized for the wider problem
    HCP_DYN_SMEM(smem);
    hcp_bf16* sL = (hcp_bf16*)smem;              // [WG_BM][WG_LS]
    hcp_bf16* sR = sL + WG_BM * WG_LS;           // [WG_BM][WG_RS]
    hcp_bf16* sL2 = sR + WG_BM * WG_RS;          // [WG_BM][WG_LS]: the residual half of a split L
    const int lo = pr.lo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q0 = qtile * WG_BQ;
    const int mb = split * rows_per_split;
    int me = mb + rows_per_split; if (me > M) me = M;
    const int fr = lane & 15, fg = lane >> 4;

    hcp_f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }

    for (int m0 = mb; m0 < me; m0 += WG_BM) {
        // stage L tile: 64 rows x 32 cols = 256 chunks of 8
        {
            int r = tid >> 2, c = (tid & 3) * 8;
            hcp_bf16x8 v = hcp_zero8();
            // only the 8-column pieces that hold this problem's rank slots [pcol0, pcol0 + P): a caller may hand in a column-offset view
            // of a 32-wide T / U (several LoRA blocks on one host), whose last pieces would otherwise run past the row
            const bool live = m0 + r < me && c + 8 > pcol0 && c < pcol0 + P;
            if (live) v = *(const hcp_bf16x8*)(L + (size_t)(m0 + r) * ldl + c);
            *(hcp_bf16x8*)(sL + r * WG_LS + c) = v;
            if (lo) {
                hcp_bf16x8 v2 = hcp_zero8();
                if (live) v2 = *(const hcp_bf16x8*)(L + (size_t)(m0 + r) * ldl + lo + c);
                *(hcp_bf16x8*)(sL2 + r * WG_LS + c) = v2;
            }
        }
        // stage R tile: 64 rows x 128 cols = 1024 chunks
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            int cidx = tid + 256 * it;
            int r = cidx >> 4, c = (cidx & 15) * 8;
            hcp_bf16x8 v = hcp_zero8();
            if (m0 + r < me && q0 + c < Q) v = *(const hcp_bf16x8*)(R + (size_t)(m0 + r) * ldr + q0 + c);
            *(hcp_bf16x8*)(sR + r * WG_RS + c) = v;
        }
        HCP_SYNC();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                                   // two k-steps of 32 token rows
            hcp_bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = wg_frag(sL, WG_LS, ks * 32, i * 16, fr, fg);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = wg_frag(sR, WG_RS, ks * 32, wave * 32 + j * 16, fr, fg);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = hcp_mfma16(fa[i], fb[j], acc[i][j]);
            if (lo) {                                                      // (workgroup-uniform)
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = wg_frag(sL2, WG_LS, ks * 32, i * 16, fr, fg);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = hcp_mfma16(fa[i], fb[j], acc[i][j]);
            }
        }
        HCP_SYNC();
    }
    // lane holds D[p = i*16 + 4*fg + r][q = q0 + wave*32 + j*16 + fr]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = q0 + wave * 32 + j * 16 + fr;
            if (q >= Q) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = i * 16 + 4 * fg + r - pcol0;
                if (p < 0 || p >= P) continue;
                float* dst = transpose_out ? out + (size_t)q * ldo + p : out + (size_t)p * ldo + q;
                hcp_atomic_add(dst, acc[i][j][r] * scale);
            }
        }
}

HCP_KERNEL(256) lora_wgrad_kernel(WgradProb pr0, WgradProb pr1, int M, int P, float scale, int rows_per_split) {
    wgrad_block(blockIdx.z == 0 ? pr0 : pr1, M, P, scale, rows_per_split, blockIdx.x, blockIdx.y);
}

// One launch for the weight gradients of MANY LoRA layers (all 160 of an SD1.5 step): workgroup -> (layer, problem,
// column tile, token split) through a prefix table.  144-byte descriptors, device array:
struct WgradGroupDesc {
    WgradProb down;        // grad_down[r,K] += s U^T x      (56 bytes each: {L, ldl, lo, R, ldr, pad, out, ldo, Q, transpose_out, pcol0})
    WgradProb up;          // grad_up[N,r]  += s dY^T T
    int M, P; float scale; int rows_per_split;
    int qt, splits;        // grid shape of this layer: qt column tiles x splits token ranges x 2 problems
    int block_begin;       // first workgroup index of this layer in the grouped grid
    int pad;
};

HCP_KERNEL(256) lora_wgrad_grouped_kernel(const WgradGroupDesc* descs, int count) {
    // binary search the layer whose block range contains blockIdx.x (wave-uniform)
    int lo = 0, hi = count - 1;
    const int bid = blockIdx.x;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (descs[mid].block_begin <= bid) lo = mid; else hi = mid - 1;
    }
    const WgradGroupDesc d = descs[lo];
    int local = bid - d.block_begin;
    const int per_prob = d.qt * d.splits;
    const int z = local / per_prob; local -= z * per_prob;
    const int split = local / d.qt, qtile = local - split * d.qt;
    wgrad_block(z == 0 ? d.down : d.up, d.M, d.P, d.scale, d.rows_per_split, qtile, split);
}

struct LoraPackDesc {
    const float* w_down;   // [r, K]  fp32 master
    const float* w_up;     // [N, r]
    hcp_bf16* ad;          // [32, K]       rows slot0.. : W_down                         (B operand of T = x Ad^T)
    hcp_bf16* adt;         // [K, 32]       cols slot0.. : alpha * W_down^T               (side-path operand of dX)
    hcp_bf16* bu;          // [Ntot, 32]    rows n0.., cols slot0.. : alpha * W_up        (side-path operand of y)
    hcp_bf16* but;         // [32, Ntot]    rows slot0.., cols n0.. : W_up^T              (B operand of U = dY Bu)
    int K, N, r;
    float alpha;
    int slot0, n0, Ntot, bu_ld;   // placement inside a (possibly shared) operand image; images are zero-initialised once.
                                  // bu_ld: row stride of `bu` in elements (0 = 32); ad / adt / but may be null (image not wanted)
};

HCP_KERNEL(256) lora_pack_kernel(const LoraPackDesc* descs) {
    const LoraPackDesc d = descs[blockIdx.x];
    const int nchunk = gridDim.y, chunk = blockIdx.y;
    const int kper = (d.K + nchunk - 1) / nchunk, k0 = chunk * kper;
    int k1 = k0 + kper; if (k1 > d.K) k1 = d.K;
    for (int i = threadIdx.x; i < d.r * (k1 - k0); i += blockDim.x) {
        int kk = i / d.r, pp = i - kk * d.r;
        int k = k0 + kk;
        float w = d.w_down[(size_t)pp * d.K + k];
        if (d.adt) d.adt[(size_t)k * 32 + d.slot0 + pp] = hcp_f2bf(w * d.alpha);
        if (d.ad) d.ad[(size_t)(d.slot0 + pp) * d.K + k] = hcp_f2bf(w);
    }
    const int bu_ld = d.bu_ld ? d.bu_ld : 32;
    const int nper = (d.N + nchunk - 1) / nchunk, n0 = chunk * nper;
    int n1 = n0 + nper; if (n1 > d.N) n1 = d.N;
    for (int i = threadIdx.x; i < d.r * (n1 - n0); i += blockDim.x) {
        int nn = i / d.r, pp = i - nn * d.r;
        int n = n0 + nn;
        float w = d.w_up[(size_t)n * d.r + pp];
        d.bu[(size_t)(d.n0 + n) * bu_ld + d.slot0 + pp] = hcp_f2bf(w * d.alpha);
        if (d.but) d.but[(size_t)(d.slot0 + pp) * d.Ntot + d.n0 + n] = hcp_f2bf(w);
    }
}

// fp32 [M, C] -> bf16 [M, 2C] = (hi | lo), hi = bf16(v), lo = bf16(v - hi): the split form of a T / U that was produced by a GEMM of
// its own (lora.CtxBatch: every cross-attention layer's x W_down^T from one launch) instead of inside a fused-LoRA kernel.
HCP_KERNEL(256) split_hi_lo_kernel(const float* src, hcp_bf16* dst, long M, int C) {
    const int cv = C / 4;
    const long total = M * cv;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long m = i / cv; const int c = (int)(i - m * cv) * 4;
        const hcp_f32x4 v = *(const hcp_f32x4*)(src + m * C + c);
        hcp_bf16x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned short h = hcp_f2bf(v[q]);
            hi[q] = (short)h; lo[q] = (short)hcp_f2bf(v[q] - hcp_bf2f(h));
        }
        *(hcp_bf16x4*)(dst + m * 2 * C + c) = hi;
        *(hcp_bf16x4*)(dst + m * 2 * C + C + c) = lo;
    }
}

}  // namespace

// dst[M, 2C] (bf16) = (bf16(src) | bf16(src - bf16(src))) for src [M, C] fp32, C % 4 == 0: a rank-r LoRA intermediate (T = x W_down^T,
// U = dY W_up; reference lora_base_patch.py:61-74 never rounds it) carried to 16 mantissa bits as two bf16 MFMA operands.
HCP_API int hcp_split_hi_lo_bf16(const float* src, void* dst, long M, int C, hipStream_t stream) {
    HCP_REQUIRE(src && dst && M > 0 && C > 0 && C % 4 == 0, "hcp_split_hi_lo_bf16: bad arguments (C %% 4 == 0)");
    long g = (M * (C / 4) + 255) / 256; if (g > 2048) g = 2048;
    HCP_LAUNCH(split_hi_lo_kernel, dim3((int)g), dim3(256), 0, stream, src, (hcp_bf16*)dst, M, C);
    HCP_LAUNCH_CHECK("split_hi_lo");
}

static int wgrad_launch(const WgradProb& a, const WgradProb& b, int nprob, int M, int P, float scale, hipStream_t stream) {
    const int qmax = nprob == 2 && b.Q > a.Q ? b.Q : a.Q;
    const int qt = hcp_cdiv(qmax, WG_BQ);
    int splits = hcp_cdiv(1024, qt * nprob);
    int maxs = hcp_cdiv(M, 2 * WG_BM);
    if (splits > maxs) splits = maxs;
    if (splits < 1) splits = 1;
    int rows = hcp_cdiv(hcp_cdiv(M, splits), WG_BM) * WG_BM;
    splits = hcp_cdiv(M, rows);
    size_t smem = (size_t)(2 * WG_BM * WG_LS + WG_BM * WG_RS) * sizeof(hcp_bf16);
    HCP_LAUNCH(lora_wgrad_kernel, dim3(qt, splits, nprob), dim3(256), smem, stream, a, b, M, P, scale, rows);
    HCP_LAUNCH_CHECK("lora_wgrad");
}

// out (fp32, accumulated atomically; caller zeroes the bucket once per step)
//   [p, q] += scale * sum_m L[m,p] R[m,q],  L:[M,32] bf16 (ldl), R:[M,Q] bf16 (ldr), p < P <= 32.
// dW_down: L = U = dY W_up, R = x, out = grad[r,K] (transpose_out=0, ldo=K)
// dW_up  : L = T = x W_down^T, R = dY, out = grad[N,r] (transpose_out=1, ldo=r)
// l_lo: 0, or the column offset of L's residual half (a split T / U of hcp_gemm_lora_bf16, ldt = 64: l_lo = 32, ldl = 64).
HCP_API int hcp_lora_wgrad(const void* L, int ldl, int l_lo, const void* R, int ldr, float* out, int ldo, int M, int P, int Q,
                           float scale, int transpose_out, hipStream_t stream) {
    HCP_REQUIRE(L && R && out && M > 0 && Q > 0, "hcp_lora_wgrad: bad arguments");
    HCP_REQUIRE(P > 0 && P <= 32 && ldl % 8 == 0 && ldl >= 32 && ldr % 8 == 0 && Q % 8 == 0, "hcp_lora_wgrad: P<=32, ldl>=32, 8-aligned leading dims required");
    HCP_REQUIRE(l_lo == 0 || (l_lo % 8 == 0 && l_lo >= 32 && ldl >= l_lo + 32), "hcp_lora_wgrad: l_lo (%d) must leave 32 columns inside ldl (%d)", l_lo, ldl);
    WgradProb a = {(const hcp_bf16*)L, ldl, l_lo, (const hcp_bf16*)R, ldr, out, ldo, Q, transpose_out, 0};
    return wgrad_launch(a, a, 1, M, P, scale, stream);
}

// Both LoRA weight gradients of one layer in ONE launch:
//   grad_down[r,K] += scale * U^T x   (U = dY W_up [M,32], x [M,K])
//   grad_up  [N,r] += scale * dY^T T  (T = x W_down^T [M,32], dY [M,N])
// ldu / ldt: 32 (bf16 U / T) or 64 (split: hi | lo, as hcp_gemm_lora_bf16 writes them with ldt = 64).
HCP_API int hcp_lora_wgrad_pair(const void* U, int ldu, const void* x, int ldx, int K, float* grad_down, const void* T, int ldt, const void* dY,
                                int ldy, int N, float* grad_up, int M, int r, float scale, hipStream_t stream) {
    HCP_REQUIRE(U && x && grad_down && T && dY && grad_up && M > 0 && K > 0 && N > 0, "hcp_lora_wgrad_pair: bad arguments");
    HCP_REQUIRE(r > 0 && r <= 32 && ldx % 8 == 0 && ldy % 8 == 0 && K % 8 == 0 && N % 8 == 0, "hcp_lora_wgrad_pair: r<=32, 8-aligned dims required");
    HCP_REQUIRE((ldu == 32 || ldu == 64) && (ldt == 32 || ldt == 64), "hcp_lora_wgrad_pair: ldu (%d) / ldt (%d) are 32 or 64 (split)", ldu, ldt);
    WgradProb a = {(const hcp_bf16*)U, ldu, ldu == 64 ? 32 : 0, (const hcp_bf16*)x, ldx, grad_down, K, K, 0, 0};
    WgradProb b = {(const hcp_bf16*)T, ldt, ldt == 64 ? 32 : 0, (const hcp_bf16*)dY, ldy, grad_up, r, N, 1, 0};
    return wgrad_launch(a, b, 2, M, r, scale, stream);
}

// Geometry the grouped launch uses for one layer: returns workgroups needed, fills qt / splits / rows_per_split.
HCP_API int hcp_lora_wgrad_group_geometry(int M, int K, int N, int* qt, int* splits, int* rows_per_split) {
    const int qmax = K > N ? K : N;
    const int t = hcp_cdiv(qmax, WG_BQ);
    int s = hcp_cdiv(256, t * 2);                 // grouped: many layers share the grid, fewer splits per layer
    int maxs = hcp_cdiv(M, 4 * WG_BM);
    if (s > maxs) s = maxs;
    if (s < 1) s = 1;
    int rows = hcp_cdiv(hcp_cdiv(M, s), WG_BM) * WG_BM;
    s = hcp_cdiv(M, rows);
    *qt = t; *splits = s; *rows_per_split = rows;
    return t * s * 2;
}
HCP_API int hcp_lora_wgrad_group_desc_bytes(void) { return (int)sizeof(WgradGroupDesc); }

// All layers' LoRA weight gradients in ONE launch.  descs: device array of `count` descriptors (layout: struct
// WgradGroupDesc above; host builders: hcp_diffusion_amd/ops.py), total_blocks = sum of the per-layer workgroup counts.
HCP_API int hcp_lora_wgrad_grouped(const void* descs, int count, int total_blocks, hipStream_t stream) {
    HCP_REQUIRE(descs && count > 0 && total_blocks > 0, "hcp_lora_wgrad_grouped: bad arguments");
    size_t smem = (size_t)(2 * WG_BM * WG_LS + WG_BM * WG_RS) * sizeof(hcp_bf16);
    HCP_LAUNCH(lora_wgrad_grouped_kernel, dim3(total_blocks), dim3(256), smem, stream, (const WgradGroupDesc*)descs, count);
    HCP_LAUNCH_CHECK("lora_wgrad_grouped");
}

// One launch converts the fp32 master LoRA factors of `count` layers into the four bf16 operand
// layouts the GEMMs consume. `descs` is a DEVICE array of 64-byte descriptors:
//   { const float* w_down; const float* w_up; bf16* ad; bf16* adt; bf16* bu; bf16* but; int K; int N; int r; float alpha;
//     int slot0; int n0; int Ntot; int bu_ld; }   (80 bytes; the operand images must be zero-initialised once by the caller;
//   bu_ld = row stride of bu in elements, 0 = 32; adt / but may be null)
HCP_API int hcp_lora_pack(const void* descs, int count, hipStream_t stream) {
    HCP_REQUIRE(descs && count > 0, "hcp_lora_pack: bad arguments");
    HCP_LAUNCH(lora_pack_kernel, dim3(count, 16), dim3(256), 0, stream, (const LoraPackDesc*)descs);
    HCP_LAUNCH_CHECK("lora_pack");
}
HCP_API int hcp_lora_pack_desc_bytes(void) { return (int)sizeof(LoraPackDesc); }
