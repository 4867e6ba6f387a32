// lora.hip — LoRA-specific kernels: operand packing and the rank-r weight gradients.
//
// Reference semantics (hcpdiff/models/lora_base_patch.py:59-74, lora_layers_patch.py:31-57):
//   y = x (W + alpha * W_up W_down)^T + b,  alpha = cfg_alpha / rank, W_down[r,in], W_up[out,r].
// Native formulation (side path):  T = x W_down^T ; y = x W^T + T (alpha W_up)^T
//   dW_down = alpha * (dY W_up)^T x = alpha * U^T x,   dW_up = alpha * dY^T T
// Both gradients are skinny reductions over the token dimension M (HBM-bound: x and dY are read once),
// computed with MFMA 16x16x32 (reduction dim = tokens).  NO ATOMICS (round 6): every workgroup owns one (column tile, token range)
// and stores its fp32 partial [P x 128] as a slab in the caller's workspace; a second kernel sums a tile's slabs IN SPLIT ORDER and adds
// scale * sum into the flat gradient bucket the optimizer / all-reduce operate on — one thread per gradient element, so the LoRA
// gradients of a step are bit-reproducible run to run (rounds 1-5 added the partials with global_atomic_add_f32 in arrival order).
#include "hcp_common.h"

namespace {

constexpr int WG_BQ = 128;        // columns of R per block
constexpr int WG_BM = 64;         // token rows per LDS tile
// Row strides (bf16 elements) of the two row-major LDS images: an odd number of 16-byte granules (17 / 5), so that the 16-byte staging
// stores stay aligned and the transposed fragment reads of four consecutive rows fall on distinct banks (the attention images' rule,
// tools/attn_lab/bank_model.py).
constexpr int WG_RS = WG_BQ + 8;
constexpr int WG_LS = 32 + 8;
#ifndef HCP_WGRAD_SLAB_ROWS
#define HCP_WGRAD_SLAB_ROWS 64
#endif
constexpr int WGRAD_SLAB_ROWS = HCP_WGRAD_SLAB_ROWS;   // token rows per rank column a split must stream to earn its slab (slab bytes <= 1/16 of its operand bytes)

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

// slab: this workgroup's own [P x 128] fp32 tile in the workspace, laid out like the output it feeds: [p][128 q], or [128 q][P] when
// transpose_out (a lane's four consecutive p are then 16 contiguous bytes).  Every (p < P, q < Q) of the tile is written exactly once.
// slab == nullptr (the layer has ONE token range): the workgroup owns its gradient elements outright and adds scale * acc itself.
HCP_DEVICE void wgrad_block(const WgradProb& pr, int M, int P, float scale, int rows_per_split, int qtile, int split, float* slab) {
    const hcp_bf16* L = pr.L; const int ldl = pr.ldl; const hcp_bf16* R = pr.R; const int ldr = pr.ldr;
    const int Q = pr.Q; const int transpose_out = pr.transpose_out;
    const int pcol0 = pr.pcol0;                // this layer's rank slots are columns [pcol0, pcol0 + P) of L
    if (qtile * WG_BQ >= Q) return;          // (the per-layer pair launch shares one grid sized for the wider problem)
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
            const int ql = q - q0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = i * 16 + 4 * fg + r - pcol0;
                if (p < 0 || p >= P) continue;
                if (slab) slab[transpose_out ? ql * P + p : p * WG_BQ + ql] = acc[i][j][r];
                else {
                    float* dst = transpose_out ? pr.out + (size_t)q * pr.ldo + p : pr.out + (size_t)p * pr.ldo + q;
                    *dst += acc[i][j][r] * scale;
                }
            }
        }
}

// out[tile] += scale * (slab_0 + slab_1 + ... + slab_{splits-1}), summed in that order by ONE thread per element (deterministic; the
// read-modify-write of `out` needs no atomic: within a launch a gradient element belongs to one tile of one problem).
// slabs: the layer-problem's first slab (split 0, column tile 0); consecutive column tiles are P*128 floats apart, splits qt tiles apart.
HCP_DEVICE void wgrad_reduce_tile(const WgradProb& pr, int P, float scale, int qt, int splits, int qtile, const float* slabs) {
    const int q0 = qtile * WG_BQ;
    if (q0 >= pr.Q || splits == 1) return;          // one token range: the producing workgroup already added its tile
    const int tile = P * WG_BQ;
    const float* s0 = slabs + (size_t)qtile * tile;
    const size_t sstride = (size_t)qt * tile;
    float* out = pr.out; const int ldo = pr.ldo; const int Q = pr.Q;
    const bool vec = (!pr.transpose_out || P % 4 == 0) && ldo % 4 == 0 && (((size_t)out) & 15) == 0;
    if (vec) {
        for (int e4 = threadIdx.x; e4 < tile / 4; e4 += 256) {
            const int e = 4 * e4;
            int p, ql;
            if (pr.transpose_out) { ql = e / P; p = e - ql * P; } else { p = e / WG_BQ; ql = e - p * WG_BQ; }
            if (q0 + ql >= Q) continue;                                       // Q % 8 == 0: a 4-run of q is valid as a whole
            hcp_f32x4 a = *(const hcp_f32x4*)(s0 + e);
            for (int s = 1; s < splits; ++s) { const hcp_f32x4 b = *(const hcp_f32x4*)(s0 + s * sstride + e); a += b; }
            float* dst = pr.transpose_out ? out + (size_t)(q0 + ql) * ldo + p : out + (size_t)p * ldo + q0 + ql;
            hcp_f32x4 o = *(hcp_f32x4*)dst;
            o += a * scale;
            *(hcp_f32x4*)dst = o;
        }
    } else {
        for (int e = threadIdx.x; e < tile; e += 256) {
            int p, ql;
            if (pr.transpose_out) { ql = e / P; p = e - ql * P; } else { p = e / WG_BQ; ql = e - p * WG_BQ; }
            if (q0 + ql >= Q) continue;
            float a = s0[e];
            for (int s = 1; s < splits; ++s) a += s0[s * sstride + e];
            float* dst = pr.transpose_out ? out + (size_t)(q0 + ql) * ldo + p : out + (size_t)p * ldo + q0 + ql;
            *dst += a * scale;
        }
    }
}

HCP_KERNEL(256) lora_wgrad_kernel(WgradProb pr0, WgradProb pr1, int M, int P, float scale, int rows_per_split, float* ws) {
    const int local = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;       // (problem, split, column tile)
    wgrad_block(blockIdx.z == 0 ? pr0 : pr1, M, P, scale, rows_per_split, blockIdx.x, blockIdx.y,
                gridDim.y == 1 ? nullptr : ws + (size_t)local * P * WG_BQ);
}
HCP_KERNEL(256) lora_wgrad_reduce_kernel(WgradProb pr0, WgradProb pr1, int P, float scale, int qt, int splits, const float* ws) {
    wgrad_reduce_tile(blockIdx.y == 0 ? pr0 : pr1, P, scale, qt, splits, blockIdx.x, ws + (size_t)blockIdx.y * qt * splits * P * WG_BQ);
}

// One launch for the weight gradients of MANY LoRA layers (all 160 of an SD1.5 step): workgroup -> (layer, problem,
// column tile, token split) through a prefix table.  152-byte descriptors, device array:
struct WgradGroupDesc {
    WgradProb down;        // grad_down[r,K] += s U^T x      (56 bytes each: {L, ldl, lo, R, ldr, pad, out, ldo, Q, transpose_out, pcol0})
    WgradProb up;          // grad_up[N,r]  += s dY^T T
    int M, P; float scale; int rows_per_split;
    int qt0, qt1, splits;  // grid shape of this layer: (qt0 column tiles of grad_down + qt1 of grad_up) x splits token ranges
    int block_begin;       // first workgroup index of this layer in the grouped grid; also its first slab (a slab = P x 128 floats ...
    int slab_begin;        // ... at workspace + slab_begin * 512 bytes: layers of different rank share the table, so units of 128 floats)
    int tile_begin;        // first workgroup index of this layer in the reduce grid (qt0 + qt1 tiles per layer)
};

HCP_DEVICE int wgrad_find(const WgradGroupDesc* descs, int count, int bid, bool tiles) {
    int lo = 0, hi = count - 1;                       // binary search (wave-uniform) for the layer whose range contains bid
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if ((tiles ? descs[mid].tile_begin : descs[mid].block_begin) <= bid) lo = mid; else hi = mid - 1;
    }
    return lo;
}

HCP_KERNEL(256) lora_wgrad_grouped_reduce_kernel(const WgradGroupDesc* descs, int count, const float* ws) {
    const WgradGroupDesc d = descs[wgrad_find(descs, count, blockIdx.x, true)];
    const int local = blockIdx.x - d.tile_begin;
    const int z = local >= d.qt0, qtile = z ? local - d.qt0 : local, qt = z ? d.qt1 : d.qt0;
    wgrad_reduce_tile(z == 0 ? d.down : d.up, d.P, d.scale, qt, d.splits, qtile,
                      ws + ((size_t)d.slab_begin + (size_t)z * d.qt0 * d.splits * d.P) * WG_BQ);
}

HCP_KERNEL(256) lora_wgrad_grouped_kernel(const WgradGroupDesc* descs, int count, float* ws) {
    const int bid = blockIdx.x;
    const WgradGroupDesc d = descs[wgrad_find(descs, count, bid, false)];
    int local = bid - d.block_begin;
    float* slab = ws + ((size_t)d.slab_begin + (size_t)local * d.P) * WG_BQ;       // slab index = (problem, split, column tile) = local
    const int z = local >= d.qt0 * d.splits;
    const int qt = z ? d.qt1 : d.qt0;
    if (z) local -= d.qt0 * d.splits;
    const int split = local / qt, qtile = local - split * qt;
    wgrad_block(z == 0 ? d.down : d.up, d.M, d.P, d.scale, d.rows_per_split, qtile, split, d.splits == 1 ? nullptr : slab);
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

static int wgrad_launch(const WgradProb& a, const WgradProb& b, int nprob, int M, int P, float scale, void* ws, size_t ws_bytes,
                        hipStream_t stream) {
    const int qmax = nprob == 2 && b.Q > a.Q ? b.Q : a.Q;
    const int qt = hcp_cdiv(qmax, WG_BQ);
    int splits = hcp_cdiv(1024, qt * nprob);
    int maxs = hcp_cdiv(M, 2 * WG_BM);
    if (splits > maxs) splits = maxs;
    if (splits > M / (WGRAD_SLAB_ROWS * P)) splits = M / (WGRAD_SLAB_ROWS * P);
    if (splits < 1) splits = 1;
    int rows = hcp_cdiv(hcp_cdiv(M, splits), WG_BM) * WG_BM;
    splits = hcp_cdiv(M, rows);
    const size_t need = splits == 1 ? 0 : (size_t)qt * splits * nprob * P * WG_BQ * sizeof(float);
    HCP_REQUIRE(need == 0 || (ws && (((size_t)ws) & 15) == 0 && ws_bytes >= need),
                "lora_wgrad: workspace of %zu bytes (16-byte aligned) required, got %zu", need, ws_bytes);
    size_t smem = (size_t)(2 * WG_BM * WG_LS + WG_BM * WG_RS) * sizeof(hcp_bf16);
    HCP_LAUNCH(lora_wgrad_kernel, dim3(qt, splits, nprob), dim3(256), smem, stream, a, b, M, P, scale, rows, (float*)ws);
    if (splits > 1)
        HCP_LAUNCH(lora_wgrad_reduce_kernel, dim3(qt, nprob), dim3(256), 0, stream, a, b, P, scale, qt, splits, (const float*)ws);
    HCP_LAUNCH_CHECK("lora_wgrad");
}

// out (fp32, out += ...; the caller zeroes the bucket once per step; no atomics: partial slabs in `workspace`, then an ordered reduce)
//   [p, q] += scale * sum_m L[m,p] R[m,q],  L:[M,32] bf16 (ldl), R:[M,Q] bf16 (ldr), p < P <= 32.
// dW_down: L = U = dY W_up, R = x, out = grad[r,K] (transpose_out=0, ldo=K)
// dW_up  : L = T = x W_down^T, R = dY, out = grad[N,r] (transpose_out=1, ldo=r)
// l_lo: 0, or the column offset of L's residual half (a split T / U of hcp_gemm_lora_bf16, ldt = 64: l_lo = 32, ldl = 64).
HCP_API int hcp_lora_wgrad(const void* L, int ldl, int l_lo, const void* R, int ldr, float* out, int ldo, int M, int P, int Q,
                           float scale, int transpose_out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    HCP_REQUIRE(L && R && out && M > 0 && Q > 0, "hcp_lora_wgrad: bad arguments");
    HCP_REQUIRE(P > 0 && P <= 32 && ldl % 8 == 0 && ldl >= 32 && ldr % 8 == 0 && Q % 8 == 0, "hcp_lora_wgrad: P<=32, ldl>=32, 8-aligned leading dims required");
    HCP_REQUIRE(l_lo == 0 || (l_lo % 8 == 0 && l_lo >= 32 && ldl >= l_lo + 32), "hcp_lora_wgrad: l_lo (%d) must leave 32 columns inside ldl (%d)", l_lo, ldl);
    WgradProb a = {(const hcp_bf16*)L, ldl, l_lo, (const hcp_bf16*)R, ldr, out, ldo, Q, transpose_out, 0};
    return wgrad_launch(a, a, 1, M, P, scale, workspace, workspace_bytes, stream);
}

// Both LoRA weight gradients of one layer in ONE launch:
//   grad_down[r,K] += scale * U^T x   (U = dY W_up [M,32], x [M,K])
//   grad_up  [N,r] += scale * dY^T T  (T = x W_down^T [M,32], dY [M,N])
// ldu / ldt: 32 (bf16 U / T) or 64 (split: hi | lo, as hcp_gemm_lora_bf16 writes them with ldt = 64).
HCP_API int hcp_lora_wgrad_pair(const void* U, int ldu, const void* x, int ldx, int K, float* grad_down, const void* T, int ldt, const void* dY,
                                int ldy, int N, float* grad_up, int M, int r, float scale, void* workspace, size_t workspace_bytes,
                                hipStream_t stream) {
    HCP_REQUIRE(U && x && grad_down && T && dY && grad_up && M > 0 && K > 0 && N > 0, "hcp_lora_wgrad_pair: bad arguments");
    HCP_REQUIRE(r > 0 && r <= 32 && ldx % 8 == 0 && ldy % 8 == 0 && K % 8 == 0 && N % 8 == 0, "hcp_lora_wgrad_pair: r<=32, 8-aligned dims required");
    HCP_REQUIRE((ldu == 32 || ldu == 64) && (ldt == 32 || ldt == 64), "hcp_lora_wgrad_pair: ldu (%d) / ldt (%d) are 32 or 64 (split)", ldu, ldt);
    WgradProb a = {(const hcp_bf16*)U, ldu, ldu == 64 ? 32 : 0, (const hcp_bf16*)x, ldx, grad_down, K, K, 0, 0};
    WgradProb b = {(const hcp_bf16*)T, ldt, ldt == 64 ? 32 : 0, (const hcp_bf16*)dY, ldy, grad_up, r, N, 1, 0};
    return wgrad_launch(a, b, 2, M, r, scale, workspace, workspace_bytes, stream);
}

// Geometry the grouped launch uses for one layer of rank P: returns workgroups needed, fills the column tiles of the two problems,
// splits and rows_per_split.  target = workgroups aimed at for this layer (the grid is shared by every layer of the model: the host
// divides what fills the chip by the number of layers).  Every token range beyond the first costs one [P x 128] fp32 slab written and
// read back per column tile against rows * 128 bf16 streamed in: the split count is capped where the slabs reach 1/16 of the
// operand bytes (M / (64 P)); a layer with one range writes its gradient itself and needs no slab.
HCP_API int hcp_lora_wgrad_group_geometry(int M, int K, int N, int P, int target, int* qt_down, int* qt_up, int* splits, int* rows_per_split) {
    const int t0 = hcp_cdiv(K, WG_BQ), t1 = hcp_cdiv(N, WG_BQ);
    if (P < 1) P = 1;
    int s = hcp_cdiv(target > 0 ? target : 256, t0 + t1);
    int maxs = hcp_cdiv(M, 4 * WG_BM);
    if (s > maxs) s = maxs;
    if (s > M / (WGRAD_SLAB_ROWS * P)) s = M / (WGRAD_SLAB_ROWS * P);
    if (s < 1) s = 1;
    int rows = hcp_cdiv(hcp_cdiv(M, s), WG_BM) * WG_BM;
    s = hcp_cdiv(M, rows);
    *qt_down = t0; *qt_up = t1; *splits = s; *rows_per_split = rows;
    return (t0 + t1) * s;
}
HCP_API int hcp_lora_wgrad_group_desc_bytes(void) { return (int)sizeof(WgradGroupDesc); }

// All layers' LoRA weight gradients in TWO launches (partials, ordered reduce).  descs: device array of `count` descriptors (layout:
// struct WgradGroupDesc above; host builder: hcp_diffusion_amd/kernels.py), total_blocks = sum of the per-layer workgroup counts,
// total_tiles = sum of qt0 + qt1, workspace >= slab_units * 512 bytes where slab_units = sum of blocks * P.  Two descriptors of one call must
// not name the same gradient rows (the reduce adds without atomics); consecutive calls on one stream may.
HCP_API int hcp_lora_wgrad_grouped(const void* descs, int count, int total_blocks, int total_tiles, long slab_units, void* workspace,
                                   size_t workspace_bytes, hipStream_t stream) {
    HCP_REQUIRE(descs && count > 0 && total_blocks > 0 && total_tiles > 0 && slab_units >= 0, "hcp_lora_wgrad_grouped: bad arguments");
    const size_t need = (size_t)slab_units * WG_BQ * sizeof(float);
    HCP_REQUIRE(need == 0 || (workspace && (((size_t)workspace) & 15) == 0 && workspace_bytes >= need),
                "hcp_lora_wgrad_grouped: workspace of %zu bytes (16-byte aligned) required, got %zu", need, workspace_bytes);
    size_t smem = (size_t)(2 * WG_BM * WG_LS + WG_BM * WG_RS) * sizeof(hcp_bf16);
    HCP_LAUNCH(lora_wgrad_grouped_kernel, dim3(total_blocks), dim3(256), smem, stream, (const WgradGroupDesc*)descs, count, (float*)workspace);
    if (slab_units > 0)                             // (no layer with more than one token range: every workgroup added its own tile)
        HCP_LAUNCH(lora_wgrad_grouped_reduce_kernel, dim3(total_tiles), dim3(256), 0, stream, (const WgradGroupDesc*)descs, count,
                   (const float*)workspace);
    HCP_LAUNCH_CHECK("lora_wgrad_grouped_reduce");
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
