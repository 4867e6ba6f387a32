// attention.hip — C-ABI entry points and dispatch of the flash-style attention kernels (gfx950, bf16, fp32 accumulation).
//
// Replaces diffusers' CrossAttention core (softmax(Q K^T / sqrt(d)) V; AttnProcessor2_0 -> SDPA, or xformers
// when `enable_xformers` — reference train_ac.py:258-260; call sites unet_struct.txt:17-43) for
// self-attention (Nk = Nq = H*W tokens) and cross-attention (Nk = 77*r text tokens), head_dim 40/64/80/160.
// Tensors stay in the token-major [B, N, heads*d] layout the QKV GEMMs write — no head permutes.
//
// The kernels live in attn_dma.h (forward, dQ, dK/dV: LDS-DMA tile fills, XCD-aware work order, one exp per score);
// here: the fp32 -> bf16 conversion of the query-split dK/dV pass, the choice of
// rows per wave, argument checks.  Backward = dQ kernel (which also produces delta) + dK/dV kernel: no atomics on the
// self-attention path, deterministic.
#include "attn_dma.h"     // second generation: LDS-DMA tiles, one VALU op per score + shared parameter block

namespace {
using namespace hcp_attn;

// fp32 slabs of the query-split dK/dV pass (one per split, [B, Nk, C] each), added in split order -> bf16 outputs (token-major, strided)
HCP_KERNEL(256) attn_dkv_convert_kernel(AttnParams p, int B, int C) {
    const int cv = C / 4;
    const long total = (long)B * p.Nk * cv;
    const size_t slab = (size_t)B * p.Nk * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 4; long r = i / cv; const int n = (int)(r % p.Nk); const int b = (int)(r / p.Nk);
        const size_t e = ((size_t)b * p.Nk + n) * C + c;
        hcp_f32x4 k4 = *(const hcp_f32x4*)(p.dk32 + e);
        hcp_f32x4 v4 = *(const hcp_f32x4*)(p.dv32 + e);
        for (int s = 1; s < p.qsplit; ++s) {
            const hcp_f32x4 ks = *(const hcp_f32x4*)(p.dk32 + s * slab + e); k4 += ks;
            const hcp_f32x4 vs = *(const hcp_f32x4*)(p.dv32 + s * slab + e); v4 += vs;
        }
        hcp_bf16x4 wk, wv;
#pragma unroll
        for (int q = 0; q < 4; ++q) { wk[q] = (short)hcp_f2bf(k4[q]); wv[q] = (short)hcp_f2bf(v4[q]); }
        *(hcp_bf16x4*)(p.dK + (size_t)b * p.k_bs + (size_t)n * p.k_rs + c) = wk;
        *(hcp_bf16x4*)(p.dV + (size_t)b * p.v_bs + (size_t)n * p.v_rs + c) = wv;
    }
}

constexpr int kMinQTilesPerSplit = 16, kSplitTargetWgs = 256;   // cross-attention dK/dV query-loop split (tools/tune_attn_split.py: 8 / 512 gave 78.5 us for the 64x64 backward, 16 / 256 gives 64.0; finer splits lose to the fp32 atomics)
HCP_TUNABLE(int, g_attn_cfg, -1);   // tools: bit0 fwd rows/wave 32 (else 16), bit1 dQ 32, bit2 dK/dV 32, bit3 fwd 8-wave workgroups, bit4 = keep the heuristic for bits 0-3, bits 8-15 / 16-19 = min query tiles per dK/dV split / target workgroups / 256; -1 = heuristic

constexpr int VAR_FWD = VAR_PRODUCT | VAR_LSUM;      // forward: lazy rescale from the row sums (exact fallback inside the kernel)
constexpr int VAR_PRE_BASE = VAR_XCD | VAR_ONES | VAR_PRE;   // Q pre-scaled by its projection: no multiply in front of v_exp_f32
constexpr int VAR_FWD_PRE = VAR_PRE_BASE | VAR_LSUM;
template <int D, int QT, int NW>
int launch_fwd(AttnParams& p, int B, hipStream_t stream) {
    const int n = hcp_cdiv(p.Nq, 16 * QT * NW) * p.H * B;   // one-dimensional: the kernel maps workgroup id -> (query tile, head, batch) XCD-aware
    // (the masked / causal instantiations keep the per-score maximum: short problems, and their scalar-register budget leaves no
    //  room for the fallback pass's DMA descriptors)
    if (p.kbias || p.causal) HCP_LAUNCH((attn2_fwd_kernel<D, QT, true, VAR_PRODUCT, NW>), dim3(n), dim3(64 * NW), fwd_smem<D>(), stream, p);
    else if (p.pre) HCP_LAUNCH((attn2_fwd_kernel<D, QT, false, VAR_FWD_PRE, NW>), dim3(n), dim3(64 * NW), fwd_smem<D>(), stream, p);
    else HCP_LAUNCH((attn2_fwd_kernel<D, QT, false, VAR_FWD, NW>), dim3(n), dim3(64 * NW), fwd_smem<D>(), stream, p);
    HCP_LAUNCH_CHECK("attn_fwd");
}
template <int D, int QT>
int launch_dq(AttnParams& p, int B, hipStream_t stream) {
    const int n = hcp_cdiv(p.Nq, 64 * QT) * p.H * B;
    if (p.kbias || p.causal) HCP_LAUNCH((attn2_bwd_dq_kernel<D, QT, true, VAR_PRODUCT>), dim3(n), dim3(256), fwd_smem<D>(), stream, p);
    else if (p.pre) HCP_LAUNCH((attn2_bwd_dq_kernel<D, QT, false, VAR_PRE_BASE>), dim3(n), dim3(256), fwd_smem<D>(), stream, p);
    else HCP_LAUNCH((attn2_bwd_dq_kernel<D, QT, false, VAR_PRODUCT>), dim3(n), dim3(256), fwd_smem<D>(), stream, p);
    HCP_LAUNCH_CHECK("attn_bwd_dq");
}
// few key tiles (cross-attention: 77 keys): split the query loop of the dK/dV kernel so the grid still fills the chip; every split leaves
// its partial sums as an fp32 slab in the workspace and the convert kernel adds them in split order (no atomics, nothing to clear)
template <int D, int KT>
void plan_dkv(AttnParams& p, int B, float* ws, size_t ws_bytes) {
    const int nkv = hcp_cdiv(p.Nk, 64 * KT);
    const int nqt = hcp_cdiv(p.Nq, KVT);
    int qsplit = 1;
    const long base = (long)nkv * p.H * B;
    const size_t need1 = (size_t)2 * B * p.Nk * p.H * D * sizeof(float);          // one split's dK + dV slab
    if (base < 256 && nqt >= 8 && ws && ws_bytes >= 2 * need1) {
        const bool ovr = g_attn_cfg >= 0 && (g_attn_cfg >> 8);                    // tools: bits 8-15 min tiles, bits 16-19 target / 256
        const int min_tiles = ovr ? ((g_attn_cfg >> 8) & 255) : kMinQTilesPerSplit;
        const int target = ovr ? 256 * ((g_attn_cfg >> 16) & 15) : kSplitTargetWgs;
        qsplit = (int)((target + base - 1) / base);
        if (qsplit > nqt / min_tiles) qsplit = nqt / min_tiles;       // >= min_tiles query tiles per workgroup, else the slabs + convert dominate
        if ((size_t)qsplit > ws_bytes / need1) qsplit = (int)(ws_bytes / need1);
        if (qsplit < 2) qsplit = 1;
    }
    p.qsplit = qsplit;
    if (qsplit > 1) { p.dk32 = ws; p.dv32 = ws + (size_t)qsplit * B * p.Nk * p.H * D; }
}
template <int D, int KT>
int launch_dkv(AttnParams& p, int B, hipStream_t stream) {
    const int nkv = hcp_cdiv(p.Nk, 64 * KT);
    const int qsplit = p.qsplit;
    const int n = nkv * qsplit * p.H * B;
    if (p.kbias || p.causal) HCP_LAUNCH((attn2_bwd_dkv_kernel<D, KT, true, VAR_PRODUCT>), dim3(n), dim3(256), dkv_smem<D>(), stream, p);
    else if (p.pre) HCP_LAUNCH((attn2_bwd_dkv_kernel<D, KT, false, VAR_PRE_BASE>), dim3(n), dim3(256), dkv_smem<D>(), stream, p);
    else HCP_LAUNCH((attn2_bwd_dkv_kernel<D, KT, false, VAR_PRODUCT>), dim3(n), dim3(256), dkv_smem<D>(), stream, p);
    if (qsplit > 1) {
        long tot = (long)B * p.Nk * (p.H * D / 4);
        int g = (int)((tot + 255) / 256); if (g > 2048) g = 2048;
        HCP_LAUNCH(attn_dkv_convert_kernel, dim3(g), dim3(256), 0, stream, p, B, p.H * D);
    }
    HCP_LAUNCH_CHECK("attn_bwd_dkv");
}

// rows per wave: 32 only where the grid still fills the chip twice over AND registers allow >= 2 waves/SIMD
template <int D> constexpr bool kWide = D <= 64;

// Forward shape choice, measured on MI355X (tools/attn_lab): 32 rows per wave where registers allow 3-4 waves per SIMD (d <= 64)
// and the grid still fills the chip; 8-wave workgroups (twice the query rows per K/V tile: half the L2 -> LDS traffic per
// CU) at d = 40 / 80 once there are >= 256 of them.
template <int D>
int run_fwd(AttnParams& p, int B, hipStream_t stream) {
    const long bh = (long)B * p.H;
    bool wide = kWide<D> && bh * hcp_cdiv(p.Nq, 128) >= 512;
    bool w8 = (D == 40 || D == 80) && bh * hcp_cdiv(p.Nq, D == 40 ? 256 : 128) >= 256;
    if (g_attn_cfg >= 0 && !(g_attn_cfg & 16)) { wide = kWide<D> && (g_attn_cfg & 1); w8 = (D == 40 || D == 80) && (g_attn_cfg & 8); }
    if constexpr (D == 40) {
        if (w8) return wide ? launch_fwd<D, 2, 8>(p, B, stream) : launch_fwd<D, 1, 8>(p, B, stream);
        return wide ? launch_fwd<D, 2, 4>(p, B, stream) : launch_fwd<D, 1, 4>(p, B, stream);
    } else if constexpr (D == 80) {
        return w8 ? launch_fwd<D, 1, 8>(p, B, stream) : launch_fwd<D, 1, 4>(p, B, stream);
    } else if constexpr (kWide<D>) {
        return wide ? launch_fwd<D, 2, 4>(p, B, stream) : launch_fwd<D, 1, 4>(p, B, stream);
    } else {
        return launch_fwd<D, 1, 4>(p, B, stream);
    }
}
template <int D>
int run_bwd(AttnParams& p, int B, float* ws, size_t ws_bytes, hipStream_t stream) {
    // (delta = rowsum(dO * O) is produced by the dQ kernel's prologue and read by the dK/dV kernel behind it)
    // measured on MI355X: 32 rows per wave pay off once the grid has >= 512 such workgroups
    bool wq = kWide<D> && (long)B * p.H * hcp_cdiv(p.Nq, 128) >= 512, wk = kWide<D> && (long)B * p.H * hcp_cdiv(p.Nk, 128) >= 512;
    if (g_attn_cfg >= 0 && !(g_attn_cfg & 16)) { wq = kWide<D> && (g_attn_cfg & 2); wk = kWide<D> && (g_attn_cfg & 4); }
    if constexpr (kWide<D>) { if (wk) plan_dkv<D, 2>(p, B, ws, ws_bytes); else plan_dkv<D, 1>(p, B, ws, ws_bytes); }
    else plan_dkv<D, 1>(p, B, ws, ws_bytes);
    int e;
    if constexpr (kWide<D>) { e = wq ? launch_dq<D, 2>(p, B, stream) : launch_dq<D, 1>(p, B, stream); }
    else e = launch_dq<D, 1>(p, B, stream);
    if (e) return e;
    if constexpr (kWide<D>) { return wk ? launch_dkv<D, 2>(p, B, stream) : launch_dkv<D, 1>(p, B, stream); }
    return launch_dkv<D, 1>(p, B, stream);
}

int attn_check(const AttnParams& p, int B, int D) {
    HCP_REQUIRE(B > 0 && p.H > 0 && p.Nq > 0 && p.Nk > 0, "attention: empty problem");
    HCP_REQUIRE(D == 40 || D == 64 || D == 80 || D == 160, "attention: head_dim %d unsupported (40/64/80/160)", D);
    HCP_REQUIRE(p.q_rs % 8 == 0 && p.k_rs % 8 == 0 && p.v_rs % 8 == 0 && p.o_rs % 8 == 0, "attention: row strides must be multiples of 8");
    HCP_REQUIRE(p.q_bs % 8 == 0 && p.k_bs % 8 == 0 && p.v_bs % 8 == 0 && p.o_bs % 8 == 0, "attention: batch strides must be multiples of 8");
    return 0;
}

}  // namespace

#if defined(HCP_TOOLS)
// TOOLS ONLY: bit0 forward / bit1 dQ / bit2 dK,dV use 32 rows per wave, bit3 forward 8-wave workgroups; -1 restores the heuristic.
HCP_API int hcp_debug_set_attention_config(int cfg) { g_attn_cfg = cfg; return 0; }
#endif

// O[b,q,h,:] = softmax_k(scale * Q[b,q,h,:].K[b,k,h,:]) V[b,k,h,:];  lse[b,h,q] = logsumexp of the scaled scores.
// All tensors bf16, token-major: element (b, n, h, c) at  base + b*bs + n*rs + h*D + c.
HCP_API int hcp_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int B, int H, int Nq, int Nk,
                              int D, long q_bs, int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs,
                              float scale, const float* key_bias, long key_bias_bs, int causal, hipStream_t stream) {
    AttnParams p = {};
    p.kbias = key_bias; p.kb_bs = key_bias_bs; p.causal = causal & 1; p.pre = (causal >> 1) & 1; p.B = B;
    p.Q = (const hcp_bf16*)Q; p.K = (const hcp_bf16*)K; p.V = (const hcp_bf16*)V; p.Out = (hcp_bf16*)O; p.lse = lse;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.qsplit = 1;
    HCP_REQUIRE(Q && K && V && O && lse, "hcp_attention_fwd: null pointer");
    HCP_REQUIRE(!(causal & 1) || Nq == Nk, "hcp_attention_fwd: causal masking is defined for self-attention (Nq == Nk)");
    HCP_REQUIRE(!(causal & ~3), "hcp_attention_fwd: unknown flag bits (1 = causal, 2 = Q pre-scaled)");
    HCP_REQUIRE(!p.pre || !(key_bias || p.causal), "hcp_attention_fwd: pre-scaled Q is provided for the unmasked kernels only");
    if (int e = attn_check(p, B, D)) return e;
    switch (D) {
        case 40: return run_fwd<40>(p, B, stream);
        case 64: return run_fwd<64>(p, B, stream);
        case 80: return run_fwd<80>(p, B, stream);
        default: return run_fwd<160>(p, B, stream);
    }
}

// Gradients of hcp_attention_fwd.  delta is a [B,H,Nq] fp32 scratch buffer; workspace (optional, 2*B*Nk*H*D floats)
// lets short-key problems (cross-attention) split the query loop of the dK/dV pass across workgroups.
HCP_API int hcp_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                              float* delta, void* dQ, void* dK, void* dV, int B, int H, int Nq, int Nk, int D, long q_bs,
                              int q_rs, long k_bs, int k_rs, long v_bs, int v_rs, long o_bs, int o_rs, float scale,
                              const float* key_bias, long key_bias_bs, int causal, void* workspace, size_t workspace_bytes,
                              hipStream_t stream) {
    AttnParams p = {};
    p.kbias = key_bias; p.kb_bs = key_bias_bs; p.causal = causal & 1; p.pre = (causal >> 1) & 1; p.B = B;
    p.Q = (const hcp_bf16*)Q; p.K = (const hcp_bf16*)K; p.V = (const hcp_bf16*)V; p.O = (const hcp_bf16*)O;
    p.dO = (const hcp_bf16*)dO; p.lse = (float*)lse; p.delta = delta;
    p.dQ = (hcp_bf16*)dQ; p.dK = (hcp_bf16*)dK; p.dV = (hcp_bf16*)dV;
    p.q_bs = q_bs; p.q_rs = q_rs; p.k_bs = k_bs; p.k_rs = k_rs; p.v_bs = v_bs; p.v_rs = v_rs; p.o_bs = o_bs; p.o_rs = o_rs;
    p.H = H; p.Nq = Nq; p.Nk = Nk; p.scale = scale; p.qsplit = 1;
    HCP_REQUIRE(Q && K && V && O && dO && lse && delta && dQ && dK && dV, "hcp_attention_bwd: null pointer");
    HCP_REQUIRE(!(causal & 1) || Nq == Nk, "hcp_attention_bwd: causal masking is defined for self-attention (Nq == Nk)");
    HCP_REQUIRE(!(causal & ~3), "hcp_attention_bwd: unknown flag bits (1 = causal, 2 = Q pre-scaled)");
    HCP_REQUIRE(!p.pre || !(key_bias || p.causal), "hcp_attention_bwd: pre-scaled Q is provided for the unmasked kernels only");
    if (int e = attn_check(p, B, D)) return e;
    float* ws = (float*)workspace; const size_t wb = workspace ? workspace_bytes : 0;
    switch (D) {
        case 40: return run_bwd<40>(p, B, ws, wb, stream);
        case 64: return run_bwd<64>(p, B, ws, wb, stream);
        case 80: return run_bwd<80>(p, B, ws, wb, stream);
        default: return run_bwd<160>(p, B, ws, wb, stream);
    }
}
