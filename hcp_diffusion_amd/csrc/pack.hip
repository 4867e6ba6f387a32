// pack.hip — refresh of the bf16 operand copies of trainable host weights after an optimizer step (full fine-tuning).
//
// The fp32 masters live in one flat bucket (fullft.HostBucket); the GEMM / conv kernels read bf16 operands in two
// layouts per layer: row-major [N][K] (forward) and its transpose [K][N] (data gradient) — for 3x3 convolutions per
// tap: [Cout][tap][Cin] and [Cin][tap][Cout].  ONE grouped launch converts and transposes every layer: a workgroup
// looks its 64x64 tile up in a prefix table of 2-D pieces (a Linear is one piece, a 3x3 conv nine).  HBM-bound:
// 4 B read + 2 x 2 B written per parameter (SD1.5: 6.9 GB per step).  The conv (LoCon) LoRA factors use the same launch to
// build their operand images (W_down per tap -> [32][tap][Cin] and [Cin][tap][32]; alpha * W_up -> [Cout][32] and [32][Cout]).
#include "hcp_common.h"

namespace {

struct PackPiece {              // 56 bytes, mirrored by fullft.py (numpy structured dtype)
    const float* src;           // fp32 [rows][src_ld]
    hcp_bf16* dst_rm;           // bf16 [rows][rm_ld]   (may be null)
    hcp_bf16* dst_tr;           // bf16 [cols][tr_ld]   (may be null)
    int rows, cols, src_ld, rm_ld, tr_ld;
    int tile0;                  // index of this piece's first tile in the launch
    int tiles_c;                // tiles along the column dimension
    float scale;                // multiplies every element (LoRA: alpha folded into the W_up operand); 1.0 for plain copies
};
static_assert(sizeof(PackPiece) == 56, "descriptor layout is part of the ABI");

HCP_KERNEL(256) pack_weights_kernel(const PackPiece* pieces, int count) {
    HCP_DYN_SMEM(smem);
    hcp_bf16* tile = (hcp_bf16*)smem;                       // [64][66]
    constexpr int TS = 66;
    int lo = 0, hi = count - 1;                             // last piece with tile0 <= blockIdx.x
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pieces[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const PackPiece pc = pieces[lo];
    const int t = b - pc.tile0;
    const int r0 = (t / pc.tiles_c) * 64, c0 = (t % pc.tiles_c) * 64;
    const int tid = threadIdx.x;
    const int cc = tid & 63, rg = tid >> 6;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int rr = rg + 4 * i;
        const int r = r0 + rr, c = c0 + cc;
        hcp_bf16 v = 0;
        if (r < pc.rows && c < pc.cols) {
            v = hcp_f2bf(pc.src[(size_t)r * pc.src_ld + c] * pc.scale);
            if (pc.dst_rm) pc.dst_rm[(size_t)r * pc.rm_ld + c] = v;
        }
        tile[rr * TS + cc] = v;
    }
    if (!pc.dst_tr) return;                                 // block-uniform
    HCP_SYNC();
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int ci = rg + 4 * i;                          // transposed row = source column
        const int c = c0 + ci, r = r0 + cc;
        if (c < pc.cols && r < pc.rows) pc.dst_tr[(size_t)c * pc.tr_ld + r] = tile[cc * TS + ci];
    }
}

}  // namespace

HCP_API int hcp_pack_piece_bytes(void) { return (int)sizeof(PackPiece); }

// pieces: device array of `count` PackPiece descriptors sorted by tile0; total_tiles = sum of their 64x64 tiles.
HCP_API int hcp_pack_weights(const void* pieces, int count, int total_tiles, hipStream_t stream) {
    HCP_REQUIRE(pieces && count > 0 && total_tiles > 0, "hcp_pack_weights: bad arguments");
    HCP_LAUNCH(pack_weights_kernel, dim3(total_tiles), dim3(256), 64 * 66 * sizeof(hcp_bf16), stream, (const PackPiece*)pieces, count);
    HCP_LAUNCH_CHECK("pack_weights");
}
