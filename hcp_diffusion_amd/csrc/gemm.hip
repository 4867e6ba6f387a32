// gemm.hip — bf16 MFMA GEMM (NT) and 3x3 implicit-GEMM convolution for gfx950.
//
//   D[M,N] = alpha * ( A[M,K] * B[N,K]^T  +  A2[M,K2] * B2[N,K2]^T ) + bias[N]
//            + rowbias[m / rows_per_group, N] + residual[M,N]
//
// * A, B are K-contiguous bf16.  The optional (A2,B2) pair extends the reduction: it is the
//   LoRA low-rank side path  y = x W^T + (x A^T)(alpha B)^T  (replaces the reference's
//   per-forward weight merge W + alpha*B@A, lora_base_patch.py:61-74) and, on the conv path,
//   nothing else is needed for the concat-free skip connection (two source tensors).
// * MODE 1/2 replace the A operand by an on-the-fly im2col gather of an NHWC tensor:
//   3x3 forward (stride 1/2, fused nearest-2x upsample, two-pointer channel concat) and
//   3x3 data-gradient (transposed gather, stride 1/2).  Replaces F.conv2d reached through
//   diffusers ResnetBlock2D/Downsample2D/Upsample2D.
// * 256 threads = 4 waves, BK=64, LDS double buffer (row stride 72 bf16 = conflict-free
//   ds_read_b128), register-staged prefetch of the next K tile, one barrier per K tile.
//   MFMA is issued with swapped operands so each lane owns 4 consecutive N of one row:
//   8-byte bf16x4 stores and float4 bias loads in the epilogue.
#include "hcp_common.h"

namespace {

struct ConvDesc {
    const hcp_bf16* X1; int C1;   // first source tensor  [B, Hs, Ws, C1]
    const hcp_bf16* X2; int C2;   // optional second source (channel concat), else null/0
    int Hs, Ws;                   // source spatial dims (memory)
    int Ho, Wo;                   // output spatial dims (rows of the implicit A matrix)
    int stride;                   // 1 or 2
    int up;                       // 1: source is nearest-upsampled 2x before the conv (fwd only)
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
    float alpha;
    int tiles_m;
    ConvDesc cv;
};

constexpr int BK = 64;
constexpr int LDS_STRIDE = BK + 8;  // bf16 elements per LDS row

template <int BM, int BN, int MODE>
HCP_KERNEL(256) gemm_kernel(GemmParams p) {
    constexpr int WTM = BM / 2, WTN = BN / 2;     // 2x2 waves
    constexpr int TM = WTM / 16, TN = WTN / 16;   // MFMA tiles per wave
    constexpr int A_IT = BM / 32, B_IT = BN / 32; // 16-byte chunks per thread per K tile
    HCP_DYN_SMEM(smem);
    hcp_bf16* lds = (hcp_bf16*)smem;
    constexpr int A_ELEMS = BM * LDS_STRIDE, B_ELEMS = BN * LDS_STRIDE;
    // layout: [buf0: A | B][buf1: A | B]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile_m = blockIdx.x % p.tiles_m, tile_n = blockIdx.x / p.tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kc = tid & 7;        // which 8-element chunk of the 64-wide K tile
    const int lrow = tid >> 3;     // 0..31

    const int nk1 = (p.K + BK - 1) / BK;
    const int nk2 = (p.K2 + BK - 1) / BK;
    const int nk = nk1 + nk2;

    // ---- per-thread row descriptors for the A operand
    int a_pix[A_IT];          // MODE 0: unused; conv: packed (b<<20 | py<<10 | px), or -1 if row >= M
    if (MODE != 0) {
#pragma unroll
        for (int i = 0; i < A_IT; ++i) {
            int m = m0 + lrow + 32 * i;
            if (m < p.M) {
                int hw = p.cv.Ho * p.cv.Wo;
                int b = m / hw; int rem = m - b * hw;
                int py = rem / p.cv.Wo; int px = rem - py * p.cv.Wo;
                a_pix[i] = (b << 20) | (py << 10) | px;
            } else a_pix[i] = -1;
        }
    }
    const int Ctot = p.cv.C1 + p.cv.C2;

    hcp_bf16x8 ra[A_IT], rb[B_IT];

    auto load_tile = [&](int kt) {
        const bool ext = kt >= nk1;
        const int k = (ext ? (kt - nk1) : kt) * BK + kc * 8;
        const int klim = ext ? p.K2 : p.K;
        // ---- B operand (always plain row-major [N, K])
        {
            const hcp_bf16* Bp = ext ? p.B2 : p.B; const int ld = ext ? p.ldb2 : p.ldb;
#pragma unroll
            for (int i = 0; i < B_IT; ++i) {
                int n = n0 + lrow + 32 * i;
                if (n < p.N && k < klim) rb[i] = *(const hcp_bf16x8*)(Bp + (size_t)n * ld + k);
                else rb[i] = hcp_zero8();
            }
        }
        // ---- A operand
        if (MODE == 0 || ext) {
            const hcp_bf16* Ap = ext ? p.A2 : p.A; const int ld = ext ? p.lda2 : p.lda;
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                int m = m0 + lrow + 32 * i;
                if (m < p.M && k < klim) ra[i] = *(const hcp_bf16x8*)(Ap + (size_t)m * ld + k);
                else ra[i] = hcp_zero8();
            }
        } else {
            // im2col: k = tap * Ctot + ci, tap = ky*3 + kx
            int tap = k / Ctot; int ci = k - tap * Ctot;
            int ky = tap / 3, kx = tap - ky * 3;
            const hcp_bf16* src; int cs, c;
            if (ci < p.cv.C1) { src = p.cv.X1; cs = p.cv.C1; c = ci; }
            else { src = p.cv.X2; cs = p.cv.C2; c = ci - p.cv.C1; }
#pragma unroll
            for (int i = 0; i < A_IT; ++i) {
                hcp_bf16x8 v = hcp_zero8();
                int pix = a_pix[i];
                if (pix >= 0 && k < klim) {
                    int b = pix >> 20, py = (pix >> 10) & 1023, px = pix & 1023;
                    int sy, sx; bool ok;
                    if (MODE == 1) {       // forward gather
                        sy = py * p.cv.stride + ky - 1; sx = px * p.cv.stride + kx - 1;
                        int He = p.cv.Hs << p.cv.up, We = p.cv.Ws << p.cv.up;
                        ok = sy >= 0 && sy < He && sx >= 0 && sx < We;
                        sy >>= p.cv.up; sx >>= p.cv.up;
                    } else {               // data-gradient gather: source pixel o with o*stride + k - 1 == p
                        int ty = py + 1 - ky, tx = px + 1 - kx;
                        ok = ty >= 0 && tx >= 0;
                        if (p.cv.stride == 2) { ok = ok && ((ty | tx) & 1) == 0; ty >>= 1; tx >>= 1; }
                        ok = ok && ty < p.cv.Hs && tx < p.cv.Ws;
                        sy = ty; sx = tx;
                    }
                    if (ok) v = *(const hcp_bf16x8*)(src + ((size_t)(b * p.cv.Hs + sy) * p.cv.Ws + sx) * cs + c);
                }
                ra[i] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        hcp_bf16* la = lds + buf * (A_ELEMS + B_ELEMS);
        hcp_bf16* lb = la + A_ELEMS;
#pragma unroll
        for (int i = 0; i < A_IT; ++i) *(hcp_bf16x8*)(la + (lrow + 32 * i) * LDS_STRIDE + kc * 8) = ra[i];
#pragma unroll
        for (int i = 0; i < B_IT; ++i) *(hcp_bf16x8*)(lb + (lrow + 32 * i) * LDS_STRIDE + kc * 8) = rb[i];
    };

    hcp_f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { hcp_f32x4 z = {0.f, 0.f, 0.f, 0.f}; acc[i][j] = z; }

    load_tile(0);
    store_tile(0);
    HCP_SYNC();

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const hcp_bf16* la = lds + cur * (A_ELEMS + B_ELEMS);
        const hcp_bf16* lb = la + A_ELEMS;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            hcp_bf16x8 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                fa[i] = *(const hcp_bf16x8*)(la + (wm * WTM + i * 16 + fr) * LDS_STRIDE + (ks * 4 + fg) * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                fb[j] = *(const hcp_bf16x8*)(lb + (wn * WTN + j * 16 + fr) * LDS_STRIDE + (ks * 4 + fg) * 8);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = hcp_mfma16(fb[j], fa[i], acc[i][j]);   // swapped: lane owns 4 consecutive n
        }
        if (kt + 1 < nk) store_tile(cur ^ 1);
        HCP_SYNC();
    }

    // ---- epilogue: lane holds D[m = .. + fr][n = .. + 4*fg + r]
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm * WTM + i * 16 + fr;
        if (m >= p.M) continue;
        const float* rbp = p.rowbias ? p.rowbias + (size_t)(m / p.rows_per_group) * p.rowbias_ld : nullptr;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 16 + 4 * fg;
            if (n >= p.N) continue;
            hcp_f32x4 v = acc[i][j] * p.alpha;
            if (p.bias) v += *(const hcp_f32x4*)(p.bias + n);
            if (rbp) v += *(const hcp_f32x4*)(rbp + n);
            if (p.residual) {
                hcp_bf16x4 r = *(const hcp_bf16x4*)(p.residual + (size_t)m * p.ldr + n);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += hcp_bf2f((unsigned short)r[q]);
            }
            if (p.out_f32) {
                *(hcp_f32x4*)((float*)p.D + (size_t)m * p.ldd + n) = v;
            } else {
                hcp_bf16x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = (short)hcp_f2bf(v[q]);
                *(hcp_bf16x4*)((hcp_bf16*)p.D + (size_t)m * p.ldd + n) = o;
            }
        }
    }
}

template <int BM, int BN, int MODE>
int launch_gemm(GemmParams& p, hipStream_t stream) {
    p.tiles_m = hcp_cdiv(p.M, BM);
    const int tiles_n = hcp_cdiv(p.N, BN);
    const size_t smem = (size_t)2 * (BM + BN) * LDS_STRIDE * sizeof(hcp_bf16);
    HCP_LAUNCH((gemm_kernel<BM, BN, MODE>), dim3(p.tiles_m * tiles_n), dim3(256), smem, stream, p);
    HCP_LAUNCH_CHECK("gemm_kernel");
}

template <int MODE>
int dispatch_gemm(GemmParams& p, hipStream_t stream) {
    // Tile choice: fill >= 256 CUs where the problem allows it.
    const long t128 = (long)hcp_cdiv(p.M, 128) * hcp_cdiv(p.N, 128);
    const long t12864 = (long)hcp_cdiv(p.M, 128) * hcp_cdiv(p.N, 64);
    if (p.N > 64 && t128 >= 256 && (p.N % 128 == 0 || p.N >= 1024)) return launch_gemm<128, 128, MODE>(p, stream);
    if (t12864 >= 256) return launch_gemm<128, 64, MODE>(p, stream);
    return launch_gemm<64, 64, MODE>(p, stream);
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

}  // namespace

// Replaces: torch.mm(x2d, (W_host + dW)^T) + bias  (reference lora_layers_patch.py:50-57),
// nn.Linear / 1x1 nn.Conv2d forward and their input-gradient (dX = dY W) in diffusers' UNet.
HCP_API int hcp_gemm_bf16(const void* A, int lda, const void* B, int ldb, void* D, int ldd, int M, int N, int K,
                          const void* A2, int lda2, const void* B2, int ldb2, int K2, const float* bias,
                          const float* rowbias, int rowbias_ld, int rows_per_group, const void* residual, int ldr,
                          float alpha, int out_f32, hipStream_t stream) {
    GemmParams p = {};
    p.A = (const hcp_bf16*)A; p.lda = lda; p.B = (const hcp_bf16*)B; p.ldb = ldb;
    p.A2 = (const hcp_bf16*)A2; p.lda2 = lda2; p.B2 = (const hcp_bf16*)B2; p.ldb2 = ldb2; p.K2 = K2;
    p.M = M; p.N = N; p.K = K; p.D = D; p.ldd = ldd; p.out_f32 = out_f32;
    p.bias = bias; p.rowbias = rowbias; p.rowbias_ld = rowbias_ld; p.rows_per_group = rows_per_group;
    p.residual = (const hcp_bf16*)residual; p.ldr = ldr; p.alpha = alpha;
    HCP_REQUIRE(A && B && D, "hcp_gemm_bf16: null operand");
    HCP_REQUIRE(lda % 8 == 0, "hcp_gemm_bf16: lda (%d) must be a multiple of 8", lda);
    if (int e = check_common(p)) return e;
    return dispatch_gemm<0>(p, stream);
}

// Replaces F.conv2d(x, W[Cout,Cin,3,3], stride, padding=1) on NHWC bf16 activations:
// mode 0 = forward  (Wp packed [Cout][ky][kx][Cin1+Cin2]),   output [B,Ho,Wo,Cout]
// mode 1 = data gradient (Wp packed [Cin][ky][kx][Cout]), source = dY [B,Hs,Ws,Cout], output [B,Ho,Wo,Cin]
// (diffusers ResnetBlock2D.conv1/conv2, Downsample2D.conv (stride 2), Upsample2D = nearest-2x + conv,
//  skip-concat inputs of the up blocks; structure: reference cfgs/unet_struct.txt:92-114,390-393.)
HCP_API int hcp_conv3x3_bf16(const void* X1, int C1, const void* X2, int C2, int Bn, int Hs, int Ws, int Ho, int Wo,
                             int mode, int stride, int upsample, const void* Wp, int Cout, void* D, int ldd,
                             const float* bias, const float* rowbias, int rowbias_ld, const void* residual, int ldr,
                             int out_f32, hipStream_t stream) {
    GemmParams p = {};
    HCP_REQUIRE(X1 && Wp && D, "hcp_conv3x3_bf16: null operand");
    HCP_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && (C2 == 0 || X2), "hcp_conv3x3_bf16: channels must be multiples of 8");
    HCP_REQUIRE(stride == 1 || stride == 2, "hcp_conv3x3_bf16: stride must be 1 or 2");
    HCP_REQUIRE(mode == 0 || (mode == 1 && upsample == 0 && C2 == 0), "hcp_conv3x3_bf16: bad mode/options");
    HCP_REQUIRE(Ho < 1024 && Wo < 1024 && Bn < 2048, "hcp_conv3x3_bf16: dims too large for packed pixel ids");
    p.cv.X1 = (const hcp_bf16*)X1; p.cv.C1 = C1; p.cv.X2 = (const hcp_bf16*)X2; p.cv.C2 = C2;
    p.cv.Hs = Hs; p.cv.Ws = Ws; p.cv.Ho = Ho; p.cv.Wo = Wo; p.cv.stride = stride; p.cv.up = upsample ? 1 : 0;
    p.M = Bn * Ho * Wo; p.N = Cout; p.K = 9 * (C1 + C2);
    p.B = (const hcp_bf16*)Wp; p.ldb = p.K;
    p.D = D; p.ldd = ldd; p.out_f32 = out_f32; p.bias = bias;
    p.rowbias = rowbias; p.rowbias_ld = rowbias_ld; p.rows_per_group = Ho * Wo;
    p.residual = (const hcp_bf16*)residual; p.ldr = ldr; p.alpha = 1.0f;
    if (int e = check_common(p)) return e;
    return mode == 0 ? dispatch_gemm<1>(p, stream) : dispatch_gemm<2>(p, stream);
}
