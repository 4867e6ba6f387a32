"""Raw (non-autograd) tensor-level entry points over the C-ABI.  torch is used only for memory and streams.

Every function takes/returns torch tensors that live on the HIP device, checks layout, and enqueues the
kernel on torch's *current* stream.  Nothing here computes with torch ops.
"""
import ctypes
import math
import os
import torch

from . import _lib

_backend = None  # bound CDLL; tests may inject the interpreter build through _set_backend_for_tests


def lib():
    global _backend
    if _backend is None:
        _backend = _lib.load()
    return _backend


def _set_backend_for_tests(cdll):
    """TEST HOOK ONLY: bind a different shared object exporting the same ABI (tests/emu)."""
    global _backend
    _backend = _lib.bind(cdll) if cdll is not None else None


def _chk(rc, name):
    if rc != 0:
        raise _lib.HcpError(f"{name}: {lib().hcp_last_error().decode()}")


def _p(t):
    return None if t is None else t.data_ptr()


def _stream(t):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    if not lib().hcp_is_emulated():
        raise _lib.HcpError("hcp_diffusion_amd kernels need tensors on the HIP device (no CPU path)")
    return None


def _bf16_2d(t, name):
    assert t.dtype == torch.bfloat16 and t.dim() == 2 and t.stride(1) == 1, f"{name}: need 2-D bf16, unit inner stride"
    return t


BF16 = torch.bfloat16

_WS = {}
_WGTAB = {}
_WS_BYTES = 256 << 20
WGRAD_GRID_BLOCKS = 16384      # workgroups the grouped LoRA weight-gradient launch aims at (tools/lab/wgrad_grouped_bench.py sweeps it)


def _workspace(t):
    """One persistent split-K scratch buffer per device (kernels on one stream run in order, so it is reused)."""
    key = (t.device.type, t.device.index)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(_WS_BYTES if t.is_cuda else (16 << 20), dtype=torch.uint8, device=t.device)
        _WS[key] = ws
    return ws


def atomics_selfcheck(device, workgroups=2048, nb=4099, stride=37):
    """Exact check of the device's fp32 atomic adds (hcp_selfcheck_atomics): raises HcpError when a sum of small integers is wrong.
    One launch of `workgroups` x 256 threads, two small device buffers, one synchronising read-back: ~1 ms.  Called by smoke(), bench.py and
    the GPU tests before they trust any number (every cross-workgroup sum left in the library — loss, gradient norm, bias / affine
    gradients, the query-split dK / dV — goes through this instruction)."""
    dev = torch.device(device)
    line = torch.empty(16, dtype=torch.float32, device=dev)
    bucket = torch.empty(nb * stride, dtype=torch.float32, device=dev)
    _chk(lib().hcp_selfcheck_atomics(_p(line), _p(bucket), nb, stride, workgroups, _stream(line)), "hcp_selfcheck_atomics")
    w = torch.arange(workgroups, dtype=torch.int64).view(-1, 1)
    t = torch.arange(256, dtype=torch.int64).view(1, -1)
    want_b = torch.zeros(nb, dtype=torch.int64).index_add_(0, ((37 * w + 101 * t) % nb).flatten(), ((t & 3) + 1).expand(workgroups, 256).flatten())
    want_l = workgroups * 16 * ((torch.arange(16) & 3) + 1)
    got_l, got_b = line.cpu().double(), bucket.cpu()[::stride].double()
    bad_l = int((got_l != want_l.double()).sum()); bad_b = int((got_b != want_b.double()).sum())
    if bad_l or bad_b:
        worst = float(((got_b - want_b.double()).abs() / want_b.double().clamp(min=1)).max())
        raise _lib.HcpError(f"fp32 atomics self-check FAILED on {dev}: {bad_l}/16 words of the shared line and {bad_b}/{nb} strided words differ "
                            f"from the exact integer sums (worst relative error {worst:.3g}); no result of this device can be trusted")
    return True


TRACE = None        # tools/autotune.py: a list collects the (kind, shape...) key of every GEMM-family launch


def _lo_pair(residual, residual_lo, want_lo, M, N, dev):
    """Checks of the (hi | lo) residual-stream arguments shared by gemm / gemm_lora; returns the lo output tensor (or None)."""
    if residual_lo is not None:
        assert residual is not None and residual_lo.dtype == BF16 and residual_lo.shape == (M, N) and residual_lo.stride() == residual.stride()
    return torch.empty((M, N), dtype=BF16, device=dev) if want_lo else None


def gemm(a, b, *, a2=None, b2=None, bias=None, rowbias=None, rows_per_group=1, residual=None, alpha=1.0,
         out_f32=False, out=None, residual_lo=None, want_lo=False, want_gact=False):
    """out[M,N] = alpha*(a[M,K] @ b[N,K]^T + a2 @ b2^T) + bias + rowbias[m // rows_per_group] + residual.
    (hi | lo) residual stream: residual_lo joins the sum in fp32; want_lo returns (out, out_lo) with out_lo = bf16(v - bf16(v)).
    want_gact (out = (h | g) of a GEGLU projection): returns (out, gact), gact [M, N/2] = bf16(h * gelu(g)) from the fp32 epilogue values."""
    _bf16_2d(a, "a"); _bf16_2d(b, "b")
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    K2 = 0
    if a2 is not None:
        _bf16_2d(a2, "a2"); _bf16_2d(b2, "b2")
        K2 = a2.shape[1]
        assert a2.shape[0] == M and b2.shape == (N, K2)
    if TRACE is not None:
        TRACE.append(("gemm", M, N, K, 1 if K2 else 0))
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else BF16, device=a.device)
    assert out.stride(1) == 1
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.shape[1] == N and rowbias.stride(1) == 1
    if residual is not None:
        _bf16_2d(residual, "residual"); assert residual.shape == (M, N)
    ws = _workspace(a)
    assert not want_lo or (out.dtype == BF16 and out.is_contiguous())
    out_lo = _lo_pair(residual, residual_lo, want_lo, M, N, a.device)
    gact = torch.empty((M, N // 2), dtype=BF16, device=a.device) if want_gact else None
    _chk(lib().hcp_gemm_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), M, N, K,
                             _p(a2), a2.stride(0) if a2 is not None else 0, _p(b2), b2.stride(0) if b2 is not None else 0,
                             K2, _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0, rows_per_group,
                             _p(residual), residual.stride(0) if residual is not None else 0, _p(residual_lo), _p(out_lo), _p(gact), float(alpha),
                             1 if out.dtype == torch.float32 else 0, _p(ws), ws.numel(), _stream(a)), "hcp_gemm_bf16")
    if want_gact:
        return out, gact
    return (out, out_lo) if want_lo else out


# Split T: the fused-LoRA GEMMs can hand the rank-r intermediate T = x W_down^T (forward) / U = dY W_up (backward) on
# as TWO bf16 images, hi = bf16(T) and lo = bf16(T - hi) — [M, 64] = (hi | lo) — instead of one rounded copy; the K-extension and the
# weight-gradient kernel then consume both (16 mantissa bits of the fp32 accumulator).  OPT-IN (HCP_T_SPLIT=1, or set this attribute
# before the model is built): measured on MI355X (profiles/r5_ab_t_split.md) it leaves the SDXL configs[3] error ratios against the
# reference's own bf16 mode where they were (flat 1.46 vs 1.45, worst class 2.15 vs 2.18, prediction 1.29 vs 1.29) — the rank-r
# intermediates are NOT where the native step differs from autocast — and costs +1.0 % (SD1.5) / +1.5 % (SDXL) of the step.
T_SPLIT = os.environ.get("HCP_T_SPLIT") == "1"


def t_lo(t):
    """Column offset of the residual half of a T / U that came out of gemm_lora / gemm_geglu_bwd (0: bf16-rounded T only)."""
    return 32 if (t is not None and T_SPLIT and t.shape[1] == 64 and t.stride(0) == 64) else 0


def gemm_lora(a, b, l, e, *, bias=None, residual=None, want_t=True, residual_lo=None, want_lo=False, want_gact=False):
    """(D, T): T = a @ l[32,K]^T;  D[M,N] = a @ b[N,K]^T + T @ e[N,32]^T + bias + residual — one launch.
    (hi | lo) residual stream (see gemm): residual_lo; want_lo returns ((D, D_lo), T); want_gact (see gemm) returns ((D, gact), T).
    T_SPLIT: T [M,64] = (bf16(T) | bf16(T - bf16(T))) and the product takes both halves; else T [M,32] rounded to bf16."""
    _bf16_2d(a, "a"); _bf16_2d(b, "b"); _bf16_2d(l, "l"); _bf16_2d(e, "e")
    M, Kd = a.shape
    N = b.shape[0]
    assert b.shape[1] == Kd and l.shape == (32, Kd) and e.shape == (N, 32) and l.is_contiguous() and e.is_contiguous()
    if TRACE is not None:
        TRACE.append(("lora", M, N, Kd))
    out = torch.empty((M, N), dtype=BF16, device=a.device)
    ldt = 64 if T_SPLIT else 32
    t = torch.empty((M, ldt), dtype=BF16, device=a.device) if want_t else None
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N and bias.is_contiguous()
    if residual is not None:
        _bf16_2d(residual, "residual"); assert residual.shape == (M, N)
    ws = _workspace(a)
    out_lo = _lo_pair(residual, residual_lo, want_lo, M, N, a.device)
    gact = torch.empty((M, N // 2), dtype=BF16, device=a.device) if want_gact else None
    _chk(lib().hcp_gemm_lora_bf16(_p(a), a.stride(0), _p(b), b.stride(0), _p(l), _p(e), _p(t), ldt, _p(out), N, M, N, Kd, _p(bias),
                                  _p(residual), residual.stride(0) if residual is not None else 0, _p(residual_lo), _p(out_lo), _p(gact),
                                  _p(ws), ws.numel(), _stream(a)),
         "hcp_gemm_lora_bf16")
    if want_gact:
        return (out, gact), t
    return ((out, out_lo) if want_lo else out), t


def gemm_geglu_bwd(dy, wt, hg, *, l=None, e=None, want_t=True):
    """d(h | g) [M, 2F] of  y = FFout(h * gelu(g))  from dy [M, C]: dY_ff = dy @ wt[F, C]^T (+ LoRA backward side path l [32, C], e [F, 32],
    U = dy @ l^T returned) with the GEGLU backward applied in the GEMM epilogue — dY_ff itself is never stored.  hg [M, 2F] = the forward's (h | g)."""
    _bf16_2d(dy, "dy"); _bf16_2d(wt, "wt"); _bf16_2d(hg, "hg")
    M, C = dy.shape
    Fd = wt.shape[0]
    assert wt.shape[1] == C and hg.shape == (M, 2 * Fd) and hg.is_contiguous()
    if TRACE is not None:
        TRACE.append(("lora", M, Fd, C) if l is not None else ("gemm", M, Fd, C, 0))
    dhg = torch.empty((M, 2 * Fd), dtype=BF16, device=dy.device)
    u = None
    if l is not None:
        _bf16_2d(l, "l"); _bf16_2d(e, "e")
        assert l.shape == (32, C) and e.shape == (Fd, 32) and l.is_contiguous() and e.is_contiguous()
        u = torch.empty((M, 64 if T_SPLIT else 32), dtype=BF16, device=dy.device) if want_t else None
    ws = _workspace(dy)
    _chk(lib().hcp_gemm_geglu_bwd_bf16(_p(dy), dy.stride(0), _p(wt), wt.stride(0), _p(l), _p(e), _p(u), 64 if T_SPLIT else 32, _p(hg), _p(dhg), M, Fd, C,
                                       _p(ws), ws.numel(), _stream(dy)), "hcp_gemm_geglu_bwd_bf16")
    return dhg, u


def conv3x3(x1, wp, cout, *, x2=None, stride=1, upsample=False, mode=0, out_hw=None, bias=None, rowbias=None,
            residual=None, out_f32=False, a2=None, b2=None, pad=1):
    """3x3 convolution on NHWC bf16. mode 0: forward (wp = [cout][3][3][C1+C2]); mode 1: data gradient
    (x1 = dY [B,Hs,Ws,C1], wp = [cin][3][3][C1], out_hw = spatial dims of the forward input).  pad=0 (forward only): the VAE
    encoder's asymmetric Downsample2D, F.pad(x, (0,1,0,1)) + padding 0."""
    assert x1.dtype == BF16 and x1.dim() == 4 and x1.is_contiguous()
    B, Hs, Ws, C1 = x1.shape
    C2 = 0
    if x2 is not None:
        assert x2.dtype == BF16 and x2.is_contiguous() and x2.shape[:3] == x1.shape[:3]
        C2 = x2.shape[3]
    assert wp.dtype == BF16 and wp.is_contiguous() and wp.numel() == cout * 9 * (C1 + C2)
    if mode == 0:
        up = 2 if upsample else 1
        Ho = (Hs * up + (2 if pad else 1) - 3) // stride + 1
        Wo = (Ws * up + (2 if pad else 1) - 3) // stride + 1
    else:
        Ho, Wo = out_hw
    if TRACE is not None:
        TRACE.append(("conv", mode, B, Hs, Ws, C1, C2, cout, stride, 1 if upsample else 0, Ho, Wo, 1 if a2 is not None else 0))
    out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32 if out_f32 else BF16, device=x1.device)
    if residual is not None:
        assert residual.dtype == BF16 and residual.is_contiguous() and residual.numel() == out.numel()
    if rowbias is not None:
        assert rowbias.dtype == torch.float32 and rowbias.shape == (B, cout) and rowbias.stride(1) == 1
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == cout and bias.is_contiguous()
    if a2 is not None:          # rank-32 K-extension: out += a2 [M,32] @ b2 [cout,32]^T
        assert a2.dtype == BF16 and b2.dtype == BF16 and a2.is_contiguous() and b2.is_contiguous()
        assert a2.numel() == B * Ho * Wo * 32 and tuple(b2.shape) == (cout, 32)
    ws = _workspace(x1)
    _chk(lib().hcp_conv3x3_bf16(_p(x1), C1, _p(x2), C2, B, Hs, Ws, Ho, Wo, mode, stride, 1 if upsample else 0, pad, _p(wp),
                                cout, _p(out), cout, _p(bias), _p(rowbias), rowbias.stride(0) if rowbias is not None else 0,
                                _p(residual), cout, 1 if out_f32 else 0, _p(a2), _p(b2), _p(ws), ws.numel(), _stream(x1)),
         "hcp_conv3x3_bf16")
    return out


def wgrad_linear(dy, x, dw):
    """dw [N,K] fp32 (any row stride) += dy[M,N]^T x[M,K]; dy / x bf16 2-D views with unit column stride."""
    assert dy.dtype == BF16 and x.dtype == BF16 and dw.dtype == torch.float32 and dy.dim() == 2 and x.dim() == 2
    assert dy.stride(1) == 1 and x.stride(1) == 1 and dw.stride(1) == 1 and dy.shape[0] == x.shape[0]
    M, N = dy.shape
    K = x.shape[1]
    assert tuple(dw.shape) == (N, K)
    ws = _workspace(x)
    _chk(lib().hcp_wgrad_linear_bf16(_p(dy), dy.stride(0), _p(x), x.stride(0), _p(dw), dw.stride(0), M, N, K, _p(ws), ws.numel(),
                                     _stream(x)), "hcp_wgrad_linear_bf16")


def wgrad_conv3x3(dy, x1, dw, *, x2=None, stride=1, upsample=False, cout=None, col0=0):
    """dw [Cout][3][3][Cw] fp32 contiguous (the channels_last storage of a [Cout,Cw,3,3] weight) += dY^T im2col(x1|x2).
    dy [B,Ho,Wo,ldy] bf16 (ldy >= Cout, padded columns zero), x1/x2 the forward's NHWC inputs.  col0 (multiple of 8): the Cout columns
    start at column col0 of dy — the rank slots of one of several LoRA blocks in a shared 32-wide U (the kernel only reads the 8-column
    pieces that start below Cout, so col0 + round8(Cout) <= ldy keeps every read inside its row)."""
    assert dy.dtype == BF16 and x1.dtype == BF16 and dw.dtype == torch.float32 and dy.is_contiguous() and x1.is_contiguous()
    B, Hs, Ws, C1 = x1.shape
    C2 = 0
    if x2 is not None:
        assert x2.dtype == BF16 and x2.is_contiguous() and x2.shape[:3] == x1.shape[:3]
        C2 = x2.shape[3]
    _, Ho, Wo, ldy = dy.shape
    cout = cout or ldy
    cw = dw.numel() // (cout * 9)
    assert dw.numel() == cout * 9 * cw and cw <= C1 + C2
    ws = _workspace(x1)
    assert col0 % 8 == 0 and col0 + (cout + 7) // 8 * 8 <= ldy
    dyp = ctypes.c_void_p(dy.data_ptr() + 2 * col0)
    _chk(lib().hcp_wgrad_conv3x3_bf16(dyp, ldy, _p(x1), C1, _p(x2), C2, _p(dw), cw, B, Hs, Ws, Ho, Wo, cout, stride,
                                      1 if upsample else 0, _p(ws), ws.numel(), _stream(x1)), "hcp_wgrad_conv3x3_bf16")


def colsum(y, out, rows_per_group=None):
    """out[g, :N] (fp32) += column sums of the rows of group g of y [M, N] bf16 (rows_per_group=None: one group)."""
    assert y.dtype == BF16 and y.dim() == 2 and y.stride(1) == 1 and out.dtype == torch.float32
    M, N = y.shape
    rpg = rows_per_group or M
    ldo = out.stride(0) if out.dim() == 2 else N
    _chk(lib().hcp_colsum_bf16(_p(y), y.stride(0), _p(out), ldo, M, N, rpg, _stream(y)), "hcp_colsum_bf16")


def _attn_strides(t):
    assert t.dtype == BF16 and t.dim() == 3 and t.stride(2) == 1
    return t.stride(0), t.stride(1)


def _key_bias(key_bias, B, Nk):
    if key_bias is None:
        return None, 0
    assert key_bias.dtype == torch.float32 and tuple(key_bias.shape) == (B, Nk) and key_bias.stride(1) == 1
    return key_bias, key_bias.stride(0)


def attention_fwd(q, k, v, heads, scale=None, key_bias=None, causal=False, q_prescaled=False):
    """q [B,Nq,H*d], k/v [B,Nk,H*d] (views with arbitrary batch/row strides allowed) -> (o [B,Nq,H*d], lse [B,H,Nq]).
    key_bias: optional fp32 [B,Nk] added to the scaled scores (additive key mask); causal: key k visible to query q iff k <= q;
    q_prescaled: q already holds q * scale * log2(e) (folded into the projection that produced it; unmasked problems only)."""
    B, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    o = torch.empty((B, Nq, C), dtype=BF16, device=q.device)
    lse = torch.empty((B, heads, Nq), dtype=torch.float32, device=q.device)
    qb, qr = _attn_strides(q); kb, kr = _attn_strides(k); vb, vr = _attn_strides(v); ob, orr = _attn_strides(o)
    kbt, kbs = _key_bias(key_bias, B, Nk)
    _chk(lib().hcp_attention_fwd(_p(q), _p(k), _p(v), _p(o), _p(lse), B, heads, Nq, Nk, D, qb, qr, kb, kr, vb, vr, ob, orr,
                                 float(scale), _p(kbt), kbs, (1 if causal else 0) | (2 if q_prescaled else 0), _stream(q)), "hcp_attention_fwd")
    return o, lse


def attention_bwd(q, k, v, o, do, lse, heads, scale=None, out=None, key_bias=None, causal=False, q_prescaled=False):
    """Gradients (dq, dk, dv).  q/k/v may be column-slice views of a fused projection buffer; `out` = preallocated
    (dq, dk, dv) with the SAME strides as (q, k, v) (e.g. slices of one [B,N,3C] gradient buffer)."""
    B, Nq, C = q.shape
    Nk = k.shape[1]
    D = C // heads
    scale = scale if scale is not None else 1.0 / math.sqrt(D)
    assert o.is_contiguous() and do.is_contiguous()
    if out is None:
        assert q.is_contiguous() and k.is_contiguous() and v.is_contiguous(), "strided q/k/v need matching `out` buffers"
        dq = torch.empty_like(q); dk = torch.empty_like(k); dv = torch.empty_like(v)
    else:
        dq, dk, dv = out
        assert dq.stride() == q.stride() and dk.stride() == k.stride() and dv.stride() == v.stride()
    delta = torch.empty((B, heads, Nq), dtype=torch.float32, device=q.device)
    qb, qr = _attn_strides(q); kb, kr = _attn_strides(k); vb, vr = _attn_strides(v); ob, orr = _attn_strides(o)
    ws = _workspace(q)
    kbt, kbs = _key_bias(key_bias, B, Nk)
    _chk(lib().hcp_attention_bwd(_p(q), _p(k), _p(v), _p(o), _p(do), _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), B, heads,
                                 Nq, Nk, D, qb, qr, kb, kr, vb, vr, ob, orr, float(scale), _p(kbt), kbs,
                                 (1 if causal else 0) | (2 if q_prescaled else 0), _p(ws),
                                 ws.numel(), _stream(q)), "hcp_attention_bwd")
    return dq, dk, dv


def _gn_ws(x, B, HW, C, G):
    n = lib().hcp_groupnorm_workspace_bytes(B, HW, C, G)
    return torch.empty((max(n, 4) // 4,), dtype=torch.float32, device=x.device)


def groupnorm_fwd(x, gamma, beta, groups, eps, silu):
    """x [B, HW.., C] NHWC bf16 -> (y, stats[B,G,2])."""
    assert x.dtype == BF16 and x.is_contiguous()
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    y = torch.empty_like(x)
    stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
    ws = _gn_ws(x, B, HW, C, groups)
    _chk(lib().hcp_groupnorm_silu_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), _p(ws), B, HW, C, groups, float(eps),
                                      1 if silu else 0, _stream(x)), "hcp_groupnorm_silu_fwd")
    return y, stats


def groupnorm_bwd(x, dy, gamma, beta, stats, groups, silu, addend=None):
    assert x.dtype == BF16 and x.is_contiguous() and dy.dtype == BF16 and dy.is_contiguous()
    assert addend is None or (addend.dtype == BF16 and addend.is_contiguous() and addend.numel() == x.numel())
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    dx = torch.empty_like(x)
    ws = _gn_ws(x, B, HW, C, groups)
    _chk(lib().hcp_groupnorm_silu_bwd(_p(x), _p(dy), _p(gamma), _p(beta), _p(stats), _p(addend), _p(dx), _p(ws), B, HW, C, groups,
                                      1 if silu else 0, _stream(x)), "hcp_groupnorm_silu_bwd")
    return dx


def groupnorm_affine_grad(x, dy, gamma, beta, stats, groups, silu, dgamma, dbeta):
    """dgamma / dbeta (fp32 [C], accumulated in place) for y = [silu](group_norm(x))."""
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    assert x.is_contiguous() and dy.is_contiguous() and dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32
    _chk(lib().hcp_groupnorm_affine_grad(_p(x), _p(dy), _p(gamma), _p(beta), _p(stats), _p(dgamma), _p(dbeta), B, HW, C, groups,
                                         1 if silu else 0, _stream(x)), "hcp_groupnorm_affine_grad")


def layernorm_affine_grad(x, dy, stats, dgamma, dbeta):
    C = x.shape[-1]
    M = x.numel() // C
    assert x.is_contiguous() and dy.is_contiguous() and dgamma.dtype == torch.float32 and dbeta.dtype == torch.float32
    _chk(lib().hcp_layernorm_affine_grad(_p(x), _p(dy), _p(stats), _p(dgamma), _p(dbeta), M, C, _stream(x)),
         "hcp_layernorm_affine_grad")


def _lo_like(t, ref):
    assert t is None or (t.dtype == BF16 and t.is_contiguous() and t.numel() == ref.numel())
    return t


def layernorm_fwd(x, gamma, beta, eps, x_lo=None):
    """x_lo: the lo image of a (hi | lo) residual stream — the row normalised is x + x_lo."""
    assert x.dtype == BF16 and x.is_contiguous()
    C = x.shape[-1]; M = x.numel() // C
    y = torch.empty_like(x)
    stats = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    _chk(lib().hcp_layernorm_fwd(_p(x), _p(_lo_like(x_lo, x)), _p(gamma), _p(beta), _p(y), _p(stats), M, C, float(eps), _stream(x)), "hcp_layernorm_fwd")
    return y, stats


def layernorm_bwd(x, dy, gamma, stats, addend=None, x_lo=None, addend_lo=None, want_lo=False):
    """want_lo: returns (dx, dx_lo) = (bf16(g), bf16(g - bf16(g))) for g = ln_backward(dy) + addend + addend_lo."""
    assert x.is_contiguous() and dy.is_contiguous() and dy.dtype == BF16
    assert addend is None or (addend.dtype == BF16 and addend.is_contiguous() and addend.numel() == x.numel())
    C = x.shape[-1]; M = x.numel() // C
    dx = torch.empty_like(x)
    dx_lo = torch.empty_like(x) if want_lo else None
    _chk(lib().hcp_layernorm_bwd(_p(x), _p(_lo_like(x_lo, x)), _p(dy), _p(gamma), _p(stats), _p(addend), _p(_lo_like(addend_lo, x)), _p(dx),
                                 _p(dx_lo), M, C, _stream(x)), "hcp_layernorm_bwd")
    return (dx, dx_lo) if want_lo else dx


def geglu_fwd(h):
    assert h.dtype == BF16 and h.is_contiguous()
    F2 = h.shape[-1]; F = F2 // 2; M = h.numel() // F2
    y = torch.empty(h.shape[:-1] + (F,), dtype=BF16, device=h.device)
    _chk(lib().hcp_geglu_fwd(_p(h), _p(y), M, F, _stream(h)), "hcp_geglu_fwd")
    return y


def geglu_bwd(h, dy):
    assert h.is_contiguous() and dy.is_contiguous() and dy.dtype == BF16
    F2 = h.shape[-1]; F = F2 // 2; M = h.numel() // F2
    dh = torch.empty_like(h)
    _chk(lib().hcp_geglu_bwd(_p(h), _p(dy), _p(dh), M, F, _stream(h)), "hcp_geglu_bwd")
    return dh


def add(a, b):
    assert a.dtype == BF16 and b.dtype == BF16 and a.is_contiguous() and b.is_contiguous() and a.numel() == b.numel()
    o = torch.empty_like(a)
    _chk(lib().hcp_add_bf16(_p(a), _p(b), _p(o), a.numel(), _stream(a)), "hcp_add_bf16")
    return o


def concat_channels(a, b):
    """[..., C1] ++ [..., C2] -> [..., C1+C2] (channels-last)."""
    assert a.dtype == BF16 and b.dtype == BF16 and a.is_contiguous() and b.is_contiguous() and a.shape[:-1] == b.shape[:-1]
    c1, c2 = a.shape[-1], b.shape[-1]
    M = a.numel() // c1
    out = torch.empty(a.shape[:-1] + (c1 + c2,), dtype=BF16, device=a.device)
    o2 = out.view(M, c1 + c2)
    _chk(lib().hcp_concat2_bf16(_p(a), c1, _p(b), c2, _p(o2), M, 0, _stream(a)), "hcp_concat2_bf16")
    return out


def split_channels(d, c1):
    assert d.dtype == BF16 and d.is_contiguous()
    c = d.shape[-1]; c2 = c - c1
    M = d.numel() // c
    d2 = d.view(M, c)
    a = torch.empty(d.shape[:-1] + (c1,), dtype=BF16, device=d.device)
    b = torch.empty(d.shape[:-1] + (c2,), dtype=BF16, device=d.device)
    _chk(lib().hcp_concat2_bf16(_p(a), c1, _p(b), c2, _p(d2), M, 1, _stream(d)), "hcp_concat2_bf16")
    return a, b


def silu_fwd(x):
    assert x.dtype == BF16 and x.is_contiguous()
    y = torch.empty_like(x)
    _chk(lib().hcp_silu_fwd(_p(x), _p(y), x.numel(), _stream(x)), "hcp_silu_fwd")
    return y


def silu_bwd(x, dy):
    dx = torch.empty_like(x)
    _chk(lib().hcp_silu_bwd(_p(x), _p(dy), _p(dx), x.numel(), _stream(x)), "hcp_silu_bwd")
    return dx


def nchw_to_nhwc(x, cpad=None):
    """[B,C,H,W] fp32|bf16 -> [B,H,W,Cpad] bf16 (zero padded channels)."""
    assert x.dim() == 4 and x.is_contiguous() and x.dtype in (torch.float32, BF16)
    B, C, H, W = x.shape
    cp = cpad or C
    y = torch.empty((B, H, W, cp), dtype=BF16, device=x.device)
    _chk(lib().hcp_nchw_to_nhwc_bf16(_p(x), 1 if x.dtype == torch.float32 else 0, _p(y), B, C, H * W, cp, _stream(x)),
         "hcp_nchw_to_nhwc_bf16")
    return y


def nhwc_to_nchw_f32(x, c):
    """[B,H,W,Cs] fp32 -> [B,c,H,W] fp32 (first c channels)."""
    assert x.dim() == 4 and x.is_contiguous() and x.dtype == torch.float32
    B, H, W, Cs = x.shape
    y = torch.empty((B, c, H, W), dtype=torch.float32, device=x.device)
    _chk(lib().hcp_nhwc_to_nchw_f32(_p(x), _p(y), B, c, H * W, Cs, _stream(x)), "hcp_nhwc_to_nchw_f32")
    return y


def upsample2x_bwd(dup):
    assert dup.dtype == BF16 and dup.is_contiguous()
    B, H2, W2, C = dup.shape
    dx = torch.empty((B, H2 // 2, W2 // 2, C), dtype=BF16, device=dup.device)
    _chk(lib().hcp_upsample2x_bwd(_p(dup), _p(dx), B, H2 // 2, W2 // 2, C, _stream(dup)), "hcp_upsample2x_bwd")
    return dx


def timestep_embedding(t, dim, max_period=10000.0):
    assert t.dtype in (torch.int64, torch.float32) and t.is_contiguous()
    emb = torch.empty((t.numel(), dim), dtype=BF16, device=t.device)
    if t.dtype == torch.int64:
        _chk(lib().hcp_timestep_embedding(_p(t), _p(emb), t.numel(), dim, float(max_period), _stream(t)), "hcp_timestep_embedding")
    else:
        _chk(lib().hcp_timestep_embedding_f32(_p(t), _p(emb), t.numel(), dim, float(max_period), _stream(t)),
             "hcp_timestep_embedding_f32")
    return emb


def add_noise(x0, noise, t, alphas_cumprod):
    assert x0.dtype == torch.float32 and noise.dtype == torch.float32 and x0.is_contiguous() and noise.is_contiguous()
    assert t.dtype == torch.int64 and alphas_cumprod.dtype == torch.float32
    xt = torch.empty_like(x0)
    B = x0.shape[0]
    _chk(lib().hcp_add_noise(_p(x0), _p(noise), _p(t), _p(alphas_cumprod), _p(xt), B, x0.numel() // B, _stream(x0)), "hcp_add_noise")
    return xt


def cfg_ddim_step(x, eps2, a_t, a_prev, guidance_scale=1.0, out=None):
    """One DDIM (eta = 0) step with classifier-free guidance: eps2 is the UNet output on [uncond ; cond] (2B rows) or on B rows."""
    assert x.dtype == torch.float32 and eps2.dtype == torch.float32 and x.is_contiguous() and eps2.is_contiguous()
    guided = eps2.numel() == 2 * x.numel()
    assert guided or eps2.numel() == x.numel()
    out = torch.empty_like(x) if out is None else out
    _chk(lib().hcp_cfg_ddim_step(_p(x), _p(eps2), _p(out), x.numel(), 1 if guided else 0, float(guidance_scale), float(a_t), float(a_prev),
                                 _stream(x)), "hcp_cfg_ddim_step")
    return out


def quick_gelu(x, dy=None):
    """quick_gelu(x) = x * sigmoid(1.702 x) (dy None) or dy * quick_gelu'(x); bf16, any shape with numel % 8 == 0."""
    assert x.dtype == BF16 and x.is_contiguous() and (dy is None or (dy.dtype == BF16 and dy.is_contiguous() and dy.shape == x.shape))
    out = torch.empty_like(x)
    _chk(lib().hcp_quick_gelu(_p(x), _p(dy), _p(out), x.numel(), _stream(x)), "hcp_quick_gelu")
    return out


def embedding(token_table, ids, position_table, position_ids=None):
    """[.., L] int64 ids -> bf16 [.., L, C] = token_table[ids] + position_table[position_ids or arange(L)]."""
    assert token_table.dtype == torch.float32 and position_table.dtype == torch.float32 and token_table.is_contiguous() and position_table.is_contiguous()
    assert ids.dtype == torch.int64 and ids.is_contiguous() and (position_ids is None or (position_ids.dtype == torch.int64 and position_ids.shape == ids.shape))
    C, L = token_table.shape[1], ids.shape[-1]
    out = torch.empty(tuple(ids.shape) + (C,), dtype=BF16, device=ids.device)
    _chk(lib().hcp_embedding_bf16(_p(token_table), _p(ids), _p(position_table), _p(position_ids.contiguous() if position_ids is not None else None),
                                  _p(out), ids.numel(), C, L, _stream(ids)), "hcp_embedding_bf16")
    return out


def transpose_bf16(x):
    """[b, R, C] bf16 -> [b, C, R]."""
    assert x.dtype == BF16 and x.dim() == 3 and x.is_contiguous()
    b, R, C = x.shape
    out = torch.empty((b, C, R), dtype=BF16, device=x.device)
    _chk(lib().hcp_transpose_bf16(_p(x), _p(out), b, R, C, _stream(x)), "hcp_transpose_bf16")
    return out


def softmax_rows(s, scale=1.0):
    """softmax(scale * s) over the last dim: fp32 [M, N] (row stride >= N) -> bf16 [M, N]."""
    assert s.dtype == torch.float32 and s.dim() == 2 and s.stride(1) == 1
    M, N = s.shape
    out = torch.empty((M, N), dtype=BF16, device=s.device)
    _chk(lib().hcp_softmax_rows(_p(s), s.stride(0), _p(out), N, M, N, float(scale), _stream(s)), "hcp_softmax_rows")
    return out


def vae_latent_sample(moments, wq, bq, noise, scale):
    """(mean + exp(0.5 clamp(logvar)) * noise) * scale with (mean | logvar) = quant_conv(moments); moments fp32 [B, 2L, h, w] NCHW,
    wq [2L, 2L], bq [2L], noise fp32 [B, L, h, w] or None (-> the mode)."""
    assert moments.dtype == torch.float32 and moments.is_contiguous() and moments.dim() == 4
    B, L2, h, w = moments.shape
    L = L2 // 2
    assert wq.dtype == torch.float32 and wq.is_contiguous() and wq.numel() == L2 * L2 and bq.dtype == torch.float32 and bq.numel() == L2
    if noise is not None:
        assert noise.dtype == torch.float32 and noise.is_contiguous() and tuple(noise.shape) == (B, L, h, w)
    out = torch.empty((B, L, h, w), dtype=torch.float32, device=moments.device)
    _chk(lib().hcp_vae_latent_sample(_p(moments), _p(wq), _p(bq), _p(noise), _p(out), B, L, h * w, float(scale), _stream(moments)),
         "hcp_vae_latent_sample")
    return out


SNR_LOSS_KINDS = {"min_snr": 0, "soft_min_snr": 1, "kdiff_min_snr": 2, "edm": 3}


def snr_loss_weight(t, alphas_cumprod, kind, gamma):
    """float[B] loss weights of the reference's MinSNRLoss family (hcpdiff/loss/min_snr_loss.py) for timesteps t."""
    assert t.dtype == torch.int64 and alphas_cumprod.dtype == torch.float32 and t.is_contiguous()
    w = torch.empty(t.shape[0], dtype=torch.float32, device=t.device)
    _chk(lib().hcp_snr_loss_weight(_p(t), _p(alphas_cumprod), _p(w), t.shape[0], SNR_LOSS_KINDS[kind], float(gamma), _stream(t)),
         "hcp_snr_loss_weight")
    return w


def mse_masked_mean(pred, target, mask=None, weight=1.0, want_grad=True, sample_weight=None):
    """(mean((pred-target)^2 * mask * sample_weight[b]) * weight as a device scalar, d loss / d pred)."""
    assert pred.dtype == torch.float32 and target.dtype == torch.float32 and pred.is_contiguous() and target.is_contiguous()
    B, C = pred.shape[0], pred.shape[1]
    HW = pred.numel() // (B * C)
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    grad = torch.empty_like(pred) if want_grad else None
    mc = 0
    if mask is not None:
        mask = mask.to(torch.float32).contiguous()
        mc = mask.shape[1]
    if sample_weight is not None:
        assert sample_weight.dtype == torch.float32 and sample_weight.numel() == B and sample_weight.is_contiguous()
    _chk(lib().hcp_mse_masked_mean(_p(pred), _p(target), _p(mask), mc, _p(sample_weight), _p(loss), _p(grad), B, C, HW, float(weight),
                                   _stream(pred)),
         "hcp_mse_masked_mean")
    return loss, grad


def split_hi_lo(src):
    """fp32 [M, C] -> bf16 [M, 2C] = (bf16(src) | bf16(src - bf16(src))): the split form of a rank-r intermediate (T_SPLIT)."""
    assert src.dtype == torch.float32 and src.dim() == 2 and src.is_contiguous() and src.shape[1] % 4 == 0
    M, C = src.shape
    dst = torch.empty((M, 2 * C), dtype=BF16, device=src.device)
    _chk(lib().hcp_split_hi_lo_bf16(_p(src), _p(dst), M, C, _stream(src)), "hcp_split_hi_lo_bf16")
    return dst


def lora_wgrad(L, R, out, P, scale, transpose_out, out_col0=0, lo=0):
    """out (fp32) += scale * L[:, :P]^T @ R (partial slabs in the workspace + an ordered reduce: no atomics); transpose_out writes out[q, out_col0 + p]
    (out_col0: first rank column of a 32-wide block when the rank exceeds one slot group).  lo: column offset of the residual half of a
    split L (t_lo(L)), 0 = none."""
    _bf16_2d(L, "L"); _bf16_2d(R, "R")
    M, Q = R.shape
    assert L.shape[0] == M and out.dtype == torch.float32 and out.is_contiguous()
    assert out_col0 == 0 or transpose_out
    ldo = out.shape[1]
    ptr = ctypes.c_void_p(out.data_ptr() + 4 * out_col0)
    ws = _workspace(L)
    _chk(lib().hcp_lora_wgrad(_p(L), L.stride(0), int(lo), _p(R), R.stride(0), ptr, ldo, M, P, Q, float(scale),
                              1 if transpose_out else 0, _p(ws), ws.numel(), _stream(L)), "hcp_lora_wgrad")


def lora_wgrad_pair(U, x, grad_down, T, dy, grad_up, r, scale):
    _bf16_2d(U, "U"); _bf16_2d(x, "x"); _bf16_2d(T, "T"); _bf16_2d(dy, "dy")
    M, Kd = x.shape
    N = dy.shape[1]
    assert U.shape in ((M, 32), (M, 64)) and T.shape in ((M, 32), (M, 64)) and U.is_contiguous() and T.is_contiguous()      # 64: split (hi | lo)
    assert grad_down.shape == (r, Kd) and grad_up.shape == (N, r) and grad_down.is_contiguous() and grad_up.is_contiguous()
    ws = _workspace(x)
    _chk(lib().hcp_lora_wgrad_pair(_p(U), U.shape[1], _p(x), x.stride(0), Kd, _p(grad_down), _p(T), T.shape[1], _p(dy), dy.stride(0), N,
                                   _p(grad_up), M, r, float(scale), _p(ws), ws.numel(), _stream(x)), "hcp_lora_wgrad_pair")


def lora_wgrad_grouped(items):
    """items: list of (U, x, grad_down, T, dy, grad_up, r, scale[, slot0[, u_lo, t_lo]]) — every layer's LoRA weight gradients as ONE
    partial-tile launch + ONE ordered reduce (no atomics: a step's LoRA gradients are bit-reproducible).
    dy, U and T may be column-slice views (row stride = their stride(0)); slot0 = first rank column of the layer in U / T;
    u_lo / t_lo = column offset of the residual half of a split U / T (t_lo(.)), 0 = none.
    A gradient tensor named twice (a layer that ran twice in the forward) is taken by a second call on the same stream: inside one
    call the reduce adds into the bucket without atomics."""
    seen, first, rest = set(), [], []
    for item in items:
        key = (item[2].data_ptr(), item[5].data_ptr())
        (rest if key in seen else first).append(item)
        seen.add(key)
    if rest:
        a = lora_wgrad_grouped(first)
        b = lora_wgrad_grouped(rest)
        return a + b
    import struct
    L = lib()
    assert L.hcp_lora_wgrad_group_desc_bytes() == 152
    qt0, qt1, sp, rows = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    ws = _workspace(items[0][1])
    target = max(8, min(256, WGRAD_GRID_BLOCKS // len(items)))     # the layers share one grid: a few thousand workgroups fill the chip
    while True:
        buf = bytearray()
        begin = tiles = units = 0
        for item in items:
            (U, x, gd, T, dy, gu, r, scale) = item[:8]
            slot0 = item[8] if len(item) > 8 else 0          # first rank column of this layer inside U / T (fused groups)
            ulo, tlo = (item[9], item[10]) if len(item) > 10 else (0, 0)
            M, Kd = x.shape
            N = dy.shape[1]
            nb = L.hcp_lora_wgrad_group_geometry(M, Kd, N, r, target, ctypes.byref(qt0), ctypes.byref(qt1), ctypes.byref(sp), ctypes.byref(rows))
            assert U.stride(1) == 1 and T.stride(1) == 1 and U.stride(0) % 8 == 0 and T.stride(0) % 8 == 0
            buf += struct.pack("<QiiQi4xQiiii", U.data_ptr(), U.stride(0), ulo, x.data_ptr(), x.stride(0), gd.data_ptr(), Kd, Kd, 0, slot0)
            buf += struct.pack("<QiiQi4xQiiii", T.data_ptr(), T.stride(0), tlo, dy.data_ptr(), dy.stride(0), gu.data_ptr(), r, N, 1, slot0)
            buf += struct.pack("<iifiiiiiii", M, r, float(scale), rows.value, qt0.value, qt1.value, sp.value, begin, units, tiles)
            begin += nb
            tiles += qt0.value + qt1.value
            if sp.value > 1:
                units += nb * r                           # a workgroup's slab: r x 128 floats (a layer with one token range writes its gradient itself)
        if units * 512 <= ws.numel() or target <= 1:      # the slabs fit the workspace (a large model at a large batch: fewer token ranges per layer)
            break
        target //= 2
    dev = items[0][1].device
    src = torch.frombuffer(buf, dtype=torch.uint8)
    if dev.type == "cuda":
        # Pinned staging buffer + device table from a pool (no allocation while a hipGraph is being captured).  A captured
        # H2D copy re-reads its HOST buffer on every replay, so a slot used under capture is frozen for good, and within one
        # step (wgrad_staging_begin_step) every call takes its own slot: two same-length launches in one capture (two
        # datasets per step, or per-layer launches) must not share a staging buffer.
        pool = _WGTAB.setdefault((dev.index, len(buf)), {"slots": [], "cursor": 0})
        capturing = torch.cuda.is_current_stream_capturing()
        slots, i = pool["slots"], pool["cursor"]
        if i > 4096 and not capturing:            # nobody marks step boundaries: recycle
            i = 0
        while i < len(slots) and slots[i]["frozen"]:
            i += 1
        if i == len(slots):                       # the eager warm-up steps before a capture grow the pool to what a step needs
            slots.append({"host": torch.empty(len(buf), dtype=torch.uint8).pin_memory(),
                          "table": torch.empty(len(buf), dtype=torch.uint8, device=dev), "event": None, "frozen": False})
        slot = slots[i]
        pool["cursor"] = i + 1
        if slot["event"] is not None and not capturing:
            slot["event"].synchronize()           # the previous use's H2D copy has consumed the staging buffer
        slot["host"].copy_(src)
        slot["table"].copy_(slot["host"], non_blocking=True)
        if capturing:
            slot["frozen"] = True
        else:
            slot["event"] = torch.cuda.Event(); slot["event"].record()
        host, table = slot["host"], slot["table"]
    else:
        host, table = src, src.clone()
    _chk(L.hcp_lora_wgrad_grouped(_p(table), len(items), begin, tiles, units, _p(ws), ws.numel(), _stream(table)), "hcp_lora_wgrad_grouped")
    return host, table


def wgrad_staging_begin_step():
    """Step boundary for lora_wgrad_grouped's staging pool: the next calls start again from the first free slot."""
    for pool in _WGTAB.values():
        pool["cursor"] = 0


def lora_pack(desc_tensor, count):
    _chk(lib().hcp_lora_pack(_p(desc_tensor), count, _stream(desc_tensor)), "hcp_lora_pack")


def pack_weights(pieces, count, total_tiles):
    assert pieces.dtype == torch.uint8 and pieces.is_contiguous() and pieces.numel() >= count * lib().hcp_pack_piece_bytes()
    _chk(lib().hcp_pack_weights(_p(pieces), count, total_tiles, _stream(pieces)), "hcp_pack_weights")


def sumsq(g, out):
    assert g.dtype == torch.float32 and g.is_contiguous()
    _chk(lib().hcp_sumsq_f32(_p(g), g.numel(), _p(out), _stream(g)), "hcp_sumsq_f32")
    return out


def adamw_clip_fused(p, g, m, v, lr, step, *, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, sumsq_t=None,
                     grad_scale=1.0, max_norm=0.0):
    for t in (p, g, m, v):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == p.numel()
    assert lr.dtype == torch.float32 and step.dtype == torch.int32
    _chk(lib().hcp_adamw_clip_fused(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr), beta1, beta2, eps, weight_decay,
                                    _p(sumsq_t), float(grad_scale), float(max_norm), _p(step), _stream(p)), "hcp_adamw_clip_fused")


def cast_f32_bf16(src, dst, scale=1.0, zero_src=False):
    """dst (bf16) <- src (fp32) * scale; zero_src clears src in the same pass (wire format of the sharded exchange)."""
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    _chk(lib().hcp_cast_f32_bf16(_p(src), _p(dst), src.numel(), float(scale), int(bool(zero_src)), _stream(src)), "hcp_cast_f32_bf16")
    return dst


def cast_bf16_f32(src, dst):
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and src.is_contiguous() and dst.is_contiguous() and src.numel() == dst.numel()
    _chk(lib().hcp_cast_bf16_f32(_p(src), _p(dst), src.numel(), _stream(src)), "hcp_cast_bf16_f32")
    return dst


def ema_update(ema, p, step, inv_gamma=1.0, power=2.0 / 3.0, decay_max=0.9997):
    assert ema.dtype == torch.float32 and p.dtype == torch.float32 and ema.is_contiguous() and p.is_contiguous() and ema.numel() == p.numel()
    assert step.dtype == torch.int32
    _chk(lib().hcp_ema_update(_p(ema), _p(p), p.numel(), _p(step), float(inv_gamma), float(power), float(decay_max), _stream(p)),
         "hcp_ema_update")
