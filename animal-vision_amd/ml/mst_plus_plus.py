"""MST++ (Cai et al., NTIRE 2022) inference module for PyTorch-ROCm.

Same network, same state-dict keys as the reference's
ml/MST_plus_plus/predict_code/architecture/MST_Plus_Plus.py (MS_MSA :88-139, FeedForward :141-158,
MSAB :160-186, MST :188-268, MST_Plus_Plus :270-293; 227 tensors, 1,619,625 parameters), written
inference-only and NHWC-first for MI355X instead of as a module tree:

  * activations stay channels-last (N*H*W, C) so every 1x1 conv / Linear is one GEMM on a 2-D view;
  * to_q/to_k/to_v run as ONE GEMM against the stacked (3C x C) weight (one pass over x, not three);
  * spectral attention works on the d x d Gram matrix: L2-normalising q and k over all pixels
    (F.normalize(dim=-1), :127-128) is folded into the 31x31 result as a division by the outer product
    of column norms, so the normalised N x C tensors are never materialised;
  * `attn @ v` followed by `proj` (:132-135) collapses to one GEMM: v @ M + b with
    M = blockdiag(attn_h^T) @ W_proj^T (C x C), exact in real arithmetic;
  * every 31-channel group is stored 32 wide (weights zero-padded once, at first use): rows of 62 B become 64 B, so
    GEMM operands, depthwise convs and elementwise ops are aligned and vectorise; the padding channel stays exactly
    zero through every linear op / GELU / LayerNorm (statistics over the 31 real channels) and is masked out of the
    attention softmax;
  * weights live in a flat dict under the reference's key names; `load_reference_state_dict` strips the
    DataParallel 'module.' prefix like architecture/__init__.py:36-40.

Half precision mirrors predict_torch.py:109 (fp16 autocast): GEMMs/convs in fp16, norms/softmax/
LayerNorm statistics in fp32."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

DIM = 31
PAD = 32  # storage width of one 31-channel group


def padded(n: int) -> int:
    """Storage width of n = m * 31 channels."""
    assert n % DIM == 0
    return n // DIM * PAD


def pad_index(n: int, device=None) -> torch.Tensor:
    """Position of original channel i in the padded layout: group i // 31 starts at 32 * (i // 31)."""
    i = torch.arange(n, device=device)
    return (i // DIM) * PAD + i % DIM


def pad_channels(t: torch.Tensor, dims) -> torch.Tensor:
    """Zero-pad the given dims of a weight from m*31 to m*32 entries (group-wise)."""
    for d in dims:
        n = t.shape[d]
        shape = list(t.shape)
        shape[d] = padded(n)
        out = torch.zeros(shape, dtype=t.dtype, device=t.device)
        out.index_copy_(d, pad_index(n, t.device), t)
        t = out
    return t


def pack_fragments(w_kn: torch.Tensor, transposed: bool) -> torch.Tensor:
    """(K x N) float16 weight of `y = x @ w` -> v_mfma_f32_32x32x8_f16 fragment order [N/32][K/8][64 lanes][4]
    (csrc/mst_mfma.hip): MFMA step s, lane half h, element j carries input channel h*K/2 + 4s + j (the order in which a
    lane holds its contiguous half row of x).  transposed=False: tile column = lane % 32 ("pixels x channels" results);
    transposed=True: the weight is the A operand and its 32 rows are permuted so a result lane ends up with 16
    CONTIGUOUS output channels (row m <-> channel 16*((m % 8) // 4) + 4*(m // 8) + m % 4)."""
    K, N = w_kn.shape
    assert K % 32 == 0 and N % 32 == 0
    dev = w_kn.device
    key = (K, N, transposed, str(dev))
    idx = _FRAG_INDEX.get(key)
    if idx is None:  # flat gather index, cached: packing a per-frame matrix is then ONE take() launch
        lane = torch.arange(64, device=dev)
        c, h = lane % 32, lane // 32
        if transposed:
            c = 16 * ((c % 8) // 4) + 4 * (c // 8) + c % 4
        s = torch.arange(K // 8, device=dev)
        j = torch.arange(4, device=dev)
        k = (h[None, :, None] * (K // 2) + 4 * s[:, None, None] + j[None, None, :])[None]              # (1, K/8, 64, 4)
        col = 32 * torch.arange(N // 32, device=dev)[:, None, None, None] + c[None, None, :, None]       # (N/32, 1, 64, 1)
        idx = (k * N + col).contiguous()                                                                 # (N/32, K/8, 64, 4)
        _FRAG_INDEX[key] = idx
    return torch.take(w_kn.contiguous(), idx)


_FRAG_INDEX: Dict[tuple, torch.Tensor] = {}


def pack_fragments16(w_kn: torch.Tensor, halfrow: bool = False) -> torch.Tensor:
    """(K x N) float16 weight of `y = x @ w` as the A operand of v_mfma_f32_32x32x16_f16 ("channels x pixels" results,
    csrc/mst_fused.hip): [N/32][K/16][64 lanes][8]; lane (m = lane % 32, h = lane // 32), element j of step s carries input
    channel 16 s + 8 h + j (natural K order) of output channel 32 t + perm(m), perm(m) = 16 ((m % 8) // 4) + 4 (m // 8) + m % 4 --
    the row permutation that leaves a result lane with 16 CONTIGUOUS output channels (D row = (reg & 3) + 8 (reg >> 2) + 4 h)."""
    K, N = w_kn.shape
    assert K % 16 == 0 and N % 32 == 0
    dev = w_kn.device
    key = (K, N, "k16h" if halfrow else "k16", str(dev))
    idx = _FRAG_INDEX.get(key)
    if idx is None:
        lane = torch.arange(64, device=dev)
        m, h = lane % 32, lane // 32
        col = 16 * ((m % 8) // 4) + 4 * (m // 8) + m % 4
        s = torch.arange(K // 16, device=dev)
        j = torch.arange(8, device=dev)
        if halfrow:  # step s of lane half h carries input channel h K/2 + 8 s + j: the order in which a lane holds its contiguous half row of x (operands straight from global memory)
            k = (8 * s[:, None, None] + (K // 2) * h[None, :, None] + j[None, None, :])[None]
        else:
            k = (16 * s[:, None, None] + 8 * h[None, :, None] + j[None, None, :])[None]                    # (1, K/16, 64, 8)
        n = 32 * torch.arange(N // 32, device=dev)[:, None, None, None] + col[None, None, :, None]      # (N/32, 1, 64, 1)
        idx = (k * N + n).contiguous()
        _FRAG_INDEX[key] = idx
    return torch.take(w_kn.contiguous(), idx)


def pack_qkv16(wqkv: torch.Tensor) -> torch.Tensor:
    """(c x 3c) float16 weight [W_q^T | W_k^T | W_v^T] of `qkv = x @ w` -> the v_mfma_f32_32x32x16_f16 fragments of
    csrc/mst_mfma.hip::k_mst_qkv16: [3c/32 tiles][c/16 steps][64 lanes][8]; step q, lane half h, element j carries input channel
    h c/2 + 8 q + j (the order in which a lane holds its contiguous half row of x); lane % 32 = m is output channel 32 t + m for the
    q and k tiles ("pixels x channels" results) and 32 t + perm(m) for the v tiles (the weight is the A operand there and its rows are
    permuted so a result lane ends up with 16 contiguous channels, perm as in pack_fragments16)."""
    K, N = wqkv.shape
    assert N == 3 * K and K % 32 == 0
    dev = wqkv.device
    key = (K, N, "qkv16", str(dev))
    idx = _FRAG_INDEX.get(key)
    if idx is None:
        lane = torch.arange(64, device=dev)
        m, h = lane % 32, lane // 32
        perm = 16 * ((m % 8) // 4) + 4 * (m // 8) + m % 4
        t = torch.arange(N // 32, device=dev)
        col = 32 * t[:, None] + torch.where(t[:, None] >= 2 * (K // 32), perm[None, :], m[None, :])                       # (N/32, 64)
        q = torch.arange(K // 16, device=dev)
        j = torch.arange(8, device=dev)
        k = h[None, :, None] * (K // 2) + 8 * q[:, None, None] + j[None, None, :]                                         # (K/16, 64, 8)
        idx = (k[None] * N + col[:, None, :, None]).contiguous()                                                          # (N/32, K/16, 64, 8)
        _FRAG_INDEX[key] = idx
    return torch.take(wqkv.contiguous(), idx)


def pack_dw_mfma(w: torch.Tensor) -> torch.Tensor:
    """(Ch, 1, 3, 3) float16 depthwise 3x3 weight -> the A fragments of csrc/mst_fused.hip's matrix-pipe depthwise conv:
    [Ch/8 octets][3 column shifts][64 lanes][8].  The 16 result rows of a v_mfma_f32_16x16x32_f16 are m = 8 s + c: channel c of the octet
    at output row s (two vertically adjacent rows); K slot q = lane // 16 of MFMA i carries input row r = q (of the four rows the pair
    needs) shifted by i columns, element j = channel j of the octet.  The weight matrix is diagonal in the channels: lane (m, q) holds
    w[8 o + c][r - s][i] at element c when 0 <= r - s <= 2 and zeros elsewhere."""
    ch = w.shape[0]
    assert ch % 8 == 0 and tuple(w.shape[1:]) == (1, 3, 3)
    dev = w.device
    lane = torch.arange(64, device=dev)
    m, q = lane % 16, lane // 16
    s_, c = m // 8, m % 8
    dy = q - s_                                                     # (64,)
    ok = (dy >= 0) & (dy <= 2)
    w8 = w.reshape(ch // 8, 8, 3, 3)                                # [o][c][dy][dx]
    vals = w8[:, c, dy.clamp(0, 2), :]                              # (O, 64, 3): [o][lane][dx]
    vals = torch.where(ok[None, :, None], vals, torch.zeros_like(vals))
    out = torch.zeros((ch // 8, 3, 64, 8), dtype=w.dtype, device=dev)
    out[:, :, lane, c] = vals.permute(0, 2, 1)                      # element c of lane
    return out.contiguous()


def fold_layernorm(w1_kn: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, scale: float) -> torch.Tensor:
    """(K = c input channels, N) weight of the 1x1 conv behind a LayerNorm -> float32 (K, N) weight for avx_mst_ffn_fused_mx, whose LayerNorm only centres and
    scales (include/avx.h): row k times gamma[k], row 31 (the first 32-wide group's padding channel; the kernel feeds it a 1) = beta @ W, everything / scale."""
    w = w1_kn.float()
    out = w * gamma.float()[:, None]
    out[31, :] = beta.float() @ w
    return (out / scale).contiguous()


def pack_down4x4(w: torch.Tensor) -> torch.Tensor:
    """(Cout, Cin, 4, 4) float16 stride-2 conv weight -> [16 taps][Cout/32][Cin/16][64][8] fragments (tap = 4 ky + kx), the A operands
    of csrc/mst_fused.hip::k_mst_down4x4_dma."""
    return torch.stack([pack_fragments16(w[:, :, t // 4, t % 4].t().contiguous()) for t in range(16)]).contiguous()


class _AvxOps:
    """Binds the hand-written gfx950 kernels of csrc/mst.hip for CUDA tensors (data_ptr hand-off on torch's
    current stream).  AVX_MST_TORCH_ONLY=1 keeps every op in torch (A/B and debugging)."""

    def __init__(self):
        import os

        self.enabled = os.environ.get("AVX_MST_TORCH_ONLY", "") == ""
        self._mfma = os.environ.get("AVX_MST_NO_MFMA", "") == ""  # A/B: the fused matrix-core kernels of csrc/mst_mfma.hip
        self._ffn2 = os.environ.get("AVX_MST_NO_FFN2", "") == ""  # A/B: the fused depthwise-conv + second GEMM kernel
        self._posemb = os.environ.get("AVX_MST_NO_POSEMB", "") == ""  # A/B: pos_emb's two depthwise convs in one pass
        self._conv_lds = os.environ.get("AVX_MST_NO_CONV_LDS", "") == ""  # A/B: the dense 3x3 convs through an LDS halo tile
        self._conv_in = os.environ.get("AVX_MST_NO_CONV_IN", "") == ""  # A/B: uint8 frame -> conv_in output in one kernel
        self._upfuse = os.environ.get("AVX_MST_NO_UPFUSE", "") == ""  # A/B: transposed conv + fusion conv in one pass
        self._tail = os.environ.get("AVX_MST_NO_TAIL", "") == ""  # A/B: the attention tail (pos_emb + v @ M + bias + x) in one pass
        self._tailx = os.environ.get("AVX_MST_NO_TAILX", "") == ""  # A/B: the tail forms v = x W_v^T itself, the Gram pass writes no v
        self._down = os.environ.get("AVX_MST_NO_DOWN4X4", "") == ""  # A/B: hand-written 32 -> 64 stride-2 conv instead of MIOpen's implicit GEMM
        self._qkv16 = os.environ.get("AVX_MST_NO_QKV16", "") == ""  # A/B: the Gram pass on K = 16 MFMAs
        self._convgram = os.environ.get("AVX_MST_NO_CONVGRAM", "") == ""  # A/B: the embedding conv carries the first block's Gram pass as its epilogue
        self._ffn = os.environ.get("AVX_MST_NO_FFN_FUSED", "") == ""  # A/B: the whole FeedForward in one kernel, hidden tile in LDS
        self._gelu_pre = None
        self._dwmx = os.environ.get("AVX_MST_NO_DW_MFMA", "") == ""  # A/B: depthwise 3x3 convs on the matrix pipe (round 3)
        self._specfuse = os.environ.get("AVX_MST_NO_SPEC_FUSE", "") == ""  # A/B: the spectral integration as conv_out's epilogue (the cube is never written)
        self.FFN_FUSED_C = tuple(int(v) for v in os.environ.get("AVX_MST_FFN_FUSED_C", "32,64,128").split(",") if v)
        self._ctx = {}

    def ctx(self, device: torch.device):
        """The process-wide libavx context (one process per GPU); it must sit on the tensor's device."""
        from ..runtime import get_context

        idx = device.index if device.index is not None else torch.cuda.current_device()
        c = get_context(idx)
        if c.device != idx:
            raise RuntimeError(f"libavx context is on device {c.device}, tensor on cuda:{idx} (one process per GPU)")
        return c

    @staticmethod
    def _dt(t: torch.Tensor) -> int:
        if t.dtype == torch.float32:
            return 0
        if t.dtype == torch.float16:
            return 1
        raise TypeError(f"csrc/mst.hip kernels take float32/float16, got {t.dtype}")

    def gram(self, qkv: torch.Tensor, heads: int):
        """qkv (b, n, 3c) contiguous -> gram (b, heads, d, d), nq (b, c), nk (b, c), all float32."""
        from .._lib import lib

        b, n, c3 = qkv.shape
        c = c3 // 3
        d = c // heads
        ctx = self.ctx(qkv.device)
        g = torch.empty((b, heads, 32, 32), dtype=torch.float32, device=qkv.device)
        nq = torch.empty((b, c), dtype=torch.float32, device=qkv.device)
        nk = torch.empty((b, c), dtype=torch.float32, device=qkv.device)
        st = torch.cuda.current_stream(qkv.device).cuda_stream
        for i in range(b):
            ctx._check(lib.avx_mst_gram(ctx._h, qkv[i].data_ptr(), self._dt(qkv), n, c, heads, g[i].data_ptr(), nq[i].data_ptr(), nk[i].data_ptr(), st))
        return g[:, :, :d, :d], nq, nk

    def gelu_prescale(self) -> float:
        """s of include/avx.h::avx_mst_gelu_prescale: the matrix-pipe kernels' GELU takes x / s and returns gelu(x) / s."""
        if self._gelu_pre is None:
            from .. import _lib
            self._gelu_pre = float(_lib.lib.avx_mst_gelu_prescale())
        return self._gelu_pre

    def fused_ok(self, x: torch.Tensor) -> bool:
        """The matrix-core kernels take float16 rows of 32, 64 or 128 channels."""
        return self.enabled and x.is_cuda and x.dtype == torch.float16 and x.shape[-1] in (32, 64, 128) and self._mfma

    def qkv_gram(self, x2: torch.Tensor, wpack: torch.Tensor, heads: int, want_v: bool = True, k16: bool = False):
        """x2 (b, n, c) float16 -> v (b, n, c) (None when not wanted: the pass then only reads), gram (b, heads, 32, 32), nq (b, c), nk (b, c)
        (csrc/mst_mfma.hip).  k16: wpack is pack_qkv16's (the K = 16 MFMA kernel), else pack_fragments' (K = 8)."""
        from .._lib import lib

        b, n, c = x2.shape
        x2 = x2.contiguous()
        ctx = self.ctx(x2.device)
        v = torch.empty_like(x2) if want_v else None
        g = torch.empty((b, heads, 32, 32), dtype=torch.float32, device=x2.device)
        nq = torch.empty((b, c), dtype=torch.float32, device=x2.device)
        nk = torch.empty((b, c), dtype=torch.float32, device=x2.device)
        st = torch.cuda.current_stream(x2.device).cuda_stream
        fn = lib.avx_mst_qkv_gram16 if k16 else lib.avx_mst_qkv_gram
        for i in range(b):
            ctx._check(fn(ctx._h, x2[i].data_ptr(), wpack.data_ptr(), n, c, v[i].data_ptr() if want_v else None, g[i].data_ptr(), nq[i].data_ptr(), nk[i].data_ptr(), st))
        return v, g, nq, nk

    def ln_gemm_gelu(self, x2: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, wpack: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
        """x2 (rows, c) float16 -> gelu(layernorm(x2) @ w1) (rows, 4c) float16 (csrc/mst_mfma.hip)."""
        from .._lib import lib

        rows, c = x2.shape
        x2 = x2.contiguous()
        ctx = self.ctx(x2.device)
        out = torch.empty((rows, 4 * c), dtype=torch.float16, device=x2.device)
        ctx._check(lib.avx_mst_ln_gemm_gelu(ctx._h, x2.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, wpack.data_ptr(), rows, c, out.data_ptr(),
                                            torch.cuda.current_stream(x2.device).cuda_stream))
        return out

    def convt2x2(self, x: torch.Tensor, wpack: torch.Tensor, bias: torch.Tensor, skip: torch.Tensor = None, wskip: torch.Tensor = None) -> torch.Tensor:
        """ConvTranspose2d(c -> c/2, 2, stride 2) + bias on (b, h, w, c) float16 -> (b, 2h, 2w, c/2) (csrc/mst_mfma.hip); with
        skip / wskip: + skip @ W_skip^T in the same pass (the decoder's fusion conv, its `up` half folded into wpack by the caller)."""
        from .._lib import lib

        b, h, w, c = x.shape
        x = x.contiguous()
        out = torch.empty((b, 2 * h, 2 * w, c // 2), dtype=torch.float16, device=x.device)
        ctx = self.ctx(x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        if skip is None:
            ctx._check(lib.avx_mst_convt2x2(ctx._h, x.data_ptr(), wpack.data_ptr(), bias.data_ptr(), out.data_ptr(), b, h, w, c, st))
        else:
            assert skip.is_contiguous() and skip.shape == out.shape
            ctx._check(lib.avx_mst_convt2x2_fuse(ctx._h, x.data_ptr(), wpack.data_ptr(), bias.data_ptr(), skip.data_ptr(), wskip.data_ptr(), out.data_ptr(), b, h, w, c, st))
        return out

    def convt2x2_gram(self, x: torch.Tensor, wpack: torch.Tensor, bias: torch.Tensor, skip: torch.Tensor, wskip: torch.Tensor, wqk16: torch.Tensor):
        """convt2x2 with skip / wskip at c = 64 and, as its epilogue, the Gram pass of the MSAB block that follows (csrc/mst_mfma.hip::k_mst_convt2x2<64, true>):
        -> (out, gram (b, 1, 32, 32), nq (b, 32), nk (b, 32))."""
        from .._lib import lib

        b, h, w, c = x.shape
        assert c == 64 and skip.is_contiguous() and skip.shape == (b, 2 * h, 2 * w, 32)
        x = x.contiguous()
        out = torch.empty((b, 2 * h, 2 * w, 32), dtype=torch.float16, device=x.device)
        g = torch.empty((b, 1, 32, 32), dtype=torch.float32, device=x.device)
        nq = torch.empty((b, 32), dtype=torch.float32, device=x.device)
        nk = torch.empty((b, 32), dtype=torch.float32, device=x.device)
        ctx = self.ctx(x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        for i in range(b):
            ctx._check(lib.avx_mst_convt2x2_fuse_gram(ctx._h, x[i].data_ptr(), wpack.data_ptr(), bias.data_ptr(), skip[i].data_ptr(), wskip.data_ptr(), out[i].data_ptr(), h, w, c,
                                                      wqk16.data_ptr(), g[i].data_ptr(), nq[i].data_ptr(), nk[i].data_ptr(), st))
        return out, g, nq, nk

    def posemb(self, v: torch.Tensor, w1_c9: torch.Tensor, w2_c9: torch.Tensor, residual: torch.Tensor = None, bias: torch.Tensor = None) -> torch.Tensor:
        """dw3x3(gelu(dw3x3(v))) + residual + bias on (b, h, w, c) float16 in one pass (csrc/mst_mfma.hip::k_mst_posemb)."""
        from .._lib import lib

        b, h, w, c = v.shape
        v = v.contiguous()
        residual = residual.contiguous() if residual is not None else None
        out = torch.empty_like(v)
        ctx = self.ctx(v.device)
        ctx._check(lib.avx_mst_posemb(ctx._h, v.data_ptr(), w1_c9.data_ptr(), w2_c9.data_ptr(), residual.data_ptr() if residual is not None else None,
                                      bias.data_ptr() if bias is not None else None, out.data_ptr(), b, h, w, c, torch.cuda.current_stream(v.device).cuda_stream))
        return out

    def conv3x3(self, x: torch.Tensor, wpack: torch.Tensor, add: torch.Tensor = None) -> torch.Tensor:
        """Dense 3x3 conv (padding 1) [+ add] on a (b, h, w, 32) float16 tensor (csrc/mst_mfma.hip, implicit GEMM on MFMA)."""
        from .._lib import lib

        b, h, w, c = x.shape
        x = x.contiguous()
        add = add.contiguous() if add is not None else None
        out = torch.empty_like(x)
        ctx = self.ctx(x.device)
        ctx._check(lib.avx_mst_conv3x3_add(ctx._h, x.data_ptr(), wpack.data_ptr(), add.data_ptr() if add is not None else None, out.data_ptr(), b, h, w, c,
                                           torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def conv_in_u8(self, frame: torch.Tensor, pads, w_27x32: torch.Tensor) -> torch.Tensor:
        """uint8 (H, W, 3) frame -> conv_in's output (1, H + pads, W + pads, 32) float16 in one kernel (csrc/mst_fused.hip::k_mst_conv_in_u8):
        / 255, float16 rounding, reflect pad (top, bottom, left, right), 3x3 conv with zero padding."""
        from .._lib import lib

        H, W, _ = frame.shape
        t, b, l, r = pads
        frame = frame.contiguous()
        out = torch.empty((1, H + t + b, W + l + r, 32), dtype=torch.float16, device=frame.device)
        ctx = self.ctx(frame.device)
        ctx._check(lib.avx_mst_conv_in_u8(ctx._h, frame.data_ptr(), H, W, t, b, l, r, w_27x32.data_ptr(), out.data_ptr(),
                                          torch.cuda.current_stream(frame.device).cuda_stream))
        return out

    def down4x4(self, x: torch.Tensor, wpack16: torch.Tensor) -> torch.Tensor:
        """Conv2d(c -> 2c, 4, stride 2, padding 1), c = 32 or 64, on (b, h, w, c) float16: halo tiles brought in by LDS-direct loads,
        the wave's weights held in registers (csrc/mst_fused.hip::k_mst_down4x4_dma)."""
        from .._lib import lib

        b, h, w, c = x.shape
        assert x.is_contiguous() and c in (32, 64) and h % 2 == 0 and w % 2 == 0
        out = torch.empty((b, h // 2, w // 2, 2 * c), dtype=torch.float16, device=x.device)
        ctx = self.ctx(x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        per = max(1, (2**31 - 1) // (h * w * c * 2))  # the kernel addresses its input with 32-bit byte offsets: frames per launch
        for i in range(0, b, per):
            ctx._check(lib.avx_mst_down4x4(ctx._h, x[i : i + per].data_ptr(), wpack16.data_ptr(), out[i : i + per].data_ptr(), min(per, b - i), h, w, c, st))
        return out

    def conv3x3_lds_gram(self, x: torch.Tensor, wpack16: torch.Tensor, add: torch.Tensor, wqk16: torch.Tensor):
        """conv3x3_lds and, as its epilogue, the Gram pass of the MSAB block that follows (csrc/mst_fused.hip::k_mst_conv3x3_lds<., true>):
        -> (out, gram (b, 1, 32, 32), nq (b, 32), nk (b, 32)) as qkv_gram(out, ..., want_v=False) would return them."""
        from .._lib import lib

        b, h, w, c = x.shape
        x = x.contiguous()
        add = add.contiguous() if add is not None else None
        out = torch.empty_like(x)
        g = torch.empty((b, 1, 32, 32), dtype=torch.float32, device=x.device)
        nq = torch.empty((b, c), dtype=torch.float32, device=x.device)
        nk = torch.empty((b, c), dtype=torch.float32, device=x.device)
        ctx = self.ctx(x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        for i in range(b):
            ctx._check(lib.avx_mst_conv3x3_lds_gram(ctx._h, x[i].data_ptr(), wpack16.data_ptr(), add[i].data_ptr() if add is not None else None, out[i].data_ptr(), h, w, c,
                                                    wqk16.data_ptr(), g[i].data_ptr(), nq[i].data_ptr(), nk[i].data_ptr(), st))
        return out, g, nq, nk

    def conv3x3_lds_spectral(self, x: torch.Tensor, wpack16: torch.Tensor, add: torch.Tensor, weights, crop):
        """conv3x3_lds with the spectral integration of its output as the epilogue (csrc/mst_fused.hip::k_mst_conv3x3_lds<., false, true>): the cube is never
        written.  x / add: (1, h, w, 32) float16; weights: (3, 32) float32 NumPy (illuminant folded in); crop = (top, left, H, W) of the unpadded frame.
        -> (planes (3, H, W) float32, partials (n, 3, 2) float64-sized records, n): what HoneybeeOp.run_device(catches=...) takes."""
        import ctypes

        import numpy as np

        from .._lib import lib

        b, h, w, c = x.shape
        assert b == 1 and c == 32 and x.is_contiguous() and (add is None or add.is_contiguous())
        t, l, H, W = crop
        ctx = self.ctx(x.device)
        planes = torch.empty((3, H, W), dtype=torch.float32, device=x.device)
        ncu = torch.cuda.get_device_properties(x.device).multi_processor_count
        partials = torch.empty((ncu * 3, 3, 2), dtype=torch.float64, device=x.device)  # 16-byte records {float min, max; double sum}: at most 3 workgroups per CU
        wh = np.ascontiguousarray(weights, dtype=np.float32)
        assert wh.shape == (3, 32)
        n = ctypes.c_int(0)
        ctx._check(lib.avx_mst_conv3x3_lds_spectral(ctx._h, x.data_ptr(), wpack16.data_ptr(), add.data_ptr() if add is not None else None, h, w, c,
                                                    wh.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), t, l, H, W, planes.data_ptr(), partials.data_ptr(), ctypes.byref(n),
                                                    torch.cuda.current_stream(x.device).cuda_stream))
        return planes, partials, int(n.value)

    def conv3x3_lds(self, x: torch.Tensor, wpack16: torch.Tensor, add: torch.Tensor = None) -> torch.Tensor:
        """conv3x3 through an LDS halo tile (csrc/mst_fused.hip::k_mst_conv3x3_lds): every input pixel is fetched once."""
        from .._lib import lib

        b, h, w, c = x.shape
        x = x.contiguous()
        add = add.contiguous() if add is not None else None
        out = torch.empty_like(x)
        ctx = self.ctx(x.device)
        ctx._check(lib.avx_mst_conv3x3_lds(ctx._h, x.data_ptr(), wpack16.data_ptr(), add.data_ptr() if add is not None else None, out.data_ptr(), b, h, w, c,
                                           torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def attn_pack(self, gram: torch.Tensor, nq: torch.Tensor, nk: torch.Tensor, rescale: torch.Tensor, wproj_t: torch.Tensor) -> torch.Tensor:
        """gram (heads, 32, 32), nq / nk (c,), rescale (heads,), W_proj^T (c, c), all float32 -> the fused attention @ projection
        matrix M (c x c) as float16 MFMA fragments (csrc/mst_mfma.hip::k_mst_attn_pack)."""
        from .._lib import lib

        c = nq.numel()
        ctx = self.ctx(gram.device)
        out = torch.empty((c // 32, c // 8, 64, 4), dtype=torch.float16, device=gram.device)
        ctx._check(lib.avx_mst_attn_pack(ctx._h, gram.data_ptr(), nq.data_ptr(), nk.data_ptr(), rescale.data_ptr(), wproj_t.data_ptr(), c, out.data_ptr(),
                                         torch.cuda.current_stream(gram.device).cuda_stream))
        return out

    def attn_pack16(self, gram: torch.Tensor, nq: torch.Tensor, nk: torch.Tensor, rescale: torch.Tensor, wproj_t: torch.Tensor) -> torch.Tensor:
        """attn_pack in the K = 16 fragment order avx_mst_attn_tail takes (csrc/mst_mfma.hip::k_mst_attn_pack, layout 1)."""
        from .._lib import lib

        c = nq.numel()
        ctx = self.ctx(gram.device)
        out = torch.empty((c // 32, c // 16, 64, 8), dtype=torch.float16, device=gram.device)
        ctx._check(lib.avx_mst_attn_pack16(ctx._h, gram.data_ptr(), nq.data_ptr(), nk.data_ptr(), rescale.data_ptr(), wproj_t.data_ptr(), c, out.data_ptr(),
                                           torch.cuda.current_stream(gram.device).cuda_stream))
        return out

    def attn_pack_mx(self, gram: torch.Tensor, nq: torch.Tensor, nk: torch.Tensor, rescale: torch.Tensor, wproj_t: torch.Tensor) -> torch.Tensor:
        """attn_pack as the block-diagonal 16x16x32 fragments avx_mst_attn_tail_mx takes (csrc/mst_mfma.hip::k_mst_attn_pack, layout 2)."""
        from .._lib import lib

        c = nq.numel()
        ctx = self.ctx(gram.device)
        out = torch.empty((c // 8, c // 16, 64, 8), dtype=torch.float16, device=gram.device)
        ctx._check(lib.avx_mst_attn_pack_mx(ctx._h, gram.data_ptr(), nq.data_ptr(), nk.data_ptr(), rescale.data_ptr(), wproj_t.data_ptr(), c, out.data_ptr(),
                                            torch.cuda.current_stream(gram.device).cuda_stream))
        return out

    def attn_tail_mx(self, x: torch.Tensor, wvpack16: torch.Tensor, mpack_mx: torch.Tensor, dw1: torch.Tensor, dw2: torch.Tensor, bias: torch.Tensor,
                     out: torch.Tensor = None) -> torch.Tensor:
        """pos_emb(v) + v @ M + bias + x with v = float16(x @ W_v) formed inside the kernel and both depthwise convs on the matrix pipe, on one
        (h, w, c) float16 frame (csrc/mst_fused.hip::k_mst_attn_tail_mx)."""
        from .._lib import lib

        h, w, c = x.shape
        assert x.is_contiguous()
        out = torch.empty_like(x) if out is None else out
        ctx = self.ctx(x.device)
        ctx._check(lib.avx_mst_attn_tail_mx(ctx._h, x.data_ptr(), wvpack16.data_ptr(), mpack_mx.data_ptr(), dw1.data_ptr(), dw2.data_ptr(), bias.data_ptr(),
                                            out.data_ptr(), 1, h, w, c, torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def attn_tail(self, v: torch.Tensor, x: torch.Tensor, mpack16: torch.Tensor, taps1: torch.Tensor, taps2: torch.Tensor, bias: torch.Tensor,
                  out: torch.Tensor = None) -> torch.Tensor:
        """pos_emb(v) + v @ M + bias + x on one (h, w, c) float16 frame in one pass (csrc/mst_fused.hip::k_mst_attn_tail)."""
        from .._lib import lib

        h, w, c = v.shape
        assert v.is_contiguous() and x.is_contiguous() and x.shape == v.shape
        out = torch.empty_like(v) if out is None else out
        ctx = self.ctx(v.device)
        ctx._check(lib.avx_mst_attn_tail(ctx._h, v.data_ptr(), x.data_ptr(), mpack16.data_ptr(), taps1.data_ptr(), taps2.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                         1, h, w, c, torch.cuda.current_stream(v.device).cuda_stream))
        return out

    def attn_tail_x(self, x: torch.Tensor, wvpack16: torch.Tensor, mpack16: torch.Tensor, taps1: torch.Tensor, taps2: torch.Tensor, bias: torch.Tensor,
                    out: torch.Tensor = None) -> torch.Tensor:
        """pos_emb(v) + v @ M + bias + x with v = float16(x @ W_v) formed inside the kernel, on one (h, w, c) float16 frame
        (csrc/mst_fused.hip::k_mst_attn_tail, wvpack path): v never exists in HBM."""
        from .._lib import lib

        h, w, c = x.shape
        assert x.is_contiguous()
        out = torch.empty_like(x) if out is None else out
        ctx = self.ctx(x.device)
        ctx._check(lib.avx_mst_attn_tail_x(ctx._h, x.data_ptr(), wvpack16.data_ptr(), mpack16.data_ptr(), taps1.data_ptr(), taps2.data_ptr(), bias.data_ptr(),
                                           out.data_ptr(), 1, h, w, c, torch.cuda.current_stream(x.device).cuda_stream))
        return out

    def rowgemm_add(self, a: torch.Tensor, wpack: torch.Tensor, add: torch.Tensor = None, a2: torch.Tensor = None, wpack2: torch.Tensor = None) -> torch.Tensor:
        """[add +] a @ W [+ a2 @ W2] ((rows, c) float16, weights in fragment order; csrc/mst_mfma.hip); in place on `add`
        when it is given, else a new tensor."""
        from .._lib import lib

        rows, c = a.shape
        assert a.is_contiguous() and (add is None or (add.is_contiguous() and add.shape == a.shape)) and (a2 is None or (a2.is_contiguous() and a2.shape == a.shape))
        out = add if add is not None else torch.empty_like(a)
        ctx = self.ctx(a.device)
        ctx._check(lib.avx_mst_rowgemm_add(ctx._h, a.data_ptr(), wpack.data_ptr(), a2.data_ptr() if a2 is not None else None,
                                           wpack2.data_ptr() if a2 is not None else None, add.data_ptr() if add is not None else None, out.data_ptr(), rows, c,
                                           torch.cuda.current_stream(a.device).cuda_stream))
        return out

    def ffn_fused(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, w1p: torch.Tensor, taps: torch.Tensor, w2p: torch.Tensor,
                  eps: float = 1e-5, dwpack: torch.Tensor = None) -> torch.Tensor:
        """x + FeedForward(LayerNorm(x)) on (b, h, w, c) float16 in ONE kernel, the 4c hidden tile in LDS (csrc/mst_fused.hip).
        dwpack (pack_dw_mfma): the depthwise conv runs on the matrix pipe (avx_mst_ffn_fused_mx); else taps ([9][4c]) on the vector unit."""
        from .._lib import lib

        b, h, w, c = x.shape
        x = x.contiguous()
        out = torch.empty_like(x)
        ctx = self.ctx(x.device)
        st = torch.cuda.current_stream(x.device).cuda_stream
        if dwpack is not None:
            ctx._check(lib.avx_mst_ffn_fused_mx(ctx._h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, w1p.data_ptr(), dwpack.data_ptr(), w2p.data_ptr(),
                                                out.data_ptr(), b, h, w, c, st))
        else:
            ctx._check(lib.avx_mst_ffn_fused(ctx._h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, w1p.data_ptr(), taps.data_ptr(), w2p.data_ptr(),
                                             out.data_ptr(), b, h, w, c, st))
        return out

    def dw_gemm_add(self, hidden: torch.Tensor, w_c9: torch.Tensor, w2pack: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """x += gelu(dwconv3x3(hidden)) @ w2, in place (hidden (b, h, w, 4c), x (b, h, w, c), float16; csrc/mst_mfma.hip)."""
        from .._lib import lib

        b, h, w, c = x.shape
        assert hidden.is_contiguous() and x.is_contiguous() and hidden.shape == (b, h, w, 4 * c)
        ctx = self.ctx(x.device)
        ctx._check(lib.avx_mst_dw_gemm_add(ctx._h, hidden.data_ptr(), w_c9.data_ptr(), w2pack.data_ptr(), x.data_ptr(), x.data_ptr(), b, h, w, c,
                                           torch.cuda.current_stream(x.device).cuda_stream))
        return x

    def dwconv(self, x: torch.Tensor, w_c9: torch.Tensor, gelu: bool, residual: torch.Tensor = None, bias: torch.Tensor = None) -> torch.Tensor:
        from .._lib import lib

        b, h, w, c = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        ctx = self.ctx(x.device)
        if residual is not None or bias is not None:  # fused epilogue: + residual + bias (float16 kernels only)
            residual = residual.contiguous() if residual is not None else None
            ctx._check(lib.avx_dwconv3x3_nhwc_add(ctx._h, x.data_ptr(), w_c9.data_ptr(), y.data_ptr(), b, h, w, c, 1 if gelu else 0,
                                                  residual.data_ptr() if residual is not None else None, bias.data_ptr() if bias is not None else None,
                                                  torch.cuda.current_stream(x.device).cuda_stream))
            return y
        ctx._check(lib.avx_dwconv3x3_nhwc(ctx._h, x.data_ptr(), w_c9.data_ptr(), y.data_ptr(), self._dt(x), b, h, w, c, 1 if gelu else 0,
                                          torch.cuda.current_stream(x.device).cuda_stream))
        return y

    def layernorm(self, x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5, group: int = 0, real: int = 0) -> torch.Tensor:
        """LayerNorm over the last dim; with group/real, over the channels c with (c % group) < real (padded groups)."""
        from .._lib import lib

        c = x.shape[-1]
        x = x.contiguous()
        y = torch.empty_like(x)
        ctx = self.ctx(x.device)
        ctx._check(lib.avx_layernorm_rows_grouped(ctx._h, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), self._dt(x), x.numel() // c, c, eps,
                                                  group or c, real or c, torch.cuda.current_stream(x.device).cuda_stream))
        return y


_AVX = _AvxOps()


def _msab_keys(prefix: str, dim: int) -> Dict[str, tuple]:
    a, f = f"{prefix}.blocks.0.0", f"{prefix}.blocks.0.1"
    return {
        f"{a}.to_q.weight": (dim, dim), f"{a}.to_k.weight": (dim, dim), f"{a}.to_v.weight": (dim, dim),
        f"{a}.rescale": (dim // DIM, 1, 1), f"{a}.proj.weight": (dim, dim), f"{a}.proj.bias": (dim,),
        f"{a}.pos_emb.0.weight": (dim, 1, 3, 3), f"{a}.pos_emb.2.weight": (dim, 1, 3, 3),
        f"{f}.fn.net.0.weight": (4 * dim, dim, 1, 1), f"{f}.fn.net.2.weight": (4 * dim, 1, 3, 3),
        f"{f}.fn.net.4.weight": (dim, 4 * dim, 1, 1), f"{f}.norm.weight": (dim,), f"{f}.norm.bias": (dim,),
    }


def reference_key_names(stage: int = 3) -> Dict[str, tuple]:
    """name -> shape of every tensor in the reference's MST_Plus_Plus().state_dict()."""
    keys: Dict[str, tuple] = {"conv_in.weight": (DIM, 3, 3, 3)}
    for s in range(stage):
        p = f"body.{s}"
        keys[f"{p}.embedding.weight"] = (DIM, DIM, 3, 3)
        d = DIM
        for i in range(2):
            keys.update(_msab_keys(f"{p}.encoder_layers.{i}.0", d))
            keys[f"{p}.encoder_layers.{i}.1.weight"] = (2 * d, d, 4, 4)
            d *= 2
        keys.update(_msab_keys(f"{p}.bottleneck", d))
        for i in range(2):
            keys[f"{p}.decoder_layers.{i}.0.weight"] = (d, d // 2, 2, 2)
            keys[f"{p}.decoder_layers.{i}.0.bias"] = (d // 2,)
            keys[f"{p}.decoder_layers.{i}.1.weight"] = (d // 2, d, 1, 1)
            keys.update(_msab_keys(f"{p}.decoder_layers.{i}.2", d // 2))
            d //= 2
        keys[f"{p}.mapping.weight"] = (DIM, DIM, 3, 3)
    keys["conv_out.weight"] = (DIM, DIM, 3, 3)
    return keys


class MSTPlusPlus(torch.nn.Module):
    def __init__(self, stage: int = 3):
        super().__init__()
        self.stage = stage
        self._shapes = reference_key_names(stage)
        self.w = torch.nn.ParameterDict({self._pk(k): torch.nn.Parameter(torch.zeros(s), requires_grad=False) for k, s in self._shapes.items()})
        self._prepared: Dict[str, torch.Tensor] = {}
        # Diagnostics (tests/test_mstpp.py): when a list, every MS_MSA appends its (heads, 31, 31) attention matrix -- the
        # softmax output of MST_Plus_Plus.py:131 -- recomputed from the block's Gram matrix and norms.
        self.capture_attn = None

    def _capture(self, gram: torch.Tensor, nq: torch.Tensor, nk: torch.Tensor, p: str, heads: int):
        """gram (b, heads, >=31, >=31), nq / nk (b, heads * 32): the attention matrix the fused kernels never materialise."""
        b = gram.shape[0]
        a = gram[:, :, :DIM, :DIM].float() / (nk.reshape(b, heads, PAD, 1)[:, :, :DIM].clamp_min(1e-12) * nq.reshape(b, heads, 1, PAD)[..., :DIM].clamp_min(1e-12))
        a = a * self._p(p + ".rescale").float().reshape(1, heads, 1, 1)
        self.capture_attn.append(a.softmax(dim=-1).cpu())

    @staticmethod
    def _pk(k: str) -> str:
        return k.replace(".", "__")

    # ---- weights -----------------------------------------------------------------------------------
    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True) -> "MSTPlusPlus":
        """Accepts the reference's checkpoint['state_dict'] (or a bare state_dict), 'module.' prefixes and all."""
        if "state_dict" in sd and not any(k.endswith(".weight") for k in sd):
            sd = sd["state_dict"]
        sd = {k.replace("module.", "", 1) if k.startswith("module.") else k: v for k, v in sd.items()}
        missing = [k for k in self._shapes if k not in sd]
        extra = [k for k in sd if k not in self._shapes]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:3]}..., unexpected {extra[:3]}...")
        with torch.no_grad():
            for k, shape in self._shapes.items():
                if k in sd:
                    t = torch.as_tensor(sd[k])
                    if tuple(t.shape) != tuple(shape):
                        raise ValueError(f"{k}: shape {tuple(t.shape)} != {shape}")
                    self.w[self._pk(k)].copy_(t.to(self.w[self._pk(k)].dtype))
        self._prepared.clear()
        return self

    def reference_state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.w[self._pk(k)].detach() for k in self._shapes}

    def init_seeded(self, seed: int = 0) -> "MSTPlusPlus":
        """Random weights (no checkpoint ships with the reference): N(0, 0.02) linears like
        MST._init_weights (:231-238), unit LayerNorm, Kaiming-uniform-scale convs."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for k, shape in self._shapes.items():
                p = self.w[self._pk(k)]
                if k.endswith("norm.weight") or k.endswith("rescale"):
                    p.fill_(1.0)
                elif k.endswith("bias"):
                    p.zero_()
                elif len(shape) == 2:
                    p.copy_(torch.randn(shape, generator=g) * 0.02)
                else:
                    fan_in = shape[1] * shape[2] * shape[3]
                    bound = (1.0 / fan_in) ** 0.5
                    p.copy_((torch.rand(shape, generator=g) * 2 - 1) * bound)
        self._prepared.clear()
        return self

    def _p(self, k: str) -> torch.Tensor:
        return self.w[self._pk(k)]

    def _prep(self, name: str, build):
        """Derived weights (stacked QKV, 2-D views of 1x1 convs), cached per dtype/device."""
        ref = self._p("conv_in.weight")
        key = f"{name}|{ref.dtype}|{ref.device}"
        t = self._prepared.get(key)
        if t is None:
            t = build()
            self._prepared[key] = t
        return t

    def _w(self, k: str, dims) -> torch.Tensor:
        """Reference weight `k` with the channel dims `dims` zero-padded to 32-wide groups (cached)."""
        return self._prep(k + ".pad" + "".join(map(str, dims)), lambda: pad_channels(self._p(k), dims).contiguous())

    # ---- blocks (x is NHWC, channel groups 32 wide) -------------------------------------------------
    def _conv3(self, x: torch.Tensor, key: str, add: torch.Tensor = None) -> torch.Tensor:
        """31 -> 31 channel 3x3 conv (padding 1) [+ add]: MFMA implicit GEMM for float16 on the GPU, F.conv2d otherwise."""
        if _AVX.fused_ok(x) and x.shape[-1] == 32 and _AVX._conv_lds:
            wq = self._prep(key + ".frag9k16", lambda: torch.stack([pack_fragments16(self._w(key, (0, 1))[:, :, t // 3, t % 3].t().contiguous()) for t in range(9)]).contiguous())
            return _AVX.conv3x3_lds(x, wq, add)
        if _AVX.fused_ok(x) and x.shape[-1] == 32:
            wp = self._prep(key + ".frag9", lambda: torch.stack([pack_fragments(self._w(key, (0, 1))[:, :, t // 3, t % 3].t().contiguous(), True) for t in range(9)]).contiguous())
            return _AVX.conv3x3(x, wp, add)
        y = self._conv_nhwc(x, self._w(key, (0, 1)), padding=1)
        return y + add if add is not None else y

    def _conv_nhwc(self, x: torch.Tensor, w: torch.Tensor, **kw) -> torch.Tensor:
        y = F.conv2d(x.permute(0, 3, 1, 2), w, **kw)  # NCHW view of channels-last memory: no copy
        return y.permute(0, 2, 3, 1)

    def _ms_msa(self, x: torch.Tensor, p: str, heads: int, pre=None) -> torch.Tensor:
        b, h, w, c = x.shape  # c = heads * 32
        n = h * w
        x2 = x.reshape(b, n, c)
        wqkv = self._prep(p + ".qkv", lambda: torch.cat([self._w(p + ".to_q.weight", (0, 1)), self._w(p + ".to_k.weight", (0, 1)),
                                                         self._w(p + ".to_v.weight", (0, 1))], 0).t().contiguous())
        d = c // heads  # 32: 31 real channels + the zero padding
        if _AVX.fused_ok(x):  # one pass over x: q, k live only in MFMA accumulators, v is the only tensor written
            tailx = _AVX._tail and _AVX._tailx and (c in (32, 64) or (c == 128 and _AVX._dwmx))  # the tail forms v itself: this pass writes nothing but the Gram partials
            if _AVX._qkv16:  # K = 16 MFMAs: a quarter of the matrix-core cycles (the K = 8 kernel is bound by them)
                wpk = self._prep(p + ".qkv.frag16", lambda: pack_qkv16(wqkv))
            else:
                wpk = self._prep(p + ".qkv.frag", lambda: torch.cat([pack_fragments(wqkv[:, : 2 * c], False), pack_fragments(wqkv[:, 2 * c :], True)], 0).contiguous())
            if pre is not None and tailx:  # (gram, nq, nk) came with x, from the epilogue of the conv that produced it
                v, (gram, nq, nk) = None, pre
            else:
                v, gram, nq, nk = _AVX.qkv_gram(x2, wpk, heads, want_v=not tailx, k16=_AVX._qkv16)
            if self.capture_attn is not None:
                self._capture(gram, nq, nk, p, heads)
            # softmax(gram / (nk nq^T) * rescale) and M = blockdiag(attn_h^T) @ W_proj^T in one small launch, M already in fragment order
            resc = self._prep(p + ".rescale32", lambda: self._p(p + ".rescale").float().reshape(heads).contiguous())
            wpt = self._prep(p + ".proj.t32", lambda: self._w(p + ".proj.weight", (0, 1)).t().float().contiguous())
            bias32 = self._prep(p + ".proj.bias32", lambda: self._w(p + ".proj.bias", (0,)).float().contiguous())
            vi = v.reshape(b, h, w, c) if v is not None else None
            if _AVX._tail and (c in (32, 64) or tailx):  # v @ M + bias + pos_emb(v) + x in ONE pass over v and x (the block's `msa(x) + x`, :183)
                k1, k2 = p + ".pos_emb.0.weight", p + ".pos_emb.2.weight"
                t1 = self._prep(k1 + ".t9h", lambda: self._w(k1, (0,)).reshape(c, 9).t().contiguous())  # [9][c] float16, tap-major
                t2 = self._prep(k2 + ".t9h", lambda: self._w(k2, (0,)).reshape(c, 9).t().contiguous())
                xc = x.contiguous()
                out = torch.empty_like(xc)
                if tailx and _AVX._dwmx:  # both depthwise convs and the projection on the matrix pipe
                    wv16 = self._prep(p + ".wv.frag16h", lambda: pack_fragments16(wqkv[:, 2 * c :].contiguous(), halfrow=True))
                    gs = _AVX.gelu_prescale()  # the GELU between the two convs works on x / gs (csrc/mst_common.h): first conv / gs, second * gs
                    d1 = self._prep(k1 + f".dwmx/{gs}", lambda: pack_dw_mfma((self._w(k1, (0,)).float() / gs).to(x.dtype)))
                    d2 = self._prep(k2 + f".dwmx*{gs}", lambda: pack_dw_mfma((self._w(k2, (0,)).float() * gs).to(x.dtype)))
                    for i in range(b):
                        _AVX.attn_tail_mx(xc[i], wv16, _AVX.attn_pack_mx(gram[i], nq[i], nk[i], resc, wpt), d1, d2, bias32, out[i])
                    return out
                if tailx:
                    wv16 = self._prep(p + ".wv.frag16", lambda: pack_fragments16(wqkv[:, 2 * c :].contiguous()))
                    for i in range(b):
                        _AVX.attn_tail_x(xc[i], wv16, _AVX.attn_pack16(gram[i], nq[i], nk[i], resc, wpt), t1, t2, bias32, out[i])
                    return out
                for i in range(b):
                    _AVX.attn_tail(vi[i], xc[i], _AVX.attn_pack16(gram[i], nq[i], nk[i], resc, wpt), t1, t2, bias32, out[i])
                return out
            # the block's `msa(x) + x` (:183): pos_emb's second conv adds x and the bias, the projection GEMM accumulates onto it in place
            if _AVX._posemb:
                k1, k2 = p + ".pos_emb.0.weight", p + ".pos_emb.2.weight"
                w91 = self._prep(k1 + ".c9f32", lambda: self._w(k1, (0,)).reshape(c, 9).float().contiguous())
                w92 = self._prep(k2 + ".c9f32", lambda: self._w(k2, (0,)).reshape(c, 9).float().contiguous())
                pe = _AVX.posemb(vi, w91, w92, x, bias32)
            else:
                pe = self._dw(self._dw(vi, p + ".pos_emb.0.weight", gelu=True), p + ".pos_emb.2.weight", gelu=False, residual=x, bias=bias32)
            for i in range(b):
                _AVX.rowgemm_add(v[i], _AVX.attn_pack(gram[i], nq[i], nk[i], resc, wpt), pe[i].reshape(n, c))
            return pe
        qkv = x2 @ wqkv  # (b, n, 3c): one GEMM, one pass over x
        q, k, v = qkv[..., :c], qkv[..., c : 2 * c], qkv[..., 2 * c :]
        # Gram matrix per head over ALL pixels (the global contraction of :129), fp32 accumulate
        if x.is_cuda and _AVX.enabled:
            gram, nq, nk = _AVX.gram(qkv, heads)  # one pass over qkv: k^T q and every column norm (csrc/mst.hip)
            nq = nq.reshape(b, heads, 1, d).clamp_min(1e-12)  # F.normalize eps
            nk = nk.reshape(b, heads, d, 1).clamp_min(1e-12)
        else:
            qh = q.reshape(b, n, heads, d).permute(0, 2, 1, 3)  # (b, heads, n, d)
            kh = k.reshape(b, n, heads, d).permute(0, 2, 1, 3)
            gram = torch.matmul(kh.transpose(-2, -1), qh).float()  # (b, heads, d, d): [i][j] = k_i . q_j
            nq = torch.linalg.vector_norm(q.float(), dim=1).reshape(b, heads, 1, d).clamp_min(1e-12)
            nk = torch.linalg.vector_norm(k.float(), dim=1).reshape(b, heads, d, 1).clamp_min(1e-12)
        attn = gram / (nk * nq)
        attn = attn * self._p(p + ".rescale").float().reshape(1, heads, 1, 1)
        attn[..., DIM:] = float("-inf")  # the padding column takes no part in the softmax over j
        attn = attn.softmax(dim=-1)
        if self.capture_attn is not None:
            self.capture_attn.append(attn[:, :, :DIM, :DIM].float().cpu())
        # out_c = proj(concat_h(attn_h @ v_h)) == v @ M + bias,  M = blockdiag(attn_h^T) @ W_proj^T
        wp = self._w(p + ".proj.weight", (0, 1))  # (c_out, c_in)
        wp_h = wp.t().reshape(heads, d, c).float()  # rows of W_proj^T grouped by head (padding rows are zero)
        M = torch.matmul(attn.transpose(-2, -1), wp_h.unsqueeze(0)).reshape(b, c, c).to(x.dtype)  # (b, c, c)
        vi = v.reshape(b, h, w, c)
        out_c = torch.baddbmm(self._w(p + ".proj.bias", (0,)).to(x.dtype).reshape(1, 1, c), v, M).reshape(b, h, w, c)
        pe = self._dw(self._dw(vi, p + ".pos_emb.0.weight", gelu=True), p + ".pos_emb.2.weight", gelu=False)
        return out_c + pe + x

    def _dw(self, x: torch.Tensor, key: str, gelu: bool, residual: torch.Tensor = None, bias: torch.Tensor = None) -> torch.Tensor:
        """Depthwise 3x3 (+ GELU) (+ residual + per-channel bias) on an NHWC tensor: csrc/mst.hip on the GPU, torch ops otherwise."""
        c = x.shape[-1]
        if x.is_cuda and _AVX.enabled:
            w9 = self._prep(key + ".c9f32", lambda: self._w(key, (0,)).reshape(c, 9).float().contiguous())
            return _AVX.dwconv(x, w9, gelu, residual, bias)
        y = self._conv_nhwc(x, self._w(key, (0,)), padding=1, groups=c)
        y = F.gelu(y) if gelu else y
        if residual is not None:
            y = y + residual
        return y + bias.to(y.dtype) if bias is not None else y

    def _ffn(self, x: torch.Tensor, p: str) -> torch.Tensor:
        b, h, w, c = x.shape
        g32 = self._prep(p + ".g32", lambda: self._w(p + ".norm.weight", (0,)).float().contiguous())
        b32 = self._prep(p + ".b32", lambda: self._w(p + ".norm.bias", (0,)).float().contiguous())
        w2 = self._prep(p + ".w2", lambda: self._w(p + ".fn.net.4.weight", (0, 1)).reshape(c, 4 * c).t().contiguous())
        if _AVX.fused_ok(x) and _AVX._ffn and c in _AVX.FFN_FUSED_C:  # the whole FeedForward + residual in one kernel: the hidden tensor stays in LDS
            key = p + ".fn.net.2.weight"
            if _AVX._dwmx:
                gs = _AVX.gelu_prescale()  # both GELUs work on x / gs (csrc/mst_common.h): W1 / gs, the depthwise conv as it is (linear: x2 / gs again), W2 * gs
                w1s = self._prep(p + f".w1.frag16.ln/{gs}", lambda: pack_fragments16(fold_layernorm(self._w(p + ".fn.net.0.weight", (0, 1)).reshape(4 * c, c).t(), g32, b32, gs).to(x.dtype)))
                w2s = self._prep(p + f".w2.frag16*{gs}", lambda: pack_fragments16((w2.float() * gs).to(x.dtype).contiguous()))
                return _AVX.ffn_fused(x, g32, b32, w1s, None, w2s, dwpack=self._prep(key + ".dwmx", lambda: pack_dw_mfma(self._w(key, (0,)))))
            w1q = self._prep(p + ".w1.frag16", lambda: pack_fragments16(self._w(p + ".fn.net.0.weight", (0, 1)).reshape(4 * c, c).t().contiguous()))
            w2q = self._prep(p + ".w2.frag16", lambda: pack_fragments16(w2))
            t9 = self._prep(key + ".t9h", lambda: self._w(key, (0,)).reshape(4 * c, 9).t().contiguous())  # [9][4c], tap-major, the model's own float16 values
            return _AVX.ffn_fused(x, g32, b32, w1q, t9, w2q)
        if _AVX.fused_ok(x):  # LayerNorm -> 1x1 conv -> GELU on the matrix cores, the hidden tensor is written once
            w1p = self._prep(p + ".w1.frag", lambda: pack_fragments(self._w(p + ".fn.net.0.weight", (0, 1)).reshape(4 * c, c).t().contiguous(), True))
            y = _AVX.ln_gemm_gelu(x.reshape(b * h * w, c), g32, b32, w1p).reshape(b, h, w, 4 * c)
            if c in (32, 64) and x.is_contiguous() and _AVX._ffn2:  # depthwise 3x3 -> GELU -> 1x1 conv -> + x in one kernel, in place on x
                key = p + ".fn.net.2.weight"
                w9 = self._prep(key + ".c9f32", lambda: self._w(key, (0,)).reshape(4 * c, 9).float().contiguous())
                w2p = self._prep(p + ".w2.frag", lambda: pack_fragments(w2, True))
                return _AVX.dw_gemm_add(y, w9, w2p, x)
            y = self._dw(y, p + ".fn.net.2.weight", gelu=True)
            # `ffn(x) + x` (:184) inside the GEMM, in place: x is the MSA half's own output buffer, nobody else holds it
            return x.reshape(b * h * w, c).addmm_(y.reshape(b * h * w, 4 * c), w2).reshape(b, h, w, c)
        if x.is_cuda and _AVX.enabled:
            y = _AVX.layernorm(x, g32, b32, group=PAD, real=DIM)
        else:  # LayerNorm over the real channels only; the padding channels stay zero
            real = self._prep(p + ".real", lambda: pad_index(c // PAD * DIM, x.device))
            yr = F.layer_norm(x.float().index_select(-1, real), (real.numel(),), g32.index_select(0, real), b32.index_select(0, real))
            y = torch.zeros_like(x, dtype=torch.float32).index_copy_(-1, real, yr).to(x.dtype)
        w1 = self._prep(p + ".w1", lambda: self._w(p + ".fn.net.0.weight", (0, 1)).reshape(4 * c, c).t().contiguous())
        y = F.gelu(y.reshape(b, h * w, c) @ w1).reshape(b, h, w, 4 * c)
        y = self._dw(y, p + ".fn.net.2.weight", gelu=True)
        return (y.reshape(b, h * w, 4 * c) @ w2).reshape(b, h, w, c) + x

    def _msab(self, x: torch.Tensor, p: str, heads: int, pre=None) -> torch.Tensor:
        """MSAB :176-186; both residual adds happen inside _ms_msa / _ffn (fused into their last launch on the GPU)."""
        x = self._ms_msa(x, p + ".blocks.0.0", heads, pre)
        return self._ffn(x, p + ".blocks.0.1")

    def _mst(self, x: torch.Tensor, p: str) -> torch.Tensor:
        pre = None
        ekey, a0 = p + ".embedding.weight", f"{p}.encoder_layers.0.0.blocks.0.0"
        if _AVX.fused_ok(x) and x.shape[-1] == 32 and _AVX._conv_lds and _AVX._convgram and _AVX._qkv16 and _AVX._tail and _AVX._tailx:
            # the embedding conv hands the first block its Gram matrix and column norms: the block's own pass over the conv's output is not run
            wq = self._prep(ekey + ".frag9k16", lambda: torch.stack([pack_fragments16(self._w(ekey, (0, 1))[:, :, t // 3, t % 3].t().contiguous()) for t in range(9)]).contiguous())
            wqkv = self._prep(a0 + ".qkv", lambda: torch.cat([self._w(a0 + ".to_q.weight", (0, 1)), self._w(a0 + ".to_k.weight", (0, 1)),
                                                              self._w(a0 + ".to_v.weight", (0, 1))], 0).t().contiguous())
            wqk = self._prep(a0 + ".qkv.frag16", lambda: pack_qkv16(wqkv))  # tiles 0 (q) and 1 (k) are what the epilogue reads
            fea, g, nq, nk = _AVX.conv3x3_lds_gram(x, wq, None, wqk)
            pre = (g, nq, nk)
        else:
            fea = self._conv3(x, ekey)
        skips: List[torch.Tensor] = []
        heads = 1
        for i in range(2):
            fea = self._msab(fea, f"{p}.encoder_layers.{i}.0", heads, pre if i == 0 else None)
            skips.append(fea)
            dkey = f"{p}.encoder_layers.{i}.1.weight"
            if _AVX.fused_ok(fea) and _AVX._down and fea.shape[-1] in (32, 64) and fea.is_contiguous():  # both encoder steps: the hand-written implicit GEMM
                fea = _AVX.down4x4(fea, self._prep(dkey + ".frag16", lambda: pack_down4x4(self._w(dkey, (0, 1)))))
            else:
                fea = self._conv_nhwc(fea, self._w(dkey, (0, 1)), stride=2, padding=1)
            heads *= 2
        fea = self._msab(fea, p + ".bottleneck", heads)
        for i in range(2):
            kw = f"{p}.decoder_layers.{i}.0.weight"
            if _AVX.fused_ok(fea) and fea.shape[-1] in (64, 128) and _AVX._upfuse:
                # transposed conv + the fusion conv's [up | skip] product in ONE pass: the conv's `up` half folded into the four taps
                # (W_tap @ W_up^T: exact in real arithmetic, one float16 rounding less than the reference's up -> cat -> conv)
                ch = fea.shape[-1] // 2
                wfk = f"{p}.fuse{i}"
                wf_ = self._prep(wfk, lambda: self._w(f"{p}.decoder_layers.{i}.1.weight", (0, 1)).reshape(ch, 2 * ch).t().contiguous())
                gt = self._prep(kw + ".upfuse4", lambda: torch.stack([pack_fragments((self._w(kw, (0, 1))[:, :, t // 2, t % 2].float() @ wf_[:ch].float()).half().contiguous(), True)
                                                                     for t in range(4)]).contiguous())
                gb = self._prep(kw + ".upfuse.bias", lambda: (self._w(f"{p}.decoder_layers.{i}.0.bias", (0,)).float() @ wf_[:ch].float()).contiguous())
                wbot_ = self._prep(f"{p}.fuse{i}.bot", lambda: pack_fragments(wf_[ch:].contiguous(), True))
                heads //= 2
                a2 = f"{p}.decoder_layers.{i}.2.blocks.0.0"
                if ch == 32 and _AVX._convgram and _AVX._qkv16 and _AVX._tail and _AVX._tailx:  # back at full resolution: the block's Gram pass rides on this launch
                    wqkv2 = self._prep(a2 + ".qkv", lambda: torch.cat([self._w(a2 + ".to_q.weight", (0, 1)), self._w(a2 + ".to_k.weight", (0, 1)),
                                                                       self._w(a2 + ".to_v.weight", (0, 1))], 0).t().contiguous())
                    wqk2 = self._prep(a2 + ".qkv.frag16", lambda: pack_qkv16(wqkv2))
                    fea, g2, nq2, nk2 = _AVX.convt2x2_gram(fea, gt, gb, skips[1 - i].contiguous(), wbot_, wqk2)
                    fea = self._msab(fea, f"{p}.decoder_layers.{i}.2", heads, (g2, nq2, nk2))
                    continue
                fea = _AVX.convt2x2(fea, gt, gb, skips[1 - i].contiguous(), wbot_)
                fea = self._msab(fea, f"{p}.decoder_layers.{i}.2", heads)
                continue
            if _AVX.fused_ok(fea) and fea.shape[-1] in (64, 128):  # four 1x1 products on the matrix cores (kernel size == stride)
                wt = self._prep(kw + ".frag4", lambda: torch.stack([pack_fragments(self._w(kw, (0, 1))[:, :, t // 2, t % 2].contiguous(), True) for t in range(4)]).contiguous())
                bt = self._prep(kw + ".bias32", lambda: self._w(f"{p}.decoder_layers.{i}.0.bias", (0,)).float().contiguous())
                up = _AVX.convt2x2(fea, wt, bt)
            else:
                up = F.conv_transpose2d(fea.permute(0, 3, 1, 2), self._w(kw, (0, 1)), self._w(f"{p}.decoder_layers.{i}.0.bias", (0,)), stride=2).permute(0, 2, 3, 1)
            heads //= 2
            skip = skips[1 - i]
            b, h, w, ch = up.shape
            c2 = 2 * ch
            # the fusion conv's 2c input channels are [up | skip], each half group-wise padded
            wf = self._prep(f"{p}.fuse{i}", lambda: self._w(f"{p}.decoder_layers.{i}.1.weight", (0, 1)).reshape(c2 // 2, c2).t().contiguous())
            if _AVX.fused_ok(skip):  # two half-K products, the concatenation is never built
                up, skip = up.contiguous(), skip.contiguous()
                wtop = self._prep(f"{p}.fuse{i}.top", lambda: pack_fragments(wf[:ch].contiguous(), True))
                wbot = self._prep(f"{p}.fuse{i}.bot", lambda: pack_fragments(wf[ch:].contiguous(), True))
                fea = _AVX.rowgemm_add(up.reshape(b * h * w, ch), wtop, None, skip.reshape(b * h * w, ch), wbot).reshape(b, h, w, ch)
            else:
                cat = torch.cat([up, skip], dim=-1)
                fea = (cat.reshape(b, h * w, c2) @ wf).reshape(b, h, w, c2 // 2)
            fea = self._msab(fea, f"{p}.decoder_layers.{i}.2", heads)
        return self._conv3(fea, p + ".mapping.weight", add=x)

    @torch.no_grad()
    def forward_from_u8(self, frame: torch.Tensor, pads, spectral=None):
        """uint8 (H, W, 3) device frame -> (1, Hp, Wp, 32) cube on the reflect-padded frame (pads = (top, bottom, left, right), a
        multiple of 8 in both directions afterwards): conv_in fused with the input preparation, then the body of forward_nhwc.
        spectral = (3, 32) float32 weights: the cube is not returned (nor written) but integrated in conv_out's epilogue ->
        (planes (3, H, W) float32 of the UNPADDED frame, partials, n_partials) (see _AvxOps.conv3x3_lds_spectral)."""
        w27 = self._prep("conv_in.w27x32", lambda: self._w("conv_in.weight", (0,)).permute(2, 3, 1, 0).reshape(27, PAD).float().contiguous())
        x = _AVX.conv_in_u8(frame, pads, w27)
        assert x.shape[1] % 8 == 0 and x.shape[2] % 8 == 0, "pad the frame to a multiple of 8 (predict_torch.py pads to 16)"
        if spectral is None:
            return self._body(x)
        H, W, _ = frame.shape
        return self._body(x, spectral=(spectral, (pads[0], pads[2], H, W)))

    def can_fuse_spectral(self) -> bool:
        return self.can_fuse_conv_in() and _AVX._conv_lds and _AVX._specfuse

    def _body(self, x: torch.Tensor, spectral=None):
        hfe = x
        for s in range(self.stage):
            hfe = self._mst(hfe, f"body.{s}")
        if spectral is not None:  # conv_out + x with the spectral integration as its epilogue
            key = "conv_out.weight"
            wq = self._prep(key + ".frag9k16", lambda: torch.stack([pack_fragments16(self._w(key, (0, 1))[:, :, t // 3, t % 3].t().contiguous()) for t in range(9)]).contiguous())
            return _AVX.conv3x3_lds_spectral(hfe.contiguous(), wq, x.contiguous(), spectral[0], spectral[1])
        return self._conv3(hfe, "conv_out.weight", add=x)

    def can_fuse_conv_in(self) -> bool:
        ref = self._p("conv_in.weight")
        return _AVX.enabled and _AVX._mfma and _AVX._conv_in and ref.is_cuda and ref.dtype == torch.float16

    @torch.no_grad()
    def forward_nhwc(self, x: torch.Tensor) -> torch.Tensor:
        """x: (b, 3, H, W) in [0, 1] -> (b, H, W, 32) channels-last, bands 0..30 + one zero padding channel (the layout
        the device hand-off consumes).  Same pad-to-8 / crop as MST_Plus_Plus.forward :279-293."""
        b, c, h_inp, w_inp = x.shape
        pad_h, pad_w = (8 - h_inp % 8) % 8, (8 - w_inp % 8) % 8
        if pad_h or pad_w:
            x = F.pad(x, [0, pad_w, 0, pad_h], mode="reflect")
        x = x.to(self._p("conv_in.weight").dtype).contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)  # NHWC view
        x = self._conv_nhwc(x, self._w("conv_in.weight", (0,)), padding=1)
        return self._body(x)[:, :h_inp, :w_inp, :]

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (b, 3, H, W) in [0, 1] -> (b, 31, H, W), the reference's signature."""
        return self.forward_nhwc(x)[..., :DIM].permute(0, 3, 1, 2)


def hbm_bytes_per_px(stage: int = 3) -> float:
    """Algorithmic HBM bytes of ONE float16 forward pass per full-resolution pixel at the CURRENT fusion level: for every
    launch of `MSTPlusPlus.forward_nhwc` on the GPU, the tensors it must read plus the tensors it must write (weights and
    halo re-reads excluded), summed over the U-shaped stages.  This is the denominator of bench.py's `roofline.hbm` for the
    MST++ route; it changes when kernels are fused, so it lives next to them.  A C-channel float16 tensor costs
    2C bytes per pixel of ITS resolution: 1 (C = 32), 1/4 (C = 64), 1/16 (C = 128) of the full-resolution pixel count."""

    def msab(c: int) -> float:
        t = 2.0 * c  # one activation tensor
        if _AVX._tail and _AVX._tailx and (c in (32, 64) or (c == 128 and _AVX._dwmx)):
            b = t                      # qkv + Gram: read x only (q, k never leave the matrix cores; v is formed in the tail)
            b += 2 * t                 # attention tail in one pass: read x, write x1 (v = x W_v^T on the tile's halo, in LDS)
            return b + ((2 * t) if (_AVX._ffn and c in _AVX.FFN_FUSED_C) else (t + 4 * t + ((4 * t + 2 * t) if _AVX._ffn2 else (4 * t + 4 * t + 4 * t + 2 * t))))
        b = 2 * t                      # qkv + Gram: read x, write v (q, k never leave the matrix cores)
        if _AVX._tail and c in (32, 64):
            b += 3 * t                 # attention tail in one pass: read v, read x, write x1
        else:
            b += (3 * t if _AVX._posemb else 5 * t)  # pos_emb (one pass: v, x -> pe; else two depthwise passes)
            b += 3 * t                 # projection GEMM: read v, read + write the accumulator
        if _AVX._ffn and c in _AVX.FFN_FUSED_C:
            b += 2 * t                 # whole FeedForward in one kernel: read x1, write x2 (the 4C hidden tile lives in LDS)
        else:
            b += t + 4 * t             # LayerNorm -> 1x1 -> GELU: read x, write the 4C hidden tensor
            b += (4 * t + 2 * t) if (c in (32, 64) and _AVX._ffn2) else (4 * t + 4 * t + 4 * t + 2 * t)  # dw3x3 -> GELU -> 1x1 -> + x
        return b

    per_stage = 2 * msab(32) + 2 * msab(64) / 4 + msab(128) / 16
    if _AVX._convgram and _AVX._qkv16 and _AVX._tail and _AVX._tailx and _AVX._conv_lds and _AVX._upfuse:
        per_stage -= 2 * 64.0          # both full-resolution blocks take their Gram matrix from the conv that produces their input: no read of x for it
    t32, t64, t128 = 64.0, 128.0 / 4, 256.0 / 16
    convs = (2 * t32) + (3 * t32)                # embedding; mapping + x
    convs += (t32 + t64) + (t64 + t128)          # two strided 4x4 convs
    if _AVX._upfuse:
        convs += (t128 + t64 + t64) + (t64 + t32 + t32)   # transposed conv + fusion conv in one pass: read x and skip, write the level's input
    else:
        convs += (t128 + t64) + (2 * t64 + t64)      # transposed conv 128 -> 64, fusion 1x1 over [up | skip]
        convs += (t64 + t32) + (2 * t32 + t32)       # transposed conv 64 -> 32, fusion 1x1
    # uint8 frame -> conv_in output (one kernel, else float32 NCHW + conv); conv_out + x (with the spectral integration as its epilogue the cube is
    # not written: 12 bytes of catch planes per pixel instead of 64)
    head = ((3 + t32) if _AVX._conv_in else (3 + 12) + (12 + t32)) + ((2 * t32 + 12) if (_AVX._specfuse and _AVX._conv_lds and _AVX._conv_in) else 3 * t32)
    return stage * (per_stage + convs) + head
