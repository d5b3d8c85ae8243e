"""CPU emulation of the MX-FP8 contraction of csrc/conv_gemm_f8.hip (TEST INFRASTRUCTURE: only tests/ and the experiment's bench leg
import it).

OCP Microscaling (MX) v1.0, FP8 element type e4m3 (max 448, emax 8), scale type e8m0, block size 32 along the contraction index:
    shared exponent  e = floor(log2(max |x| of the block)) - 8        (clamped to >= -127; an all-zero block takes -127)
    element          q = RNE_e4m3(clamp(x * 2^-e, -448, +448))
    value used by the matrix core = q * 2^e
The products q_a q_w are exact in fp32 and the block scales are powers of two, so a GEMM on the matrix cores equals the fp32
(here: fp64) product of the DEQUANTISED operands up to the accumulation order -- that is what the kernel test compares with.
There is no reference code to pin this to (the reference computes in fp32, /root/reference/code/models.py:1331-1354): PARITY
UNPINNED by construction, the experiment reports its deviation from the fp32-exact path instead of claiming parity."""
import torch

BLOCK = 32


def mx_dequantized(x, block=BLOCK):
    """x [..., K] fp32 -> the values an MX-FP8 matrix core multiplies (fp32), blocks of `block` along the last axis."""
    K = x.shape[-1]
    if K % block:
        raise ValueError(f"K={K} is not a multiple of the MX block size {block}")
    xb = x.detach().to(torch.float32).reshape(-1, K // block, block)
    amax = xb.abs().amax(-1, keepdim=True)
    e = ((amax.contiguous().view(torch.int32) >> 23) & 0xFF) - 127 - 8
    e = e.clamp_min(-127)
    q = torch.ldexp(xb, -e).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)
    return torch.ldexp(q, e).reshape(x.shape)


def mx_linear(a, w, block=BLOCK):
    """[M, K] x [N, K]^T on emulated MX-FP8 operands, accumulated in fp64."""
    return mx_dequantized(a, block).double() @ mx_dequantized(w, block).double().T


def mx_conv2d_nhwc(x, w, stride=1, pad=1, block=BLOCK):
    """x [B, H, W, C] (C % block == 0), w [N, KH, KW, C]: the implicit GEMM of conv_gemm_f8.hip -- K = (tap, channel), zero
    padding enters the blocks as zeros -- on emulated MX-FP8 operands, fp64 accumulate.  Returns [B, OH, OW, N]."""
    B, H, W, C = x.shape
    N, KH, KW, _ = w.shape
    cols = torch.nn.functional.unfold(x.permute(0, 3, 1, 2).double(), (KH, KW), padding=pad, stride=stride)    # [B, C*KH*KW, L]
    L = cols.shape[-1]
    cols = cols.reshape(B, C, KH * KW, L).permute(0, 3, 2, 1).reshape(B * L, KH * KW * C).float()                # (tap, channel)
    out = mx_linear(cols, w.reshape(N, KH * KW * C), block)
    OH = (H + 2 * pad - KH) // stride + 1
    return out.reshape(B, OH, L // OH, N)
