"""3x3 stride-1 convolutions of the float32 trunk as Winograd F(2x2, 3x3) (HIP, ``csrc/winograd.hip``).

The reference runs its ResNet bottlenecks through ``torch.nn.Conv2d`` (``network/basenetworks.py:71-150``); at 641 px the
thirteen stride-1 3x3 convolutions of ResNet-50 are half of the float32 step.  ``conv3x3`` computes the same convolution
with 16 instead of 36 multiplications per 2x2 outputs, in float32 throughout; the filter side ``G g G^T`` is computed once
per weight, in float64, and stored in the order the kernel's lanes read it.
"""
import ctypes
import os

import torch

from . import _lib

# variant -> (K-chunk, 32-wide channel blocks per workgroup): must match launch_winograd_f23 (csrc/winograd.hip)
VARIANTS = {0: (16, 2), 1: (8, 1), 2: (16, 2), 3: (16, 2)}     # 2: variant 0's tile and filter layout, eight waves in two shifts
DEFAULT_VARIANT = 2         # eight waves in two shifts (csrc/winograd.hip): 5-8 % faster than variant 0 on every ResNet-50 shape
MIN_WORKGROUPS = 256        # below one workgroup per compute unit the launch does not fill the chip: MIOpen's convolution
_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))


_mode = os.environ.get('OPA_CONV3X3', 'auto')      # read ONCE, when the module is imported; set_mode() afterwards
if _mode not in ('auto', 'conv', 'winograd'):
    raise ValueError('OPA_CONV3X3 must be auto, conv or winograd, not %r' % _mode)


def set_mode(mode):
    """'auto' (the shape rule of :func:`takes`), 'conv' (always MIOpen's convolution) or 'winograd' (the kernel wherever the
    operands qualify, however small the launch) -> the previous mode.  The two paths round differently: a job's ranks must agree."""
    global _mode
    if mode not in ('auto', 'conv', 'winograd'):
        raise ValueError(mode)
    previous, _mode = _mode, mode
    return previous


def get_mode():
    return _mode


def workgroups(x_shape, c_out, variant=DEFAULT_VARIANT):
    B, _, H, W = x_shape
    return ((B * ((H + 1) // 2) * ((W + 1) // 2) + 63) // 64) * (c_out // (32 * VARIANTS[variant][1]))


def takes(conv, x, u, variant=DEFAULT_VARIANT):
    """True if the Winograd kernel runs ``conv(x)`` for this bias-free 3x3 ``torch.nn.Conv2d``: ``u`` (its transformed filter)
    is there, the operands qualify and the launch fills the chip -- a rule on shapes, not a timing, so that every rank of a job
    and every run take the same path (the two round differently); :func:`set_mode` (default: ``OPA_CONV3X3`` at import) forces one."""
    forced = _mode
    return (u is not None and forced != 'conv' and u.dtype == torch.float32 and conv.bias is None
            and supported(x, conv.weight, variant, conv.stride, conv.padding, conv.groups, conv.dilation)
            and (forced == 'winograd' or workgroups(x.shape, conv.out_channels, variant) >= MIN_WORKGROUPS)
            and not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)))


def conv_or_fallback(conv, x, u, variant=DEFAULT_VARIANT):
    """``conv(x)``: through the kernel where :func:`takes` says so, else the module itself (MIOpen)."""
    if takes(conv, x, u, variant):
        return conv3x3(x, u, conv.out_channels, variant=variant)
    return conv(x)


def supported(x, weight, variant=0, stride=(1, 1), padding=(1, 1), groups=1, dilation=(1, 1)):
    kc, nb = VARIANTS[variant]
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 4
            and tuple(weight.shape[2:]) == (3, 3) and tuple(stride) == (1, 1) and tuple(padding) == (1, 1)
            and groups == 1 and tuple(dilation) == (1, 1) and weight.shape[1] == x.shape[1]
            and x.is_contiguous(memory_format=torch.channels_last)
            and weight.shape[1] % kc == 0 and weight.shape[0] % (32 * nb) == 0
            and x.numel() < 2 ** 32 and x.data_ptr() % 16 == 0 and _lib.available())


def transform_filter(weight, variant=0):
    """``[C_out, C_in, 3, 3]`` -> the kernel's operand: U = G g G^T per (c_out, c_in) in float64, rounded to float32 once,
    laid out ``[channel block][K-chunk][position 16][j][kq][lane 64][4]`` -- lane ``l`` of a wave multiplies
    ``U[k = 2 * (4 * kq + e) + l // 32][c = 32 * j + l % 32]`` in its e-th MFMA of group kq, so a lane's four values are
    one 16-byte load and a wave's load is 1 KB contiguous."""
    kc, nb = VARIANTS[variant]
    cout, cin = weight.shape[:2]
    g = torch.tensor(_G, dtype=torch.float64, device=weight.device)
    u = torch.einsum('ar,oirs,bs->aboi', g, weight.detach().double(), g)        # [4, 4, cout, cin]
    u = u.permute(0, 1, 3, 2).reshape(16, cin, cout)                            # [position, k, c]
    u = u.reshape(16, cin // kc, kc // 8, 4, 2, cout // (32 * nb), nb, 32)      # [pos, chunk, kq, e, khalf, block, j, c]
    u = u.permute(5, 1, 0, 6, 2, 4, 7, 3)                                       # [block, chunk, pos, j, kq, khalf, c, e]
    return u.contiguous().float().reshape(-1)


# direct-convolution flops of the launches that went through the kernel since the last reset: a measurement (bench.py) that
# quotes a convolutional network's FLOP/s has to say how many of its multiplications F(2x2, 3x3) did not execute
_direct_flops = 0.0


def reset_flop_counter():
    global _direct_flops
    _direct_flops = 0.0


def direct_flops():
    """Flops (2 x MACs of the DIRECT form) of the convolutions run by the kernel since ``reset_flop_counter``; the kernel's
    MFMAs execute 1 / 2.25 of them."""
    return _direct_flops


def conv3x3(x, u, c_out, bias=None, relu=False, variant=0, order=0, out=None):
    """``conv2d(x, weight, padding=1)`` (+ bias, ReLU) for the ``u = transform_filter(weight, variant)`` of a 3x3 weight.
    ``x``: ``[B, C_in, H, W]`` float32 channels_last on the GPU -> ``[B, c_out, H, W]`` channels_last."""
    global _direct_flops
    B, cin, H, W = x.shape
    _direct_flops += 18.0 * B * H * W * cin * c_out
    if out is None:
        out = torch.empty((B, c_out, H, W), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().opa_conv3x3_winograd_f32(
        ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(u.data_ptr()),
        ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, ctypes.c_void_p(out.data_ptr()),
        B, H, W, cin, c_out, int(bool(relu)), variant, order,
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_conv3x3_winograd_f32')
    return out


def reference_f23(x, weight):
    """The same algorithm in plain PyTorch (float64 on the CPU): the model the CPU tests check the transform matrices and
    the filter layout against, tile by tile like the kernel.  ``x`` ``[B, C, H, W]`` -> ``[B, C_out, H, W]``."""
    x = x.double()
    B, C, H, W = x.shape
    th, tw = (H + 1) // 2, (W + 1) // 2
    xp = torch.zeros((B, C, 2 * th + 2, 2 * tw + 2), dtype=torch.float64)
    xp[:, :, 1:H + 1, 1:W + 1] = x
    bt = torch.tensor(((1, 0, -1, 0), (0, 1, 1, 0), (0, -1, 1, 0), (0, 1, 0, -1)), dtype=torch.float64)
    at = torch.tensor(((1, 1, 1, 0), (0, 1, -1, -1)), dtype=torch.float64)
    g = torch.tensor(_G, dtype=torch.float64)
    u = torch.einsum('ar,oirs,bs->aboi', g, weight.double(), g)
    tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [B, C, th, tw, 4, 4]
    v = torch.einsum('ai,bcyxij,dj->adbcyx', bt, tiles, bt)                       # [4, 4, B, C, th, tw]
    m = torch.einsum('adoc,adbcyx->adboyx', u, v)
    yt = torch.einsum('pa,adboyx,qd->boypxq', at, m, at)                          # [B, O, th, 2, tw, 2]
    return yt.reshape(B, -1, 2 * th, 2 * tw)[:, :, :H, :W]
