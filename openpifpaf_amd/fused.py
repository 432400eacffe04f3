"""Fused inference epilogues for the field-producing network (HIP, ``csrc/epilogue.hip``)."""
import ctypes

import torch

from . import _lib

_DTYPES = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _nhwc_rows(x):
    """(rows, channels) if ``x`` is physically [rows, channels]-contiguous, else None."""
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return x.shape[0] * x.shape[2] * x.shape[3], x.shape[1]
    if x.dim() == 2 and x.is_contiguous():
        return x.shape[0], x.shape[1]
    return None


def bias_act_(x, bias, residual=None, relu=True):
    """In place ``x = act(x + bias[c] (+ residual))`` for a channels_last activation.

    One HIP kernel on the GPU; the equivalent PyTorch ops elsewhere (CPU tests, odd layouts)."""
    rc = _nhwc_rows(x) if x.is_cuda else None
    per_vec = 4 if x.dtype == torch.float32 else 8
    ok = (rc is not None and x.dtype in _DTYPES and rc[1] % per_vec == 0 and bias.dtype == x.dtype
          and bias.is_contiguous() and x.data_ptr() % 16 == 0 and bias.data_ptr() % 16 == 0
          and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                    and _nhwc_rows(residual) is not None and residual.data_ptr() % 16 == 0)))
    if not ok:
        x.add_(bias.view(1, -1, 1, 1) if x.dim() == 4 else bias)
        if residual is not None:
            x.add_(residual)
        return torch.relu_(x) if relu else x
    _lib.check(_lib.lib().opa_bias_act(
        ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
        ctypes.c_void_p(residual.data_ptr()) if residual is not None else None,
        rc[0], rc[1], _DTYPES[x.dtype], int(bool(relu)),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_bias_act')
    return x
