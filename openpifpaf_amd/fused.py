"""Fused inference epilogues for the field-producing network (HIP, ``csrc/epilogue.hip``)."""
import ctypes
import os

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


def conv1x1_supported(x, weight, bias=None, residual=None, a_bias=None):
    """True if ``conv1x1_bias_act`` can run the HIP GEMM for these operands.  The kernels read raw buffers:
    EVERY operand must have the activation's dtype, bfloat16 or float32 (autocast keeps parameters in float32
    next to bfloat16 activations -- those take the PyTorch path), the activation channels_last, the residual
    channels_last of the output's shape."""
    dt = x.dtype
    if not (x.is_cuda and dt in (torch.bfloat16, torch.float32) and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last)
            and weight.dtype == dt and weight.is_cuda
            and weight.shape[1] % (64 if dt == torch.bfloat16 else 32) == 0 and weight.shape[0] % 64 == 0
            and weight.shape[1] == x.shape[1]):
        return False
    for vec, n in ((bias, weight.shape[0]), (a_bias, weight.shape[1])):
        if vec is not None and not (vec.dtype == dt and vec.is_cuda and vec.is_contiguous()
                                    and vec.numel() == n and vec.data_ptr() % 16 == 0):
            return False
    if residual is not None:
        if not (residual.dtype == dt and residual.is_cuda
                and tuple(residual.shape) == (x.shape[0], weight.shape[0], x.shape[2], x.shape[3])
                and residual.is_contiguous(memory_format=torch.channels_last) and residual.data_ptr() % 16 == 0):
            return False
    return x.data_ptr() % 16 == 0 and weight.data_ptr() % 16 == 0


def conv1x1_bias_act(x, weight2d, bias, residual=None, relu=True, a_bias=None):
    """``act(conv1x1(x, weight) + bias (+ residual))`` as ONE MFMA GEMM kernel with fused epilogue.

    :param x: ``[B, C_in, H, W]`` bfloat16 or float32, channels_last
    :param weight2d: ``[C_out, C_in]`` of the same dtype, contiguous
    :param a_bias: ``[C_in]``: ``x`` is the RAW output of the preceding convolution and
        ``relu(x + a_bias)`` -- that convolution's epilogue -- is applied while the operand is staged
    :returns: ``[B, C_out, H, W]`` of that dtype, channels_last
    """
    B, K, H, W = x.shape
    N = weight2d.shape[0]
    out = torch.empty((B, N, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    if x.dtype == torch.float32:
        _lib.check(_lib.lib().opa_gemm_bias_act_f32(
            ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(a_bias.data_ptr()) if a_bias is not None else None,
            ctypes.c_void_p(weight2d.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
            ctypes.c_void_p(residual.data_ptr()) if residual is not None else None,
            ctypes.c_void_p(out.data_ptr()), B * H * W, N, K, int(bool(relu)),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_gemm_bias_act_f32')
        return out
    if a_bias is not None:
        _lib.check(_lib.lib().opa_gemm_pro_bias_act_bf16(
            ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(a_bias.data_ptr()), ctypes.c_void_p(weight2d.data_ptr()),
            ctypes.c_void_p(bias.data_ptr()), ctypes.c_void_p(residual.data_ptr()) if residual is not None else None,
            ctypes.c_void_p(out.data_ptr()), B * H * W, N, K, int(bool(relu)),
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_gemm_pro_bias_act_bf16')
        return out
    _lib.check(_lib.lib().opa_gemm_bias_act_bf16(
        ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(weight2d.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
        ctypes.c_void_p(residual.data_ptr()) if residual is not None else None,
        ctypes.c_void_p(out.data_ptr()), B * H * W, N, K, int(bool(relu)),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_gemm_bias_act_bf16')
    return out


def split_weight(weight2d):
    """``[N, K]`` float32 -> ``[3, N, K]`` bfloat16: the three pieces of every weight (its 24-bit significand cut into 8 + 8 + 8
    bits: ``w1`` = ``w`` with the low 16 bits cleared, ``r = w - w1``, ``w2`` = ``r`` with the low 16 bits cleared, ``w3 = r - w2``;
    every step is exact and ``w1 + w2 + w3 == w`` bit for bit) -- the operand of ``opa_gemm_bias_act_f32x3`` (``csrc/gemm_f32x3.hip``)."""
    w = weight2d.detach().to(torch.float32).contiguous()

    def top(x):
        return (x.view(torch.int32) & -65536).view(torch.float32)
    w1 = top(w)
    r1 = w - w1
    w2 = top(r1)
    w3 = r1 - w2
    out = torch.stack((w1, w2, w3)).to(torch.bfloat16)          # (exact: each piece has at most 8 significant bits)
    return out.contiguous()


def conv1x1_bias_act_x3(x, w3, bias, residual=None, relu=True, a_bias=None, terms=9):
    """``conv1x1_bias_act`` for float32 through the split-operand kernel: ``w3`` = ``split_weight(weight2d)``; ``terms`` 9 or 6."""
    B, K, H, W = x.shape
    N = w3.shape[1]
    out = torch.empty((B, N, H, W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().opa_gemm_bias_act_f32x3(
        ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(a_bias.data_ptr()) if a_bias is not None else None,
        ctypes.c_void_p(w3.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
        ctypes.c_void_p(residual.data_ptr()) if residual is not None else None,
        ctypes.c_void_p(out.data_ptr()), B * H * W, N, K, int(bool(relu)), int(terms),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_gemm_bias_act_f32x3')
    return out


# float32 1x1 convolutions may take the split-operand kernel (choice 'gemm3'): float32 in and out, every product formed from
# the operands' three bfloat16 pieces on the bf16 MFMA pipe (csrc/gemm_f32x3.hip).  X3_TERMS = 6 leaves out the three smallest of
# the nine partial products (< 2^-23 of a product each); measured error against float64 BELOW the float32 MFMA kernel's and
# torch's own float32 convolution on every ResNet-50 shape (tools/gpu/gemm_x3_probe.py, tests/test_gpu_gemm_x3.py).
# OPA_GEMM3=0 (or fused.X3_TERMS = 0) takes the choice away.
X3_TERMS = {'0': 0, '6': 6, '9': 9}.get(os.environ.get('OPA_GEMM3', '6'), 6)


def _x3_supported(x, weight):
    return X3_TERMS in (6, 9) and x.dtype == torch.float32 and weight.shape[1] % 64 == 0


def _split_weight_of(conv, w2d):
    """``split_weight(w2d)``, computed once per convolution and kept on the module (inference: the weight does not change; a weight
    replaced or moved since is split again)."""
    key = (w2d.data_ptr(), w2d._version, str(w2d.device))
    cached = getattr(conv, '_opa_w3', None)
    if cached is None or cached[0] != key:
        cached = (key, split_weight(w2d))
        conv._opa_w3 = cached
    return cached[1]


FORCE_PICK = os.environ.get('OPA_GEMM3_PICK') or None        # 'x3' | 'conv': every pick() takes that side (tests, A/B)


def pick(kind, m, k, n, flag_a, flag_b, run_x3, run_other):
    """One of two ways to compute the same tensor -- a split-operand kernel (``run_x3``) or what the trunk did before
    (``run_other``: MIOpen's convolution + the fused passes) -- chosen ONCE per shape like ``conv_bias_act`` chooses its GEMM: from
    the shipped table (``conv1x1_pinned.json``, key dtype ``'torch.float32/<kind>'``), else by timing both on the first call; while
    a stream is being captured or in a job of several ranks, where timing is not an option, by size (the split-operand kernels win
    from ~16 000 output pixels: a batch of one 641-px image keeps MIOpen in layers 3-4).  Returns the chosen function's result."""
    if FORCE_PICK in ('x3', 'conv'):
        return run_x3() if FORCE_PICK == 'x3' else run_other()
    key = ('torch.float32/' + kind, int(m), int(k), int(n), bool(flag_a), bool(flag_b))
    choice = _CHOICE.get(key)
    if choice is None:
        if torch.cuda.is_current_stream_capturing() or _in_multi_rank_job():
            choice = 'x3' if m >= 16384 else 'conv'
        else:
            choice = 'x3' if _time_ms(run_x3) <= _time_ms(run_other) else 'conv'
        _CHOICE[key] = choice
    return run_x3() if choice == 'x3' else run_other()


# ... and the block's LAST 1x1 convolution together with its downsampling convolution as one product (OPA_GEMM3_PAIR=0: off)
X3_PAIR = os.environ.get('OPA_GEMM3_PAIR', '1') != '0'


def pair_supported(conv, dconv, h, x, bias, a_bias=None):
    """Can ``conv(h) + dconv(x)`` run as ONE split-operand product (``conv1x1_pair_bias_act_x3``)?  float32, channels_last,
    both 1x1 without bias / groups / padding, ``conv`` of stride 1, ``dconv`` of any stride."""
    k1, k2 = conv.in_channels, dconv.in_channels
    ok = (X3_PAIR and X3_TERMS in (6, 9) and h.is_cuda and h.dtype == torch.float32 and x.dtype == torch.float32
          and conv.kernel_size == (1, 1) and dconv.kernel_size == (1, 1) and conv.stride == (1, 1) and dconv.stride[0] == dconv.stride[1]
          and conv.groups == 1 and dconv.groups == 1 and conv.padding == (0, 0) and dconv.padding == (0, 0)
          and conv.bias is None and dconv.bias is None and conv.out_channels == dconv.out_channels
          and k1 % 32 == 0 and (k1 + k2) % 64 == 0 and k2 % 4 == 0 and conv.out_channels % 64 == 0
          and h.dim() == 4 and x.dim() == 4 and h.is_contiguous(memory_format=torch.channels_last)
          and x.is_contiguous(memory_format=torch.channels_last) and h.shape[1] == k1 and x.shape[1] == k2
          and h.shape[0] == x.shape[0] and h.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
          and bias.dtype == torch.float32 and bias.is_contiguous() and bias.data_ptr() % 16 == 0
          and x.shape[0] * x.shape[2] * x.shape[3] < 2 ** 31)
    if not ok:
        return False
    s = dconv.stride[0]
    if (h.shape[2], h.shape[3]) != ((x.shape[2] - 1) // s + 1, (x.shape[3] - 1) // s + 1):
        return False
    return a_bias is None or (a_bias.dtype == torch.float32 and a_bias.numel() == k1 and a_bias.is_contiguous())


def conv1x1_pair_bias_act_x3(conv, dconv, h, x, bias, relu=True, a_bias=None):
    """``act(conv(h) + dconv(x) + bias)`` -- the last 1x1 convolution of a ResNet block and the block's downsampling convolution
    (reference ``network/basenetworks.py:71-150``: torchvision's Bottleneck) -- as ONE product ``[h | x at stride] * [W ; Wd]^T`` of
    the split-operand kernel (``opa_gemm2_bias_act_f32x3``): the identity tensor is neither written nor read back.  With
    ``a_bias``, ``h`` is the raw output of the preceding convolution and ``relu(h + a_bias)`` is applied while it is staged (``x``
    gets zeros: it is non-negative).  ``pair_supported`` says whether this can run."""
    k1, k2, n = conv.in_channels, dconv.in_channels, conv.out_channels
    w1, w2 = conv.weight.reshape(n, k1), dconv.weight.reshape(n, k2)
    key = (w1.data_ptr(), w1._version, w2.data_ptr(), w2._version, str(w1.device), None if a_bias is None else (a_bias.data_ptr(), a_bias._version))
    cached = getattr(conv, '_opa_w3_pair', None)
    if cached is None or cached[0] != key:
        ab = None if a_bias is None else torch.cat((a_bias.detach().float(), torch.zeros(k2, device=a_bias.device)))
        cached = (key, split_weight(torch.cat((w1.detach(), w2.detach()), dim=1)), ab)
        conv._opa_w3_pair = cached
    _, w3, ab = cached
    B, _, H, W = x.shape
    out = torch.empty((B, n, h.shape[2], h.shape[3]), dtype=torch.float32, device=h.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().opa_gemm2_bias_act_f32x3(
        ctypes.c_void_p(h.data_ptr()), k1, ctypes.c_void_p(x.data_ptr()), k2, B, H, W, dconv.stride[0],
        ctypes.c_void_p(ab.data_ptr()) if ab is not None else None, ctypes.c_void_p(w3.data_ptr()), ctypes.c_void_p(bias.data_ptr()),
        ctypes.c_void_p(out.data_ptr()), n, int(bool(relu)), int(X3_TERMS),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_gemm2_bias_act_f32x3')
    return out


# ... and the STRIDED 3x3 convolutions (the head of ResNet layers 2-4) as an implicit GEMM of the same kernel (OPA_GEMM3_3X3=0: off)
X3_CONV3 = os.environ.get('OPA_GEMM3_3X3', '1') != '0'


def split_weight_3x3(weight):
    """``[N, C, 3, 3]`` float32 -> ``split_weight`` of ``[N, (ky, kx, c)]``: the operand of ``opa_conv3x3_f32x3``."""
    n, c = weight.shape[0], weight.shape[1]
    return split_weight(weight.detach().permute(0, 2, 3, 1).reshape(n, 9 * c))


def conv3x3_x3_supported(conv, x, bias):
    return (X3_CONV3 and X3_TERMS in (6, 9) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous(memory_format=torch.channels_last) and conv.kernel_size == (3, 3) and conv.padding == (1, 1)
            and conv.stride[0] == conv.stride[1] and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0 and x.shape[1] == conv.in_channels
            and x.data_ptr() % 16 == 0 and bias.dtype == torch.float32 and bias.is_contiguous() and bias.data_ptr() % 16 == 0
            and (x.shape[0] * x.shape[2] * x.shape[3] + x.shape[3] + 1) * x.shape[1] * 4 < 2 ** 31)


def conv3x3_bias_act_x3(conv, x, bias, relu=True):
    """``act(conv(x) + bias)`` for a 3x3 convolution with padding 1 and any stride (reference ``network/basenetworks.py:71-150``: the
    strided convolution of a ResNet block) as an implicit GEMM of the split-operand kernel -- float32 in and out."""
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device))
    cached = getattr(conv, '_opa_w3_3x3', None)
    if cached is None or cached[0] != key:
        cached = (key, split_weight_3x3(w))
        conv._opa_w3_3x3 = cached
    B, C, H, W = x.shape
    s = conv.stride[0]
    out = torch.empty((B, conv.out_channels, (H - 1) // s + 1, (W - 1) // s + 1), dtype=torch.float32, device=x.device,
                      memory_format=torch.channels_last)
    _lib.check(_lib.lib().opa_conv3x3_f32x3(
        ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(cached[1].data_ptr()), ctypes.c_void_p(bias.data_ptr()),
        ctypes.c_void_p(out.data_ptr()), B, H, W, C, conv.out_channels, s, int(bool(relu)), int(X3_TERMS),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_conv3x3_f32x3')
    return out


# ... and the heads' 1x1 convolutions, whose output channels are no multiple of the kernel's 64-wide tile (OPA_GEMM3_HEAD=0: off)
X3_HEAD = os.environ.get('OPA_GEMM3_HEAD', '1') != '0'


def head_conv_x3_supported(conv, x):
    return (X3_HEAD and X3_TERMS in (6, 9) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and not x.requires_grad
            and x.is_contiguous(memory_format=torch.channels_last) and conv.kernel_size == (1, 1) and conv.stride == (1, 1)
            and conv.padding == (0, 0) and conv.groups == 1 and conv.in_channels % 64 == 0 and x.shape[1] == conv.in_channels
            and conv.weight.dtype == torch.float32 and x.data_ptr() % 16 == 0)


def head_conv_x3(conv, x):
    """``conv(x)`` for a head's biased 1x1 convolution (reference ``network/heads.py:272-378``: ``CompositeField4.conv``) through the
    split-operand GEMM: the output channels are padded to the next multiple of 64 with zero weights, the product is written with
    that pitch and the real channels are copied out (0.2 GB for both COCO heads at batch 32)."""
    w = conv.weight
    n, k = w.shape[0], w.shape[1]
    npad = (n + 63) // 64 * 64
    key = (w.data_ptr(), w._version, str(w.device), None if conv.bias is None else conv.bias._version)
    cached = getattr(conv, '_opa_w3_head', None)
    if cached is None or cached[0] != key:
        wp = torch.zeros((npad, k), dtype=torch.float32, device=w.device)
        wp[:n] = w.detach().reshape(n, k)
        bp = torch.zeros(npad, dtype=torch.float32, device=w.device)
        if conv.bias is not None:
            bp[:n] = conv.bias.detach()
        cached = (key, split_weight(wp), bp)
        conv._opa_w3_head = cached
    out = conv1x1_bias_act_x3(x, cached[1], cached[2], None, False, None, X3_TERMS)
    return out[:, :n].contiguous(memory_format=torch.channels_last) if npad != n else out


# ... and the 7x7 stride-2 stem (OPA_GEMM3_STEM=0: off)
X3_STEM = os.environ.get('OPA_GEMM3_STEM', '1') != '0'


def stem_x3_supported(conv, x, bias):
    return (X3_STEM and X3_TERMS in (6, 9) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (7, 7)
            and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.in_channels == 3 and x.shape[1] == 3 and conv.out_channels % 64 == 0
            and bias.dtype == torch.float32 and bias.is_contiguous() and bias.data_ptr() % 16 == 0
            and x.shape[0] * (x.shape[2] + 7) * (x.shape[3] + 7) * 16 < 2 ** 31)


def stem7x7_bias_act_x3(conv, x, bias, relu=True):
    """``act(conv(x) + bias)`` for the 7x7 stride-2 padding-3 stem of a ResNet on RGB input (reference ``network/basenetworks.py:71-150``)
    as an implicit GEMM of the split-operand kernel: the image is copied once into a zero-padded 4-channel NHWC tensor (3 pixels
    before, 4 behind; ~1 % of the step), a window ROW -- 8 pixels x 4 channels = 32 contiguous floats -- is one K-step, the
    eighth row and column and the fourth channel meet zero weights.  K = 8 x 32 = 256."""
    w = conv.weight
    key = (w.data_ptr(), w._version, str(w.device))
    cached = getattr(conv, '_opa_w3_stem', None)
    if cached is None or cached[0] != key:
        n = w.shape[0]
        wp = torch.zeros((n, 8, 8, 4), dtype=torch.float32, device=w.device)
        wp[:, :7, :7, :3] = w.detach().permute(0, 2, 3, 1)
        cached = (key, split_weight(wp.reshape(n, 256)))
        conv._opa_w3_stem = cached
    B, _, H, W = x.shape
    xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, 1, 3, 4, 3, 4)).contiguous()        # [B, H + 7, W + 7, 4]
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((B, conv.out_channels, ho, wo), dtype=torch.float32, device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().opa_conv_rows_f32x3(
        ctypes.c_void_p(xp.data_ptr()), ctypes.c_void_p(cached[1].data_ptr()), ctypes.c_void_p(bias.data_ptr()),
        ctypes.c_void_p(out.data_ptr()), B, H + 7, W + 7, 4, ho, wo, 2, 8, 32, conv.out_channels, int(bool(relu)), int(X3_TERMS),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_conv_rows_f32x3')
    return out


# (dtype, M, K, N, has_residual, has_a_bias) -> 'gemm' | 'gemm3' | 'pass+gemm' | 'conv'.  The three paths round
# differently, so the choice is part of the result: it is made once per shape (the key holds no device index: a
# table exported on rank 0 must match the lookups of every other rank), never by timing while a stream is being
# captured (timing synchronises; the capture-time default is remembered, so a captured graph and a later eager
# run use the same kernel), can be pinned with OPA_CONV1X1=gemm|conv, and can be exported / imported
# (choices / set_choices; distributed.broadcast_conv_choices) so that every rank of a job runs the same kernels.
_CHOICE = {}
PINNED_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'conv1x1_pinned.json')


def load_pinned(path=None):
    """The table the package ships (``conv1x1_pinned.json``: measured on an MI355X by ``tools/gpu/dump_conv_choices.py`` for the
    shapes of the BASELINE configurations, float32 and bfloat16): the same choice on every rank of a multi-GPU job WITHOUT a
    collective (round 5 broadcast rank 0's wall-clock choices, ``distributed.broadcast_conv_choices``).  Loaded when this module
    is imported; entries made later (``set_choices``, timing) go on top.  -> number of entries adopted."""
    import json
    try:
        with open(path or PINNED_FILE) as f:
            table = json.load(f)['table']
    except (OSError, ValueError, KeyError):
        return 0
    for dtype, m, k, n, res, a_bias, choice in table:
        _CHOICE.setdefault((dtype, int(m), int(k), int(n), bool(res), bool(a_bias)), choice)
    return len(table)


def _in_multi_rank_job():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:              # noqa: BLE001
        return False


def choices():
    return dict(_CHOICE)


def set_choices(table, *, replace=False):
    if replace:
        _CHOICE.clear()
    _CHOICE.update(table)


def _time_ms(fn, reps=3):
    """Best of two rounds of ``reps`` calls (one round alone flips close calls between runs: third session, batch-1 shapes)."""
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(2):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            fn()
        end.record()
        end.synchronize()
        ms = start.elapsed_time(end) / reps
        best = ms if best is None or ms < best else best
    return best


def conv_bias_act(conv, x, bias, residual=None, relu=True, a_bias=None):
    """``act(conv(x) + bias (+ residual))`` for a bias-free ``conv`` module: 1x1 stride-1 convolutions go
    to the fused MFMA GEMM when it is faster than MIOpen's convolution + the fused epilogue pass
    (decided once per shape by timing both on the first call); everything else is conv + ``bias_act_``.

    With ``a_bias``, ``x`` is the raw output of the preceding convolution whose epilogue
    ``relu(x + a_bias[c])`` has not been applied yet: the GEMM applies it to its operand on the fly, the
    fallback applies it in place first."""
    w = conv.weight
    if (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and conv.padding == (0, 0)
            and conv1x1_supported(x, w, bias, residual, a_bias)):
        M = x.shape[0] * x.shape[2] * x.shape[3]
        key = (str(x.dtype), M, w.shape[1], w.shape[0], residual is not None, a_bias is not None)
        w2d = w.reshape(w.shape[0], w.shape[1])
        if not w2d.is_contiguous():
            w2d = w2d.contiguous()
        choice = _CHOICE.get(key)
        if choice is None:
            forced = os.environ.get('OPA_CONV1X1', 'auto')
            if forced in ('gemm', 'conv'):
                choice = _CHOICE[key] = forced
            elif torch.cuda.is_current_stream_capturing():
                choice = _CHOICE[key] = 'gemm'   # no timing inside a capture; remembered, so eager runs agree with the graph
            elif _in_multi_rank_job():
                # a shape the shipped table does not know, in a job of several ranks: every rank takes the SAME default instead of
                # timing (wall clocks differ from rank to rank, and the three paths round differently) -- no collective needed
                choice = _CHOICE[key] = 'gemm'
        if choice is None:
            times = {'gemm': _time_ms(lambda: conv1x1_bias_act(x, w2d, bias, residual, relu, a_bias))}
            if _x3_supported(x, w):
                w3 = _split_weight_of(conv, w2d)
                times['gemm3'] = _time_ms(lambda: conv1x1_bias_act_x3(x, w3, bias, residual, relu, a_bias, X3_TERMS))
            if a_bias is None:
                times['conv'] = _time_ms(lambda: bias_act_(conv(x), bias, residual, relu))
            else:       # timing only: the separate epilogue pass runs on a scratch copy
                scratch = x.clone()
                times['pass+gemm'] = _time_ms(
                    lambda: conv1x1_bias_act(bias_act_(scratch, a_bias), w2d, bias, residual, relu))
                times['conv'] = _time_ms(lambda: bias_act_(conv(bias_act_(scratch, a_bias)), bias, residual, relu))
            choice = _CHOICE[key] = min(times, key=times.get)
        if choice == 'gemm3' and not _x3_supported(x, w):
            choice = 'gemm'                  # (switched off after the table was made: the float32 MFMA kernel)
        if choice == 'gemm3':
            return conv1x1_bias_act_x3(x, _split_weight_of(conv, w2d), bias, residual, relu, a_bias, X3_TERMS)
        if choice == 'gemm':
            return conv1x1_bias_act(x, w2d, bias, residual, relu, a_bias)
        if choice == 'pass+gemm':     # the prologue's VALU work is repeated per N-tile: cheaper as its own pass here
            return conv1x1_bias_act(bias_act_(x, a_bias), w2d, bias, residual, relu)
    if a_bias is not None:
        x = bias_act_(x, a_bias)
    return bias_act_(conv(x), bias, residual, relu)


def head_epilogue_supported(x, meta, training=False):
    """True if :func:`head_epilogue` can run for this convolution output and head meta."""
    us = meta.upsample_stride
    return (not training and x.is_cuda and x.dim() == 4 and x.dtype in _DTYPES and _lib.available()
            and not (torch.is_grad_enabled() and x.requires_grad)           # the kernel has no backward
            and x.is_contiguous(memory_format=torch.channels_last) and us in (1, 2)
            and x.shape[1] % (us ** 2) == 0 and x.data_ptr() % 16 == 0
            and 16 * us * us * (x.shape[3] + 1) * 4 <= 64 * 1024)           # the kernel's LDS row buffer (head.hip)


def head_epilogue(x, meta):
    """Everything ``CompositeField4`` does after its 1x1 convolution, in ONE kernel (reference
    ``network/heads.py:330-378``): PixelShuffle -> crop -> ``[B, F, C, H, W]`` float32 -> sigmoid / index offsets /
    softplus.  ``x``: the convolution output ``[B, F*C*us^2, hc, wc]``, channels_last."""
    B, ctot, hc, wc = x.shape
    us = meta.upsample_stride
    n_comp = 1 + meta.n_confidences + meta.n_vectors * 2 + meta.n_scales
    n_fields = ctot // (n_comp * us * us)
    low_cut = (us - 1) // 2
    high_cut = us - 1 - low_cut
    H, W = hc * us - low_cut - high_cut, wc * us - low_cut - high_cut
    out = torch.empty((B, n_fields, n_comp, H, W), dtype=torch.float32, device=x.device)
    mask = sum(1 << i for i, on in enumerate(meta.vector_offsets) if on)
    _lib.check(_lib.lib().opa_head_epilogue(
        ctypes.c_void_p(x.data_ptr()), _DTYPES[x.dtype], B, hc, wc, n_fields, n_comp, us, meta.n_confidences,
        meta.n_vectors, mask, meta.n_scales, ctypes.c_void_p(out.data_ptr()),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_head_epilogue')
    return out


def _pixel_stride(x):
    """Elements between neighbouring pixels of a channels-innermost (NHWC in memory) 4-d tensor or channel slice of
    one; None if ``x`` is laid out differently."""
    if x.dim() != 4 or x.stride(1) != 1:
        return None
    B, C, H, W = x.shape
    ps = x.stride(3) if W > 1 else (x.stride(2) if H > 1 else C)
    if ps < C or (H > 1 and x.stride(2) != W * ps) or (B > 1 and x.stride(0) != H * W * ps):
        return None
    return ps


def dwconv_supported(x, kernel_size, stride):
    if not (x.is_cuda and x.dim() == 4):
        return False
    rows_out = (x.shape[2] + 2 * (kernel_size // 2) - kernel_size) // stride + 1
    return (x.dtype in (torch.float32, torch.bfloat16) and kernel_size in (3, 5) and stride in (1, 2)
            and x.shape[0] * rows_out <= 65535                              # grid.y of the stencil kernel (dwconv.hip)
            and _pixel_stride(x) is not None and _lib.available())


def dwconv_bias_act(x, w_taps, bias, kernel_size, stride, relu=False):
    """Depthwise ``kernel_size`` x ``kernel_size`` convolution (padding k//2) + bias (+ ReLU) of a channels-last
    activation or channel slice, one HIP stencil kernel.  ``w_taps``: ``[k*k, C]`` (tap-major) in ``x``'s dtype."""
    B, C, H, W = x.shape
    pad = kernel_size // 2
    Ho, Wo = (H + 2 * pad - kernel_size) // stride + 1, (W + 2 * pad - kernel_size) // stride + 1
    out = torch.empty((B, C, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().opa_dwconv_bias_act(
        ctypes.c_void_p(x.data_ptr()), _pixel_stride(x), ctypes.c_void_p(w_taps.data_ptr()),
        ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, ctypes.c_void_p(out.data_ptr()), C,
        B, H, W, C, kernel_size, stride, _DTYPES[x.dtype], int(bool(relu)),
        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'opa_dwconv_bias_act')
    return out


def channel_interleave(a, b):
    """``channel_shuffle(torch.cat((a, b), 1), groups=2)`` in one pass: out[:, 2i] = a[:, i], out[:, 2i+1] = b[:, i]."""
    pa, pb = _pixel_stride(a), _pixel_stride(b)
    if not (a.is_cuda and a.dtype in _DTYPES and a.dtype == b.dtype and a.shape == b.shape and pa and pb
            and _lib.available()):
        x = torch.cat((a, b), dim=1)
        n, c, h, w = x.shape
        return x.view(n, 2, c // 2, h, w).transpose(1, 2).reshape(n, c, h, w)
    B, half, H, W = a.shape
    out = torch.empty((B, 2 * half, H, W), dtype=a.dtype, device=a.device, memory_format=torch.channels_last)
    _lib.check(_lib.lib().opa_channel_interleave(
        ctypes.c_void_p(a.data_ptr()), pa, ctypes.c_void_p(b.data_ptr()), pb, ctypes.c_void_p(out.data_ptr()),
        B * H * W, half, _DTYPES[a.dtype], ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
        'opa_channel_interleave')
    return out


load_pinned()
