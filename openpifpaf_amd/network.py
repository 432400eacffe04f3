"""Field producers on PyTorch-ROCm: backbones + CompositeField4 heads.

This is the "plumbing" in front of the decode path: plain ``torch.nn`` modules
(MIOpen / hipBLASLt convolutions, no custom kernels) restated without
torchvision, with random initialisation (no checkpoints offline).  Architecture
follows the reference so that the field tensors have the reference's shapes:

* ``Resnet``: torchvision ResNet topology with the input max-pool removed, so
  the overall stride is 16 (reference ``network/basenetworks.py:71-150``,
  factories ``network/factory.py:51-57``).
* ``ShuffleNetV2K``: reference ``network/basenetworks.py:186-355`` (k16: stages
  [4,8,4], channels [24,348,696,1392,1392]; k30: [8,16,6], [32,512,1024,2048,2048]).
* ``CompositeField4``: 1x1 conv -> PixelShuffle(2) -> crop last row/col -> view
  ``[B, F, C, H, W]`` -> sigmoid / index-add / softplus (reference
  ``network/heads.py:272-378``); 641 px -> 41 -> 82 -> 81.
* ``Shell``: reference ``network/nets.py:11-48``.
"""
import math

import torch
from torch import nn

from . import fused, headmeta, winograd


def _take_biases(block, conv_names):
    """Move the (BN-folded) conv biases of a residual block into buffers ``fb1..`` applied by the
    fused epilogue; the downsample conv's bias is merged into the last one (both are added
    before the block's final ReLU)."""
    for i, name in enumerate(conv_names, 1):
        conv = getattr(block, name)
        bias = conv.bias.detach().clone() if conv.bias is not None else torch.zeros(conv.out_channels)
        if i == len(conv_names) and block.downsample is not None:
            dconv = block.downsample[0]
            if dconv.bias is not None:
                bias = bias + dconv.bias.detach().to(bias.device)
                dconv.bias = None
        conv.bias = None
        block.register_buffer('fb%d' % i, bias)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.fused = False

    def enable_fused_(self):
        _take_biases(self, ('conv1', 'conv2', 'conv3'))
        if self.conv2.stride == (1, 1):     # the stride-1 3x3: G g G^T once, in the Winograd kernel's operand order
            self.register_buffer('wino_u', winograd.transform_filter(self.conv2.weight, winograd.DEFAULT_VARIANT),
                                 persistent=False)
        self.fused = True

    def forward(self, x):
        if self.fused:      # conv (no bias) -> ONE fused bias(+residual)+ReLU pass each
            out = fused.conv_bias_act(self.conv1, x, self.fb1)            # 1x1: fused MFMA GEMM
            # 3x3: float32 stride 1 -> Winograd F(2x2, 3x3) in one HIP kernel, whose output transform adds the bias and applies
            # the ReLU on the way out (nothing extra to read or write), so the expanding 1x1 is the plain GEMM ...
            u = getattr(self, 'wino_u', None)
            wino = winograd.takes(self.conv2, out, u)
            if wino:
                out = winograd.conv3x3(out, u, self.conv2.out_channels, bias=self.fb2, relu=True, variant=winograd.DEFAULT_VARIANT)
            elif fused.conv3x3_x3_supported(self.conv2, out, self.fb2):  # ... float32 strided: implicit GEMM of the split-operand kernel
                h = out                                                   # (or MIOpen + the epilogue pass, whichever is faster for the shape)
                s = self.conv2.stride[0]
                out = fused.pick('conv3', h.shape[0] * ((h.shape[2] - 1) // s + 1) * ((h.shape[3] - 1) // s + 1), 9 * h.shape[1],
                                 self.conv2.out_channels, s > 1, False,
                                 lambda: fused.conv3x3_bias_act_x3(self.conv2, h, self.fb2, True),
                                 lambda: fused.bias_act_(self.conv2(h), self.fb2))
                wino = True                                               # (bias + ReLU applied: nothing left for the next operand)
            else:
                out = self.conv2(out)                                     # ... bfloat16: MIOpen, raw output ...
            a_bias = None if wino else self.fb2

            def two_launches():
                identity = x if self.downsample is None else self.downsample[0](x)
                # (a_bias: conv2's bias + ReLU applied by the 1x1 GEMM while it stages its operand)
                return fused.conv_bias_act(self.conv3, out, self.fb3, identity, a_bias=a_bias)
            # a block WITH a downsampling convolution, float32: conv3(out) + downsample(x) as ONE product (the identity tensor
            # is never written), conv2's bias + ReLU applied to the operand where conv2 left them out -- where that is faster
            if self.downsample is not None and fused.pair_supported(self.conv3, self.downsample[0], out, x, self.fb3, a_bias):
                return fused.pick('pair', out.shape[0] * out.shape[2] * out.shape[3], self.conv3.in_channels + self.downsample[0].in_channels,
                                  self.conv3.out_channels, self.downsample[0].stride[0] > 1, a_bias is not None,
                                  lambda: fused.conv1x1_pair_bias_act_x3(self.conv3, self.downsample[0], out, x, self.fb3, True, a_bias),
                                  two_launches)
            return two_launches()
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class _BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.fused = False

    def enable_fused_(self):
        _take_biases(self, ('conv1', 'conv2'))
        for i, conv in ((1, self.conv1), (2, self.conv2)):   # stride-1 3x3: the Winograd kernel's operand (see _Bottleneck)
            if conv.stride == (1, 1) and conv.in_channels % 16 == 0 and conv.out_channels % 64 == 0:
                self.register_buffer('wino_u%d' % i, winograd.transform_filter(conv.weight, winograd.DEFAULT_VARIANT),
                                     persistent=False)
        self.fused = True

    def forward(self, x):
        if self.fused:
            identity = x if self.downsample is None else self.downsample[0](x)
            u1, u2 = getattr(self, 'wino_u1', None), getattr(self, 'wino_u2', None)
            if winograd.takes(self.conv1, x, u1):             # bias + ReLU in the kernel's output transform
                out = winograd.conv3x3(x, u1, self.conv1.out_channels, bias=self.fb1, relu=True, variant=winograd.DEFAULT_VARIANT)
            else:
                out = fused.bias_act_(self.conv1(x), self.fb1)
            return fused.bias_act_(winograd.conv_or_fallback(self.conv2, out, u2), self.fb2, identity)
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + identity)


class BaseNetwork(nn.Module):
    def __init__(self, name, *, stride, out_features):
        super().__init__()
        self.name = name
        self.stride = stride
        self.out_features = out_features


class Resnet(BaseNetwork):
    """ResNet-18/34/50/101 without the input max-pool: stride 16."""
    CONFIGS = {
        'resnet18': (_BasicBlock, [2, 2, 2, 2]),
        'resnet34': (_BasicBlock, [3, 4, 6, 3]),
        'resnet50': (_Bottleneck, [3, 4, 6, 3]),
        'resnet101': (_Bottleneck, [3, 4, 23, 3]),
    }

    def __init__(self, name='resnet50'):
        block, layers = self.CONFIGS[name]
        super().__init__(name, stride=16, out_features=512 * block.expansion)
        self.inplanes = 64
        self.input_block = nn.Sequential(
            nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True))
        self.block2 = self._make_layer(block, 64, layers[0], 1)
        self.block3 = self._make_layer(block, 128, layers[1], 2)
        self.block4 = self._make_layer(block, 256, layers[2], 2)
        self.block5 = self._make_layer(block, 512, layers[3], 2)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def _make_layer(self, block, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, 1, stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def enable_fused_(self):
        conv = self.input_block[0]
        self.register_buffer('fb0', conv.bias.detach().clone())
        conv.bias = None
        self.fused = True

    def forward(self, x):
        if getattr(self, 'fused', False):
            if fused.stem_x3_supported(self.input_block[0], x, self.fb0):     # float32: the stem as an implicit GEMM, bias + ReLU inside
                conv, x0 = self.input_block[0], x
                x = fused.pick('stem', x0.shape[0] * ((x0.shape[2] - 1) // 2 + 1) * ((x0.shape[3] - 1) // 2 + 1), 256, conv.out_channels,
                               True, False, lambda: fused.stem7x7_bias_act_x3(conv, x0, self.fb0),
                               lambda: fused.bias_act_(conv(x0), self.fb0))
            else:
                x = fused.bias_act_(self.input_block[0](x), self.fb0)
        else:
            x = self.input_block(x)
        return self.block5(self.block4(self.block3(self.block2(x))))


def _channel_shuffle(x, groups=2):
    b, c, h, w = x.shape
    return x.view(b, groups, c // groups, h, w).transpose(1, 2).reshape(b, c, h, w)


class _InvertedResidualK(nn.Module):
    """ShuffleNetV2 unit with a k x k depthwise conv (reference ``basenetworks.py:186-268``)."""

    def __init__(self, inp, oup, first_in_stage, *, stride=1, kernel_size=5):
        super().__init__()
        assert stride in (1, 2) and (stride != 1 or inp == oup)
        self.first_in_stage = first_in_stage
        bf = oup // 2
        pad = (kernel_size - 1) // 2
        self.branch1 = None
        if first_in_stage:
            self.branch1 = nn.Sequential(
                nn.Conv2d(inp, inp, kernel_size, stride, pad, groups=inp, bias=False), nn.BatchNorm2d(inp),
                nn.Conv2d(inp, bf, 1, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))
        self.branch2 = nn.Sequential(
            nn.Conv2d(inp if first_in_stage else bf, bf, 1, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True),
            nn.Conv2d(bf, bf, kernel_size, stride, pad, groups=bf, bias=False), nn.BatchNorm2d(bf),
            nn.Conv2d(bf, bf, 1, bias=False), nn.BatchNorm2d(bf), nn.ReLU(inplace=True))

    fused = False

    def enable_fused_(self):
        """After conv+BN folding: the depthwise convolutions run as one HIP stencil kernel each (bias fused) and
        cat + channel_shuffle as one interleave pass.  Keeps the tap-major copies of the depthwise weights."""
        for branch in (self.branch1, self.branch2):
            if branch is None:
                continue
            for m in branch:
                if isinstance(m, nn.Conv2d) and m.groups == m.in_channels and m.groups > 1:
                    k = m.kernel_size[0]
                    m.register_buffer('w_taps', m.weight.detach().reshape(m.out_channels, k * k).t().contiguous())
        self.fused = True

    @staticmethod
    def _run(branch, x):
        for m in branch:
            if isinstance(m, nn.Conv2d) and hasattr(m, 'w_taps') and m.w_taps.dtype == x.dtype \
                    and fused.dwconv_supported(x, m.kernel_size[0], m.stride[0]):
                x = fused.dwconv_bias_act(x, m.w_taps, m.bias, m.kernel_size[0], m.stride[0])
            else:
                x = m(x)
        return x

    def forward(self, x):
        if self.fused and x.is_cuda:
            if self.branch1 is None:
                x1, x2 = x.chunk(2, dim=1)
                return fused.channel_interleave(x1, self._run(self.branch2, x2))
            return fused.channel_interleave(self._run(self.branch1, x), self._run(self.branch2, x))
        if self.branch1 is None:
            x1, x2 = x.chunk(2, dim=1)
            out = torch.cat((x1, self.branch2(x2)), dim=1)
        else:
            out = torch.cat((self.branch1(x), self.branch2(x)), dim=1)
        return _channel_shuffle(out, 2)


class ShuffleNetV2K(BaseNetwork):
    """ShuffleNetV2 with 5x5 depthwise kernels, stride 16 (reference ``basenetworks.py:271-355``)."""
    CONFIGS = {
        'shufflenetv2k16': ([4, 8, 4], [24, 348, 696, 1392, 1392]),
        'shufflenetv2k20': ([5, 10, 5], [32, 512, 1024, 2048, 2048]),
        'shufflenetv2k30': ([8, 16, 6], [32, 512, 1024, 2048, 2048]),
        'shufflenetv2k44': ([12, 24, 8], [32, 512, 1024, 2048, 2048]),
    }

    def __init__(self, name='shufflenetv2k16'):
        repeats, ch = self.CONFIGS[name]
        super().__init__(name, stride=16, out_features=ch[-1])
        self.input_block = nn.Sequential(
            nn.Conv2d(3, ch[0], 3, 2, 1, bias=False), nn.BatchNorm2d(ch[0]), nn.ReLU(inplace=True))
        stages, inp = [], ch[0]
        for rep, oup in zip(repeats, ch[1:4]):
            # the first stage keeps resolution (the reference drops the max-pool and uses stride 16)
            seq = [_InvertedResidualK(inp, oup, True, stride=2)]
            seq += [_InvertedResidualK(oup, oup, False) for _ in range(rep - 1)]
            stages.append(nn.Sequential(*seq))
            inp = oup
        self.stage2, self.stage3, self.stage4 = stages
        self.conv5 = nn.Sequential(
            nn.Conv2d(inp, ch[-1], 1, bias=False), nn.BatchNorm2d(ch[-1]), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.conv5(self.stage4(self.stage3(self.stage2(self.input_block(x)))))


BASE_FACTORIES = {
    **{n: (lambda n=n: Resnet(n)) for n in Resnet.CONFIGS},
    **{n: (lambda n=n: ShuffleNetV2K(n)) for n in ShuffleNetV2K.CONFIGS},
}


class CompositeField4(nn.Module):
    """Head: features -> ``[B, n_fields, n_components, H, W]`` in the layout the decoder reads."""

    def __init__(self, meta: headmeta.Base, in_features):
        super().__init__()
        self.meta = meta
        self.n_components = 1 + meta.n_confidences + meta.n_vectors * 2 + meta.n_scales
        self.conv = nn.Conv2d(in_features, meta.n_fields * self.n_components * (meta.upsample_stride ** 2), 1)
        self.upsample_op = nn.PixelShuffle(meta.upsample_stride) if meta.upsample_stride > 1 else None

    fused_epilogue = True      # one HIP kernel for everything behind the convolution (inference, channels_last)

    def forward(self, x):
        if not self.training and fused.head_conv_x3_supported(self.conv, x):     # float32 inference: the head's 1x1 convolution through
            conv, x0 = self.conv, x                                                # the split-operand GEMM where that is faster
            x = fused.pick('head', x0.shape[0] * x0.shape[2] * x0.shape[3], conv.in_channels, conv.out_channels, False, False,
                           lambda: fused.head_conv_x3(conv, x0), lambda: conv(x0))
        else:
            x = self.conv(x)
        if self.fused_epilogue and fused.head_epilogue_supported(x, self.meta, self.training):
            return fused.head_epilogue(x, self.meta)
        if self.upsample_op is not None:
            x = self.upsample_op(x)
            us = self.meta.upsample_stride
            low_cut = (us - 1) // 2
            high_cut = math.ceil((us - 1) / 2.0)
            x = x[:, :, low_cut:x.shape[2] - high_cut, low_cut:x.shape[3] - high_cut]
        B, _, H, W = x.shape
        m = self.meta
        x = x.float().reshape(B, m.n_fields, self.n_components, H, W).contiguous()
        if not self.training:
            nc = m.n_confidences
            x[:, :, 1:1 + nc].sigmoid_()
            ii = torch.arange(W, device=x.device, dtype=x.dtype)
            jj = torch.arange(H, device=x.device, dtype=x.dtype).unsqueeze(1)
            for i, do_offset in enumerate(m.vector_offsets):
                if do_offset:
                    x[:, :, 1 + nc + 2 * i] += ii
                    x[:, :, 1 + nc + 2 * i + 1] += jj
            first_scale = 1 + nc + m.n_vectors * 2
            x[:, :, first_scale:first_scale + m.n_scales] = nn.functional.softplus(
                x[:, :, first_scale:first_scale + m.n_scales])
        return x


class Shell(nn.Module):
    """base_net + head_nets (reference ``network/nets.py:11-48``)."""

    def __init__(self, base_net, head_nets):
        super().__init__()
        self.base_net = base_net
        self.head_nets = None
        self.set_head_nets(head_nets)

    @property
    def head_metas(self):
        return [hn.meta for hn in self.head_nets] if self.head_nets is not None else None

    def set_head_nets(self, head_nets):
        if not isinstance(head_nets, nn.ModuleList):
            head_nets = nn.ModuleList(head_nets)
        for i, hn in enumerate(head_nets):
            hn.meta.head_index = i
            hn.meta.base_stride = self.base_net.stride
        self.head_nets = head_nets

    def forward(self, image_batch):
        x = self.base_net(image_batch)
        return tuple(hn(x) for hn in self.head_nets)


def factory(base_name='resnet50', head_metas=None, *, seed=0):
    """Randomly initialised ``Shell`` for ``head_metas`` (default: cocokp CIF+CAF, upsample 2)."""
    if head_metas is None:
        head_metas = headmeta.cocokp_metas()
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        base = BASE_FACTORIES[base_name]()
        heads = [CompositeField4(m, base.out_features) for m in head_metas]
        net = Shell(base, heads)
    finally:
        torch.random.set_rng_state(gen_state)
    return net.eval()


def _fold(conv, bn):
    """conv <- conv followed by eval-mode batch norm."""
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        conv.weight.mul_(scale.reshape(-1, 1, 1, 1))
        bias = bn.bias - bn.running_mean * scale
        if conv.bias is not None:
            bias = bias + conv.bias * scale
        conv.bias = nn.Parameter(bias)


def fuse_conv_bn_(model):
    """Inference-time folding of every (Conv2d, BatchNorm2d) pair into the convolution
    (removes one full activation read+write per layer, the dominant non-GEMM cost at 641 px)."""
    assert not model.training
    for module in model.modules():
        if isinstance(module, nn.Sequential):
            prev_name, prev = None, None
            for name, child in list(module.named_children()):
                if isinstance(child, nn.BatchNorm2d) and isinstance(prev, nn.Conv2d):
                    _fold(prev, child)
                    setattr(module, name, nn.Identity())
                prev_name, prev = name, child
        for i in (1, 2, 3):
            conv, bn = getattr(module, 'conv%d' % i, None), getattr(module, 'bn%d' % i, None)
            if isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d):
                _fold(conv, bn)
                setattr(module, 'bn%d' % i, nn.Identity())
    return model


def optimize_for_inference_(model):
    """Fold every conv+BN pair, then switch the ResNet blocks to the fused-epilogue forward
    (conv without bias followed by ONE ``fused.bias_act_`` pass) and the ShuffleNetV2K units to the HIP depthwise /
    interleave kernels."""
    fuse_conv_bn_(model)
    for m in model.modules():
        if isinstance(m, (_Bottleneck, _BasicBlock, Resnet, _InvertedResidualK)):
            m.enable_fused_()
    return model
