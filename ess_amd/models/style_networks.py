"""
Task decoder and image encoder of ESS on the HIP kernels (reference: models/style_networks.py).

SemSegE2VID / StyleEncoderE2VID / ReLUINSConv2d / INSResBlock keep the reference's constructor signatures,
forward contracts (dicts keyed by scale) and state_dict key layout.  The nn.Conv2d / nn.BatchNorm2d
children only hold parameters; forward/backward run through ess_amd.functional (fused concat +
nearest-upsample inside the conv tile loader, InstanceNorm/BatchNorm plane kernels, MFMA dgrad/wgrad).
"""
import torch
import torch.nn as nn

from .. import functional as Fn
from .. import hip
from .submodules import InterpolationLayer


def gaussian_weights_init(m):
    """N(0, 0.02) on every module whose class name starts with 'Conv' (reference :152-155)."""
    if m.__class__.__name__.find('Conv') == 0:
        m.weight.data.normal_(0.0, 0.02)


class ReLUINSConv2d(nn.Module):
    """Conv(bias) -> InstanceNorm2d(affine=False) -> ReLU (reference :158-169)."""

    def __init__(self, n_in, n_out, kernel_size, stride, padding=0):
        super().__init__()
        self.model = nn.Sequential(nn.Conv2d(n_in, n_out, kernel_size=kernel_size, stride=stride, padding=padding, bias=True),
                                   nn.InstanceNorm2d(n_out, affine=False), nn.ReLU(inplace=True))
        self.model.apply(gaussian_weights_init)

    def forward_fused(self, x, skip=None, up=False):
        """conv over cat(nearest_up2(x) if up else x, skip) without materialising either."""
        c = self.model[0]
        pre = Fn.pre_norm_fmt() if hip.is_c8(x) else None  # (the conv output is a pre-norm tensor: F16_C8 in the bf16 configuration)
        y = Fn.conv2d(x, c.weight, c.bias, c.stride[0], c.padding[0], x1=skip,
                      mode0=hip.SRC_NEAREST_UP2 if up else hip.SRC_DIRECT, out_c8=pre, half=True)  # (half: the mixed configuration's forward)
        return Fn.instance_norm(y, None, True, self.model[1].eps, x_f16=pre == Fn.PRE_NORM)

    def forward(self, x):
        return self.forward_fused(x)


class INSResBlock(nn.Module):
    """conv3x3 -> IN -> ReLU -> conv3x3 -> IN, plus the identity (reference :172-193)."""

    def conv3x3(self, inplanes, out_planes, stride=1):
        return [nn.Conv2d(inplanes, out_planes, kernel_size=3, stride=stride, padding=1)]

    def __init__(self, inplanes, planes, stride=1, dropout=0.0):
        super().__init__()
        if dropout > 0:
            raise NotImplementedError('INSResBlock dropout is never enabled by SemSegE2VID')
        layers = self.conv3x3(inplanes, planes, stride) + [nn.InstanceNorm2d(planes), nn.ReLU(inplace=True)]
        layers += self.conv3x3(planes, planes) + [nn.InstanceNorm2d(planes)]
        self.model = nn.Sequential(*layers)
        self.model.apply(gaussian_weights_init)

    def forward(self, x, first=False):
        """first (mixed configuration): the block reads the event latents -- its first pre-norm tensor has channel means of 6-17 standard
        deviations and leaves the convolution as a [hi | lo] half pair"""
        c1, c2 = self.model[0], self.model[3]
        # x enters the graph ONCE: conv1 hands it through as the skip operand, so the skip gradient is added inside
        # conv1's data-gradient kernel instead of by a separate elementwise pass (functional.Conv2dFn.forward)
        pre = Fn.pre_norm_fmt() if hip.is_c8(x) else None  # (the conv outputs are pre-norm tensors: F16_C8 in the bf16 configuration)
        pre1 = pre
        if first and pre == Fn.PRE_NORM and Fn.mixed() and c1.out_channels % 64 == 0:
            pre1 = Fn.PRE_NORM_HILO
        y, skip = Fn.conv2d_passthrough(x, c1.weight, c1.bias, c1.stride[0], 1, out_c8=pre1, half=True)
        hx = hip.h16_of(x)
        if hx is not None and hip.h16_of(skip) is None:  # (an input handed through an autograd.Function comes back as a new alias: keep its half copy)
            hip.attach_h16(skip, hx[0], hx[1])
        y = Fn.instance_norm(y, None, True, self.model[1].eps, x_f16=pre == Fn.PRE_NORM)
        y = Fn.conv2d(y, c2.weight, c2.bias, 1, 1, out_c8=pre, half=True)
        return Fn.instance_norm(y, skip, False, self.model[4].eps, x_f16=pre == Fn.PRE_NORM)  # IN(.) + residual in one pass


class SemSegE2VID(nn.Module):
    """Shared segmentation decoder (reference :9-107).  forward({1,2,4,8}) -> {8,4,2,1}."""

    def __init__(self, input_c, output_c, skip_connect=False, skip_type='sum', input_index_map=False):
        super().__init__()
        if input_index_map:
            raise NotImplementedError('input_index_map is never enabled by the ESS trainers')
        if skip_connect and skip_type != 'concat':
            raise ValueError("skip_connect=True only works with skip_type='concat' (decoder_scale_2 is built for "
                             "2x channels, reference :25)")
        self.skip_connect, self.skip_type, self.input_index_map = skip_connect, skip_type, input_index_map
        self.index_coords = None
        tch = input_c
        if skip_connect:
            blocks = [INSResBlock(tch, tch) for _ in range(5)]
            blocks += [ReLUINSConv2d(tch, tch // 2, kernel_size=3, stride=1, padding=1)]
            self.decoder_scale_1 = nn.Sequential(*blocks)
            self.decoder_scale_2 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, kernel_size=3, stride=1, padding=1),
                                                 ReLUINSConv2d(tch // 2, tch // 4, kernel_size=3, stride=1, padding=1))
            tch //= 2
            self.decoder_scale_3 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, kernel_size=3, stride=1, padding=1),
                                                 ReLUINSConv2d(tch // 2, tch // 2, kernel_size=3, stride=1, padding=1))
            tch //= 2
            self.decoder_scale_4 = nn.Sequential(ReLUINSConv2d(tch, tch // 2, kernel_size=3, stride=1, padding=1))
            tch //= 2
        else:
            self.decoder_scale_1 = nn.Sequential(*[INSResBlock(tch, tch) for _ in range(3)])
            for name in ('decoder_scale_2', 'decoder_scale_3', 'decoder_scale_4'):
                setattr(self, name, nn.Sequential(InterpolationLayer(scale_factor=2, mode='nearest'),
                                                  ReLUINSConv2d(tch, tch // 2, kernel_size=3, stride=1, padding=1)))
                tch //= 2
        self.decoder_scale_5 = nn.Sequential(nn.Conv2d(tch, output_c, kernel_size=1, stride=1, padding=0))

    def update_skip_dict(self, skips, x, sz_in):
        rem, scale = sz_in % x.shape[3], sz_in // x.shape[3]
        assert rem == 0
        skips[scale] = x

    def forward(self, input_dict):
        """bf16 configuration (hip.set_compute('bf16')): every activation between the latents and the logits is a BF16_C8
        tensor (bfloat16 [N, C/8, H, W, 8]) -- the latents are taken as they come (BF16_C8 from the image encoder, the
        frozen encoder's staging copies, or converted), out[2] / out[4] are returned in that form (hip.from_bf16_c8 gives
        the reference's fp32 NCHW view), out[1] (the logits) is fp32 NCHW in either configuration."""
        sz_in = input_dict[1].shape[3]
        x = input_dict[8]
        out = {8: x}
        c8 = Fn.c8_mode()
        lat = (lambda t, deep=False: Fn.as_c8(t, want_hilo=deep).contiguous()) if c8 else (lambda t, deep=False: t.contiguous())
        x = lat(x, True)  # (mixed configuration: the 1/8 latent enters as a [hi | lo] half pair)
        if self.skip_connect:
            x = self.decoder_scale_1[0](x, first=True)
            for blk in list(self.decoder_scale_1)[1:]:
                x = blk(x)
            x = self.decoder_scale_2[0].forward_fused(x, lat(input_dict[4]), up=True)
            # out[4] / out[2] feed the next stage AND (through the returned dict) the cycle losses: Fn.fork sums the two
            # gradients in one library launch instead of autograd's accumulation add
            x, xo = Fn.fork(self.decoder_scale_2[1](x))
            self.update_skip_dict(out, xo, sz_in)
            x = self.decoder_scale_3[0].forward_fused(x, lat(input_dict[2]), up=True)
            x, xo = Fn.fork(self.decoder_scale_3[1](x))
            self.update_skip_dict(out, xo, sz_in)
            x = self.decoder_scale_4[0].forward_fused(x, None, up=True)
        else:
            x = self.decoder_scale_1[0](x, first=True)
            for blk in list(self.decoder_scale_1)[1:]:
                x = blk(x)
            x, xo = Fn.fork(self.decoder_scale_2[1].forward_fused(x, None, up=True))
            self.update_skip_dict(out, xo, sz_in)
            x, xo = Fn.fork(self.decoder_scale_3[1].forward_fused(x, None, up=True))
            self.update_skip_dict(out, xo, sz_in)
            x = self.decoder_scale_4[1].forward_fused(x, None, up=True)
        c5 = self.decoder_scale_5[0]
        x = Fn.conv2d(x, c5.weight, c5.bias, 1, 0, out_c8=False, half=True)  # the logits: fp32 NCHW for the loss / metric kernels
        self.update_skip_dict(out, x, sz_in)
        return out


# ------------------------------------------------------------------------------------------------
_PENDING_BN_COUNTERS = []


def flush_bn_counters():
    """num_batches_tracked += 1 for every train-mode BatchNorm that ran since the last flush, as one multi-tensor launch."""
    if _PENDING_BN_COUNTERS:
        with torch.no_grad():
            torch._foreach_add_(_PENDING_BN_COUNTERS, 1)
        _PENDING_BN_COUNTERS.clear()


class _ConvBN(nn.Module):
    """Bias-free conv + BatchNorm2d (+residual) (+ReLU): train mode = conv kernel + BN plane kernel with batch
    statistics; eval mode = one fused conv kernel with the running statistics folded into its epilogue."""

    @staticmethod
    def run(conv, bn, x, residual=None, relu=True, passthrough=False):
        """passthrough (train mode): -> (out, x handed through conv's autograd node) for an identity skip.
        bf16 configuration: the output (and a residual) is a BF16_C8 tensor whatever the input's format (the stem reads the
        fp32 image)."""
        out_c8 = Fn.c8_mode()
        if bn.training:
            pre = Fn.pre_norm_fmt()  # (the conv output is read by the BatchNorm kernels only: F16_C8 in the bf16 configuration)
            if passthrough:
                y, skip = Fn.Conv2dFn.apply(x, None, conv.weight, None, conv.stride[0], conv.padding[0], hip.SRC_DIRECT,
                                            hip.SRC_DIRECT, True, pre)
            else:
                y = Fn.conv2d(x, conv.weight, None, conv.stride[0], conv.padding[0], out_c8=pre)
            out = Fn.batch_norm_train(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, relu,
                                      bn.momentum, bn.eps, x_f16=pre == Fn.PRE_NORM)
            _PENDING_BN_COUNTERS.append(bn.num_batches_tracked)  # incremented together (flush_bn_counters): 1 launch, not 15
            return (out, skip) if passthrough else out
        if passthrough:
            raise NotImplementedError('passthrough is a train-mode (autograd) feature')
        if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
            raise NotImplementedError('eval-mode BatchNorm is forward-only here (validation runs under no_grad, '
                                      'training/base_trainer.py:419)')
        N, C, H, W = x.shape[0], Fn._channels(x), x.shape[2], x.shape[3]
        spec = hip.conv_spec(N, H, W, C, 0, conv.out_channels, conv.kernel_size[0], conv.stride[0], conv.padding[0],
                             act=hip.ACT_RELU if relu else hip.ACT_NONE)
        with torch.no_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
        out = Fn._empty_act(N, conv.out_channels, spec.H_out, spec.W_out, x.device, out_c8)
        return hip.conv_forward(spec, x.contiguous(), None, Fn.packed_weight(spec, conv.weight),
                                hip.pack_rows(spec, scale.contiguous(), fill=1.0), hip.pack_rows(spec, shift.contiguous()),
                                residual, out=out, src_fmt=Fn._fmt(x), out_fmt=Fn._fmt(out))


class BasicBlock(nn.Module):
    """torchvision (0.7.0) ResNet BasicBlock: conv3x3(s)-BN-ReLU-conv3x3-BN + identity / 1x1(s)-BN downsample, ReLU.
    torchvision is not vendored by the reference (models/style_networks.py:3,117-121 import it); restated here."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        self.stride = stride

    def forward(self, x):
        if self.downsample is not None:
            xa, xb = Fn.fork(x)  # two consumers (1x1 downsample, conv1): their data-gradients are summed by one library launch
            identity = _ConvBN.run(self.downsample[0], self.downsample[1], xa, None, relu=False)
            out = _ConvBN.run(self.conv1, self.bn1, xb, None, relu=True)
        elif self.bn1.training:
            # identity skip: x enters the graph once, conv1 hands it through (skip gradient added in its dgrad epilogue)
            out, identity = _ConvBN.run(self.conv1, self.bn1, x, None, relu=True, passthrough=True)
        else:
            identity = x
            out = _ConvBN.run(self.conv1, self.bn1, x, None, relu=True)
        return _ConvBN.run(self.conv2, self.bn2, out, identity, relu=True)


def _resnet_init(module):
    """torchvision's ResNet initialisation (kaiming_normal fan_out for convs, BN weight 1 / bias 0)."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class _Stem(nn.Sequential):
    """conv7x7/s2 (1->64, no bias), bn1, relu, layer1 -- indices 0,1,2,3 as in the reference's Sequential."""

    def forward(self, x):
        x = _ConvBN.run(self[0], self[1], x, None, relu=True)
        return self[3](x)


class StyleEncoderE2VID(nn.Module):
    """Image encoder = conv7x7/s2 + ResNet-18 {bn1, relu, layer1, layer2, layer3}, no maxpool (reference :110-145).
    The reference takes those layers from torchvision.models.resnet18(pretrained=True); ImageNet weights are not
    available offline, so the layers start from torchvision's default initialisation and `pretrained_state_dict`
    (a torchvision resnet18 state_dict) can be passed to reproduce the reference's starting point."""

    def __init__(self, input_dim, skip_connect=False, pretrained_state_dict=None):
        super().__init__()
        self.skip_connect = skip_connect
        layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.encoder_scale_1 = _Stem(nn.Conv2d(input_dim, 64, kernel_size=(7, 7), stride=(2, 2), padding=(3, 3), bias=False),
                                     nn.BatchNorm2d(64), nn.ReLU(inplace=True), layer1)
        self.encoder_scale_2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.encoder_scale_3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        _resnet_init(self)
        if pretrained_state_dict is not None:
            self.load_resnet18(pretrained_state_dict)

    def load_resnet18(self, sd):
        """Copy bn1/layer1/layer2/layer3 of a torchvision resnet18 state_dict (the conv7x7 stays freshly initialised,
        exactly as in the reference where it is a new nn.Conv2d)."""
        remap = {}
        for k, v in sd.items():
            if k.startswith('bn1.'):
                remap['encoder_scale_1.1.' + k[4:]] = v
            elif k.startswith('layer1.'):
                remap['encoder_scale_1.3.' + k[7:]] = v
            elif k.startswith('layer2.'):
                remap['encoder_scale_2.' + k[7:]] = v
            elif k.startswith('layer3.'):
                remap['encoder_scale_3.' + k[7:]] = v
        missing = [k for k in self.state_dict() if k not in remap and k != 'encoder_scale_1.0.weight']
        if missing:
            raise KeyError(f'resnet18 state_dict lacks {missing[:4]}...')
        self.load_state_dict(remap, strict=False)

    def update_skip_dict(self, skips, x, sz_in):
        rem, scale = sz_in % x.shape[3], sz_in // x.shape[3]
        assert rem == 0
        skips[scale] = x

    def forward(self, x):
        """bf16 configuration: out[2] / out[4] / out[8] are BF16_C8 tensors (see SemSegE2VID.forward); out[1] is the image."""
        out = {1: x}
        sz_in = x.shape[3]
        x = self.encoder_scale_1(x.contiguous())
        if self.skip_connect:  # (a returned latent and the next stage: two consumers, see Fn.fork)
            x, xo = Fn.fork(x)
            self.update_skip_dict(out, xo, sz_in)
        x = self.encoder_scale_2(x)
        if self.skip_connect:
            x, xo = Fn.fork(x)
            self.update_skip_dict(out, xo, sz_in)
        x = self.encoder_scale_3(x)
        self.update_skip_dict(out, x, sz_in)
        flush_bn_counters()
        return out


def skip_concat(x1, x2):
    return torch.cat([x1, x2], dim=1)


def skip_sum(x1, x2):
    return hip.add(x1, x2)
