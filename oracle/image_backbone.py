"""Oracle restatement of the reference's image feature extractor (TEST INFRASTRUCTURE - never imported by the product).

The reference detector calls `self.img_backbone(img.float())` and `self.img_neck(img_feats)`
(`models/detectors/deepinteraction.py:100-118`); `Fusion_0075_refactor.py:132-145` configures them as mmdet's
`ResNet(depth=50, num_stages=4, out_indices=(0,1,2,3), frozen_stages=1, norm_eval=True, style='pytorch')` and
`FPN(in_channels=[256,512,1024,2048], out_channels=256, num_outs=5)`.  Both classes live in mmdet 2.14.0 (pinned by the
reference's install.md:14), which is NOT under /root/reference and not installed here; this file restates their
published architecture in plain, unfused fp32 torch modules whose parameter names are mmdet's, so that a `state_dict()`
of these modules has the layout of an mmdet checkpoint:

  backbone: conv1.weight, bn1.{weight,bias,running_mean,running_var,num_batches_tracked},
            layer{1..4}.{j}.conv{1,2,3}.weight, layer{1..4}.{j}.bn{1,2,3}.*, layer{1..4}.0.downsample.{0.weight,1.*}
  neck:     lateral_convs.{i}.conv.{weight,bias}, fpn_convs.{i}.conv.{weight,bias}

PARITY UNPINNED: the reference holds no test or golden vector for this boundary and mmdet cannot be imported here.  The
one outside anchor is the entry count: ResNet-50 without its classifier has 318 state entries (torchvision's 320 minus
`fc.weight`, `fc.bias`), which `tests/test_image_backbone.py` checks.
"""
import torch
import torch.nn.functional as F
from torch import nn

STAGE_BLOCKS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}


class Bottleneck(nn.Module):
    """mmdet `models/backbones/resnet.py` Bottleneck, style='pytorch': the stride sits on the 3x3 convolution."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        if downsample:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))
        else:
            self.downsample = None

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = F.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return F.relu(out + identity)


class ResNet(nn.Module):
    """mmdet ResNet, depth 50/101, four stages, all four stage outputs; evaluation-mode BatchNorm (`norm_eval=True`)."""

    def __init__(self, depth=50):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for s, blocks in enumerate(STAGE_BLOCKS[depth]):
            planes, stride = 64 << s, 1 if s == 0 else 2
            layer = []
            for j in range(blocks):
                first = j == 0
                layer.append(Bottleneck(inplanes, planes, stride if first else 1,
                                        downsample=first and (stride != 1 or inplanes != planes * 4)))
                inplanes = planes * 4
            setattr(self, f'layer{s + 1}', nn.Sequential(*layer))
        self.eval()

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
        outs = []
        for s in range(4):
            x = getattr(self, f'layer{s + 1}')(x)
            outs.append(x)
        return tuple(outs)


class _Conv(nn.Module):
    """mmcv ConvModule without norm / activation: the convolution is the child `conv`."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2)

    def forward(self, x):
        return self.conv(x)


class FPN(nn.Module):
    """mmdet `models/necks/fpn.py` with start_level=0, add_extra_convs=False, nearest up-sampling to the finer map's
    size, extra levels by `max_pool2d(kernel 1, stride 2)` of the last output."""

    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256, num_outs=5):
        super().__init__()
        self.num_outs = num_outs
        self.lateral_convs = nn.ModuleList(_Conv(c, out_channels, 1) for c in in_channels)
        self.fpn_convs = nn.ModuleList(_Conv(out_channels, out_channels, 3) for _ in in_channels)

    def forward(self, inputs):
        lat = [conv(x) for conv, x in zip(self.lateral_convs, inputs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        outs = [conv(x) for conv, x in zip(self.fpn_convs, lat)]
        while len(outs) < self.num_outs:
            outs.append(F.max_pool2d(outs[-1], 1, stride=2))
        return tuple(outs)


def randomize(module, seed=0):
    """Random weights and NON-TRIVIAL BatchNorm statistics (a fresh BatchNorm is the identity: folding errors would hide)."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            fan = m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3]
            m.weight.data = torch.randn(m.weight.shape, generator=g) * (2.0 / fan) ** 0.5
            if m.bias is not None:
                m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data = 0.5 + torch.rand(m.weight.shape, generator=g)
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var.data = 0.5 + torch.rand(m.running_var.shape, generator=g)
    return module
