"""Restatement of timm.models.regnet.RegStage as configured by projector.py:153-161 / 176-184
(stride 1, dilation 1, act SiLU, norm LayerNorm2d; timm defaults bottle_ratio=1, group_size=1,
se_ratio=0.25, downsample='conv1x1', linear_out=False).  State-dict key names follow timm 1.0.3
(`bN.conv1.conv.weight`, `bN.conv1.bn.weight`, `bN.se.fc1.weight` ...).  Parity unpinned."""
import torch.nn as nn

from .layers import LayerNormAct2d


class ConvNormAct(nn.Module):
    def __init__(self, in_chs, out_chs, kernel_size=1, stride=1, dilation=1, groups=1, apply_act=True,
                 act_layer=nn.ReLU):
        super().__init__()
        pad = ((stride - 1) + dilation * (kernel_size - 1)) // 2
        self.conv = nn.Conv2d(in_chs, out_chs, kernel_size, stride=stride, padding=pad, dilation=dilation,
                              groups=groups, bias=False)
        self.bn = LayerNormAct2d(out_chs, apply_act=apply_act, act_layer=act_layer)

    def forward(self, x):
        return self.bn(self.conv(x))


class SEModule(nn.Module):
    def __init__(self, channels, rd_channels, act_layer=nn.ReLU):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, rd_channels, 1, bias=True)
        self.bn = nn.Identity()
        self.act = act_layer()
        self.fc2 = nn.Conv2d(rd_channels, channels, 1, bias=True)
        self.gate = nn.Sigmoid()

    def forward(self, x):
        s = x.mean((2, 3), keepdim=True)
        s = self.fc2(self.act(self.bn(self.fc1(s))))
        return x * self.gate(s)


class Bottleneck(nn.Module):
    def __init__(self, in_chs, out_chs, stride=1, dilation=(1, 1), act_layer=nn.ReLU, norm_layer=None):
        super().__init__()
        self.conv1 = ConvNormAct(in_chs, out_chs, 1, act_layer=act_layer)
        self.conv2 = ConvNormAct(out_chs, out_chs, 3, stride=stride, dilation=dilation[0], groups=out_chs,
                                 act_layer=act_layer)
        self.se = SEModule(out_chs, int(round(in_chs * 0.25)), act_layer=act_layer)
        self.conv3 = ConvNormAct(out_chs, out_chs, 1, apply_act=False, act_layer=act_layer)
        self.act3 = act_layer()
        if in_chs != out_chs or stride != 1:
            self.downsample = ConvNormAct(in_chs, out_chs, 1, stride=stride, apply_act=False, act_layer=act_layer)
        else:
            self.downsample = nn.Identity()

    def forward(self, x):
        shortcut = x
        x = self.conv1(x)
        x = self.conv2(x)
        x = self.se(x)
        x = self.conv3(x)
        return self.act3(x + self.downsample(shortcut))


class RegStage(nn.Module):
    def __init__(self, depth, in_chs, out_chs, stride, dilation, act_layer=nn.ReLU, norm_layer=None, **_):
        super().__init__()
        for i in range(depth):
            self.add_module(f"b{i + 1}", Bottleneck(in_chs if i == 0 else out_chs, out_chs,
                                                    stride=stride if i == 0 else 1, dilation=(dilation, dilation),
                                                    act_layer=act_layer, norm_layer=norm_layer))

    def forward(self, x):
        for blk in self.children():
            x = blk(x)
        return x
