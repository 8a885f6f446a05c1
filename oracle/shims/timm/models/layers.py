"""Restatement of timm.models.layers.{LayerNorm, LayerNorm2d, LayerNormAct2d} (timm 1.0.3).
Used only as the oracle for projector.py:22-23,153-161,176-184.  Parity unpinned (no timm on this box)."""
import torch.nn as nn
import torch.nn.functional as F

LayerNorm = nn.LayerNorm  # imported by projector.py:23, never instantiated there


class LayerNorm2d(nn.LayerNorm):
    """Channel-wise LN over NCHW; projector.py passes this class as the `norm_layer` type tag."""

    def __init__(self, num_channels, eps=1e-6, affine=True):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        return x.permute(0, 3, 1, 2)


class LayerNormAct2d(nn.LayerNorm):
    """What timm's ConvNormAct builds when norm_layer=LayerNorm2d: LN over C (eps 1e-5) then the activation."""

    def __init__(self, num_channels, eps=1e-5, affine=True, apply_act=True, act_layer=nn.ReLU):
        super().__init__(num_channels, eps=eps, elementwise_affine=affine)
        self.drop = nn.Identity()
        self.act = act_layer() if apply_act else nn.Identity()

    def forward(self, x):
        x = x.permute(0, 2, 3, 1)
        x = F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        x = x.permute(0, 3, 1, 2)
        return self.act(self.drop(x))
