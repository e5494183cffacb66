"""VoVNet (one-shot aggregation) V1 / V2(eSE) on libvt_amd.

Module tree and variant tables follow the reference
(vision_toolbox/backbones/vovnet.py:20-136).  The OSA block's defining cost in the
reference is `torch.cat(outputs, dim=1)` over the input and all n intermediate maps
(ref :55) -- up to 2144 channels copied per block.  Here the concat buffer is allocated
FIRST and every producer (the max-pool / previous block for the input slice, each 3x3
unit for its slice) writes its channels in place; the 1x1 aggregation conv then reads the
buffer directly, and in backward its data gradient lands in the mirrored gradient buffer
where each 3x3 unit accumulates into its predecessor's slice.
"""
from __future__ import annotations

from typing import NamedTuple, Union

from torch import nn

from ..components import ConvNormAct, HipModule
from .base import BaseBackbone

__all__ = [
    "VoVNet", "OSABlock", "ESEBlock", "VoVNetStageConfig",
    "vovnet27_slim", "vovnet39", "vovnet57",
    "vovnet19_slim_ese", "vovnet19_ese", "vovnet39_ese", "vovnet57_ese", "vovnet99_ese",
]  # fmt: skip

_RELEASE = "https://github.com/gau-nernst/vision-toolbox/releases/download/v0.0.1/"


class ESEBlock(HipModule):
    """effective squeeze-excitation: x * hardsigmoid(conv1x1(avgpool(x)))."""

    def __init__(self, num_channels: int) -> None:
        super().__init__()
        self.pool = nn.AdaptiveAvgPool2d((1, 1))
        self.linear = nn.Conv2d(num_channels, num_channels, 1)
        self.gate = nn.Hardsigmoid(inplace=True)

    def _vt_emit(self, b, x, out=None, residual=None, name: str = "ese"):
        return b.ese(x, self.linear, residual=residual, out=out, name=name)

    def _vt_emit_maps(self, b, x):
        return [self._vt_emit(b, x)]

    def _eager_maps(self, x):
        return [x * self.gate(self.linear(self.pool(x)))]


class OSABlock(HipModule):
    def __init__(self, in_channels: int, mid_channels: int, num_layers: int, out_channels: int,
                 ese: bool = True) -> None:
        super().__init__()
        self.convs = nn.ModuleList(
            ConvNormAct(mid_channels if i else in_channels, mid_channels) for i in range(num_layers)
        )
        self.out_conv = ConvNormAct(in_channels + mid_channels * num_layers, out_channels, 1)
        self.ese = ESEBlock(out_channels) if ese else None
        self.residual = in_channels == out_channels

    # geometry of the aggregation buffer -------------------------------------------------
    @property
    def _in_channels(self) -> int:
        return self.convs[0].conv.in_channels

    def _vt_alloc_concat(self, b, B, H, W, name: str):
        return b.act(B, H, W, self.out_conv.conv.in_channels, name + ".cat")

    def _vt_emit_from_concat(self, b, joined, out=None, name: str = "osa", pool_out=None):
        """`joined[:, :in]` already holds the block input; fill the rest and aggregate.  `pool_out`: the next stage's
        max-pool of this block's output goes there (fused into out_conv's normalise pass: ConvNormAct._vt_emit)."""
        cin = self._in_channels
        x = joined.sl(0, cin)
        prev, off = x, cin
        for i, conv in enumerate(self.convs):
            mid = conv.conv.out_channels
            prev = conv._vt_emit(b, prev, out=joined.sl(off, mid), name=f"{name}.convs.{i}")
            off += mid
        shortcut = x if self.residual else None
        if self.ese is None:
            return self.out_conv._vt_emit(b, joined, out=out, residual=shortcut, name=name + ".out_conv", pool_out=pool_out)
        t = self.out_conv._vt_emit(b, joined, name=name + ".out_conv")
        y = self.ese._vt_emit(b, t, out=out, residual=shortcut, name=name + ".ese")
        if pool_out is not None:
            b.maxpool3x3s2(y, out=pool_out, name=name + ".max_pool")
        return y

    def _vt_emit(self, b, x, out=None, name: str = "osa"):
        joined = self._vt_alloc_concat(b, x.B, x.H, x.W, name)
        b.copy(x, joined.sl(0, x.C))
        return self._vt_emit_from_concat(b, joined, out=out, name=name)

    def _vt_emit_maps(self, b, x):
        return [self._vt_emit(b, x)]

    def _eager_maps(self, x):
        import torch

        feats = [x]
        for conv in self.convs:
            feats.append(conv._eager(feats[-1]))
        y = self.out_conv._eager(torch.cat(feats, dim=1))
        if self.ese is not None:
            y = self.ese._eager(y)
        return [y + x if self.residual else y]


class VoVNetStageConfig(NamedTuple):
    n_blocks: int
    mid_channels: int
    n_layers: int
    out_channels: int


_StageCfg = Union[VoVNetStageConfig, "tuple[int, int, int, int]"]


def _pooled(n: int) -> int:
    return (n + 2 - 3) // 2 + 1  # MaxPool2d(3, 2, 1)


class VoVNet(BaseBackbone):
    def __init__(self, stem_channels: int, stage_configs: "list[_StageCfg]", ese: bool = True) -> None:
        super().__init__()
        self.out_channels_list = (stem_channels,) + tuple(int(cfg[3]) for cfg in stage_configs)
        self.stride = 2 ** len(self.out_channels_list)

        half = stem_channels // 2
        self.stem = nn.Sequential(
            ConvNormAct(3, half, 3, 2),
            ConvNormAct(half, half),
            ConvNormAct(half, stem_channels),
        )
        self.stages = nn.ModuleList()
        width = stem_channels
        for n_blocks, mid_ch, n_layers, out_ch in stage_configs:
            stage = nn.Sequential()
            stage.add_module("max_pool", nn.MaxPool2d(3, 2, 1))
            for i in range(n_blocks):
                stage.add_module(f"module_{i}", OSABlock(width, mid_ch, n_layers, out_ch, ese))
                width = out_ch
            self.stages.append(stage)

    def _vt_emit_maps(self, b, x):
        stage_blocks = [[m for m in stage.children() if isinstance(m, OSABlock)] for stage in self.stages]

        def pooled_concat(src_B, src_H, src_W, si):
            """the aggregation buffer of stage si's first block, whose first slice the stage's max-pool fills"""
            return stage_blocks[si][0]._vt_alloc_concat(b, src_B, _pooled(src_H), _pooled(src_W), f"stages.{si}.module_0")

        o = x
        stem = list(self.stem)
        for i, unit in enumerate(stem[:-1]):
            o = unit._vt_emit(b, o, name=f"stem.{i}")
        # every stage opens with MaxPool2d(3, 2, 1) of the previous stage's output (ref :94): it is written by the unit
        # that produces that output -- the last stem unit, then each stage's last block -- straight into the first slice
        # of the next aggregation buffer
        last = stem[-1]
        s_ = last.conv.stride[0]
        Ho0 = (o.H + 2 * last.conv.padding[0] - last.conv.kernel_size[0]) // s_ + 1
        Wo0 = (o.W + 2 * last.conv.padding[0] - last.conv.kernel_size[0]) // s_ + 1
        joined = pooled_concat(o.B, Ho0, Wo0, 0) if stage_blocks else None
        o = last._vt_emit(b, o, name=f"stem.{len(stem) - 1}",
                          pool_out=joined.sl(0, last.conv.out_channels) if joined is not None else None)
        maps = [o]
        for si, blocks in enumerate(stage_blocks):
            H, W = joined.H, joined.W
            for bi, blk in enumerate(blocks):
                nxt, pool_next, joined_next_stage = None, None, None
                out_ch = blk.out_conv.conv.out_channels
                if bi + 1 < len(blocks):
                    nxt = blocks[bi + 1]._vt_alloc_concat(b, joined.B, H, W, f"stages.{si}.module_{bi + 1}")
                elif si + 1 < len(stage_blocks):
                    joined_next_stage = pooled_concat(joined.B, H, W, si + 1)
                    pool_next = joined_next_stage.sl(0, out_ch)
                out = nxt.sl(0, out_ch) if nxt is not None else None
                o = blk._vt_emit_from_concat(b, joined, out=out, name=f"stages.{si}.module_{bi}", pool_out=pool_next)
                joined = nxt if nxt is not None else joined_next_stage
            maps.append(o)
        return maps

    def _eager_maps(self, x):
        h = x
        for unit in self.stem:
            h = unit._eager(h)
        maps = [h]
        for stage in self.stages:
            for m in stage.children():
                h = m._eager(h) if isinstance(m, HipModule) else m(h)  # MaxPool2d, then the OSA blocks
            maps.append(h)
        return maps

    _DEPTHS = {
        # variant: (blocks per stage, 3x3 layers per block)
        19: ((1, 1, 1, 1), (3, 3, 3, 3)),
        27: ((1, 1, 1, 1), (5, 5, 5, 5)),
        39: ((1, 1, 2, 2), (5, 5, 5, 5)),
        57: ((1, 1, 4, 3), (5, 5, 5, 5)),
        99: ((1, 3, 9, 3), (5, 5, 5, 5)),
    }
    _CKPTS = {
        (27, True, False): "vovnet27_slim-dd43306a.pth",
        (39, False, False): "vovnet39-4c79d629.pth",
        (57, False, False): "vovnet57-ecb9cc34.pth",
        (19, True, True): "vovnet19_slim_ese-f8075640.pth",
        (19, False, True): "vovnet19_ese-a077657e.pth",
        (39, False, True): "vovnet39_ese-9ce81b0d.pth",
        (57, False, True): "vovnet57_ese-ae1a7f89.pth",
        (99, False, True): "vovnet99_ese-713f3062.pth",
    }

    @staticmethod
    def from_config(variant: int, slim: bool = False, ese: bool = False, pretrained: bool = False) -> "VoVNet":
        n_blocks, n_layers = VoVNet._DEPTHS[variant]
        mids = (64, 80, 96, 112) if slim else (128, 160, 192, 224)
        outs = (128, 256, 384, 512) if slim else (256, 512, 768, 1024)
        m = VoVNet(128, list(zip(n_blocks, mids, n_layers, outs)), ese)
        if pretrained:
            m._load_state_dict_from_url(_RELEASE + VoVNet._CKPTS[(variant, slim, ese)])
        return m


def vovnet27_slim(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(27, slim=True, ese=False, pretrained=pretrained)


def vovnet39(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(39, slim=False, ese=False, pretrained=pretrained)


def vovnet57(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(57, slim=False, ese=False, pretrained=pretrained)


def vovnet19_slim_ese(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(19, slim=True, ese=True, pretrained=pretrained)


def vovnet19_ese(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(19, slim=False, ese=True, pretrained=pretrained)


def vovnet39_ese(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(39, slim=False, ese=True, pretrained=pretrained)


def vovnet57_ese(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(57, slim=False, ese=True, pretrained=pretrained)


def vovnet99_ese(pretrained: bool = False) -> VoVNet:
    return VoVNet.from_config(99, slim=False, ese=True, pretrained=pretrained)
