"""Darknet-19/53, CSPDarknet-53 and the YOLOv5 Darknet variants on libvt_amd.

Module tree, child names and variant tables follow the reference
(vision_toolbox/backbones/darknet.py:20-137) so that state_dict keys and the published
checkpoints match; what each module contributes here is not a Python `forward` over
ATen ops but `_vt_emit`, which appends its launches to a static launch list:

* DarknetBlock (ref :20-28): the `x + conv2(conv1(x))` add is the residual operand of
  conv2's normalise+ReLU pass -- no separate add kernel.
* CSPDarknetStage (ref :39-55): `torch.cat` is elided -- conv1 and the last block write
  straight into the two channel halves of the buffer out_conv reads.
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Sequence, Union

from torch import nn

from ..components import ConvNormAct, HipModule
from .base import BaseBackbone

__all__ = [
    "Darknet", "DarknetYOLOv5", "DarknetBlock", "DarknetStage", "CSPDarknetStage", "DarknetStageConfig",
    "darknet19", "darknet53", "cspdarknet53",
    "darknet_yolov5n", "darknet_yolov5s", "darknet_yolov5m", "darknet_yolov5l", "darknet_yolov5x",
]  # fmt: skip

_RELEASE = "https://github.com/gau-nernst/vision-toolbox/releases/download/v0.0.1/"


class DarknetBlock(HipModule):
    """1x1 squeeze -> 3x3 expand with identity shortcut (added after the ReLU)."""

    def __init__(self, in_channels: int, expansion: float = 0.5) -> None:
        super().__init__()
        self.conv1 = ConvNormAct(in_channels, int(in_channels * expansion), 1)
        self.conv2 = ConvNormAct(self.conv1.conv.out_channels, in_channels)

    def _vt_emit(self, b, x, out=None, name: str = "block"):
        h = self.conv1._vt_emit(b, x, name=name + ".conv1")
        return self.conv2._vt_emit(b, h, out=out, residual=x, name=name + ".conv2")

    def _vt_emit_maps(self, b, x):
        return [self._vt_emit(b, x)]

    def _eager_maps(self, x):
        return [x + self.conv2._eager(self.conv1._eager(x))]


def _emit_chain(blocks: Sequence[DarknetBlock], b, x, out, name: str):
    last = len(blocks) - 1
    for i, blk in enumerate(blocks):
        x = blk._vt_emit(b, x, out=out if i == last else None, name=f"{name}.{i}")
    return x


class DarknetStage(nn.Sequential, HipModule):
    """stride-2 3x3 conv followed by n residual blocks (YOLOv2/v3)."""

    def __init__(self, n: int, in_channels: int, out_channels: int) -> None:
        super().__init__()
        self.add_module("conv", ConvNormAct(in_channels, out_channels, stride=2))
        self.add_module("blocks", nn.Sequential(*(DarknetBlock(out_channels) for _ in range(n))))

    def _vt_emit(self, b, x, out=None, name: str = "stage"):
        blocks = list(self.blocks)
        o = self.conv._vt_emit(b, x, out=None if blocks else out, name=name + ".conv")
        return _emit_chain(blocks, b, o, out, name + ".blocks") if blocks else o

    def _vt_emit_maps(self, b, x):
        return [self._vt_emit(b, x)]

    def _eager_maps(self, x):
        h = self.conv._eager(x)
        for blk in self.blocks:
            h = blk._eager(h)
        return [h]

    def forward(self, x):
        return HipModule.forward(self, x)


class CSPDarknetStage(HipModule):
    """Cross-stage-partial stage: half the channels bypass the residual blocks."""

    def __init__(self, n: int, in_channels: int, out_channels: int) -> None:
        assert n > 0
        super().__init__()
        half = out_channels // 2
        self.conv = ConvNormAct(in_channels, out_channels, stride=2)
        self.conv1 = ConvNormAct(out_channels, half, 1)
        self.conv2 = ConvNormAct(out_channels, half, 1)
        self.blocks = nn.Sequential(*(DarknetBlock(half, expansion=1) for _ in range(n)))
        self.out_conv = ConvNormAct(out_channels, out_channels, 1)

    def _vt_emit(self, b, x, out=None, name: str = "csp"):
        o = self.conv._vt_emit(b, x, name=name + ".conv")
        width = self.out_conv.conv.in_channels
        half = self.conv1.conv.out_channels
        joined = b.act(o.B, o.H, o.W, width, name + ".cat")  # the tensor torch.cat would have produced
        # conv1 and conv2 read the same tensor: one N = C launch per pass where the pointwise kernels apply
        _, t = b.conv_unit_pair(o, (self.conv1, joined.sl(0, half), name + ".conv1"), (self.conv2, None, name + ".conv2"))
        _emit_chain(list(self.blocks), b, t, joined.sl(half, width - half), name + ".blocks")
        return self.out_conv._vt_emit(b, joined, out=out, name=name + ".out_conv")

    def _vt_emit_maps(self, b, x):
        return [self._vt_emit(b, x)]

    def _eager_maps(self, x):
        import torch

        o = self.conv._eager(x)
        bypass, h = self.conv1._eager(o), self.conv2._eager(o)
        for blk in self.blocks:
            h = blk._eager(h)
        return [self.out_conv._eager(torch.cat((bypass, h), dim=1))]


class DarknetStageConfig(NamedTuple):
    n_blocks: int
    out_channels: int


_StageCfg = Union[DarknetStageConfig, "tuple[int, int]"]


class Darknet(BaseBackbone):
    def __init__(
        self,
        stem_channels: int,
        stage_configs: "list[_StageCfg]",
        stage_cls: Callable[..., nn.Module] = DarknetStage,
    ):
        assert len(stage_configs) > 0
        super().__init__()
        self.out_channels_list = tuple(int(cfg[1]) for cfg in stage_configs)
        self.stride = 32
        self.stem = ConvNormAct(3, stem_channels)
        self.stages = nn.ModuleList()
        width = stem_channels
        for n_blocks, out_ch in stage_configs:
            # a stage without blocks is a bare stride-2 unit (darknet19's first stage, ref :79)
            self.stages.append(stage_cls(n_blocks, width, out_ch) if n_blocks else ConvNormAct(width, out_ch, 3, 2))
            width = out_ch

    def _vt_emit_maps(self, b, x):
        o = self.stem._vt_emit(b, x, name="stem")
        maps = []
        for i, stage in enumerate(self.stages):
            o = stage._vt_emit(b, o, name=f"stages.{i}")
            maps.append(o)
        return maps  # the stem is not a feature map (ref :87)

    def _eager_maps(self, x):
        h, maps = self.stem._eager(x), []
        for stage in self.stages:
            h = stage._eager(h)
            maps.append(h)
        return maps

    _VARIANTS = {
        # name: (blocks per stage, stage class, checkpoint)
        "darknet19": ((0, 1, 1, 2, 2), DarknetStage, "darknet19-2cb641ca.pth"),
        "darknet53": ((1, 2, 8, 8, 4), DarknetStage, "darknet53-94427f5b.pth"),
        "cspdarknet53": ((1, 2, 8, 8, 4), CSPDarknetStage, "cspdarknet53-3bfa0423.pth"),
    }

    @staticmethod
    def from_config(variant: str, pretrained: bool = False) -> "Darknet":
        depths, stage_cls, ckpt = Darknet._VARIANTS[variant]
        widths = (64, 128, 256, 512, 1024)
        m = Darknet(32, list(zip(depths, widths)), stage_cls)
        if pretrained:
            m._load_state_dict_from_url(_RELEASE + ckpt)
        return m


class DarknetYOLOv5(BaseBackbone):
    def __init__(self, stem_channels: int, stage_configs: "list[_StageCfg]") -> None:
        super().__init__()
        self.out_channels_list = (stem_channels,) + tuple(int(cfg[1]) for cfg in stage_configs)
        self.stride = 2 ** len(self.out_channels_list)
        self.stem = ConvNormAct(3, stem_channels, 6, 2)
        self.stages = nn.ModuleList()
        width = stem_channels
        for n_blocks, out_ch in stage_configs:
            self.stages.append(CSPDarknetStage(n_blocks, width, out_ch))
            width = out_ch

    def _vt_emit_maps(self, b, x):
        maps = [self.stem._vt_emit(b, x, name="stem")]  # the stem IS returned here (ref :120)
        for i, stage in enumerate(self.stages):
            maps.append(stage._vt_emit(b, maps[-1], name=f"stages.{i}"))
        return maps

    def _eager_maps(self, x):
        maps = [self.stem._eager(x)]
        for stage in self.stages:
            maps.append(stage._eager(maps[-1]))
        return maps

    _SCALES = {
        # name: (depth multiple, width multiple, checkpoint)
        "n": (1 / 3, 1 / 4, "darknet_yolov5n-68f182f1.pth"),
        "s": (1 / 3, 1 / 2, "darknet_yolov5s-175f7462.pth"),
        "m": (2 / 3, 3 / 4, "darknet_yolov5m-9866aa40.pth"),
        "l": (1 / 1, 1 / 1, "darknet_yolov5l-8e25d388.pth"),
        "x": (4 / 3, 5 / 4, "darknet_yolov5x-0ed0c035.pth"),
    }

    @staticmethod
    def from_config(variant: str, pretrained: bool = False) -> "DarknetYOLOv5":
        depth, width, ckpt = DarknetYOLOv5._SCALES[variant]
        cfgs = [(int(d * depth), int(w * width)) for d, w in zip((3, 6, 9, 3), (128, 256, 512, 1024))]
        m = DarknetYOLOv5(int(64 * width), cfgs)
        if pretrained:
            m._load_state_dict_from_url(_RELEASE + ckpt)
        return m


# named factories: the surface README.md:27 / classifier.py:58 / the checkpoint names use
def darknet19(pretrained: bool = False) -> Darknet:
    return Darknet.from_config("darknet19", pretrained)


def darknet53(pretrained: bool = False) -> Darknet:
    return Darknet.from_config("darknet53", pretrained)


def cspdarknet53(pretrained: bool = False) -> Darknet:
    return Darknet.from_config("cspdarknet53", pretrained)


def darknet_yolov5n(pretrained: bool = False) -> DarknetYOLOv5:
    return DarknetYOLOv5.from_config("n", pretrained)


def darknet_yolov5s(pretrained: bool = False) -> DarknetYOLOv5:
    return DarknetYOLOv5.from_config("s", pretrained)


def darknet_yolov5m(pretrained: bool = False) -> DarknetYOLOv5:
    return DarknetYOLOv5.from_config("m", pretrained)


def darknet_yolov5l(pretrained: bool = False) -> DarknetYOLOv5:
    return DarknetYOLOv5.from_config("l", pretrained)


def darknet_yolov5x(pretrained: bool = False) -> DarknetYOLOv5:
    return DarknetYOLOv5.from_config("x", pretrained)
