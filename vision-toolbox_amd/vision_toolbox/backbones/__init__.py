"""Backbones of the MI355X hot path: Darknet / CSPDarknet / DarknetYOLOv5 / VoVNet.

Both surfaces the reference snapshot exposes are exported: the classes with
`from_config` (reference backbones/__init__.py:3,10, tests/test_backbones.py:25-30) and the
named factories that classifier.py:58 / README.md:27 / the checkpoint file names use.
The reference's other backbones (ViT, Swin, ConvNeXt, torchvision extractors, ...) are
outside this build's scope.
"""
from .base import BaseBackbone
from .darknet import (
    CSPDarknetStage,
    Darknet,
    DarknetBlock,
    DarknetStage,
    DarknetYOLOv5,
    cspdarknet53,
    darknet19,
    darknet53,
    darknet_yolov5l,
    darknet_yolov5m,
    darknet_yolov5n,
    darknet_yolov5s,
    darknet_yolov5x,
)
from .vovnet import (
    ESEBlock,
    OSABlock,
    VoVNet,
    vovnet19_ese,
    vovnet19_slim_ese,
    vovnet27_slim,
    vovnet39,
    vovnet39_ese,
    vovnet57,
    vovnet57_ese,
    vovnet99_ese,
)
