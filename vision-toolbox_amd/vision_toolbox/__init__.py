"""vision_toolbox -- MI355X-native build of the Darknet / VoVNet backbone hot path.

Import name and module layout mirror the reference package (vision_toolbox/__init__.py:1-3)
so `from vision_toolbox import backbones` keeps working; the arithmetic runs in
libvt_amd.so (hand-written HIP for gfx950) and only on the GPU.
"""
from . import backbones, checkpoint, components, necks
from .components import ConvNormAct
from .necks import FPN, PAN

__version__ = "0.1.0+mi355x"
