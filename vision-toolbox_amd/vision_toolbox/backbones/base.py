"""BaseBackbone -- the drop-in boundary of the hot path.

Mirrors the reference contract (vision_toolbox/backbones/base.py:14-25):
`get_feature_maps(x) -> list[Tensor]`, `forward(x) = get_feature_maps(x)[-1]`,
`out_channels_list`, `stride`, and checkpoint loading through torch.hub.  The legacy
`get_last_out_channels()` that classifier.py:63 still calls is provided too.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from ..components import HipModule

__all__ = ["BaseBackbone"]


class BaseBackbone(HipModule):
    out_channels_list: tuple
    stride: int

    def get_feature_maps(self, x: Tensor) -> list[Tensor]:
        return self._vt_runner()(x, all_maps=True, compute_dtype=self.compute_dtype)

    def forward(self, x: Tensor) -> Tensor:
        if type(self).get_feature_maps is not BaseBackbone.get_feature_maps:
            return self.get_feature_maps(x)[-1]  # user subclass overriding the reference's abstract hook
        return self._vt_runner()(x, all_maps=False, compute_dtype=self.compute_dtype)[-1]

    def get_last_out_channels(self) -> int:
        return int(self.out_channels_list[-1])

    def _load_state_dict_from_url(self, url: str) -> None:
        state_dict = torch.hub.load_state_dict_from_url(url)
        self.load_state_dict(state_dict)
