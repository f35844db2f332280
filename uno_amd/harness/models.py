"""Darcy-flow U-NO (the 5-block "UNO_9" of the reference, darcy_flow_uno2d.py:27-141) built on the
MI355X-native operator blocks.  Sub-module names, shapes and registration order follow the reference,
so its state_dict loads with strict=True.  Differences are host-side hygiene only:
  * the positional grid is built once per (shape, device) and cached on the device instead of being rebuilt
    on the host and copied every forward (reference :135-141);
  * lift and projection run channels-first (the nn.Linear weights applied as batched GEMMs on the
    (B, C, pixels) view), which removes the two full-size permute copies around the U (reference :104, :126);
  * the point-wise projection runs before the crop (they commute), so the crop touches one channel.
The arithmetic is the reference's up to float32 summation order."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..integral_operators import OperatorBlock_2D, channel_mix


class UNO_9(nn.Module):
    """in_width = 3 ([a(x,y), x, y]); width = lifted channel count; pad = domain padding (scaled by
    ceil(S/85)); factor = channel growth per level.  Input (B, S, S, 1) -> output (B, S, S, 1)."""

    def __init__(self, in_width, width, pad=5, factor=1, block_cls=OperatorBlock_2D):
        super().__init__()
        self.in_width = in_width
        self.width = width
        self.padding = pad
        w, f = width, factor
        self.fc_n1 = nn.Linear(in_width, w // 2)
        self.fc0 = nn.Linear(w // 2, w)
        # (in, out, default grid, modes): grids are overridden at call time, modes are fixed
        self.conv0 = block_cls(w, 2 * f * w, 40, 40, 18, 18)
        self.conv1 = block_cls(2 * f * w, 4 * f * w, 20, 20, 8, 8, Normalize=True)
        self.conv2 = block_cls(4 * f * w, 4 * f * w, 20, 20, 8, 8)
        self.conv4 = block_cls(4 * f * w, 2 * f * w, 40, 40, 8, 8, Normalize=True)
        self.conv5 = block_cls(4 * f * w, w, 85, 85, 18, 18)
        self.fc1 = nn.Linear(2 * w, w)
        self.fc2 = nn.Linear(w, 1)
        self._grid_cache = {}

    def get_grid(self, shape, device):
        key = (tuple(shape[:3]), str(device))
        grid = self._grid_cache.get(key)
        if grid is None:
            b, sx, sy = shape[0], shape[1], shape[2]
            gx = torch.linspace(0, 1, sx, dtype=torch.float64).to(torch.float32).reshape(1, sx, 1, 1).expand(b, sx, sy, 1)
            gy = torch.linspace(0, 1, sy, dtype=torch.float64).to(torch.float32).reshape(1, 1, sy, 1).expand(b, sx, sy, 1)
            grid = torch.cat((gx, gy), dim=-1).contiguous().to(device)
            self._grid_cache = {key: grid}
        return grid

    def forward(self, x):
        S1, S2 = x.shape[1], x.shape[2]
        x = torch.cat((x, self.get_grid(x.shape, x.device)), dim=-1).permute(0, 3, 1, 2).contiguous()   # (B, 3, S, S): tiny
        lifted = F.gelu(channel_mix(F.gelu(channel_mix(x, self.fc_n1.weight, self.fc_n1.bias)), self.fc0.weight, self.fc0.bias))
        scale = math.ceil(S2 / 85)
        margin = scale * self.padding
        lifted = F.pad(lifted, [0, margin, 0, margin])
        d1, d2 = lifted.shape[-2], lifted.shape[-1]

        c0 = self.conv0(lifted, d1 // 2, d2 // 2)
        c1 = self.conv1(c0, d1 // 4, d2 // 4)
        c2 = self.conv2(c1, d1 // 4, d2 // 4)
        c4 = torch.cat([self.conv4(c2, d1 // 2, d2 // 2), c0], dim=1)
        c5 = torch.cat([self.conv5(c4, d1, d2), lifted], dim=1)
        out = channel_mix(F.gelu(channel_mix(c5, self.fc1.weight, self.fc1.bias)), self.fc2.weight, self.fc2.bias)
        return out[:, :, :S1, :S2].permute(0, 2, 3, 1).contiguous()     # crop the padding, back to (B, S, S, 1) (one channel: tiny)
