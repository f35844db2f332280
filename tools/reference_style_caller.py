"""The Darcy U-NO driven the way the reference's model file drives the operator blocks (darcy_flow_uno2d.py:94-141):
channels-last `nn.Linear` lift and projection, `F.gelu`, `permute`, `F.pad`, positional `block(x, d1, d2)` calls,
`torch.cat` skip connections, crop, and a positional grid rebuilt on the host on every forward.

It exists to answer one question: what does a user get who keeps the reference's model code UNCHANGED and only swaps
`integral_operators` for this package?  (The fused harness model `UNO_9` additionally uses this package's own helpers -
channels-first lift, two-source blocks, fused GELU - which a drop-in user would not call.)  Same parameters and
registration order as `UNO_9`, so the reference's state_dict loads into both.

A MEASUREMENT COMPARATOR (bench.py `extras.darcy_reference_style_caller`, two tests): it lives beside the bench, not in the product
package - a drop-in user runs the reference's own darcy_flow_uno2d.py, not this file."""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

from uno_amd.harness.models import UNO_9


class UNO_9_ReferenceStyle(UNO_9):
    def forward(self, a):
        b, sx, sy = a.shape[0], a.shape[1], a.shape[2]
        # positional features, host-built and copied per call like the reference's get_grid
        gx = torch.tensor(np.linspace(0, 1, sx), dtype=torch.float).reshape(1, sx, 1, 1).repeat([b, 1, sy, 1])
        gy = torch.tensor(np.linspace(0, 1, sy), dtype=torch.float).reshape(1, 1, sy, 1).repeat([b, sx, 1, 1])
        h = torch.cat((a, torch.cat((gx, gy), dim=-1).to(a.device)), dim=-1)
        h = F.gelu(self.fc_n1(h))
        h = F.gelu(self.fc0(h))
        lifted = h.permute(0, 3, 1, 2)
        margin = math.ceil(lifted.shape[-1] / 85) * self.padding
        lifted = F.pad(lifted, [0, margin, 0, margin])
        d1, d2 = lifted.shape[-2], lifted.shape[-1]
        c0 = self.conv0(lifted, d1 // 2, d2 // 2)
        c1 = self.conv1(c0, d1 // 4, d2 // 4)
        c2 = self.conv2(c1, d1 // 4, d2 // 4)
        c4 = torch.cat([self.conv4(c2, d1 // 2, d2 // 2), c0], dim=1)
        c5 = torch.cat([self.conv5(c4, d1, d2), lifted], dim=1)
        if self.padding != 0:
            c5 = c5[..., :-margin, :-margin]
        out = F.gelu(self.fc1(c5.permute(0, 2, 3, 1)))
        return self.fc2(out)
