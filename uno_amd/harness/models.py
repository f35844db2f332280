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

from ..integral_operators import (GradJoin, OperatorBlock_2D, OperatorBlock_3D, channel_mix, channel_mix_cat, channel_mix_cat_project,
                                  gelu_channel_mix, gelu_channel_mix_pad, gelu_pad2d, gelu_project, lift_gelu_pad)


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
        x = torch.cat((x, self.get_grid(x.shape, x.device).to(x.dtype)), dim=-1).permute(0, 3, 1, 2).contiguous()   # (B, 3, S, S): tiny
        # lift + activation + domain padding: one forward kernel; neither fc_n1's nor fc0's output is stored (both are recomputed from
        # the 3-channel input where the backward pass needs them)
        scale = math.ceil(S2 / 85)
        margin = scale * self.padding
        product = hasattr(self.conv5, "forward_cat")          # MI355X operator blocks (the CPU baseline builds the model on oracle blocks)
        fused = product and self.conv5.non_lin and not self.conv5.normalize
        jl = GradJoin() if fused else None
        lifted = lift_gelu_pad(x, self.fc_n1, self.fc0, margin, margin, grad_join=jl)
        d1, d2 = lifted.shape[-2], lifted.shape[-1]

        if fused:
            # `lifted` and `c0` feed two layers each (skip connections).  Their gradients are JOINED: the later consumer leaves its
            # contribution (a truncated spectrum + accumulating closures) to the first consumer, which transforms the summed spectrum
            # once and returns the complete gradient - no second gradient tensor, no element-wise sum (GradJoin)
            # conv0 / conv2 end in a GELU (no normalisation): the block that completes the gradient of their output (conv1 with the
            # join of c0; conv4, c2's only consumer) applies gelu'(pre) in its last accumulating kernel (`out_join`)
            jc, j2 = GradJoin(), GradJoin()
            c0 = self.conv0(lifted, d1 // 2, d2 // 2, join=jl, out_join=jc)
            c1 = self.conv1(c0, d1 // 4, d2 // 4, join=jc)
            c2 = self.conv2(c1, d1 // 4, d2 // 4, out_join=j2)
            # skip connections: conv5 consumes cat([conv4 output, c0]) and fc1 cat([conv5 output, lifted]) from their two
            # sources; the concatenations are never built.  conv5's GELU is deferred to its only consumer: fc1 applies it while
            # reading the pre-activation tensor
            skip5 = [self.conv4(c2, d1 // 2, d2 // 2, join=j2), c0]
            # fc2(gelu(fc1(cat([gelu(conv5 pre), lifted])))): one forward kernel (the fc1 pass also reduces its 64 channels to the output)
            # ... on the S1 x S2 domain only (the padding is cropped from the result: its points are never computed)
            out = channel_mix_cat_project([self.conv5.forward_cat(skip5, d1, d2, defer_gelu=True, defer_grad=jc), lifted], self.fc1.weight,
                                          self.fc1.bias, self.fc2.weight, self.fc2.bias, gelu_first=True, defer_grad=jl, crop=(S1, S2))
            return out[:, :, :S1, :S2].permute(0, 2, 3, 1).contiguous()
        else:
            c0 = self.conv0(lifted, d1 // 2, d2 // 2)
            c1 = self.conv1(c0, d1 // 4, d2 // 4)
            c2 = self.conv2(c1, d1 // 4, d2 // 4)
            skip5 = [self.conv4(c2, d1 // 2, d2 // 2), c0]
            c5 = channel_mix_cat([_block_cat(self.conv5, skip5, d1, d2), lifted], self.fc1.weight, self.fc1.bias)
        out = gelu_project(c5, self.fc2.weight, self.fc2.bias)
        return out[:, :, :S1, :S2].permute(0, 2, 3, 1).contiguous()     # crop the padding, back to (B, S, S, 1) (one channel: tiny)


def _block_cat(block, xs, *dims):
    """block(torch.cat(xs, dim=1), *dims); product blocks take the sources as they are."""
    if hasattr(block, "forward_cat"):
        return block.forward_cat(xs, *dims)
    return block(torch.cat(list(xs), dim=1), *dims)


def _cached(cache: dict, key, build):
    grid = cache.get(key)
    if grid is None:
        grid = build()
        cache.clear()
        cache[key] = grid
    return grid


class UNO(nn.Module):
    """Navier-Stokes 2-D U-NO (7 blocks, channel growth factor 3/4) - own counterpart of the reference's
    `UNO` (navier_stokes_uno2d.py:145-238): one autoregressive step (B, S, S, T_in) -> (B, S, S, 1).  Input
    channels = T_in + 4 positional features (sin/cos of the two coordinates, :229-238).  Sub-module names and
    registration order follow the reference (state_dict compatible).  Channels-first lift/projection as in UNO_9."""

    def __init__(self, in_width, width, pad=0, factor=3 / 4, block_cls=OperatorBlock_2D):
        super().__init__()
        self.in_width, self.width, self.factor, self.padding = in_width, width, factor, pad
        w, f = width, factor
        self.fc = nn.Linear(in_width, w // 2)
        self.fc0 = nn.Linear(w // 2, w)
        self.L0 = block_cls(w, 2 * f * w, 48, 48, 22, 22)
        self.L1 = block_cls(2 * f * w, 4 * f * w, 32, 32, 14, 14)
        self.L2 = block_cls(4 * f * w, 8 * f * w, 16, 16, 6, 6)
        self.L3 = block_cls(8 * f * w, 8 * f * w, 16, 16, 6, 6)
        self.L4 = block_cls(8 * f * w, 4 * f * w, 32, 32, 6, 6)
        self.L5 = block_cls(8 * f * w, 2 * f * w, 48, 48, 14, 14)
        self.L6 = block_cls(4 * f * w, w, 64, 64, 22, 22)
        self.fc1 = nn.Linear(2 * w, 4 * w)
        self.fc2 = nn.Linear(4 * w, 1)
        self._grid_cache = {}

    def get_grid(self, shape, device):
        def build():
            b, sx, sy = shape[0], shape[1], shape[2]
            gx = torch.linspace(0, 2 * math.pi, sx, dtype=torch.float64).to(torch.float32).reshape(1, sx, 1, 1).expand(b, sx, sy, 1)
            gy = torch.linspace(0, 2 * math.pi, sy, dtype=torch.float64).to(torch.float32).reshape(1, 1, sy, 1).expand(b, sx, sy, 1)
            return torch.cat((torch.sin(gx), torch.sin(gy), torch.cos(gx), torch.cos(gy)), dim=-1).contiguous().to(device)
        return _cached(self._grid_cache, (tuple(shape[:3]), str(device)), build)

    def forward(self, x):
        # channels-first input in ONE pass: the cat kernel reads the permuted view of x and the (cached, channels-first) grid features
        z = torch.cat((x.permute(0, 3, 1, 2), self.get_grid(x.shape, x.device).permute(0, 3, 1, 2)), dim=1)
        return self.forward_cf(z).permute(0, 2, 3, 1).contiguous()         # one channel: the permuted view IS contiguous

    def forward_cf(self, x):
        """(B, T_in + 4, S, S) channels-first window + positional features -> (B, 1, S, S).  The roll-out keeps its window in this
        layout (harness.ns2d_rollout_loss): one concatenation per step instead of one for the window and one for the layout."""
        lifted = F.gelu(gelu_channel_mix(channel_mix(x, self.fc.weight, self.fc.bias), self.fc0.weight, self.fc0.bias))
        p = self.padding
        if p != 0:              # (F.pad with zero widths still copies the tensor: 40 copies per roll-out)
            lifted = F.pad(lifted, [p, p, p, p])
        d1, d2 = lifted.shape[-2], lifted.shape[-1]
        if hasattr(self.L6, "forward_cat") and all(b.non_lin and not b.normalize for b in (self.L2, self.L3)):
            # c2 and c3 have ONE consumer each: that block applies gelu'(pre) of its producer in the kernel that completes the
            # gradient (`out_join`, as in UNO_9) - no separate GELU-backward pass for L2 / L3.  (The skip tensors are NOT joined
            # here the way UNO_9 joins them: at 64^2 x 32 samples every kernel of this model is a 10-30 us launch and the joined
            # form trades two big element-wise sums for more small launches - measured 81.4 -> 84.3 ms per step.)
            j2, j3 = GradJoin(), GradJoin()
            c0 = self.L0(lifted, int(d1 * self.factor), int(d2 * self.factor))
            c1 = self.L1(c0, d1 // 2, d2 // 2)
            c2 = self.L2(c1, d1 // 4, d2 // 4, out_join=j2)
            c3 = self.L3(c2, d1 // 4, d2 // 4, join=j2, out_join=j3)
            c4 = torch.cat([self.L4(c3, d1 // 2, d2 // 2, join=j3), c1], dim=1)
        else:
            c0 = self.L0(lifted, int(d1 * self.factor), int(d2 * self.factor))
            c1 = self.L1(c0, d1 // 2, d2 // 2)
            c2 = self.L2(c1, d1 // 4, d2 // 4)
            c3 = self.L3(c2, d1 // 4, d2 // 4)
            c4 = torch.cat([self.L4(c3, d1 // 2, d2 // 2), c1], dim=1)
        c5 = torch.cat([self.L5(c4, int(d1 * self.factor), int(d2 * self.factor)), c0], dim=1)
        c6 = torch.cat([self.L6(c5, d1, d2), lifted], dim=1)
        if p != 0:      # the reference pads both sides but crops one (navier_stokes_uno2d.py:201,217-218); kept
            c6 = c6[..., :-p, :-p]
        return gelu_project(channel_mix(c6.contiguous(), self.fc1.weight, self.fc1.bias), self.fc2.weight, self.fc2.bias)


class Uno3D_T20(nn.Module):
    """Navier-Stokes 3-D (space-time) U-NO mapping 10 input steps to 20 output steps - own counterpart of the
    reference's `Uno3D_T20` (navier_stokes_uno3d.py:239-409): 7 OperatorBlock_3D that also stretch the time axis,
    skip connections through (identity) trilinear resizes, time-axis padding int(pad * 0.1 * T).
    Input (B, S, S, T, 1) -> output (B, S, S, 2T, 1); in_width = 1 + 5 positional features."""

    def __init__(self, in_width, width, pad=2, factor=1, pad_both=False, block_cls=OperatorBlock_3D):
        super().__init__()
        self.in_width, self.width, self.pad, self.pad_both = in_width, width, pad, pad_both
        w, f = width, factor
        self.fc = nn.Linear(in_width, in_width * 2)
        self.fc0 = nn.Linear(in_width * 2, w)
        self.conv0 = block_cls(w, 2 * f * w, 48, 48, 10, 22, 22, 5, Normalize=True)
        self.conv1 = block_cls(2 * f * w, 4 * f * w, 32, 32, 10, 14, 14, 5)
        self.conv2 = block_cls(4 * f * w, 8 * f * w, 16, 16, 12, 6, 6, 5)
        self.conv3 = block_cls(8 * f * w, 16 * f * w, 16, 16, 12, 6, 6, 6, Normalize=True)
        self.conv6 = block_cls(16 * f * w, 4 * f * w, 32, 32, 18, 6, 6, 6)
        self.conv7 = block_cls(8 * f * w, 2 * f * w, 48, 48, 20, 14, 14, 8, Normalize=True)
        self.conv8 = block_cls(4 * f * w, 2 * w, 64, 64, 20, 22, 22, 8)
        self.fc1 = nn.Linear(3 * w, 4 * w)
        self.fc2 = nn.Linear(4 * w, 1)
        self._grid_cache = {}

    def get_grid(self, shape, device):
        def build():
            b, sx, sy, sz = shape[0], shape[1], shape[2], shape[3]
            lin = lambda hi, n: torch.linspace(0, hi, n, dtype=torch.float64).to(torch.float32)
            gx = lin(2 * math.pi, sx).reshape(1, sx, 1, 1, 1).expand(b, sx, sy, sz, 1)
            gy = lin(2 * math.pi, sy).reshape(1, 1, sy, 1, 1).expand(b, sx, sy, sz, 1)
            gz = lin(1, sz).reshape(1, 1, 1, sz, 1).expand(b, sx, sy, sz, 1)
            return torch.cat((torch.sin(gx), torch.sin(gy), torch.cos(gx), torch.cos(gy), gz), dim=-1).contiguous().to(device)
        return _cached(self._grid_cache, (tuple(shape[:4]), str(device)), build)

    @staticmethod
    def _resize(t, like):
        # the reference resizes every skip tensor with trilinear / align_corners (navier_stokes_uno3d.py:352-372); on the
        # device this is the separable banded kernel (the stock backward kernel alone took 20 ms of a 60 ms step)
        from ..resample import resample3d_trilinear
        return resample3d_trilinear(t, tuple(like.shape[2:]))

    def forward(self, x):
        x = torch.cat((x, self.get_grid(x.shape, x.device)), dim=-1).permute(0, 4, 1, 2, 3).contiguous()
        lifted = F.gelu(gelu_channel_mix(channel_mix(x, self.fc.weight, self.fc.bias), self.fc0.weight, self.fc0.bias))
        self.padding = int(self.pad * 0.1 * lifted.shape[-1])
        lifted = F.pad(lifted, [self.padding, self.padding, 0, 0, 0, 0] if self.pad_both else [0, self.padding, 0, 0, 0, 0])
        d1, d2, d3 = lifted.shape[-3:]
        c0 = self.conv0(lifted, int(3 * d1 / 4), int(3 * d2 / 4), d3)
        c1 = self.conv1(c0, d1 // 2, d2 // 2, d3)
        c2 = self.conv2(c1, d1 // 4, d2 // 4, int(d3 * 1.2))
        c3 = self.conv3(c2, d1 // 4, d2 // 4, int(d3 * 1.2))
        c6 = self.conv6(c3, d1 // 2, d2 // 2, int(d3 * 1.8))
        c6 = torch.cat([c6, self._resize(c1, c6)], dim=1)
        c7 = self.conv7(c6, int(3 * d1 / 4), int(3 * d2 / 4), int(2.0 * d3))
        c7 = torch.cat([c7, self._resize(c0, c7)], dim=1)
        c8 = self.conv8(c7, d1, d2, 2 * d3)
        c8 = torch.cat([c8, self._resize(lifted, c8)], dim=1)
        if self.padding != 0:
            c8 = c8[..., 2 * self.padding:-2 * self.padding] if self.pad_both else c8[..., :-2 * self.padding]
        out = gelu_project(channel_mix(c8.contiguous(), self.fc1.weight, self.fc1.bias), self.fc2.weight, self.fc2.bias)
        return out.permute(0, 2, 3, 4, 1).contiguous()
