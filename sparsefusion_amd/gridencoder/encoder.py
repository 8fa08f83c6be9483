"""Multi-resolution grid encoder module on the HIP backend.

Host-side counterpart of the reference's GridEncoder / grid_encode
(external/gridencoder/grid.py:19-154): identical constructor arguments,
parameter and buffer names (`embeddings` [rows, C], `offsets` [L+1] int32),
level-size rule (:110-121), U(-1e-4, 1e-4) init (:131-133), input mapping
(x+bound)/(2*bound) (:142) and [..., L*C] output -- so reference NGP checkpoints
load unchanged.  fp32 only (the reference distillation loop never autocasts).

Differences in construction: the level table is kept on the host next to the
device buffer (no device->host sync per call), and the autograd node stores the
geometry in one small record instead of re-deriving it."""
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn as nn

from . import backend

GRIDTYPE_ID = {'hash': 0, 'tiled': 1}


@dataclass(frozen=True)
class GridGeometry:
    n_points: int
    in_dim: int
    feat: int
    levels: int
    log2_scale: float
    base_res: int
    gridtype: int
    align_corners: bool


def level_offsets(input_dim, num_levels, per_level_scale, base_resolution, log2_hashmap_size, align_corners):
    """Row offset of every level: rows = min(2^log2_hashmap_size, (res[+1])^D) rounded up to a
    multiple of 8 (grid.py:110-121)."""
    cap = 2 ** log2_hashmap_size
    rows = []
    for lvl in range(num_levels):
        res = int(np.ceil(base_resolution * per_level_scale ** lvl))
        side = res if align_corners else res + 1
        rows.append(int(np.ceil(min(cap, side ** input_dim) / 8) * 8))
    return np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)


class GridEncodeFn(torch.autograd.Function):
    """y[B, L*C] = encode(x[B, D]; table).  Backward scatters into a zeroed table gradient."""

    @staticmethod
    def forward(ctx, x, table, offsets, per_level_scale, base_resolution, want_dx=False, gridtype=0,
                align_corners=False):
        x = x.contiguous()
        table_c = table.contiguous()
        geo = GridGeometry(x.shape[0], x.shape[1], table.shape[1], offsets.shape[0] - 1,
                           float(np.log2(per_level_scale)), int(base_resolution), int(gridtype),
                           bool(align_corners))
        y_lbc = x.new_empty((geo.levels, geo.n_points, geo.feat))
        dy_dx = x.new_empty((geo.n_points, geo.levels * geo.in_dim * geo.feat)) if want_dx else None
        backend.grid_encode_forward(x, table_c, offsets, y_lbc, geo.n_points, geo.in_dim, geo.feat, geo.levels,
                                    geo.log2_scale, geo.base_res, dy_dx, geo.gridtype, geo.align_corners)
        ctx.geo = geo
        ctx.save_for_backward(x, table_c, offsets, dy_dx)
        return y_lbc.permute(1, 0, 2).reshape(geo.n_points, geo.levels * geo.feat)

    @staticmethod
    def backward(ctx, gy):
        x, table_c, offsets, dy_dx = ctx.saved_tensors
        geo = ctx.geo
        g_lbc = gy.reshape(geo.n_points, geo.levels, geo.feat).permute(1, 0, 2).contiguous()
        g_table = torch.zeros_like(table_c)
        g_x = torch.zeros_like(x) if dy_dx is not None else None
        backend.grid_encode_backward(g_lbc, x, table_c, offsets, g_table, geo.n_points, geo.in_dim, geo.feat,
                                     geo.levels, geo.log2_scale, geo.base_res, dy_dx, g_x, geo.gridtype,
                                     geo.align_corners)
        return g_x, g_table, None, None, None, None, None, None


grid_encode = GridEncodeFn.apply


class GridEncoder(nn.Module):
    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False):
        super().__init__()
        if desired_resolution is not None:
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale, base_resolution
        self.log2_hashmap_size, self.max_params = log2_hashmap_size, 2 ** log2_hashmap_size
        self.gridtype, self.gridtype_id, self.align_corners = gridtype, GRIDTYPE_ID[gridtype], align_corners
        self.output_dim = num_levels * level_dim

        self.host_offsets = level_offsets(input_dim, num_levels, per_level_scale, base_resolution,
                                          log2_hashmap_size, align_corners)
        self.register_buffer('offsets', torch.from_numpy(self.host_offsets.copy()))
        total_rows = int(self.host_offsets[-1])
        self.n_params = total_rows * level_dim
        self.embeddings = nn.Parameter(torch.empty(total_rows, level_dim))
        self.reset_parameters()

    def reset_parameters(self):
        self.embeddings.data.uniform_(-1e-4, 1e-4)

    def extra_repr(self):
        top = int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))
        return (f"D={self.input_dim} L={self.num_levels} C={self.level_dim} res {self.base_resolution}->{top} "
                f"scale={self.per_level_scale:.4f} table={tuple(self.embeddings.shape)} {self.gridtype}")

    def forward(self, inputs, bound=1):
        unit = (inputs + bound) / (2 * bound)
        lead = unit.shape[:-1]
        flat = unit.reshape(-1, self.input_dim)
        y = grid_encode(flat, self.embeddings, self.offsets, self.per_level_scale, self.base_resolution,
                        flat.requires_grad, self.gridtype_id, self.align_corners)
        return y.reshape(*lead, self.output_dim)
