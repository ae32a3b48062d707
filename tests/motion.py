"""Deterministic sphere motion for the multi-rank tests: every sphere oscillates along a direction derived from a
hash of its global id.  A step adds an exactly representable f32 increment (small integer x power of two), so the
rank-local update (torch, on whatever rank owns the sphere at that moment, CPU or GPU) and the global reference
update (NumPy) produce identical bits whatever the ownership history."""
import numpy as np


def _hash64(g):
    return (g * 2654435761) & 0xFFFFFFFF


def step_increment_numpy(gids, k, unit):
    """[n, 3] float32 increments of step k for spheres with global ids `gids` (uint32)."""
    h = _hash64(np.asarray(gids).astype(np.int64))
    sign = 1 - 2 * (((k + ((h >> 20) & 3)) >> 2) & 1)
    out = np.empty((len(h), 3), np.float32)
    for a in range(3):
        c = ((h >> (5 * a)) & 7) + 1
        out[:, a] = (sign * c).astype(np.float32) * np.float32(unit)
    return out


def advance_numpy(coords4, gids, k, unit):
    coords4[:, :3] += step_increment_numpy(gids, k, unit).astype(coords4.dtype)


def advance_torch(rows, gids_i32, k, unit):
    """In place on rows[:, :3] (any float dtype, any device); gids_i32 = int32 bit patterns of the uint32 ids."""
    import torch
    h = _hash64(gids_i32.to(torch.int64) & 0xFFFFFFFF)
    sign = 1 - 2 * (((k + ((h >> 20) & 3)) >> 2) & 1)
    for a in range(3):
        c = ((h >> (5 * a)) & 7) + 1
        rows[:, a] += (sign * c).to(torch.float32).mul(float(unit)).to(rows.dtype)


def unit_for(radius):
    """Power of two such that the fastest sphere (8 units per axis per step) moves about 2-4 radii per step."""
    return float(2.0 ** np.floor(np.log2(max(float(radius), 1e-9) / 2.0)))
