"""Film sharding across the GPUs of one node.

The path shards by independent units: every (pixel, sample) is independent given the scene
(rc/renderer.rs:96-131), and the reference already hands 16x16 tiles to threads. Here the scene is
replicated on each rank and rank r renders the 16-pixel tile rows t with t % world == r
(interleaved for load balance). There is no collective on the data path; the only exchange is the
end-of-frame merge that replaces `Film::merge_film_tile` under its mutex (rc/film.rs:177-194): a
sum-reduce of the XYZW film to rank 0 over RCCL/xGMI (gloo in the CPU tests). Rows a rank does not
own are exactly zero in its film, so with a filter radius <= 0.5 the sum is a bit-exact gather;
wider filters splat across row boundaries and genuinely need the sum.
"""
from __future__ import annotations

import numpy as np

TILE = 16  # block_size passed by rc/api.rs:1009


def owned_sample_rows(sample_y0: int, sample_y1: int, rank: int, world: int) -> np.ndarray:
    """Sample-bounds rows (absolute y) rendered by `rank`; mirrors owned_pixel() in csrc/rtx_kernels.h."""
    rows = np.arange(sample_y0, sample_y1)
    return rows[((rows - sample_y0) // TILE) % world == rank]


def owned_pixel_mask(cropped, sample_bounds, rank: int, world: int) -> np.ndarray:
    """(H, W) boolean mask over the cropped pixel bounds of the pixels whose own samples `rank` renders."""
    x0, y0, x1, y1 = [int(v) for v in cropped]
    rows = owned_sample_rows(int(sample_bounds[1]), int(sample_bounds[3]), rank, world)
    m = np.zeros((y1 - y0, x1 - x0), bool)
    sel = rows[(rows >= y0) & (rows < y1)] - y0
    m[sel, :] = True
    return m


def merge_film(film, dst: int = 0):
    """Sum-reduce the per-rank film tensors to `dst` (torch.distributed must be initialised)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
