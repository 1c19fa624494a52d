"""Film sharding across the GPUs of one node.

The path shards by independent units: every (pixel, sample) is independent given the scene
(rc/renderer.rs:96-131), and the reference already hands 16x16 tiles to threads. Here the scene is
replicated on each rank and rank r renders the 16-pixel tile rows t with t % world == r
(interleaved for load balance). There is no collective on the data path; the only exchange is the
end-of-frame merge that replaces `Film::merge_film_tile` under its mutex (rc/film.rs:177-194): a
GATHER of tile-row buffers to rank 0 - every rank sends only the film rows it can have touched (its
tile rows widened by the filter's reach) with grouped point-to-point sends (ncclSend / ncclRecv over
RCCL / xGMI; gloo in the CPU tests), and rank 0 adds them into its film in rank order. Rows outside a
rank's bands are exactly zero in its film. With the box filter a band is the rank's own rows plus one
row either side: a sample that falls exactly on a pixel edge is splatted into both pixels
(film.rs:313-321), which is also why the rows are added rather than copied. The same scheme under
the C ABI for one process driving several GPUs: rt_multi_render (include/rtx_hip.h).
"""
from __future__ import annotations

import numpy as np

TILE = 16  # block_size passed by rc/api.rs:1009


def shard_rows(n_sample_rows: int, world: int) -> int:
    """Rows per shard band - RT_SHARD_ROWS of include/rtx_hip.h: 4 rows on a sharded frame (thin bands even out what the ranks' rows show: the slowest rank sets
    the frame time), the reference's 16-row tile rows on a single device."""
    return 4 if world > 1 else 16


def owned_sample_rows(sample_y0: int, sample_y1: int, rank: int, world: int) -> np.ndarray:
    """Sample-bounds rows (absolute y) rendered by `rank`; mirrors owned_pixel() in csrc/rtx_kernels.h."""
    rows = np.arange(sample_y0, sample_y1)
    return rows[((rows - sample_y0) // shard_rows(sample_y1 - sample_y0, world)) % world == rank]


def owned_pixel_mask(cropped, sample_bounds, rank: int, world: int) -> np.ndarray:
    """(H, W) boolean mask over the cropped pixel bounds of the pixels whose own samples `rank` renders."""
    x0, y0, x1, y1 = [int(v) for v in cropped]
    rows = owned_sample_rows(int(sample_bounds[1]), int(sample_bounds[3]), rank, world)
    m = np.zeros((y1 - y0, x1 - x0), bool)
    sel = rows[(rows >= y0) & (rows < y1)] - y0
    m[sel, :] = True
    return m


def touched_rows(cropped, sample_bounds, rank: int, world: int, filter_radius_y: float = 0.5) -> np.ndarray:
    """Film rows (relative to the cropped bounds) that `rank` can have written: its tile rows widened by the filter's reach - the bands
    rt_multi_render gathers (csrc/rtx_hip.hip)."""
    y0, y1 = int(cropped[1]), int(cropped[3])
    sy0, sy1 = int(sample_bounds[1]), int(sample_bounds[3])
    halo = int(np.ceil(filter_radius_y - 0.5)) + 1
    hit = np.zeros(y1 - y0, bool)
    band = shard_rows(sy1 - sy0, world)
    n_tile_rows = (sy1 - sy0 + band - 1) // band
    for t in range(rank, n_tile_rows, world):
        a = max(0, sy0 + band * t - halo - y0)
        b = min(y1 - y0, sy0 + min(band * t + band, sy1 - sy0) + halo - y0)
        if a < b:
            hit[a:b] = True
    return np.nonzero(hit)[0]


def merge_film(film, dst: int = 0, cropped=None, sample_bounds=None, filter_radius_y: float = 0.5):
    """End-of-frame gather to `dst` (torch.distributed must be initialised): every other rank sends the rows it can have touched, `dst` adds
    them in rank order. `film`: (H, W, 4) tensor over the cropped bounds, this rank's shard. Without `cropped` / `sample_bounds` the film's
    own extent is taken for both (an uncropped frame under the box filter)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return film
    rank, world = dist.get_rank(), dist.get_world_size()
    h = int(film.shape[0])
    cropped = (0, 0, int(film.shape[1]), h) if cropped is None else cropped
    sample_bounds = cropped if sample_bounds is None else sample_bounds
    if rank == dst:
        bufs, ops = [], []
        for r in range(world):
            if r == dst:
                continue
            rows = touched_rows(cropped, sample_bounds, r, world, filter_radius_y)
            if len(rows):
                buf = torch.empty((len(rows),) + tuple(film.shape[1:]), dtype=film.dtype, device=film.device)
                bufs.append((torch.as_tensor(rows, device=film.device), buf))
                ops.append(dist.P2POp(dist.irecv, buf, r))
        if ops:
            for req in dist.batch_isend_irecv(ops):  # one grouped launch: ncclGroupStart .. ncclRecv x (world - 1) .. ncclGroupEnd
                req.wait()
        for rows, buf in bufs:
            film.index_add_(0, rows, buf)
    else:
        rows = touched_rows(cropped, sample_bounds, rank, world, filter_radius_y)
        if len(rows):
            part = film.index_select(0, torch.as_tensor(rows, device=film.device)).contiguous()
            for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, part, dst)]):
                req.wait()
    return film
