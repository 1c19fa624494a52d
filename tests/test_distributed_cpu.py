"""world_size-2 gloo tests (CPU) of the multi-GPU path: row ownership and the end-of-frame film merge."""
import os
import socket

import numpy as np
import pytest


def test_owned_rows_partition_the_frame():
    from rustracer_amd.distributed import owned_sample_rows, owned_pixel_mask, shard_rows
    for y0, y1 in ((0, 70), (-2, 1083), (5, 21), (0, 1024), (0, 1080)):
        for world in (1, 2, 3, 4, 8):
            rows = [owned_sample_rows(y0, y1, r, world) for r in range(world)]
            allr = np.sort(np.concatenate(rows))
            assert np.array_equal(allr, np.arange(y0, y1))  # every row owned exactly once
            band = shard_rows(y1 - y0, world)  # RT_SHARD_ROWS: 16-row tile rows, or 8 rows where those do not divide over the ranks
            assert band == (8 if world > 1 and ((y1 - y0 + 15) // 16) % world else 16)
            for r in range(world):
                assert np.all(((rows[r] - y0) // band) % world == r)
    # what the rule is for: 1080 rows on 8 devices - 68 tile rows would be 9 for four devices and 8 for the others (1.059 of the mean), 135 bands of 8 rows are 17 at most
    n = [len(owned_sample_rows(0, 1080, r, 8)) for r in range(8)]
    assert shard_rows(1080, 8) == 8 and max(n) / (1080 / 8) < 1.01
    assert shard_rows(1024, 8) == 16 and len({len(owned_sample_rows(0, 1024, r, 8)) for r in range(8)}) == 1
    m = [owned_pixel_mask((0, 0, 40, 70), (0, 0, 40, 70), r, 2) for r in range(2)]
    assert np.all(m[0] ^ m[1])


def test_touched_rows_cover_the_owned_rows_and_the_filter_reach():
    from rustracer_amd.distributed import owned_sample_rows, touched_rows
    for (cy0, cy1), (sy0, sy1), radius in (((0, 70), (0, 70), 0.5), ((0, 1080), (-2, 1082), 2.0), ((10, 50), (8, 52), 1.5)):
        for world in (1, 2, 3, 8):
            seen = np.zeros(cy1 - cy0, int)
            for r in range(world):
                t = touched_rows((0, cy0, 40, cy1), (0, sy0, 40, sy1), r, world, radius)
                own = owned_sample_rows(sy0, sy1, r, world)
                own = own[(own >= cy0) & (own < cy1)] - cy0
                assert np.isin(own, t).all()                       # a rank's own rows
                reach = int(np.ceil(radius - 0.5)) + 1
                grown = np.unique(np.clip(np.add.outer(own, np.arange(-reach, reach + 1)).ravel(), 0, cy1 - cy0 - 1))
                assert np.array_equal(np.sort(t), grown)           # ... widened by the filter's reach, nothing more
                seen[t] += 1
            assert (seen >= 1).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from rustracer_amd.distributed import merge_film, owned_pixel_mask, owned_sample_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)  # same "full frame" on every rank
        full = rng.uniform(0, 4, (70, 40, 4)).astype(np.float32)
        mask = owned_pixel_mask((0, 0, 40, 70), (0, 0, 40, 70), rank, world)
        mine = np.where(mask[..., None], full, np.float32(0))
        edge = int(owned_sample_rows(0, 70, 1, world)[0]) - 1  # the row above rank 1's first band: rank 0's (70 rows on 2 ranks: 5 tile rows do not divide, bands of 8)
        if rank == 1:
            mine[edge] = np.float32(0.25)  # a sample of rank 1 that sat exactly on the edge of its first row splats into the row above, which rank 0 owns
        mine = torch.from_numpy(mine)
        merge_film(mine, dst=0)
        dist.barrier()
        if rank == 0:
            want = full.copy()
            want[edge] += np.float32(0.25)
            q.put(bool(np.array_equal(mine.numpy().view(np.uint32), want.view(np.uint32))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_film_merge_over_gloo_gathers_the_touched_rows(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True
