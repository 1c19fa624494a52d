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
            band = shard_rows(y1 - y0, world)  # RT_SHARD_ROWS: 4-row bands on a sharded frame
            assert band == (4 if world > 1 else 16)
            for r in range(world):
                assert np.all(((rows[r] - y0) // band) % world == r)
    # what thin bands are for: 1080 rows on 8 devices - 68 tile rows of 16 would be 9 for four devices and 8 for the others (1.059 of the mean); 270 bands of 4 rows are 34 at most
    n = [len(owned_sample_rows(0, 1080, r, 8)) for r in range(8)]
    assert shard_rows(1080, 8) == 4 and max(n) / (1080 / 8) < 1.01
    assert shard_rows(1024, 8) == 4 and len({len(owned_sample_rows(0, 1024, r, 8)) for r in range(8)}) == 1
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


CASES = {
    # name: (cropped x0 y0 x1 y1, sample bounds x0 y0 x1 y1, filter radius y)
    "box-70-rows": ((0, 0, 40, 70), (0, 0, 40, 70), 0.5),          # 5 tile rows: bands of 8 rows for every world that does not divide 5
    "wide-filter-cropped": ((3, 10, 43, 90), (1, 8, 45, 92), 2.0),  # Film::get_sample_bounds grows the cropped window by the radius (film.rs:249-257): rows nobody owns inside the film
    "triangle-1080": ((0, 0, 8, 1080), (0, -1, 8, 1081), 1.5),      # C5's height: 270 bands of 4 rows on 8 ranks (RT_SHARD_ROWS)
}


def _rank_film(case, rank, world, edge=True):
    """What rank `rank` renders of a synthetic frame: every sample row y it owns splats a row pattern f(y, dy) into the film rows within the filter's reach
    (FilmTile::add_sample, film.rs:303-321), plus - under the box filter - one sample exactly on its first row's upper edge, which lands in the row above."""
    from rustracer_amd.distributed import owned_sample_rows
    (cx0, cy0, cx1, cy1), (sx0, sy0, sx1, sy1), radius = CASES[case]
    h, w = cy1 - cy0, cx1 - cx0
    film = np.zeros((h, w, 4), np.float32)
    reach = int(np.ceil(radius - 0.5))
    rng = np.random.default_rng(1234)
    pattern = rng.uniform(0.1, 4.0, (sy1 - sy0, 2 * reach + 1, w, 4)).astype(np.float32)  # the same on every rank
    rows = owned_sample_rows(sy0, sy1, rank, world)
    for y in rows:
        for dy in range(-reach, reach + 1):
            r = y + dy - cy0
            if 0 <= r < h:
                film[r] += pattern[y - sy0, dy + reach]
    if edge and len(rows) and 0 <= rows[0] - 1 - cy0 < h:
        film[rows[0] - 1 - cy0] += np.float32(0.25)  # a sample exactly on a pixel edge (film.rs:313-321): one row beyond the filter's nominal reach
    return film


def _worker(rank, world, port, q, case):
    import torch
    import torch.distributed as dist
    from rustracer_amd.distributed import merge_film, touched_rows
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cropped, sb, radius = CASES[case]
        mine = _rank_film(case, rank, world)
        t = touched_rows(cropped, sb, rank, world, radius)
        untouched = np.setdiff1d(np.arange(mine.shape[0]), t)
        ok = not mine[untouched].any()  # the gather's premise: a rank writes nothing outside the rows it reports as touched
        mine = torch.from_numpy(mine)
        merge_film(mine, dst=0, cropped=cropped, sample_bounds=sb, filter_radius_y=radius)
        dist.barrier()
        if rank == 0:
            want = _rank_film(case, 0, world)
            parts = _rank_film(case, 0, world, edge=False)
            for r in range(1, world):  # rank 0 adds the others' rows in rank order: bit-equal to this sum
                want = want + _rank_film(case, r, world)
                parts = parts + _rank_film(case, r, world, edge=False)
            whole = _rank_film(case, 0, 1, edge=False)   # the frame one rank renders: every sample row exactly once, up to the order of the additions
            q.put((0, ok, bool(np.array_equal(mine.numpy().view(np.uint32), want.view(np.uint32))), bool(np.allclose(parts, whole, rtol=1e-6, atol=0))))
        else:
            q.put((rank, ok, True, True))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world, case", [(2, "box-70-rows"), (3, "box-70-rows"), (8, "box-70-rows"), (2, "wide-filter-cropped"), (3, "wide-filter-cropped"),
                                         (8, "wide-filter-cropped"), (8, "triangle-1080")])
def test_film_merge_over_gloo_gathers_the_touched_rows(world, case):
    """The N > 1 data path on CPU (gloo): worlds 2, 3 and 8 - 4-row bands (RT_SHARD_ROWS) -, the box filter, a wide
    filter on a cropped film, C5's 1080 rows on 8 ranks. Rank 0's merged film equals the rank films summed in rank order bit for bit, and the whole frame
    rendered by one rank up to the order of additions."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, case)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    results = sorted(q.get(timeout=30) for _ in range(world))
    assert [r[0] for r in results] == list(range(world))
    assert all(r[1] for r in results), "a rank wrote outside the rows it reports as touched"
    assert results[0][2], "rank 0's merged film is not the rank films summed in rank order"
    assert results[0][3], "the shards do not add up to the whole frame"
