"""The host-side tables of the stackless LDS walks (rt_link_tables, rustracer_amd/csrc/rtx_link_tables.h), checked without a device: per direction
octant (and for the occlusion walk) one word per node says where the walk goes when the node's box passes and where it carries on after the node's subtree - over all
nodes, and over the nodes a calibration on synthetic path rays found worth testing. What BVH::intersect (rc/bvh/mod.rs:366-433) computes depends only on which LEAVES
are tested, in which order: the pruned walk must reach exactly the leaves whose own box the ray passes that the full walk reaches, in the same order."""
import numpy as np
import pytest

from rustracer_amd import host
from rustracer_amd.scenes import cornell_box, mis_plates, random_soup

SCENES = {
    "cornell": (lambda: cornell_box(8, 8, 1), False),
    "soup-126": (lambda: random_soup(126, seed=5, max_prims=4), False),
    "soup-40-degenerate": (lambda: random_soup(40, seed=9, max_prims=1, degenerate=True), False),
    "mis-plates": (lambda: mis_plates(spp=1), True),     # 2461 nodes: the mid-size packing (11 bits for a leaf's first primitive)
}


def _tables(name):
    make, mid = SCENES[name]
    h = host.HostScene(make())
    return h, h.bvh(), h.link_tables(mid), mid


def _leaf_fields(word, mid):
    off_bits = 11 if mid else 7
    return int(word >> 16) & ((1 << off_bits) - 1), int(word & 0x7fffffff) >> (16 + off_bits)


def _octant_order(b, o):
    """BVH::intersect's visiting order for rays of octant o with every box passed: the child on the ray's side of the split first (bvh/mod.rs:411-417)."""
    order, st, cur = [], [], 0
    while True:
        order.append(cur)
        if b["n_prims"][cur] == 0:
            neg = (o >> min(int(b["axis"][cur]), 2)) & 1
            if neg:
                st.append(cur + 1); cur = int(b["offset"][cur])
            else:
                st.append(int(b["offset"][cur])); cur += 1
        else:
            if not st:
                return order
            cur = st.pop()


@pytest.mark.parametrize("name", list(SCENES))
def test_link_tables_are_walks_of_the_tree(name):
    h, b, lt, mid = _tables(name)
    nn = len(b["offset"])
    is_leaf = b["n_prims"] > 0
    for row in range(9):
        full, kept = lt["full"][row], lt["kept"][row]
        # every node: leaf words carry the leaf's primitive range, interior words name nodes
        for t in (full, kept):
            assert np.array_equal((t >> 31) == 1, is_leaf)
            for i in np.nonzero(is_leaf)[0][:200]:
                assert _leaf_fields(t[i], mid) == (int(b["offset"][i]), int(b["n_prims"][i]))
        # the full table with every box passed: each node once; rows 0 - 7 in BVH::intersect's order for the octant
        seq, cur = [], int(lt["start_full"][row])
        while cur < nn:
            seq.append(cur)
            cur = int(full[cur] & 0xffff) if is_leaf[cur] else int(full[cur] >> 16)
            assert len(seq) <= nn
        assert sorted(seq) == list(range(nn))
        if row < 8:
            assert seq == _octant_order(b, row)
        pos = {n: k for k, n in enumerate(seq)}
        # the kept table with every box passed: a subsequence of that order holding every leaf; with every box FAILED: still forward only, and it ends
        kseq, cur = [], int(lt["start_kept"][row])
        while cur < nn:
            kseq.append(cur)
            cur = int(kept[cur] & 0xffff) if is_leaf[cur] else int(kept[cur] >> 16)
            assert len(kseq) <= nn
        assert [pos[n] for n in kseq] == sorted(pos[n] for n in kseq)
        assert set(np.nonzero(is_leaf)[0]) <= set(kseq)
        fseq, cur = [], int(lt["start_kept"][row])
        while cur < nn:
            fseq.append(cur); cur = int(kept[cur] & 0xffff)
            assert len(fseq) <= nn
        assert [pos[n] for n in fseq] == sorted(pos[n] for n in fseq) and set(fseq) <= set(kseq)


@pytest.mark.parametrize("name", ["cornell", "soup-126", "mis-plates"])
def test_pruned_walk_reaches_the_leaves_the_full_walk_reaches(name):
    """Random finite rays, the slab test in float64 (ordered products: min / max of the two bounds per axis): the walk over the kept nodes and the walk over all nodes
    pass the same leaves' boxes in the same order, whatever t_max each leaf is tested with (here: every prefix of the sequence sees the same t_max in both)."""
    h, b, lt, mid = _tables(name)
    nn = len(b["offset"])
    lo, hi = b["bounds"][:, :3].astype(np.float64), b["bounds"][:, 3:].astype(np.float64)
    is_leaf = b["n_prims"] > 0
    rng = np.random.default_rng(7)
    ext = hi[0] - lo[0]
    n_rays = 300 if nn > 500 else 1500
    org = rng.uniform(lo[0] - 0.3 * ext, hi[0] + 0.3 * ext, (n_rays, 3))
    d = rng.normal(size=(n_rays, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    t_caps = rng.choice([np.inf, 0.5 * float(np.linalg.norm(ext)), 0.1 * float(np.linalg.norm(ext))], n_rays)
    some = 0
    for r in range(n_rays):
        inv = 1.0 / d[r]
        a0, a1 = (lo - org[r]) * inv, (hi - org[r]) * inv
        tn, tf = np.minimum(a0, a1).max(1), np.maximum(a0, a1).min(1)
        passes = (tn <= tf) & (tf > 0) & (tn < t_caps[r])
        o = int(d[r, 0] < 0) | (int(d[r, 1] < 0) << 1) | (int(d[r, 2] < 0) << 2)
        for row in (o, 8):
            out = []
            for t, start in ((lt["full"][row], lt["start_full"][row]), (lt["kept"][row], lt["start_kept"][row])):
                leaves, cur = [], int(start)
                while cur < nn:
                    w = int(t[cur])
                    if passes[cur]:
                        if is_leaf[cur]:
                            leaves.append(cur); cur = w & 0xffff
                        else:
                            cur = w >> 16
                    else:
                        cur = w & 0xffff
                out.append(leaves)
            assert out[0] == out[1], (r, row)
            some += len(out[0])
    assert some > n_rays // 5   # (leaf boxes passed in all: the comparison is not vacuous; mis-plates is an open scene)


@pytest.mark.parametrize("name", ["cornell", "mis-plates"])
def test_calibration_is_deterministic_and_never_costs_tests(name):
    h, b, lt, mid = _tables(name)
    lt2 = h.link_tables(mid)
    assert all(np.array_equal(lt[k], lt2[k]) for k in lt)
    assert (lt["rays"] >= 64).all()
    assert (lt["tests_kept"] <= lt["tests_all"]).all() and lt["tests_kept"].sum() < 0.9 * lt["tests_all"].sum()
    # S1: the pruning the headline's kernels run with (29 of 39 nodes tested, ~11 node tests per closest-hit ray instead of ~15)
    if name == "cornell":
        tested = [len({int(lt["start_kept"][row])} | {int(w >> 16) for w in lt["kept"][row] if not (w >> 31)} | {int(w & 0xffff) for w in lt["kept"][row]}) - 1 for row in range(8)]
        assert all(24 <= t <= 34 for t in tested), tested


def test_link_tables_refuse_what_is_not_lds_sized():
    h = host.HostScene(mis_plates(spp=1))
    with pytest.raises(host.BackendError):
        h.link_tables(mid=False)        # 2461 nodes are not a 256-node scene
