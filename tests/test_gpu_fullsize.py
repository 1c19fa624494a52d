"""BASELINE.json configs 2-5 at their REAL sizes against the oracle (VERDICT r01 "full-size configs are not parity-checked").

The things that only exist at full size - 2^28-path passes, 2^19-pixel batches with double-buffered sampler tables, the class-wise shade
dispatch over a binned queue of 10^8 entries, the dense 1282-light voxel table, 1024^2 image pyramids and the 2048 x 1024 environment map with
its 4096 x 2048 sampling distribution - are rendered by the production kernels (no counting frame) and compared with the oracle on a window
of pixels: `pixel_bounds` (rc/integrator/path.rs:53-69, checked after start_pixel in rc/renderer.rs:96-104) keeps the oracle to seconds
while every pixel's sampler tables, indices and seeds are those of the full frame. Gates as everywhere: filter weight sums exact, linear-RGB
film of the window within 1e-3 relative L2, ray counts by class within 1e-4.
"""
import copy

import numpy as np
import pytest

from util import rel_l2

pytestmark = pytest.mark.gpu
L2_GATE = 1e-3


def _windowed(d, window, spp=None):
    d = copy.copy(d)
    d.integrator = copy.copy(d.integrator)
    d.sampler = copy.copy(d.sampler)
    d.integrator.pixel_bounds = window  # x0 x1 y0 y1
    if spp is not None:
        d.sampler.spp = spp
    return d


class _Pair:
    """Oracle scene + GPU scene of one description, built once; renders swap in variants of the description that differ in sampler /
    integrator / film parameters only (both wrappers read those from `.desc` at render time)."""

    def __init__(self, gpu_host, orc, d):
        self.gpu_host, self.orc = gpu_host, orc
        self.o, self.h = orc.OracleScene(d), gpu_host.HostScene(d)

    def compare(self, d, window=None, count_gate=1e-4):
        self.o.desc = self.h.desc = d
        fo, so = self.o.render(mode=1)
        fh, sh = self.h.render()
        assert fo.shape == fh.shape
        assert np.array_equal(fo[..., 3], fh[..., 3]), "filter weight sums differ"
        ro, rh = self.orc.film_to_rgb(fo), self.gpu_host.film_to_rgb(fh)
        assert np.isfinite(rh).all()
        if window is not None:
            x0, x1, y0, y1 = window
            outside = np.ones(fo.shape[:2], bool)
            outside[max(y0 - 1, 0):y1 + 1, max(x0 - 1, 0):x1 + 1] = False  # a sample exactly on a pixel edge splats into the neighbour too (film.rs:313-321)
            assert not fh[outside].any() and not fo[outside].any(), "samples outside pixel_bounds reached the film"
            ro, rh = ro[y0:y1, x0:x1], rh[y0:y1, x0:x1]
            assert (fo[y0:y1, x0:x1, 3] > 0).all()
        err = rel_l2(rh, ro)
        assert err < L2_GATE, err
        assert int(sh["camera_rays"]) == int(so["camera_rays"])
        for k in ("rays_closest", "rays_shadow", "rays_mis"):
            assert abs(int(sh[k]) - int(so[k])) <= count_gate * int(so[k]) + 16, (k, sh[k], so[k])
        return err, sh


def test_c2_cornell_1024sq_1024spp_window(gpu_host, orc):
    from rustracer_amd.scenes import cornell_box
    w = (480, 544, 560, 624)  # the short block's top edge and its shadow
    d = cornell_box(1024, 1024, 1024)
    err, sh = _Pair(gpu_host, orc, d).compare(_windowed(d, w), w)
    assert sh["n_passes"] >= 2  # the frame really was cut into 2^19-pixel batches


def test_c3_blob_1m_triangles_1280x720_256spp_window(gpu_host, orc):
    from rustracer_amd.scenes import blob_scene
    d = blob_scene(spp=256)
    assert d.n_tris >= 1048576 and (d.film.xres, d.film.yres) == (1280, 720)
    w = (600, 664, 300, 364)
    _Pair(gpu_host, orc, d).compare(_windowed(d, w), w)


def test_c4_mis_plates_full_emitters_512spp_window_and_whole_frame(gpu_host, orc):
    from rustracer_amd.scenes import mis_plates
    d = mis_plates(spp=512)
    assert len(d.lights) >= 1282 and (d.film.xres, d.film.yres) == (1280, 720)
    w = (560, 624, 400, 464)  # highlights on the plates
    pair = _Pair(gpu_host, orc, d)
    pair.compare(_windowed(d, w), w)
    pair.compare(_windowed(d, None, spp=4))  # every pixel of the frame: the dense light table in every voxel the frame touches


def test_c5_room_env_full_maps_1920x1080(gpu_host, orc):
    from rustracer_amd.scenes import room_env
    d = room_env(spp=1024)
    assert (d.film.xres, d.film.yres) == (1920, 1080) and max(m.data.shape[1] for m in d.mipmaps) == 2048
    w = (900, 964, 600, 664)
    pair = _Pair(gpu_host, orc, d)
    pair.compare(_windowed(d, w), w)
    d2 = _windowed(d, None, spp=2)
    pair.compare(d2)
    # the 8-way film sharding of config 5: the shards' films sum to the whole frame, bit for bit under the box filter
    h = pair.h
    whole, _ = h.render()
    acc, touched = np.zeros_like(whole), np.zeros(whole.shape[:2], np.int32)
    for r in range(8):
        f, st = h.render(rank=r, world_size=8)
        touched += f[..., 3] > 0
        acc += f
    # Shards own disjoint tile rows, but a sample exactly on a pixel edge splats into both pixels (film.rs:313-321), and y + o.y rounds to y
    # for o.y < 2^-13 at y ~ 1000: about 68 boundary rows x 1920 x 2 spp x 2^-13 ~ 30 samples reach a row of the neighbouring shard.
    shared = touched > 1
    assert shared.sum() <= 200, int(shared.sum())
    assert np.array_equal(acc[~shared], whole[~shared]) and np.allclose(acc[shared], whole[shared], rtol=1e-6, atol=0)


def test_c1_cornell_400sq_64spp(gpu_host, orc):
    """BASELINE configs[0] (the reference's own CPU-runnable case) through the GPU path, whole frame."""
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(400, 400, 64)
    _Pair(gpu_host, orc, d).compare(d)


def test_c1_with_the_references_own_sampler_stream(gpu_host, orc):
    """BASELINE configs[0] - cornell 400 x 400 x 64 spp, the one configuration the reference itself runs - with the REFERENCE'S sampler stream on the device (round 6; VERDICT
    r05 missing #2): rc/renderer.rs:83-84 reseeds one PCG32 stream per 16 x 16 tile and every pixel and sample of the tile consumes it in order, so the frame loop's parity is
    defined on a pixel-keyed variant and equals what rustracer-cli writes only statistically. RT_FLAG_REF_STREAM keeps the chain - one lane per tile, 625 of them - and must
    then reproduce the oracle's SAMPLER_REF mode sample for sample: the same samples in the same pixels (weights exact), the same paths (ray counts within what the
    radiance-only reciprocals can move Russian roulette by), the film inside north_star's 1e-3 - and far inside what separates two independent estimates of the image."""
    from rustracer_amd.scenes import cornell_box
    d = cornell_box(400, 400, 64)
    fo, so = orc.OracleScene(d).render(mode=0)
    h = gpu_host.HostScene(d)
    fr, sr = h.render(ref_stream=True)
    assert np.array_equal(fo[..., 3], fr[..., 3])
    ro, rr = orc.film_to_rgb(fo), gpu_host.film_to_rgb(fr)
    assert np.isfinite(rr).all() and rel_l2(rr, ro) < 1e-3, rel_l2(rr, ro)
    assert int(sr["camera_rays"]) == int(so["camera_rays"]) == 400 * 400 * 64
    for k in ("rays_closest", "rays_shadow", "rays_mis"):
        assert abs(int(sr[k]) - int(so[k])) <= 1e-3 * int(so[k]) + 16, (k, sr[k], so[k])
    fk, _ = h.render()  # the pixel-keyed frame of the same scene: another estimate of the same image
    assert rel_l2(gpu_host.film_to_rgb(fk), ro) > 20 * max(rel_l2(rr, ro), 1e-6)

