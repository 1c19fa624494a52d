"""Ad-hoc GPU check used during development (the real tests live in tests/)."""
import sys, time, json
import numpy as np
sys.path.insert(0, '.')
from oracle import orc
from rustracer_amd import host
from rustracer_amd.scenes import cornell_box

def rel_l2(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))

res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
d = cornell_box(res, res, spp)
o = orc.OracleScene(d); h = host.HostScene(d)
print('device', host.device_available(), host.hip_lib().rt_version())
# --- trace parity
rng = np.random.default_rng(1)
n = 200000
org = rng.uniform([0, 0, 0], [555, 548, 559], (n, 3)).astype(np.float32)
dirs = rng.normal(size=(n, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
rays = np.zeros((n, 8), np.float32); rays[:, :3] = org; rays[:, 3] = np.inf; rays[:, 4:7] = dirs
ro = o.trace(rays); rh = h.trace(rays)
print('closest prim eq', np.array_equal(ro['prim'], rh['prim']), 't eq', np.array_equal(ro['t'].view(np.uint32), rh['t'].view(np.uint32)),
      'b eq', np.array_equal(ro['b0'].view(np.uint32), rh['b0'].view(np.uint32)) and np.array_equal(ro['b1'].view(np.uint32), rh['b1'].view(np.uint32)),
      'nodes', ro['nodes'], rh['nodes'], 'tris', ro['tris'], rh['tris'])
rays[:, 3] = rng.uniform(50, 600, n).astype(np.float32)
ao = o.trace(rays, True); ah = h.trace(rays, True)
print('any eq', np.array_equal(ao['occluded'], ah['occluded']), 'nodes', ao['nodes'], ah['nodes'], 'tris', ao['tris'], ah['tris'])
# --- sampler tables
for sp in (16, 64):
    sc, pm = host.sampler_tables(sp, 4, 1000, 70)
    ok = True
    for i in (0, 1, 69):
        o1, o2, _ = orc.sampler_tables(sp, 4, 1, 1000 + i)
        # rebuild values from scramble+perm
        def brev(x):
            x = np.asarray(x, np.uint32); r = np.zeros_like(x)
            for b in range(32): r |= ((x >> np.uint32(b)) & np.uint32(1)) << np.uint32(31 - b)
            return r
        C1 = np.array([0x80000000,0xc0000000,0xa0000000,0xf0000000,0x88000000,0xcc000000,0xaa000000,0xff000000,0x80800000,0xc0c00000,0xa0a00000,0xf0f00000,0x88880000,0xcccc0000,0xaaaa0000,0xffff0000,0x80008000,0xc000c000,0xa000a000,0xf000f000,0x88008800,0xcc00cc00,0xaa00aa00,0xff00ff00,0x80808080,0xc0c0c0c0,0xa0a0a0a0,0xf0f0f0f0,0x88888888,0xcccccccc,0xaaaaaaaa,0xffffffff], np.uint32)
        def sob(k):
            g = k ^ (k >> np.uint32(1)); v = np.zeros_like(g)
            for j in range(32): v ^= np.where((g >> np.uint32(j)) & np.uint32(1), C1[j], np.uint32(0))
            return v
        def unit(v): return np.minimum(v.astype(np.float32) * np.float32(2.3283064365386963e-10), np.float32(0.99999994))
        for dd in range(4):
            k = pm[i, dd].astype(np.uint32)
            v = unit(sc[i, dd] ^ brev(k ^ (k >> np.uint32(1))))
            ok &= np.array_equal(v.view(np.uint32), o1[dd].view(np.uint32))
            k = pm[i, 4 + dd].astype(np.uint32)
            v0 = unit(sc[i, 4 + 2 * dd] ^ brev(k ^ (k >> np.uint32(1)))); v1 = unit(sc[i, 4 + 2 * dd + 1] ^ sob(k))
            ok &= np.array_equal(v0.view(np.uint32), o2[dd, :, 0].view(np.uint32)) and np.array_equal(v1.view(np.uint32), o2[dd, :, 1].view(np.uint32))
    print('sampler tables spp', sp, 'bit-exact', ok)
# --- light distribution
ldh = h.light_distribution(); ldo = o.light_distrib(max_voxels=20000)
m = ldo['func'].shape[0]
print('lightdist nvox', ldh['n_voxels'], ldo['n_voxels'], 'func eq', np.array_equal(ldh['func'][:m].view(np.uint32), ldo['func'].view(np.uint32)),
      'cdf eq', np.array_equal(ldh['cdf'][:m].view(np.uint32), ldo['cdf'].view(np.uint32)), 'maxdiff', np.abs(ldh['func'][:m] - ldo['func']).max())
# --- render parity
t = time.time(); fo, so = o.render(mode=1); to = time.time() - t
fh, sh = h.render(count_traversal=True, time_kernels=True)
ro_, rh_ = orc.film_to_rgb(fo), host.film_to_rgb(fh)
print('render rel L2 (rgb)', rel_l2(rh_, ro_), 'max abs', np.abs(rh_ - ro_).max(), 'weights eq', np.array_equal(fo[..., 3], fh[..., 3]))
print('oracle stats', {k: so[k] for k in ('camera_rays', 'rays_closest', 'rays_shadow', 'rays_mis', 'nodes_closest', 'tris_closest', 'nodes_shadow', 'tris_shadow')}, 'sec', so['seconds'])
print('gpu stats', json.dumps(sh))
fh2, sh2 = h.render()
print('deterministic', np.array_equal(fh, fh2), 'ms', sh2['ms_total'], 'Msamples/s', sh2['camera_rays'] / sh2['ms_total'] / 1e3)
