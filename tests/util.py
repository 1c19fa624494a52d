import numpy as np


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def random_rays(n, lo, hi, seed, tmax=None):
    rng = np.random.default_rng(seed)
    org = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 8), np.float32)
    rays[:, :3] = org
    rays[:, 3] = np.inf if tmax is None else rng.uniform(tmax[0], tmax[1], n).astype(np.float32)
    rays[:, 4:7] = d
    return rays


# closed form of the (0,2)-sequence value at index k (rc/sampler/lowdiscrepancy.rs:96-112)
_C1 = np.array([0x80000000, 0xc0000000, 0xa0000000, 0xf0000000, 0x88000000, 0xcc000000, 0xaa000000, 0xff000000, 0x80800000, 0xc0c00000,
                0xa0a00000, 0xf0f00000, 0x88880000, 0xcccc0000, 0xaaaa0000, 0xffff0000, 0x80008000, 0xc000c000, 0xa000a000, 0xf000f000,
                0x88008800, 0xcc00cc00, 0xaa00aa00, 0xff00ff00, 0x80808080, 0xc0c0c0c0, 0xa0a0a0a0, 0xf0f0f0f0, 0x88888888, 0xcccccccc,
                0xaaaaaaaa, 0xffffffff], np.uint32)


def _brev(x):
    x = np.asarray(x, np.uint32)
    r = np.zeros_like(x)
    for b in range(32):
        r |= ((x >> np.uint32(b)) & np.uint32(1)) << np.uint32(31 - b)
    return r


def _sobol1(k):
    g = k ^ (k >> np.uint32(1))
    v = np.zeros_like(g)
    for j in range(32):
        v ^= np.where((g >> np.uint32(j)) & np.uint32(1), _C1[j], np.uint32(0)).astype(np.uint32)
    return v


def _unit(v):
    return np.minimum(v.astype(np.float32) * np.float32(2.3283064365386963e-10), np.float32(0.99999994))


def tables_from_perm(scr, perm, dims):
    """(scrambles (3*dims,), perms (2*dims, spp)) -> (t1d (dims, spp), t2d (dims, spp, 2)) float32."""
    spp = perm.shape[1]
    t1 = np.zeros((dims, spp), np.float32)
    t2 = np.zeros((dims, spp, 2), np.float32)
    for d in range(dims):
        k = perm[d].astype(np.uint32)
        t1[d] = _unit(scr[d] ^ _brev(k ^ (k >> np.uint32(1))))
        k = perm[dims + d].astype(np.uint32)
        t2[d, :, 0] = _unit(scr[dims + 2 * d] ^ _brev(k ^ (k >> np.uint32(1))))
        t2[d, :, 1] = _unit(scr[dims + 2 * d + 1] ^ _sobol1(k))
    return t1, t2
