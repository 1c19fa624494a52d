"""Time of rt_scene_create (HostScene.upload) for LDS-resident scenes with and without the pruning calibration, and how many nodes their walks test. GPU box, repo root."""
import sys, time, os
sys.path.insert(0, '.')
import numpy as np
from rustracer_amd import host
from rustracer_amd.scenes import cornell_box, random_soup, mis_plates
for name, d in [('cornell', cornell_box(32, 32, 1)), ('soup-126', random_soup(126, seed=5, max_prims=4)), ('mis-spheres', mis_plates(spp=1, analytic_spheres=True))]:
    for prune in ('1', '0'):
        os.environ['RTX_LDS_PRUNE'] = prune
        h = host.HostScene(d)
        t0 = time.perf_counter(); h.upload(0); t1 = time.perf_counter()
        print(name, 'prune', prune, 'upload %.1f ms' % ((t1 - t0) * 1e3), 'nodes', h.bvh_sizes()[0], 'tested', host.lib().rtxh_scene_query(h.h, 1))
os.environ['RTX_LDS_PRUNE'] = '1'
os.environ['RTX_PRUNE_REPORT'] = '1'
h = host.HostScene(cornell_box(32, 32, 1)); h.upload(0)
d = mis_plates(spp=1); h = host.HostScene(d)
t0 = time.perf_counter(); h.upload(0); t1 = time.perf_counter()
print('mis-plates upload %.1f ms' % ((t1 - t0) * 1e3), 'nodes', h.bvh_sizes()[0], 'tested', host.lib().rtxh_scene_query(h.h, 1))
