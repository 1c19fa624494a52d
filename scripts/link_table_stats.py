"""What the calibration of the LDS walks' link tables decided, per scene (host only, no device): calibration rays per set (direction octants 0 - 7, occlusion segments),
simulated node tests per ray with every node tested and with the kept ones, nodes tested per row. Usage: python scripts/link_table_stats.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rustracer_amd import host  # noqa: E402
from rustracer_amd.scenes import cornell_box, mis_plates, random_soup  # noqa: E402

SCENES = [("S1 cornell", lambda: cornell_box(8, 8, 1), False), ("mis-spheres", lambda: mis_plates(spp=1, analytic_spheres=True), False),
          ("soup-126", lambda: random_soup(126, seed=5, max_prims=4), False), ("S3 mis-plates", lambda: mis_plates(spp=1), True)]
out = {}
for name, make, mid in SCENES:
    h = host.HostScene(make())
    t0 = time.perf_counter(); lt = h.link_tables(mid); ms = (time.perf_counter() - t0) * 1e3
    nn = h.bvh_sizes()[0]
    tested = [len({int(lt["start_kept"][r])} | {int(w >> 16) for w in lt["kept"][r] if not (w >> 31)} | {int(w & 0xffff) for w in lt["kept"][r]}) - 1 for r in range(9)]
    rays = np.maximum(lt["rays"], 1)
    out[name] = {"n_nodes": nn, "calibration_ms_on_this_host": round(ms, 1), "rays_per_set": lt["rays"].astype(int).tolist(),
                 "node_tests_per_ray_all_nodes": (lt["tests_all"] / rays).round(2).tolist(), "node_tests_per_ray_kept_nodes": (lt["tests_kept"] / rays).round(2).tolist(),
                 "nodes_tested_per_row": tested, "sets": "0 - 7: closest-hit rays by direction octant; 8: occlusion segments"}
    print(name, nn, "nodes;", "closest %.2f -> %.2f, occlusion %.2f -> %.2f node tests per ray" % (
        lt["tests_all"][:8].sum() / rays[:8].sum(), lt["tests_kept"][:8].sum() / rays[:8].sum(), lt["tests_all"][8] / rays[8], lt["tests_kept"][8] / rays[8]), "tested", tested)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
