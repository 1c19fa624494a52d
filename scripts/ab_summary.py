"""Medians over the repetitions of scripts/ab_bench.sh: per (variant, scene) throughput and stage times."""
import glob, json, os, re, collections
import numpy as np
rows = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/ab/*_*_*.json")):
    m = re.match(r"(.+)_([a-z0-9-]+)_(\d+)\.json$", os.path.basename(f))
    try:
        j = json.load(open(f))
    except Exception:
        continue
    rows[(m.group(2), m.group(1))].append(j)
print("medians:")
for (sc, name), js in sorted(rows.items()):
    k = lambda key: float(np.median([j["kernel_ms_per_step"].get(key, 0.0) for j in js]))
    v = [j["value"] for j in js]
    print(f"{name:>14s} {sc:8s} n={len(js)} {np.median(v):8.1f} Msamples/s [{min(v):.1f} .. {max(v):.1f}] {np.median([j['ms_per_step'] for j in js]):9.1f} ms | closest {k('trace_closest'):7.1f} any {k('trace_any'):6.1f} "
          f"mis {k('trace_mis'):6.1f} shade {k('shade'):7.1f} resolve {k('resolve'):5.1f} raygen {k('raygen'):5.1f} sampler {k('sampler'):6.1f}", flush=True)
