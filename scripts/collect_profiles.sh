#!/bin/bash
# After `bash scripts/round_profiles.sh <tag>` ran on the GPU box and gpurun merged gpurun_out/ back: copy what the judge reads into profiles/ (tracked). Usage: bash scripts/collect_profiles.sh r05
TAG=${1:-r06}
cd "$(dirname "$0")/.." || exit 1
cp gpurun_out/final/profiles/${TAG}_* gpurun_out/final/profiles/pmc_*.json profiles/ || exit 1
cp gpurun_out/final/bench_default.json profiles/${TAG}_bench_default.json
cp gpurun_out/final/bench_default_detail.json profiles/${TAG}_bench_default_detail.json
for sc in cornell room blob mis; do cp gpurun_out/${TAG}_$sc/stats_detail.json profiles/${TAG}_bench_${sc}_detail.json; done
for sc in mis-spheres instances-10k; do cp gpurun_out/final/bench_$sc.json profiles/${TAG}_bench_$sc.json; done
python - <<PY
import json, sys
sys.path.insert(0, '.')
import bench
sha = bench.source_sha()
bad = [sc for sc in ('cornell', 'blob', 'mis', 'room') if json.load(open(f'profiles/pmc_{sc}.json')).get('kernel_source_sha') != sha]
print('tree', sha, 'stale:', bad or 'none')
r = json.load(open('profiles/${TAG}_bench_default.json'))
print('S1', r['value'], r['ms_per_step'], 'frac', r['roofline']['frac'], 'cpu', r['cpu_baseline']['value'], 'x', r['speedup'], 'C1', r['config_c1'])
print({k: (v['value'], v['ms_per_step'], v['cpu']) for k, v in r['other_configs'].items()})
PY
