#!/bin/bash
# Usage (GPU box, repo root): bash scripts/tcc_scene.sh <scene> <spp>
# L2 hit rate and fabric read requests of a scene's trace kernels (is the tree resident under the streams' cache policy?): one --pmc pass
# (TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum; counters only), per kernel: hits / (hits + misses), L2-miss read requests, and requests per ray of its class.
SC=${1:-blob}; SPP=${2:-64}
OUT=gpurun_out/tcc_$SC; mkdir -p $OUT
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
export RTX_K0_OVERLAP=0
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p -- python3 bench.py --scene $SC --spp $SPP --steps 1 --warmup 0 --no-cpu-baseline --headline-only --detail $OUT/detail.json > $OUT/p.log 2>&1
python3 - "$OUT" "$SC" "$SPP" <<'PY'
import csv, glob, json, sys, collections
d, sc, spp = sys.argv[1:4]
path = glob.glob(d + '/p/*/*_counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rtx::', '')
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
det = json.load(open(d + '/detail.json'))
cls = det.get('traversal_by_ray_class', {})
print(f"{sc} at {spp} spp, one frame (+ its counting frame's launches of k_trace_big, listed apart):")
for k, v in sorted(agg.items()):
    if not k.startswith('k_trace') and not k.startswith('k_shade<'): continue
    h, m, rq = v.get('TCC_HIT_sum', 0.0), v.get('TCC_MISS_sum', 0.0), v.get('TCC_EA0_RDREQ_sum', 0.0)
    print(f"  {k[:52]:52s} L2 hit rate {h / max(h + m, 1):.3f}  L2 requests {h + m:.3e}  fabric read requests {rq:.3e}")
print("  rays per class (frame):", {k: int(v.get('rays', 0)) for k, v in cls.items()})
PY
