#!/bin/bash
# Final measurements of a round (GPU box, repo root): per-scene rocprofv3 stats + PMC passes -> profiles/<tag>_*, the default bench line, the two extra workloads, SQ counter sets. Usage: bash scripts/round_profiles.sh r03
TAG=${1:-r03}
mkdir -p gpurun_out/final
for sc in cornell room blob mis; do
  timeout 1500 bash scripts/profile_round.sh $TAG $sc > gpurun_out/final/profile_$sc.log 2>&1
  cp gpurun_out/${TAG}_$sc/stats.log gpurun_out/final/bench_$sc.log 2>/dev/null
done
timeout 900 python bench.py --detail gpurun_out/final/bench_default_detail.json > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err
for sc in mis-spheres instances-10k; do
  timeout 300 python bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --headline-only > gpurun_out/final/bench_$sc.json 2> gpurun_out/final/bench_$sc.err
done
timeout 600 bash scripts/pmc_scene.sh cornell 1024 ${TAG}_sq_cornell > gpurun_out/final/sq_cornell.txt 2>&1
timeout 600 bash scripts/pmc_scene.sh blob 256 ${TAG}_sq_blob > gpurun_out/final/sq_blob.txt 2>&1
timeout 900 bash scripts/pmc_scene.sh room 1024 ${TAG}_sq_room > gpurun_out/final/sq_room.txt 2>&1
mkdir -p gpurun_out/final/profiles && cp profiles/${TAG}_* profiles/pmc_*.json gpurun_out/final/profiles/ 2>/dev/null
ls gpurun_out/final gpurun_out/final/profiles
tail -c 600 gpurun_out/final/bench_default.json
