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
# (the SQ counter summaries are part of profile_round.sh since round 5: profiles/<tag>_<scene>_pmc_sq.txt carries the same kernel_source_sha as the traffic files)
# the 8-way static split of C5 and of the headline replayed on this one GPU: per-rank ms / mean (what the 8-GPU target will be measured on)
timeout 600 python scripts/shard_replay.py room 8 profiles/${TAG}_shard_replay_room_8.json > gpurun_out/final/shard_room.log 2>&1
timeout 300 python scripts/shard_replay.py cornell 8 profiles/${TAG}_shard_replay_cornell_8.json > gpurun_out/final/shard_cornell.log 2>&1
# film parity of the production kernels against the oracle on medium-size versions of the six workloads, and the STRICT build through the parity tests
timeout 900 python scripts/parity_report.py $TAG > gpurun_out/final/parity.log 2>&1; cp gpurun_out/${TAG}_parity.json profiles/${TAG}_parity.json 2>/dev/null
[ -f rustracer_amd/csrc/_build/strict/librtx_hip.so ] && timeout 900 bash scripts/strict_check.sh > profiles/${TAG}_strict_build_tests.txt 2>&1
mkdir -p gpurun_out/final/profiles && cp profiles/${TAG}_* profiles/pmc_*.json gpurun_out/final/profiles/ 2>/dev/null
ls gpurun_out/final gpurun_out/final/profiles
tail -c 600 gpurun_out/final/bench_default.json
