#!/bin/bash
# Usage (GPU box, repo root, after `make -C rustracer_amd/csrc ABLATE=1 OUT=_build/abl`): bash scripts/pmc_ablate.sh <scene> <spp> [dbg bits ...]
# What each part of the shade kernels costs in L2-MISS READ REQUESTS: the measurement build (RTX_DBG switches parts off) under rocprofv3 --pmc FETCH_SIZE, one pass per
# setting; prints per shade kernel the counter's bytes (64 per request) per vertex of its front-end. Counters only - never combined with tracing.
SC=${1:-room}; SPP=${2:-64}; shift 2
BITS=${@:-0 1 2 4 8 32}
B=rustracer_amd/csrc/_build
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp && cd $ROOT
mkdir -p /tmp/prod_libs gpurun_out/pmc_ablate && cp $B/librtx_hip.so $B/librtx_host.so /tmp/prod_libs/
cp $B/abl/librtx_hip.so $B/abl/librtx_host.so $B/
for d in $BITS; do
  export RTX_DBG=$d
  rm -rf gpurun_out/pmc_ablate/d$d
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_ablate/d$d -- python3 scripts/exp_ablate.py $SC $SPP child > gpurun_out/pmc_ablate/d$d.log 2>&1
  python3 scripts/pmc_ablate_summary.py gpurun_out/pmc_ablate/d$d gpurun_out/pmc_ablate/d$d.log $d
done
unset RTX_DBG
cp /tmp/prod_libs/librtx_hip.so /tmp/prod_libs/librtx_host.so $B/
