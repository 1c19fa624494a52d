#!/bin/bash
# round-3 session 1: parity after the fixes, K0 overlap price, ablation of S4 / S3 / S1 shade
mkdir -p gpurun_out/s1
python -m pytest tests -m gpu -x -q > gpurun_out/s1/pytest.log 2>&1; tail -3 gpurun_out/s1/pytest.log
bash scripts/ab_env.sh "RTX_K0_OVERLAP=0" cornell room > gpurun_out/s1/k0.log 2>&1; cat gpurun_out/s1/k0.log
LIB=rustracer_amd/csrc/_build/librtx_hip.so
cp $LIB /tmp/orig.so; cp rustracer_amd/csrc/_build/ablate.so $LIB
python scripts/exp_ablate.py room 128 0 1 2 4 6 8 16 > gpurun_out/s1/ablate_room.log 2>&1; cat gpurun_out/s1/ablate_room.log
python scripts/exp_ablate.py mis 128 0 8 > gpurun_out/s1/ablate_mis.log 2>&1; cat gpurun_out/s1/ablate_mis.log
python scripts/exp_ablate.py cornell 256 0 > gpurun_out/s1/ablate_cornell.log 2>&1; cat gpurun_out/s1/ablate_cornell.log
cp /tmp/orig.so $LIB
