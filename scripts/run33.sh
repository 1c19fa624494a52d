#!/bin/bash
mkdir -p gpurun_out/s33
LIB=rustracer_amd/csrc/_build/librtx_hip.so
cp $LIB /tmp/orig.so; cp rustracer_amd/csrc/_build/ablate.so $LIB
timeout 900 python scripts/exp_ablate.py room 128 0 1 2 4 6 8 16 31 > gpurun_out/s33/ablate_room.log 2>&1
timeout 300 python scripts/exp_ablate.py mis 128 0 > gpurun_out/s33/ablate_mis.log 2>&1
cp /tmp/orig.so $LIB
grep -v amdgpu.ids gpurun_out/s33/ablate_room.log gpurun_out/s33/ablate_mis.log | cut -c1-330
