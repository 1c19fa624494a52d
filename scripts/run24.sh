#!/bin/bash
mkdir -p gpurun_out/s24
timeout 1200 python scripts/shard_times.py cornell room > gpurun_out/s24/shard.log 2>&1; grep -v amdgpu.ids gpurun_out/s24/shard.log
