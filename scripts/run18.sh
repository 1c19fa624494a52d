#!/bin/bash
mkdir -p gpurun_out/s18
timeout 900 python scripts/exp_sort.py cornell blob room > gpurun_out/s18/sort.log 2>&1; cat gpurun_out/s18/sort.log | tail -40
