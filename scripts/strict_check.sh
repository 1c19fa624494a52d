#!/bin/bash
# Usage (GPU box, repo root, after `make -C rustracer_amd/csrc STRICT=1 OUT=_build/strict`): bash scripts/strict_check.sh
# The STRICT build (correctly rounded quotients in the radiance-only arithmetic too, rtx_dev_math.h) through the parity tests on hardware: swaps the
# strict libraries in for the product ones, runs the bit-exact and image-gate tests, swaps back.
B=rustracer_amd/csrc/_build
mkdir -p /tmp/prod_libs && cp $B/librtx_hip.so $B/librtx_host.so $B/source.sha /tmp/prod_libs/
cp $B/strict/librtx_hip.so $B/strict/librtx_host.so $B/
python -m pytest tests/test_gpu_parity.py tests/test_gpu_scenes.py tests/test_gpu_materials.py tests/test_gpu_sphere.py tests/test_gpu_instances.py -q -m gpu -k "not ten_times_inside" 2>&1 | tail -4
cp /tmp/prod_libs/librtx_hip.so /tmp/prod_libs/librtx_host.so /tmp/prod_libs/source.sha $B/
