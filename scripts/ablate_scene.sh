#!/bin/bash
# Usage (GPU box, repo root, after `make -C rustracer_amd/csrc ABLATE=1 OUT=_build/abl`): bash scripts/ablate_scene.sh <scene> <spp> [dbg bits ...]
# Swaps the measurement build (section stamps + RTX_DBG switches) in for the product libraries, runs scripts/exp_ablate.py, swaps back.
B=rustracer_amd/csrc/_build
mkdir -p /tmp/prod_libs && cp $B/librtx_hip.so $B/librtx_host.so /tmp/prod_libs/
cp $B/abl/librtx_hip.so $B/abl/librtx_host.so $B/
python scripts/exp_ablate.py "$@"
cp /tmp/prod_libs/librtx_hip.so /tmp/prod_libs/librtx_host.so $B/
