#!/bin/bash
# Builds librtx_hip.so variants for scripts/ab_bench.sh in parallel: scripts/build_variants.sh name1="-DX=1 -DY=2" name2="" ...  -> rustracer_amd/csrc/_build/ab/<name>.so
cd "$(dirname "$0")/../rustracer_amd/csrc" || exit 1
mkdir -p _build/ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-comment -Wno-pass-failed -mllvm -amdgpu-spill-vgpr-to-agpr=0"
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  [ "$name" = "$spec" ] && defs=""
  ( /opt/rocm/bin/hipcc $FL $defs -shared -o _build/ab/$name.so rtx_hip.hip 2>&1 | grep -i "error" ) &
done
wait
ls _build/ab
