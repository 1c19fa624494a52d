#!/bin/bash
# Builds librtx_hip.so variants for scripts/ab_bench.sh in parallel: scripts/build_variants.sh name1="-DX=1 -DY=2" name2="" ...  -> rustracer_amd/csrc/_build/ab/<name>.so
# (the library's three translation units of each variant compile side by side)
cd "$(dirname "$0")/../rustracer_amd/csrc" || exit 1
mkdir -p _build/ab
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Wno-unused-value -Wno-comment -Wno-pass-failed -mllvm -amdgpu-spill-vgpr-to-agpr=0"
for spec in "$@"; do
  name="${spec%%=*}"; defs="${spec#*=}"
  [ "$name" = "$spec" ] && defs=""
  ( /opt/rocm/bin/hipcc $FL $defs -c -o _build/ab/${name}_hip.o rtx_hip.hip 2>&1 | grep -i "error" ) &
  ( /opt/rocm/bin/hipcc $FL $defs -c -o _build/ab/${name}_shade.o rtx_shade.hip 2>&1 | grep -i "error" ) &
  ( /opt/rocm/bin/hipcc $FL $defs -c -o _build/ab/${name}_ref.o rtx_ref.hip 2>&1 | grep -i "error" ) &
done
wait
for spec in "$@"; do
  name="${spec%%=*}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o _build/ab/$name.so _build/ab/${name}_hip.o _build/ab/${name}_shade.o _build/ab/${name}_ref.o && rm -f _build/ab/${name}_hip.o _build/ab/${name}_shade.o _build/ab/${name}_ref.o
done
ls _build/ab
