#!/bin/bash
mkdir -p gpurun_out/s13
REPS=3 STEPS=3 timeout 900 bash scripts/ab_bench.sh mis > gpurun_out/s13/ab_mis.log 2>&1; tail -2 gpurun_out/s13/ab_mis.log
REPS=1 STEPS=2 timeout 900 bash scripts/ab_bench.sh room > gpurun_out/s13/ab_room.log 2>&1; tail -2 gpurun_out/s13/ab_room.log
cp rustracer_amd/csrc/_build/ab/b_relaxed.so rustracer_amd/csrc/_build/librtx_hip.so
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/s13/pytest_relaxed.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/s13/pytest_relaxed.log | cut -c1-200
python - <<'PY'
import sys; sys.path.insert(0,'.')
import numpy as np
from rustracer_amd import host
from oracle import orc
from rustracer_amd.scenes import mis_plates, cornell_box, room_env
for name, d in (("mis", mis_plates(320, 180, 64)), ("cornell", cornell_box(128, 128, 64)), ("room", room_env(240, 136, 32, detail=2, tex_size=128, env_size=256))):
    fo, so = orc.OracleScene(d).render(mode=1)
    fh, sh = host.HostScene(d).render()
    a, b = host.film_to_rgb(fh).astype(np.float64), orc.film_to_rgb(fo).astype(np.float64)
    print(name, "relaxed build: rel L2 vs oracle", np.linalg.norm(a - b) / np.linalg.norm(b), "weights equal", np.array_equal(fo[..., 3], fh[..., 3]),
          "rays", [int(sh[k]) - int(so[k]) for k in ("rays_closest", "rays_shadow", "rays_mis")])
PY
