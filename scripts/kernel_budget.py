#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in the BUILT librtx_hip.so (seconds; no recompile): the gfx950 code object is cut out of the
.hip_fatbin bundle and its metadata notes are read with llvm-readelf. Usage: scripts/kernel_budget.py [pattern]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rustracer_amd", "csrc", "_build", "librtx_hip.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernel_resources(lib=LIB):
    """{demangled kernel name: dict(vgpr, agpr, sgpr, scratch, lds, vgpr_spills)}"""
    b = open(lib, "rb").read()
    codes, i = [], b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while i >= 0:  # one bundle per translation unit (rtx_hip.hip, rtx_shade.hip)
        n = struct.unpack_from("<Q", b, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, p)
            p += 24
            triple = b[p:p + tl]
            p += tl
            if b"gfx950" in triple:
                codes.append(b[i + off:i + off + size])
        i = b.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)
    if not codes:
        raise RuntimeError("no gfx950 code object in " + lib)
    notes = ""
    for code in codes:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(code)
            f.flush()
            notes += subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out[name] = dict(vgpr=g("vgpr_count"), agpr=int(blk.split()[0]), sgpr=g("sgpr_count"), scratch=g("private_segment_fixed_size"),
                         lds=g("group_segment_fixed_size"), vgpr_spills=g("vgpr_spill_count"))
    names = list(out)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    return {d.split("(")[0].replace("void ", ""): out[n] for d, n in zip(dem, names)}


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for k, v in sorted(kernel_resources().items()):
        if pat in k:
            print(f"{k[:70]:70s} vgpr={v['vgpr']:4d} sgpr={v['sgpr']:4d} scratch={v['scratch']:5d} lds={v['lds']:6d} spills={v['vgpr_spills']}")
