#!/usr/bin/env python3
"""VGPR / SGPR / LDS / scratch per kernel from the device assembly (hipcc -S). Usage: scripts/kernel_resources.py [asm]"""
import re, subprocess, sys, os
asm = sys.argv[1] if len(sys.argv) > 1 else "/tmp/rtx.s"
if not os.path.exists(asm) or len(sys.argv) <= 1:
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rustracer_amd", "csrc")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-Wno-comment", "-Wno-unused-value",
                    "-S", "--cuda-device-only", "-o", asm, os.path.join(root, "rtx_hip.hip")], check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    def g(k):
        r = re.search(k + r'\s+(\d+)', body)
        return r.group(1) if r else '?'
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0].replace('void ', '')
    print(f"{dn[:64]:64s} vgpr={g('.amdhsa_next_free_vgpr'):>4s} agpr={g('.amdhsa_accum_offset'):>4s} sgpr={g('.amdhsa_next_free_sgpr'):>4s} lds={g('.amdhsa_group_segment_fixed_size'):>6s} scratch={g('.amdhsa_private_segment_fixed_size'):>5s}")
