#!/usr/bin/env python3
"""Static instruction mix of the kernels in the BUILT librtx_hip.so: the gfx950 code object is cut out of the fat binary, disassembled with llvm-objdump and
every kernel's instructions are counted by class (packed / plain VALU, transcendental, IEEE-division sequences, v_readlane, scalar, LDS, global / scratch
memory, branches). Static counts, not executed ones - a first look at what a VALU-issue-bound kernel is made of. Usage: scripts/isa_mix.py [pattern] [--top N]
[--dump DIR: also write each matching kernel's disassembly to DIR/<name>.s]"""
import collections
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rustracer_amd", "csrc", "_build", "librtx_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def code_object(lib=LIB):
    b = open(lib, "rb").read()
    i = b.find(b"__CLANG_OFFLOAD_BUNDLE__")
    n = struct.unpack_from("<Q", b, i + 24)[0]
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", b, p)
        p += 24
        triple = b[p:p + tl]
        p += tl
        if b"gfx950" in triple:
            return b[i + off:i + off + size]
    raise RuntimeError("no gfx950 code object")


def classify(op):
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "v_readlane"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")):
        return "valu_trans"
    if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
        return "valu_ieee_div"
    if op.startswith(("v_fma", "v_mad", "v_mac")):
        return "valu_fma"
    if op.startswith("v_cndmask"):
        return "valu_select"
    if op.startswith("v_cmp"):
        return "valu_cmp"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "valu_mov"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith(("s_cbranch", "s_branch", "s_call", "s_setpc", "s_swappc")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_load", "s_buffer_load")):
        return "s_load"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "vmem"
    return "other"


def kernels(lib=LIB):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object(lib))
        f.flush()
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        m = re.match(r"^\s+(\w+)\b(.*?)(?://.*)?$", line)
        if m and cur is not None:
            out[cur].append((m.group(1), m.group(2).strip()))
    names = [k for k in out if not k.endswith(".kd")]
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
    return {d.split("(")[0].replace("void ", ""): out[n] for d, n in zip(dem, names)}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    pat = args[0] if args else ""
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    for name, ins in sorted(kernels().items()):
        if pat not in name or not ins:
            continue
        c = collections.Counter(classify(op) for op, _ in ins)
        valu = sum(v for k, v in c.items() if k.startswith("valu") or k == "v_readlane")
        print(f"{name[:90]}\n    {len(ins)} instructions, {valu} VALU: " + ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        ops = collections.Counter(op for op, _ in ins if op.startswith("v_"))
        print("    top VALU ops: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))
        if dump:
            os.makedirs(dump, exist_ok=True)
            with open(os.path.join(dump, re.sub(r"[^\w]+", "_", name)[:80] + ".s"), "w") as f:
                f.write("\n".join(f"{op} {rest}" for op, rest in ins))
