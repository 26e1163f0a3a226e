#!/usr/bin/env python3
"""ISA audit of the built kernels (hipcc -save-temps .s files): per kernel, registers / LDS / scratch and — for kernels that stage LDS tiles
with the load unit (global_load_lds) — every COMPILER-emitted `s_waitcnt vmcnt(0)` inside a loop.  Such a wait drains the copy ring every
iteration (cdna_hip_programming.md section 5, "Three .s-level traps": a second __shared__ object, a mixed load kind, ...) and is invisible in
the source.  Usage: tools/isa_audit.py file.s [kernel-name-substring]
Build the .s with:  hipcc -O3 -std=c++20 --offload-arch=gfx950 -ffp-contract=off -fno-math-errno -Iinclude -Icleanba_amd/csrc -save-temps -c X.hip"""
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def audit(path, want=None):
    lines = open(path).read().split("\n")
    # kernel bodies: "<name>:" after ".type <name>,@function" up to s_endpgm's .Lfunc_end
    kernels = {}
    cur = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and i > 0 and any(("@function" in lines[j] and m.group(1) in lines[j]) for j in range(max(0, i - 6), i)):
            cur = m.group(1)
            kernels[cur] = [i, None]
        if cur and l.startswith(".Lfunc_end"):
            kernels[cur][1] = i
            cur = None
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", "\n".join(lines)):
        meta[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    out = []
    for name, (a, b) in kernels.items():
        if b is None or (want and want not in name and want not in demangle(name)):
            continue
        body = lines[a:b]
        nglds = sum("global_load_lds" in l or (" lds" in l and "buffer_load" in l) for l in body)
        lds = next((int(re.search(r"(\d+)", l).group(1)) for l in lines[b:b + 80] if "group_segment_fixed_size" in l), -1)
        scr = next((int(re.search(r"(\d+)", l).group(1)) for l in lines[b:b + 80] if "private_segment_fixed_size" in l), -1)
        in_asm = False
        in_loop = False
        bad = []
        for j, l in enumerate(body):
            if "#ASMSTART" in l:
                in_asm = True
            elif "#ASMEND" in l:
                in_asm = False
            if re.match(r"^\.LBB\d+_\d+:", l):
                in_loop = "in Loop" in l or "Loop Header" in l or (j + 1 < len(body) and "Loop" in body[j + 1])
            if not in_asm and in_loop and re.search(r"s_waitcnt\s+vmcnt\(0\)", l):
                bad.append(a + j + 1)
        out.append((name, nglds, lds, scr, bad))
    return out


if __name__ == "__main__":
    rows = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
    for name, nglds, lds, scr, bad in rows:
        d = demangle(name)
        d = d if len(d) < 150 else d[:147] + "..."
        flag = "  <-- compiler vmcnt(0) inside a loop of a load-unit-staged kernel at .s lines %s" % bad[:6] if (nglds and bad) else ""
        print("%-150s glds=%-3d lds=%-6d scratch=%-4d%s" % (d, nglds, lds, scr, flag))
