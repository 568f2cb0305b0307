#!/usr/bin/env python
"""Registers / LDS / scratch / occupancy of every kernel in a HIP source, from the compiler's own assembly comments.
usage: tools/kernel_regs.py tinybvh_amd/csrc/kernels_cwbvh.hip [filter]"""
import os, re, subprocess, sys, tempfile
src = os.path.abspath(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = tempfile.mkdtemp(prefix="kregs_")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed", "-save-temps", "-c", src, "-o", "k.o"] + sys.argv[3:],
               cwd=d, check=True, stderr=subprocess.DEVNULL)
asm = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
txt = open(os.path.join(d, asm)).read()
name = None
rows = []
cur = {}
for line in txt.split("\n"):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name = m.group(1); cur = {}
    m = re.match(r"^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)\s*[:=] (\d+)", line)
    if m and name:
        cur[m.group(1)] = int(m.group(2))
        if m.group(1) == "Occupancy":
            rows.append((name, dict(cur)))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, c), dn in zip(rows, dem):
    dn = re.sub(r"\(.*", "", dn.replace("(anonymous namespace)::", "").replace("void ", ""))
    if flt in dn:
        print(f"{dn:90s} vgpr {c.get('NumVgprs'):3d} agpr {c.get('NumAgprs', 0):2d} scratch {c.get('ScratchSize'):3d} lds {c.get('LDSByteSize'):5d} occ {c.get('Occupancy')} code {c.get('codeLenInByte')}")
