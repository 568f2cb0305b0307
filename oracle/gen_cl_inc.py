#!/usr/bin/env python3
"""Build-time helper for oracle/Makefile: reads the reference's traverse.cl FROM THE REFERENCE
CHECKOUT, expands its #include lines the way tinyocl::Kernel does for AMD (tiny_ocl.h:772-805),
and writes the text as a C string literal to a generated, git-ignored header under oracle/_ref/.
No reference source is stored in this repository.   usage: gen_cl_inc.py <reference dir> <out> [root .cl file = traverse.cl]"""
import re
import sys

ref, out = sys.argv[1], sys.argv[2]
root = sys.argv[3] if len(sys.argv) > 3 else "traverse.cl"


def read(name):
    return open(f"{ref}/{name}", encoding="latin-1").read().lstrip("\ufeff\xef\xbb\xbf")


def expand(name):   # nested: wavefront.cl includes traverse.cl, which includes the per-layout files
    return re.sub(r'#include\s+"([^"]+)"', lambda m: expand(m.group(1)), read(name))


text = expand(root)
text = text.encode("ascii", "replace").decode("ascii")  # comments contain non-ASCII author names
with open(out, "w") as f:
    for line in text.splitlines():
        f.write('"' + line.replace("\\", "\\\\").replace('"', '\\"') + '\\n"\n')
