#!/usr/bin/env python3
"""Builds an experimental flavour of the library beside the product one (A/B runs in one GPU session):
   tools/build_variant.py <name> [hipcc flags for the HIP sources, e.g. -DOG_X=1 -ffp-contract=off]  -> variants/<name>.so
Run with OPUSGPU_LIB=variants/<name>.so (opus_amd/_abi.py).  The host objects are the product build's (build/obj)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opus_amd import build  # noqa: E402

name, extra = sys.argv[1], sys.argv[2:]
build.build_lib()
objdir = os.path.join(build.ROOT, "build", "obj")
outdir = os.path.join(build.ROOT, "variants")
os.makedirs(outdir, exist_ok=True)
objs = [os.path.join(objdir, s.replace("/", "_") + ".o") for s in build.HOST_SOURCES]
common = ["-O3", "-std=c++17", "-fPIC", "-fno-strict-aliasing", "-w"]
for s in build.HIP_SOURCES:
    o = os.path.join(objdir, f"variant_{name}_" + s.replace("/", "_") + ".o")
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950"] + common + ["-ffp-contract=off"] + extra + ["-c", os.path.join(build.CSRC, s), "-o", o])
    objs.append(o)
out = os.path.join(outdir, name + ".so")
subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-o", out] + objs + ["-lpthread"])
print(out)
