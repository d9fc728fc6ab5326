"""Builds opus_amd/libopusgpu.so (HIP kernels + C++ host layer + C ABI) with hipcc for gfx950."""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "opus_amd", "csrc")
OUT = os.path.join(ROOT, "opus_amd", "libopusgpu.so")
OUT_STATIC = os.path.join(ROOT, "opus_amd", "libopusgpu.a")     # the same objects archived (north_star: "gfx950 static library")

HOST_SOURCES = ["host/packet.cpp", "host/celt_parse.cpp", "host/silk_parse.cpp", "host/stream_parser.cpp",
                "host/capi_parse.cpp", "host/leaf_sort.cpp", "host/container.cpp", "host/celt_tables.cpp", "host/silk_tables.cpp", "host/engine.cpp"]
HIP_SOURCES = ["hip/celt_launch.hip", "hip/silk_launch.hip"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _all_inputs():
    deps = []
    for d in (CSRC, os.path.join(ROOT, "include")):
        for r, _, fs in os.walk(d):
            deps += [os.path.join(r, f) for f in fs if f.endswith((".h", ".hpp", ".cpp", ".hip"))]
    return deps


def build_lib(force=False, verbose=False):
    srcs = [s for s in HOST_SOURCES + HIP_SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = _all_inputs() + [os.path.abspath(__file__)]
    if not force and not _newer(OUT, deps) and not _newer(OUT_STATIC, deps):
        return OUT
    objdir = os.path.join(ROOT, "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-strict-aliasing", "-Wall", "-Wno-unused-function", "-Wno-missing-braces"]
    objs = []
    for s in srcs:
        o = os.path.join(objdir, s.replace("/", "_") + ".o")
        flags = list(common)
        cmd = [_hipcc(), "--offload-arch=gfx950"] + flags + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(o)
    subprocess.check_call([_hipcc(), "--offload-arch=gfx950", "-shared", "-o", OUT] + objs + ["-lpthread"])
    # static flavour for a Rust / C host that links the engine in (link with hipcc or add -lamdhip64 -lstdc++ -lpthread):
    # the HIP objects carry their gfx950 code object, which the HIP runtime registers at start-up
    if os.path.exists(OUT_STATIC):
        os.remove(OUT_STATIC)
    subprocess.check_call(["ar", "rcs", OUT_STATIC] + objs)
    return OUT


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def build_emu(variant="", defines=()):
    """CPU emulation of the kernel bodies: a test fixture (tests/emu), never shipped in libopusgpu.so.
    `variant` + `defines` build a second copy with other compile-time limits (tests of the overflow paths)."""
    out = os.path.join(ROOT, "tests", "emu", f"libemu{variant}.so")
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in os.listdir(os.path.join(ROOT, "tests", "emu")) if f.endswith(".cpp")]
    srcs.append(os.path.join(CSRC, "host", "celt_tables.cpp"))
    srcs.append(os.path.join(CSRC, "host", "silk_tables.cpp"))
    srcs.append(os.path.join(CSRC, "host", "leaf_sort.cpp"))
    if not _newer(out, _all_inputs() + srcs + [os.path.abspath(__file__)]):
        return out
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-strict-aliasing", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
                           "-Wno-unused-variable", "-Wno-missing-braces", "-Wno-unknown-pragmas"] + [f"-D{d}" for d in defines] + ["-o", out] + srcs)
    return out


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    build_oracle()
    print(build_emu())
