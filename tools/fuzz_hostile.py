#!/usr/bin/env python3
"""Packets nobody would send -- random lengths (0 .. 1600), random TOCs, runs of 0x00 / 0xFF -- through the host parser + kernel
bodies (g++ build) and through the oracle: both must refuse the same packets and agree on the PCM of the rest; after a refusal
both start afresh.  (tools/parse_fuzz_asan.cpp runs the same kind of input through the host parser under ASan / UBSan.)
   usage: tools/fuzz_hostile.py <seed> <packets per channel count>      (CPU only)"""
import sys, numpy as np, ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import orc
from test_emu_silk import EmuDecoder
from gpu_util import check_pcm, RecentPeak
from opus_amd import build
lib = C.CDLL(build.build_emu())
lib.emu_silk_state_floats.restype = C.c_size_t
lib.emu_launch_silk.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
lib.emu_launch_celt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
seed, n = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
acc = ref_ = bad = 0
for ch in (1, 2):
    d, ref, recent = EmuDecoder(lib, ch), orc.Decoder(ch), RecentPeak()
    for i in range(n):
        kind = int(rng.integers(0, 16))
        ln = 0 if kind == 0 else 1 + int(rng.integers(0, 64)) if kind < 10 else 1 + int(rng.integers(0, 400)) if kind < 15 else 1 + int(rng.integers(0, 1600))
        fill = int(rng.integers(0, 8))
        p = bytearray(rng.integers(0, 256, ln, dtype=np.uint8).tobytes() if fill > 1 else bytes([0x00 if fill == 0 else 0xFF]) * ln)
        if ln and rng.integers(0, 4): p[0] = int(rng.integers(0, 256))
        p = bytes(p)
        try:
            r = ref.decode(p)
        except ValueError:
            r = None
        out = d.decode(p)
        if isinstance(out, int) or r is None:
            if not (isinstance(out, int) and r is None):
                bad += 1
                print("DISAGREE seed", seed, "ch", ch, "packet", i, "len", ln, "product", out if isinstance(out, int) else "ok", "oracle", "refused" if r is None else "ok", p[:6].hex(), flush=True)
            ref_ += 1
            d, ref, recent = EmuDecoder(lib, ch), orc.Decoder(ch), RecentPeak()
            continue
        acc += 1
        try:
            check_pcm(r, out, f"seed {seed} ch {ch} packet {i} toc {p[:1].hex()} len {ln}", recent.value())
        except AssertionError as e:
            bad += 1
            print("MISMATCH", str(e)[:200], flush=True)
            d, ref, recent = EmuDecoder(lib, ch), orc.Decoder(ch), RecentPeak()
            continue
        recent.push(r)
print("hostile: accepted", acc, "refused", ref_, "bad", bad)
sys.exit(1 if bad else 0)
