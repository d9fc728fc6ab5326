#!/usr/bin/env python3
"""Long fuzz run behind tests/test_emu_silk.py::test_random_packets_every_toc: random packets under every TOC, lost packets and
in-band FEC recoveries strewn in, host parser + kernel bodies (g++ build) against the oracle, six seeds x two channel counts x
1500 packets per call.   usage: tools/fuzz_packets.py <first seed>     (CPU only; a few minutes)"""
import sys, numpy as np, ctypes as C
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import orc
from test_emu_silk import EmuDecoder, random_packet
from gpu_util import check_pcm
from opus_amd import build
lib = C.CDLL(build.build_emu())
lib.emu_silk_state_floats.restype = C.c_size_t
lib.emu_launch_silk.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
lib.emu_launch_celt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
seed0 = int(sys.argv[1])
bad = 0
for seed in range(seed0, seed0 + 6):
    for channels in (1, 2):
        rng = np.random.default_rng(seed * 10 + channels)
        d, ref = EmuDecoder(lib, channels), orc.Decoder(channels)
        for i in range(1500):
            config, stereo, code = int(rng.integers(0, 32)), int(rng.integers(0, 2)), int(rng.integers(0, 4))
            p = random_packet(rng, config, stereo, code)
            if rng.integers(0, 25) == 0: p = b""            # a loss now and then
            if p and rng.integers(0, 12) == 0:              # the previous packet "was lost": recover from this one first
                try:
                    r0, o0 = ref.decode_fec(p), d.decode(p, fec=True)
                    check_pcm(r0, o0, "fec")
                except Exception as e:
                    bad += 1
                    print("FEC MISMATCH seed", seed, "ch", channels, "packet", i, "config", config, stereo, code, len(p), str(e)[:120], flush=True)
                    break
            try:
                r = ref.decode(p); out = d.decode(p)
                assert not isinstance(out, int), out
                check_pcm(r, out, "x")
            except Exception as e:
                bad += 1
                print("MISMATCH seed", seed, "ch", channels, "packet", i, "config", config, stereo, code, len(p), str(e)[:120], flush=True)
                break
print("done", seed0, "bad", bad, flush=True)
