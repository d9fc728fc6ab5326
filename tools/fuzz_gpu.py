#!/usr/bin/env python3
"""Long fuzz run on the GPU behind tests/test_gpu_opus.py's two fuzz tests: random packets under every TOC with losses, FEC
recoveries and sends of 1-8 packets per stream (uniform CELT-only sends among them), through the one-stream C ABI (plain and as random multistream decoders) and through the batch engine, every PCM block against the oracle.
   usage (GPU box): python3 -u tools/fuzz_gpu.py <first seed> <seconds>       prints a line per seed; exit status 1 on a mismatch"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import test_gpu_opus as T                                   # noqa: E402
import test_multistream as M                                # noqa: E402

seed, budget = int(sys.argv[1]), float(sys.argv[2])
t0, packets, sends, bad = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    try:
        for ch in (1, 2):
            T.fuzz_one_stream(ch, 1000 * seed + ch, 1500)
            packets += 1500
        T.fuzz_batch_engine(1000 * seed + 7, 60)
        T.fuzz_batch_engine(1000 * seed + 8, 60, wide=True)
        T.fuzz_batch_engine(1000 * seed + 9, 60, wide=bool(seed & 1), channels=1)
        sends += 180
        M.fuzz_multistream(1000 * seed + 11, 300)
        packets += 300
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, str(e)[:300], flush=True)
    print(f"seed {seed} done: {packets} one-stream packets, {sends} batch sends of <= 96 streams so far, {bad} mismatches, "
          f"{time.time() - t0:.0f} s", flush=True)
    seed += 1
print("fuzz_gpu: bad", bad, flush=True)
sys.exit(1 if bad else 0)
