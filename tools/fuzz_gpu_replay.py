#!/usr/bin/env python3
"""Replays one seed of tools/fuzz_gpu.py's batch-engine fuzz and prints the history of the first stream that differs.
   usage (GPU box): python3 -u tools/fuzz_gpu_replay.py <batch seed> [nsteps]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import orc                                                   # noqa: E402
from gpu_util import check_pcm                               # noqa: E402
from opus_amd.decoder import BatchDecoder                    # noqa: E402
from test_emu_silk import random_packet                      # noqa: E402

seed, nsteps, ns = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 60, 96
wide = seed % 1000 == 8                                    # (tools/fuzz_gpu.py: seeds ...008 are the wide runs)
rng = np.random.default_rng(seed)
dec = BatchDecoder(ns, 2, 0)
refs = [orc.Decoder(2) for _ in range(ns)]
hist = [[] for _ in range(ns)]


def toc(p):
    return "lost" if not p else f"cfg{p[0] >> 3}{'s' if p[0] & 4 else 'm'}c{p[0] & 3}/{len(p)}"


def refuse(s, packets, fec=False):
    try:
        for q in packets:
            refs[s].decode_fec(q) if fec else refs[s].decode(q)
        print("ORACLE ACCEPTS what the engine refused: stream", s, [toc(q) for q in packets])
    except ValueError:
        pass
    refs[s] = orc.Decoder(2)
    dec.flush(s)
    hist[s].append("REFUSED+flush")


for step in range(nsteps):
    ids = [s for s in range(ns) if rng.integers(0, 5) != 0]
    T = 2 if step % 7 == 3 else 1
    if wide and step % 5 != 2:
        T = int(rng.choice([1, 1, 2, 3, 4, 8]))
    if wide and step % 3 == 0:
        config, code = int(rng.integers(16, 32)), int(rng.integers(0, 4))
        pk = [[random_packet(rng, config, int(rng.integers(0, 2)), code) for _ in range(T)] for _ in ids]
        if code == 3:
            n0 = pk[0][0][1] & 63
            keep = [j for j, p in enumerate(pk) if all(q[1] & 63 == n0 for q in p)]
            ids, pk = [ids[j] for j in keep], [pk[j] for j in keep]
            if not ids:
                continue
    else:
        pk = [[b"" if rng.integers(0, 25) == 0 else random_packet(rng, int(rng.integers(0, 32)), int(rng.integers(0, 2)), int(rng.integers(0, 4)))
               for _ in range(T)] for _ in ids]
    bad = []
    if T == 1 and step % 5 == 2:
        fec_ids = [s for s, p in zip(ids, pk) if p[0]]
        fec_pk = [p[0] for p in pk if p[0]]
        errs = dec.send(fec_pk, fec_ids, fec=True)
        dec.decode()
        for s, p, e in zip(fec_ids, fec_pk, errs):
            hist[s].append(f"step {step} FEC {toc(p)}")
            if e:
                refuse(s, [p], fec=True)
            else:
                want, got = refs[s].decode_fec(p), dec.read(s)
                try:
                    check_pcm(want, got, "fec")
                except AssertionError as ex:
                    got = got[: len(want)]
                    d = np.abs(want - got).max(axis=1)
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    np.savez(os.path.join(ROOT, "gpurun_out", f"replay_{seed}.npz"), want=want, got=got)
                    print("   per 120 samples: rms of the reference | max abs diff")
                    for o in range(0, len(want), 120):
                        print(f"     {o:5d}  {np.sqrt((want[o:o + 120] ** 2).mean()):.3e}  {d[o:o + 120].max():.3e}")
                    bad.append((s, "FEC " + str(ex)[:100] + f"; {len(want)} samples, peak {np.abs(want).max():.3e}, rms {np.sqrt((want ** 2).mean()):.3e}, "
                                   f"max abs diff {d.max():.3e} at {int(d.argmax())}, differing > 1e-7: {int((d > 1e-7).sum())}"))
    if T == 1:
        errs = [[e] for e in dec.send([p[0] for p in pk], ids)]
    else:
        errs = dec.send_multi(pk, ids)
    dec.decode()
    for s, p, e in zip(ids, pk, errs):
        hist[s].append(f"step {step} " + "+".join(toc(q) for q in p))
        if any(e):
            refuse(s, p)
            continue
        want = np.concatenate([refs[s].decode(q) for q in p])
        got = dec.read(s)[: len(want)]
        try:
            check_pcm(want, got, "x")
        except AssertionError as ex:
            d = np.abs(want - got).max(axis=1)
            nz = np.nonzero(d > 1e-4 * max(1.0, np.abs(want).max()))[0]
            bad.append((s, str(ex)[:100] + f"; {len(want)} samples, peak {np.abs(want).max():.3e}, rms {np.sqrt((want ** 2).mean()):.3e}, max abs diff {d.max():.3e}"
                           + (f", differing {nz[0]}..{nz[-1]} ({len(nz)})" if len(nz) else "")))
    if bad:
        for s, msg in bad:
            print("step", step, "stream", s, msg)
            print("   history:", hist[s][-8:])
        break
else:
    print("clean")
