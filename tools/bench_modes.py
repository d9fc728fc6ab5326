#!/usr/bin/env python3
"""End-to-end throughput of the batch engine (layer 2 of the C ABI) on the other BASELINE / SURVEY.md 8(d) workloads:

  c2   synthetic CELT FB stereo 20 ms, 4096 streams (bench.py's workload, here through packets -> parse -> H2D -> kernels)
  c3   SILK WB mono 20 ms: the config-9 mono packets of vectors 04 and 12 looped, 16384 streams, per-stream offset
  c4   hybrid / mixed CELT: packets of vectors 05, 06, 07, 10 round-robin, 8192 streams
  c5   all 12 vectors, N replicas each (streams = 12 * N), every stream replays its vector from packet 0

Per step every stream gets one packet.  Reported per workload: audio-s/s end to end (host parse + H2D + kernels), the
same with the PCM copied back to pinned host memory (D2H), the same with step k + 1 parsed while the kernels of step k run
(no D2H), and the split of a step into host parse and device time.
One JSON line per workload.  Packets come from tests/golden/opus_vectors.bin (the committed demux of the RFC vectors).
"""
import argparse
import ctypes as C
import json
import os
import struct
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FRAME_SAMPLES = {0: 480, 1: 960, 2: 1920, 3: 2880}          # SILK configs: 10/20/40/60 ms at 48 kHz


def load_vectors():
    b = open(os.path.join(ROOT, "tests", "golden", "opus_vectors.bin"), "rb").read()
    assert b[:8] == b"OPVEC1\0\0"
    (nvec,) = struct.unpack_from("<I", b, 8)
    p, out = 12, {}
    for _ in range(nvec):
        vid, el = struct.unpack_from("<II", b, p); p += 8
        p += el
        (npk,) = struct.unpack_from("<I", b, p); p += 4
        pk = []
        for _ in range(npk):
            (l,) = struct.unpack_from("<H", b, p); p += 2
            pk.append(b[p:p + l]); p += l
        out[vid] = pk
    return out


def packet_samples(pkt):
    """48 kHz samples per channel a packet decodes to (RFC 6716 3.1 / packet.rs:288-311)."""
    toc = pkt[0]
    cfg, code = toc >> 3, toc & 3
    if cfg < 12:
        per = FRAME_SAMPLES[cfg & 3]
    elif cfg < 16:
        per = 480 << (cfg & 1)
    else:
        per = 120 << (cfg & 3)
    n = 1 if code == 0 else 2 if code < 3 else (pkt[1] & 63)
    return per * n


def workload(name, vec, streams, nsteps):
    """-> list over steps of list over streams of packet bytes"""
    if name == "c2":
        import numpy as np
        from opus_amd import synth
        pk = synth.celt_fb_stereo_packets(np.arange(streams, dtype=np.uint64), nsteps, 160)
        return [[pk[s, k].tobytes() for s in range(streams)] for k in range(nsteps)]
    if name == "c3":
        pool = [p for v in (4, 12) for p in vec[v] if (p[0] >> 3) == 9 and not (p[0] & 4) and (p[0] & 3) == 0]
    elif name == "c4":
        pool = [p for v in (5, 6, 7, 10) for p in vec[v]]
    elif name == "c5":
        ids = sorted(vec)
        return [[vec[ids[s % 12]][k % len(vec[ids[s % 12]])] for s in range(streams)] for k in range(nsteps)]
    else:
        raise SystemExit(f"unknown workload {name}")
    assert pool, name
    # stream s replays the pool from its own offset: consecutive packets of one recording stay consecutive
    return [[pool[(s * 7919 + k) % len(pool)] for s in range(streams)] for k in range(nsteps)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["c2", "c3", "c4", "c5"])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=0, help="override the workload's stream count")
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()

    import torch
    from opus_amd import _abi as A
    lib = A.load()
    torch.cuda.set_device(a.device)
    hip = C.CDLL("libamdhip64.so.7")          # the runtime torch already loaded
    hip.hipMemcpy2D.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int]
    vec = load_vectors()
    default_streams = {"c2": 4096, "c3": 16384, "c4": 8192, "c5": 12 * 512}

    for name in a.workloads:
        S = a.streams or default_streams[name]
        nsteps = a.steps + a.warmup
        steps = workload(name, vec, S, nsteps)
        # ctypes argument arrays are built outside the timed region: the C ABI is what is measured, not Python
        prepared = []
        for pk in steps:
            arr = (C.c_char_p * S)(*pk)
            lens = (C.c_uint32 * S)(*[len(p) for p in pk])
            prepared.append((arr, lens, sum(packet_samples(p) for p in pk)))
        h = C.c_void_p()
        if lib.opusgpu_batch_create(C.byref(h), S, 2, a.device) < 0:
            raise SystemExit(lib.opusgpu_last_error().decode())
        errs = (C.c_int32 * S)()
        dptr, stride, nsp = C.c_void_p(), C.c_size_t(), C.POINTER(C.c_uint32)()
        host = torch.empty(S * 5760 * 2, dtype=torch.float32).pin_memory()
        res = {}
        for d2h in (False, True):
            t_parse = t_dev = t_copy = 0.0
            samples = nerr = 0
            for k, (arr, lens, smp) in enumerate(prepared):
                t0 = time.perf_counter()
                rc = lib.opusgpu_batch_send(h, arr, lens, None, S, errs)
                t1 = time.perf_counter()
                rc |= lib.opusgpu_batch_decode(h)
                rc |= lib.opusgpu_batch_sync(h)
                t2 = time.perf_counter()
                if rc < 0:
                    raise SystemExit(lib.opusgpu_last_error().decode())
                if d2h:
                    lib.opusgpu_batch_pcm(h, C.byref(dptr), C.byref(stride), C.byref(nsp))
                    width = max(nsp[s] for s in range(0, S, max(1, S // 64))) * 2 * 4        # widest row of a sample of streams
                    width = min(max(width, 960 * 2 * 4), stride.value * 4)
                    if hip.hipMemcpy2D(host.data_ptr(), width, dptr, stride.value * 4, width, S, 2):
                        raise SystemExit("hipMemcpy2D failed")
                t3 = time.perf_counter()
                if k >= a.warmup:
                    t_parse += t1 - t0; t_dev += t2 - t1; t_copy += t3 - t2
                    samples += smp
                    nerr += sum(1 for e in errs if e)
            res[d2h] = (samples / 48000.0, t_parse, t_dev, t_copy, nerr)
        # the same steps with the host ahead of the device: batch_decode is asynchronous and the next batch_send only waits for
        # the previous step's H2D copies, so step k + 1 is parsed while the kernels of step k run; one sync at the end
        t_over = au_over = 0.0
        for k, (arr, lens, smp) in enumerate(prepared):
            if k == a.warmup:
                lib.opusgpu_batch_sync(h)
                t0 = time.perf_counter()
            rc = lib.opusgpu_batch_send(h, arr, lens, None, S, errs)
            rc |= lib.opusgpu_batch_decode(h)
            if rc < 0:
                raise SystemExit(lib.opusgpu_last_error().decode())
            if k >= a.warmup:
                au_over += smp / 48000.0
        lib.opusgpu_batch_sync(h)
        t_over = time.perf_counter() - t0
        lib.opusgpu_batch_destroy(h)
        au, tp, td, _, nerr = res[False]
        au2, tp2, td2, tc2, _ = res[True]
        print(json.dumps({
            "workload": name, "streams": S, "steps": a.steps, "warmup": a.warmup,
            "audio_s_per_s_end_to_end": round(au / (tp + td), 1),
            "audio_s_per_s_with_d2h": round(au2 / (tp2 + td2 + tc2), 1),
            "audio_s_per_s_host_ahead_of_device": round(au_over / t_over, 1),
            "ms_per_step": {"host_parse": round(tp / a.steps * 1e3, 3), "h2d_and_kernels": round(td / a.steps * 1e3, 3),
                            "d2h": round(tc2 / a.steps * 1e3, 3)},
            "audio_s_per_step": round(au / a.steps, 2), "packet_errors": nerr,
            "parser_threads": int(os.environ.get("OPUSGPU_THREADS", "0")) or "default (min(16, cores))",
        }), flush=True)


if __name__ == "__main__":
    main()
