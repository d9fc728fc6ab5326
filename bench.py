#!/usr/bin/env python3
"""bench.py -- decoded audio-seconds per second on BASELINE.json config 2:
synthetic CELT-only fullband 48 kHz stereo 20 ms frames, 4096 streams per MI355X (weak scaling over GPUs).

A step = one pass of the hot path over one batch: the next T consecutive 20 ms frames (default T = 8, 160 ms of
audio) of each of the 4096 streams through the C ABI (layer 3b), i.e. one opusgpu_launch_celt per step on one stream,
issued as its three parts (opusgpu_launch_celt_pvq: the K1 PVQ kernel over all 4096*T frames' leaves + the band-plan
kernel; opusgpu_launch_celt_spectrum: the stateless frame kernel, one workgroup per frame; opusgpu_launch_celt_tail: one
workgroup per stream walking its T frames in time order) so that HIP events on the launching stream bracket the dominant
kernel.  Inputs (side-info records, pre-parsed by the host layer) are resident in HBM before the timed region; PCM and
decoder state stay in HBM.  --frames-per-step 1 gives the one-frame-per-call figure.  After the timed region the last
step's PCM of a sample of streams is compared with the CPU oracle ("verified").

--gpus N > 1 without a torch.distributed environment starts N ranks itself (torch.distributed.run, one process per GPU,
before this process touches a GPU); RCCL is used only for the barrier and the max over the ranks' elapsed times.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STREAMS_PER_GPU = 4096
PAYLOAD = 160                       # bytes per 20 ms frame = 64 kb/s
FRAME_S = 0.02
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# SURVEY.md 8(d) algorithmic bytes per C2 frame: 1024 side-info + 7680 PCM + 1920 overlap r/w + ~700 energy/seed
# = 11324, plus 7712 of post-filter history when the comb filter is active in the frame
ALG_BYTES_BASE, ALG_BYTES_PF = 11324, 7712
K2_BYTES = 1724 + 7680 + 7680       # og_celt_kernel per frame: side-info + energies, K1's coefficient row in, sample row out


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames-per-step", type=int, default=8, help="consecutive frames of every stream decoded per step")
    ap.add_argument("--streams", type=int, default=STREAMS_PER_GPU, help="streams per GPU (BASELINE: 4096)")
    ap.add_argument("--no-t1", action="store_true", help="skip the one-frame-per-launch leg (reported as `t1`)")
    ap.add_argument("--pipeline-leg", action="store_true", default=True, help="(default) also time the steps pipelined over two HIP streams (reported as `pipelined`)")
    ap.add_argument("--no-pipeline-leg", dest="pipeline_leg", action="store_false")
    ap.add_argument("--no-overlap-leg", action="store_true", help="skip the two-sub-batch leg (reported as `two_sub_batches`)")
    ap.add_argument("--split", type=int, default=1,
                    help="issue each step as this many independent sub-batches of streams, each on a HIP stream of its own "
                         "(the sub-batches share nothing: their kernels may run side by side)")
    ap.add_argument("--cu-mask", default="", help="experiment: the pipelined leg's second stream (K1 + plan of the next step) on the first N compute units "
                                                  "('N'), the first stream on the others ('N:rest')")
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c2",
                    help="c2 (default, the headline): BASELINE configs[1], synthetic CELT FB stereo 20 ms, 4096 streams per GPU; "
                         "c3: BASELINE configs[2], SILK WB mono 20 ms, 16384 streams per GPU (SURVEY.md 8d); "
                         "c4: BASELINE configs[3], hybrid SWB/FB + CELT of every frame size, 8192 streams per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (packets in host memory -> PCM in HBM; reported as `end_to_end`)")
    ap.add_argument("--no-side-stream", action="store_true", help="c4: every launch of a step on one stream (default: the stateless CELT "
                                                                  "kernels on a side stream beside the SILK launches, as opusgpu_batch_decode does)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison of the last step's PCM")
    ap.add_argument("--parse-only", action="store_true",
                    help="host half only (packets -> side-info through the C ABI parser), no GPU; with --backend gloo this is "
                         "the CPU rehearsal of the N-rank path (tests/test_sharding_gloo.py)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL)")
    return ap.parse_args()


def spawn_ranks(a):
    """--gpus N > 1 outside torch.distributed.run: start N ranks (one process per GPU) and pass their exit code on.
    Runs before this process has made any GPU call; the ranks are child processes, not an exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def cpu_baseline(packets, budget_s=12.0):
    """The CPU oracle (restatement of lu-zero/opus + RFC 6716 synthesis tail; the Rust crate cannot be built here),
    one stream per thread on the host cores, on a bounded sample of the same packets."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    cores = min(16, os.cpu_count() or 1)          # the GPU box grants 16 CPUs per GPU
    nstreams, nframes = packets.shape[0], packets.shape[1]
    sample_streams = min(nstreams, cores * 16)     # about 17 s of CPU work (a frame takes the oracle 0.26 ms)
    orc.lib()

    def work(s):
        d = orc.Decoder(2)
        t_end = time.perf_counter() + budget_s
        n = 0
        for f in range(nframes):
            d.decode(packets[s, f].tobytes())
            n += 1
            if time.perf_counter() > t_end:
                break
        return n

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        done = list(ex.map(work, range(sample_streams)))
    dt = time.perf_counter() - t0
    frames = sum(done)
    out = {"value": round(frames * FRAME_S / dt, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
           "sample": f"{sample_streams} of the same synthetic streams x {max(done)} frames ({frames} frames) on the CPU oracle "
                     f"(oracle/liborc.so, scalar C, -O2), one stream per thread"}
    # BASELINE configs[0]: RFC vector 01 (CELT FB stereo) on one core.  The Rust decoder cannot decode it (no toolchain
    # here, and its CELT path stops at the first split band), so this too is the oracle.
    try:
        _, pk = orc.load_vectors(ids=[1])[1]
        d = orc.Decoder(2)
        t0 = time.perf_counter()
        samples = sum(len(d.decode(p)) for p in pk)
        dt1 = time.perf_counter() - t0
        out["c1_vector01"] = {"value": round(samples / 48000.0 / dt1, 2), "unit": "audio-s/s", "cores": 1,
                              "sample": f"testvector01, {len(pk)} packets, {samples / 48000.0:.1f} s of audio"}
    except Exception as e:                                  # the fixture is optional for the headline number
        out["c1_vector01"] = {"error": str(e)}
    return out


def parse_only(a, rank, world):
    """The N-rank path without a GPU: every rank parses its shard of the streams on the host (layer 3a) and the ranks
    meet for the barrier and the max, exactly as the GPU path does."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from gpu_util import HostParser
    from opus_amd import sharding, synth
    S, K, T = a.streams, a.steps, a.frames_per_step
    ids = sharding.shard_stream_ids(rank, world, S)
    packets = synth.celt_fb_stereo_packets(ids, K * T, PAYLOAD)
    parsers = [HostParser(2) for _ in range(S)]
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    leaves = 0
    for f in range(K * T):
        for s in range(S):
            n, nc, nl, _ = parsers[s].parse(packets[s, f].tobytes())
            assert n == 1 and nc == 1
            leaves += nl
    if world > 1:
        dist.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    tot = torch.tensor([leaves, int(ids[0]), int(ids[-1])], dtype=torch.int64)
    if world > 1:
        allv = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allv, tot)
    else:
        allv = [tot]
    if rank == 0:
        print(json.dumps({
            "metric": "host parse audio-sec/sec (CPU rehearsal of the N-rank path, no GPU)",
            "value": round(sharding.aggregate_audio_seconds_per_second(world, S, K, T * FRAME_S, elapsed), 1),
            "unit": "audio-s/s", "n_gpus": world, "steps": K, "warmup": 0, "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int", "data": "synthetic",
            "config": {"workload": "parse-only", "streams_per_gpu": S, "frames_per_stream_per_step": T,
                       "shards": [[int(v[1]), int(v[2])] for v in allv], "leaves": int(sum(int(v[0]) for v in allv)),
                       "parallelism": f"streams sharded over {world} rank(s), no collective in the data path; "
                                      f"{a.backend} only for the barrier and the max of the elapsed times"}}), flush=True)


C3_STREAMS_PER_GPU = 16384
# SURVEY.md 8(d), C3 (SILK WB mono 20 ms): side-info record 1 024 B (its stride there; the record here is 3 072 B)
# + PCM write 960 x 4 + history `output` / `lpc_history` (322 + 16) f32 each, read + write, 5 408 B + resampler state 128 B
C3_ALG_BYTES = 1024 + 3840 + 5408 + 128


def c3_packets():
    """the config-9 (SILK WB 20 ms) mono code-0 packets of RFC vectors 04 and 12, in file order"""
    import struct
    b = open(os.path.join(ROOT, "tests", "golden", "opus_vectors.bin"), "rb").read()
    assert b[:8] == b"OPVEC1\0\0"
    (nvec,) = struct.unpack_from("<I", b, 8)
    p, pool = 12, {}
    for _ in range(nvec):
        vid, el = struct.unpack_from("<II", b, p); p += 8 + el
        (npk,) = struct.unpack_from("<I", b, p); p += 4
        for _ in range(npk):
            (l,) = struct.unpack_from("<H", b, p); p += 2
            pkt = b[p:p + l]; p += l
            if vid in (4, 12) and (pkt[0] >> 3) == 9 and not (pkt[0] & 4) and (pkt[0] & 3) == 0:
                pool.setdefault(vid, []).append(pkt)
    return pool[4] + pool[12]


def main_c3(a, rank, local_rank, world):
    """--workload c3: one step = one 20 ms SILK frame of every stream = one opusgpu_launch_silk (og_silk_synth_kernel +
    og_silk_output_kernel), side-info resident in HBM, mono PCM stays in HBM.  Same contract and fields as the headline."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from opus_amd import BatchDecoder, _abi as A, sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = A.load()
    S = a.streams if a.streams != STREAMS_PER_GPU else C3_STREAMS_PER_GPU
    K, W = a.steps, a.warmup
    nsteps = K + W
    pool = c3_packets()
    ids = sharding.shard_stream_ids(rank, world, S)

    def packet(s, k):                                   # stream s replays the pool from its own offset
        return pool[(int(ids[s]) * 7919 + k) % len(pool)]

    parser = BatchDecoder(S, 1, local_rank)             # host half only: packets -> OgSilkFrame records (stream, pcm_off filled in)
    recs, t_parse = [], 0.0
    for k in range(nsteps):
        pk = [packet(s, k) for s in range(S)]
        lens = np.array([len(p) for p in pk], np.uint32)
        rows = np.zeros((S, int(lens.max())), np.uint8)  # one row per packet: no Python object per packet inside the timed call
        for s, p in enumerate(pk):
            rows[s, :len(p)] = np.frombuffer(p, np.uint8)
        t0 = time.perf_counter()
        errs = parser.send_packed(rows, lens)
        t_parse += time.perf_counter() - t0
        assert not errs.any()
        _, _, silk = parser.sideinfo()
        assert len(silk) == S
        recs.append(torch.from_numpy(np.frombuffer(silk, dtype=np.uint8).copy()).to(dev))
    parser.close()
    state = torch.zeros(lib.opusgpu_silk_state_bytes(S), dtype=torch.uint8, device=dev)
    pcm = torch.zeros(S * 960, dtype=torch.float32, device=dev)
    strm = torch.cuda.Stream(dev)
    ps = C.c_void_p(strm.cuda_stream)

    def step(k):
        rc = lib.opusgpu_launch_silk(recs[k].data_ptr(), S, state.data_ptr(), S, pcm.data_ptr(), 960, None, ps)
        if rc:
            raise RuntimeError(lib.opusgpu_last_error().decode())

    torch.cuda.synchronize(dev)
    for k in range(W):
        step(k)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(strm)
    for k in range(W, nsteps):
        step(k)
    ev1.record(strm)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    launch_s = ev0.elapsed_time(ev1) / K / 1e3          # both kernels of a launch, on the stream they run on
    assert a.no_verify or bool(torch.isfinite(pcm).all())      # (profiling runs that stop kernels early or skip leaf classes: --no-verify)
    verify = None
    if not a.no_verify:                                 # the last step's PCM of a few streams against the CPU oracle (bit-exact)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        rows = pcm.view(S, 960).cpu().numpy()
        sample = sorted({0, 1, S // 3, S // 2, 2 * S // 3, S - 1})
        ok = True
        for s in sample:
            d = orc.Decoder(1)
            for k in range(nsteps):
                want = d.decode(packet(s, k))
            ok = ok and bool(np.array_equal(want[:, 0], rows[s]))
        verify = {"ok": ok, "streams": sample, "bar": "bit-exact vs the CPU oracle (SILK keeps the reference's operation order)"}
        assert ok, "c3: PCM differs from the oracle"
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    alg = S * C3_ALG_BYTES
    c3_traffic, c3_pmc = pmc_info("c3", launch_s) if S == C3_STREAMS_PER_GPU else (None, None)      # collected at the default size only
    line = {"metric": "decoded audio-sec/sec (SILK WB mono 20 ms, 16384 streams)",
            "value": round(sharding.aggregate_audio_seconds_per_second(world, S, K, FRAME_S, elapsed), 1), "unit": "audio-s/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "RFC 6716 vectors 04 / 12 replayed",
            "verified": bool(verify and verify["ok"]), "verify": verify,
            "config": {"workload": "BASELINE configs[2] (SURVEY.md 8d C3): SILK WB mono 20 ms, the config-9 mono packets of RFC vectors "
                                   f"04 and 12 ({len(pool)} packets) looped with a per-stream offset, {S} streams per GPU, one frame per "
                                   "stream per step; device-only: side-info pre-parsed and resident in HBM, mono PCM stays in HBM",
                       "streams_per_gpu": S, "frame_ms": 20, "frames_per_stream_per_step": 1,
                       "parallelism": f"streams sharded over {world} GPU(s), one process per GPU, no collective in the data path; "
                                      "RCCL only for the barrier and the max of the elapsed times"},
            "roofline": {"bound": "hbm", "achieved": round(alg / launch_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / launch_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": c3_traffic,
                         "kernel": "one opusgpu_launch_silk call: og_silk_synth_kernel + og_silk_output_kernel",
                         "avg_launch_us": round(launch_s * 1e6, 2), "algorithmic_bytes_per_launch": alg},
            "host_parse": {"value": round(S * nsteps * FRAME_S / t_parse, 1), "unit": "audio-s/s",
                           "note": "packets -> side-info on the host threads (untimed in `value`)"}}
    if c3_pmc:
        line["pmc"] = c3_pmc
    if not a.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        cores = min(16, os.cpu_count() or 1)
        nst, nfr = cores * 4, 400

        def work(s):
            d = orc.Decoder(1)
            for k in range(nfr):
                d.decode(packet(s, k))
            return nfr
        orc.lib()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            done = sum(ex.map(work, range(nst)))
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(done * FRAME_S / dt, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
                                "sample": f"{nst} of the same streams x {nfr} frames on the CPU oracle (oracle/liborc.so, scalar C, -O2), "
                                          "one stream per thread"}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


C4_STREAMS_PER_GPU = 8192
C4_VECTORS = (5, 6, 7, 10)
C5_STREAMS_PER_GPU = 12 * 512
C5_VECTORS = tuple(range(1, 13))


def vector_packets(ids):
    """packets of the RFC 6716 vectors `ids` from tests/golden/opus_vectors.bin, in file order: {id: [bytes]}"""
    import struct
    b = open(os.path.join(ROOT, "tests", "golden", "opus_vectors.bin"), "rb").read()
    assert b[:8] == b"OPVEC1\0\0"
    (nvec,) = struct.unpack_from("<I", b, 8)
    p, pool = 12, {}
    for _ in range(nvec):
        vid, el = struct.unpack_from("<II", b, p); p += 8 + el
        (npk,) = struct.unpack_from("<I", b, p); p += 4
        for _ in range(npk):
            (l,) = struct.unpack_from("<H", b, p); p += 2
            if vid in ids:
                pool.setdefault(vid, []).append(b[p:p + l])
            p += l
    return pool


def c4_alg_bytes(celt, silk):
    """SURVEY.md 8(d) per-record algorithmic bytes for a mixed step (numpy views of the step's OgCeltHdr / OgSilkFrame records).
    CELT record of N samples per channel, CC output channels: 1024 side-info + 4 N CC PCM write (+ the same read where it adds
    onto the SILK output or cross-fades) + 1920 overlap r/w + 700 energies / seed, + 7712 post-filter history where the comb
    filter is on.  SILK frame: per coded channel 1024 side-info + 5408 history r/w; + 4 x 48 kHz samples x CC PCM write + 128
    resampler state."""
    import numpy as np
    from opus_amd import _abi as A
    total = 0
    if len(celt):
        h = np.frombuffer(celt, dtype=np.uint8).reshape(len(celt), C.sizeof(A.CeltHdr))
        lm = h[:, A.CeltHdr.lm.offset].astype(np.int64) & 3
        fl = h[:, A.CeltHdr.flags.offset].astype(np.int64)
        sp = h[:, A.CeltHdr.spread.offset].astype(np.int64)
        cc = np.where(fl & 32, 2, 1)
        n = 120 << lm
        g = h[:, A.CeltHdr.pf_gain_q.offset:A.CeltHdr.pf_gain_q.offset + 3].astype(np.int64)
        pf = (g[:, 0] != 0) | (g[:, 1] != 0) | ((lm != 0) & (g[:, 2] != 0))
        rmw = ((fl & 64) != 0) | (((sp >> 4) & 3) != 0)
        total += int((1024 + 4 * n * cc * np.where(rmw, 2, 1) + 1920 + 700 + np.where(pf, ALG_BYTES_PF, 0)).sum())
    if len(silk):
        k = np.frombuffer(silk, dtype=np.uint8).reshape(len(silk), C.sizeof(A.SilkFrame))
        fl = k[:, A.SilkFrame.flags.offset].astype(np.int64)
        bw = k[:, A.SilkFrame.bandwidth.offset].astype(np.int64)
        fs = k[:, A.SilkFrame.f_size.offset:A.SilkFrame.f_size.offset + 2].copy().view(np.uint16)[:, 0].astype(np.int64)
        nch = np.where(fl & 1, 2, 1)
        cc = np.where(fl & 2, 2, 1)
        out48 = fs * np.where(bw == 0, 6, np.where(bw == 1, 4, 3))
        total += int((nch * (1024 + 5408) + 4 * out48 * cc + 128).sum())
    return total


def main_c4(a, rank, local_rank, world):
    """--workload c4: BASELINE configs[3].  A step = one packet of every stream -- hybrid SWB / FB 10 and 20 ms
    (vectors 05, 06), CELT of every bandwidth and frame size, 2.5-20 ms, transients included (07), hybrid <-> CELT switching with
    redundant frames (10) -- as the launches opusgpu_batch_decode makes of it: the SILK frames by depth (og_silk_synth_kernel +
    og_silk_output_kernel), then every CELT record of the step in one K1 / plan / spectrum launch and one tail launch over the
    streams' runs.  Side-info and launch plans of all steps are pre-parsed and resident in HBM; PCM stays in HBM."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from opus_amd import BatchDecoder, _abi as A, parse_packet, sharding
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = A.load()
    c5 = a.workload == "c5"                             # BASELINE configs[4]: all twelve vectors, the same launches
    VECS, S_DEFAULT, wl = (C5_VECTORS, C5_STREAMS_PER_GPU, "c5") if c5 else (C4_VECTORS, C4_STREAMS_PER_GPU, "c4")
    S = a.streams if a.streams != STREAMS_PER_GPU else S_DEFAULT
    K, W = a.steps, a.warmup
    nsteps = K + W
    pools = vector_packets(VECS)
    ids = sharding.shard_stream_ids(rank, world, S)
    NV = len(VECS)

    def packet(s, k):                                   # stream s replays one of the vectors from its own offset, in file order
        g = int(ids[s])
        v = pools[VECS[g % NV]]
        return v[((g // NV) * 7919 + k) % len(v)]

    parser = BatchDecoder(S, 2, local_rank)             # host half only: packets -> side-info + launch plan
    steps, t_parse, audio_s, alg_total, rec_total = [], 0.0, 0.0, 0, [0, 0]
    max_celt, stride = 1, 5760 * 2
    for k in range(nsteps):
        pk = [packet(s, k) for s in range(S)]
        lens = np.array([len(p) for p in pk], np.uint32)
        rows = np.zeros((S, int(lens.max())), np.uint8)
        for s, p in enumerate(pk):
            rows[s, :len(p)] = np.frombuffer(p, np.uint8)
        t0 = time.perf_counter()
        errs = parser.send_packed(rows, lens)
        t_parse += time.perf_counter() - t0
        assert not errs.any()
        celt, leaves, silk = parser.sideinfo()
        pl = parser.plan()
        up = lambda x: torch.from_numpy(np.frombuffer(x, dtype=np.uint8).copy() if len(x) else np.zeros(16, np.uint8)).to(dev)
        st = dict(celt=up(celt), leaves=up(leaves), silk=up(silk), refs=torch.from_numpy(pl["refs"]).to(dev),
                  runs=torch.from_numpy(pl["run_off"].astype(np.uint32).view(np.uint8).copy()).to(dev), ncelt=len(celt), nruns=pl["nruns"],
                  run_len=pl["run_len"], classes=pl["class_counts"], depth_off=[int(x) for x in pl["silk_depth_off"]])
        steps.append(st)
        max_celt = max(max_celt, len(celt))
        if k >= W:
            _, _, nsamp = parser.device_pcm()
            audio_s += float(np.asarray(nsamp, np.float64).sum()) / 48000.0
            alg_total += c4_alg_bytes(celt, silk)
            rec_total[0] += len(celt); rec_total[1] += len(silk)
    parser.close()
    HS, SS = C.sizeof(A.SilkFrame), C.sizeof(A.CeltState)
    sstate = torch.zeros(lib.opusgpu_silk_state_bytes(S), dtype=torch.uint8, device=dev)
    cstate = torch.zeros(S * SS, dtype=torch.uint8, device=dev)
    work = torch.zeros(lib.opusgpu_celt_work_bytes(max_celt), dtype=torch.uint8, device=dev)
    pcm = torch.zeros(S * stride, dtype=torch.float32, device=dev)
    strm = torch.cuda.Stream(dev)
    ps = C.c_void_p(strm.cuda_stream)

    def check(rc):
        if rc:
            raise RuntimeError(lib.opusgpu_last_error().decode())

    side = torch.cuda.Stream(dev)
    pside = C.c_void_p(side.cuda_stream)
    fork, join = torch.cuda.Event(), torch.cuda.Event()

    def step(k):                                        # what opusgpu_batch_decode issues for the step (engine.cpp)
        st = steps[k]
        off = st["depth_off"]
        beside = st["ncelt"] and len(off) > 1 and not a.no_side_stream
        front = pside if beside else ps
        if beside:                                      # the stateless CELT kernels beside the SILK launches
            fork.record(strm)
            side.wait_event(fork)
        if st["ncelt"] and beside:
            check(lib.opusgpu_launch_celt_pvq(st["celt"].data_ptr(), st["leaves"].data_ptr(), st["ncelt"], st["refs"].data_ptr(), st["classes"],
                                              work.data_ptr(), None, front))
            check(lib.opusgpu_launch_celt_spectrum(st["celt"].data_ptr(), st["ncelt"], work.data_ptr(), None, front))
            join.record(side)
        for d in range(len(off) - 1):
            if off[d + 1] > off[d]:
                check(lib.opusgpu_launch_silk(st["silk"].data_ptr() + off[d] * HS, off[d + 1] - off[d], sstate.data_ptr(), S, pcm.data_ptr(),
                                              stride, None, ps))
        if st["ncelt"]:
            if beside:
                strm.wait_event(join)
            else:
                check(lib.opusgpu_launch_celt_pvq(st["celt"].data_ptr(), st["leaves"].data_ptr(), st["ncelt"], st["refs"].data_ptr(), st["classes"],
                                                  work.data_ptr(), None, ps))
                check(lib.opusgpu_launch_celt_spectrum(st["celt"].data_ptr(), st["ncelt"], work.data_ptr(), None, ps))
            if st["run_len"]:
                check(lib.opusgpu_launch_celt_tail(st["celt"].data_ptr(), st["ncelt"], st["run_len"], work.data_ptr(), cstate.data_ptr(),
                                                   pcm.data_ptr(), stride, None, ps))
            else:
                check(lib.opusgpu_launch_celt_tail_runs(st["celt"].data_ptr(), st["ncelt"], st["runs"].data_ptr(), st["nruns"], work.data_ptr(),
                                                        cstate.data_ptr(), pcm.data_ptr(), stride, None, ps))

    torch.cuda.synchronize(dev)
    for k in range(W):
        step(k)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(strm)
    for k in range(W, nsteps):
        step(k)
    ev1.record(strm)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0)
    step_s = ev0.elapsed_time(ev1) / K / 1e3
    verify = None
    if not a.no_verify:                                 # the last step's PCM of a few streams (one per vector and more) against the CPU oracle
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        from gpu_util import check_pcm
        rows = pcm.view(S, stride).cpu().numpy()
        sample = sorted(set(range(NV)) | {S // 3, S // 2, S // 2 + 1, 2 * S // 3 + 2, S - 1})    # every vector at least once
        worst = 0.0
        for s in sample:
            d = orc.Decoder(2)
            for k in range(nsteps):
                want = d.decode(packet(s, k))
            worst = max(worst, check_pcm(want, rows[s][: want.size].reshape(want.shape), f"{wl} stream {s}"))
        verify = {"ok": True, "streams": sample, "max_scaled_err": float(f"{worst:.3e}"),
                  "bar": "max abs err <= 1e-4 (scaled by the reference peak when that exceeds 1), SNR >= 90 dB vs the CPU oracle"}
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    total_audio = audio_s * world                       # (every rank decodes the same mix: weak scaling)
    alg = alg_total / K
    c4_traffic, c4_pmc = pmc_info(wl, step_s) if S == S_DEFAULT else (None, None)
    line = {"metric": "decoded audio-sec/sec (the 12 RFC 6716 vectors, 512 replicas each)" if c5 else
                      "decoded audio-sec/sec (hybrid SWB/FB + CELT 2.5-20 ms mixed, 8192 streams)",
            "value": round(total_audio / elapsed, 1), "unit": "audio-s/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "RFC 6716 vectors 01 .. 12 replayed" if c5 else "RFC 6716 vectors 05 / 06 / 07 / 10 replayed",
            "verified": bool(verify and verify["ok"]), "verify": verify,
            "config": {"workload": ("BASELINE configs[4] (SURVEY.md 8d C5): all twelve RFC 6716 vectors -- SILK NB / MB / WB 10-60 ms, hybrid, CELT of "
                                    "every bandwidth and frame size, mode switches with redundant frames, mono and stereo packets; stream s replays "
                                    f"vector 1 + s % 12 from its own offset, one packet per stream per step, {S} streams per GPU; device-only: "
                                    "side-info and launch plan pre-parsed and resident in HBM, PCM stays in HBM") if c5 else
                                   "BASELINE configs[3] (SURVEY.md 8d C4): hybrid SWB / FB 10 and 20 ms (RFC vectors 05, 06), CELT of every "
                                   "bandwidth and frame size incl. transients (07), hybrid <-> CELT switching with redundant frames (10); stream s "
                                   f"replays vector (05, 06, 07, 10)[s % 4] from its own offset, one packet per stream per step, {S} streams per "
                                   "GPU; device-only: side-info and launch plan pre-parsed and resident in HBM, PCM stays in HBM",
                       "streams_per_gpu": S, "mean_packet_ms": round(audio_s / K / S * 1e3, 2),
                       "celt_records_per_step": round(rec_total[0] / K, 1), "silk_frames_per_step": round(rec_total[1] / K, 1),
                       "parallelism": f"streams sharded over {world} GPU(s), one process per GPU, no collective in the data path; "
                                      f"{a.backend} only for the barrier and the max of the elapsed times"},
            "roofline": {"bound": "hbm", "achieved": round(alg / step_s / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / step_s / 1e9 / HBM_PEAK_GBS, 5), "traffic": c4_traffic,
                         "kernel": "one step: opusgpu_launch_silk per SILK depth (og_silk_synth_kernel + og_silk_output_kernel) + one "
                                   "opusgpu_launch_celt[_runs] (og_pvq_all_kernel + og_plan_kernel + og_celt_kernel + og_celt_tail_kernel)",
                         "avg_launch_us": round(step_s * 1e6, 2), "algorithmic_bytes_per_launch": int(alg)},
            "host_parse": {"value": round(audio_s / K * nsteps / t_parse, 1), "unit": "audio-s/s",
                           "note": "packets -> side-info + launch plan on the host threads (untimed in `value`)"}}
    if c4_pmc:
        line["pmc"] = c4_pmc
    if not a.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import orc
        cores = min(16, os.cpu_count() or 1)
        nst, nfr = cores * 4, 300

        def work(s):
            d = orc.Decoder(2)
            return sum(len(d.decode(packet(s, k))) for k in range(nfr))
        orc.lib()
        t0 = time.perf_counter()
        with ThreadPoolExecutor(cores) as ex:
            done = sum(ex.map(work, range(nst)))
        dt = time.perf_counter() - t0
        line["cpu_baseline"] = {"value": round(done / 48000.0 / dt, 2), "unit": "audio-s/s", "cores": cores, "kind": "port",
                                "sample": f"{nst} of the same streams x {nfr} packets on the CPU oracle (oracle/liborc.so, scalar C, -O2), "
                                          "one stream per thread"}
    print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cu_masked_stream(dev, cus):
    """a HIP stream whose kernels run on the compute units `cus` only (hipExtStreamCreateWithCUMask), as a torch stream"""
    import torch
    hip = C.CDLL("libamdhip64.so")
    words = [0] * 8
    for c in cus:
        words[c >> 5] |= 1 << (c & 31)
    h = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), C.c_uint32(len(words)), (C.c_uint32 * len(words))(*words))
    if rc:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(h.value, device=dev)


def pmc_info(workload, step_s=None):
    """profiles/pmc_traffic.json (tools/gpu_measure.sh + tools/pmc_traffic.py: rocprofv3 --pmc passes over this bench at its default
    sizes) for one workload -> (HBM bytes per step or None, the `pmc` object of the bench line or None).  Per kernel: HBM bytes,
    VALU issue fraction (4 x SQ_INSTS_VALU over the SIMD cycles of the dispatch) and which of the two binds it."""
    try:
        w = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(workload)
    except Exception:
        w = None
    if not w:
        return None, None
    ks = []
    for name, e in w["kernels"].items():
        k = {"kernel": name, "avg_us": e.get("avg_us"), "hbm_bytes": e.get("hbm_bytes"), "valu_issue_frac": e.get("valu_issue_frac")}
        if e.get("avg_us") and e.get("hbm_bytes"):
            k["hbm_frac"] = round(e["hbm_bytes"] / (e["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            if e.get("valu_issue_frac") is not None:
                k["bound"] = "valu" if e["valu_issue_frac"] >= max(0.5, k["hbm_frac"]) else "hbm" if k["hbm_frac"] >= 0.5 else "latency"
        ks.append(k)
    out = {"source": "profiles/pmc_traffic.json", "kernels": ks}
    if step_s and w.get("insts_valu_per_step"):
        # 1024 SIMDs x 2.4 GHz (MI355X_MICROARCH.md) x the live step time: what share of the step's SIMD cycles issue a VALU instruction
        out["valu_issue_frac_of_step"] = round(4 * w["insts_valu_per_step"] / (1024 * 2.4e9 * step_s), 4)
    return w.get("launch_bytes"), out


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus}")
    if a.workload == "c3":
        return main_c3(a, rank, local_rank, world)
    if a.workload in ("c4", "c5"):
        return main_c4(a, rank, local_rank, world)
    import numpy as np
    import torch
    import torch.distributed as dist
    from opus_amd import sharding, synth

    if a.parse_only:
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(a.backend)
        parse_only(a, rank, world)
        if world > 1:
            dist.destroy_process_group()
        return
    from opus_amd import BatchDecoder, _abi as A, steps as stepmod

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = A.load()

    S, K, W, T = a.streams, a.steps, a.warmup, a.frames_per_step
    nsteps = K + W
    nframes = nsteps * T
    ids = sharding.shard_stream_ids(rank, world, S)              # streams shard across ranks, no collective
    packets = synth.celt_fb_stereo_packets(ids, nframes, PAYLOAD)

    # ---- host half (untimed here; its rate is reported separately): packets -> side-info, copied to HBM.
    # Per step the records are laid out [stream][t] (a tail workgroup walks one stream's T frames in time order).
    # The streams are dealt to `P` sub-batches (contiguous blocks), each with its own scratch, HIP stream and launches.
    P = max(1, a.split)
    assert S % P == 0, "--split must divide --streams"
    SP = S // P
    parts = []
    t_parse = 0.0
    nleaves = [0] * nsteps
    alg_bytes = [0] * nsteps
    state = torch.zeros(S * C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
    stride = T * 960 * 2
    pcm = torch.zeros(S * stride, dtype=torch.float32, device=dev)
    SB = C.sizeof(A.CeltState)
    for q in range(P):
        parser = BatchDecoder(SP, 2, local_rank)
        host_steps, tp = stepmod.build_celt_steps(parser, packets[q * SP:(q + 1) * SP], T, nsteps)
        parser.close()
        t_parse += tp
        strm = torch.cuda.Stream(dev)
        parts.append(dict(
            hdr=[torch.from_numpy(st.hdr).to(dev) for st in host_steps], leaves=[torch.from_numpy(st.leaves).to(dev) for st in host_steps],
            refs=[torch.from_numpy(st.refs).to(dev) for st in host_steps], classes=[st.classes for st in host_steps],
            work=torch.zeros(lib.opusgpu_celt_work_bytes(SP * T), dtype=torch.uint8, device=dev),   # K1 -> plan -> frame-kernel scratch
            state=state[q * SP * SB:(q + 1) * SP * SB], pcm=pcm[q * SP * stride:(q + 1) * SP * stride],
            stream=strm, ps=C.c_void_p(strm.cuda_stream),
            evk=[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]))
        for k, st in enumerate(host_steps):
            nleaves[k] += st.nleaves
            alg_bytes[k] += int(SP * T * ALG_BYTES_BASE + st.pf_on * ALG_BYTES_PF)

    def check(rc):
        if rc:
            raise RuntimeError(lib.opusgpu_last_error().decode())

    def step(f):
        """step f = one opusgpu_launch_celt per sub-batch, issued as its three parts; og_celt_kernel is bracketed by evk[f] on
        the stream it runs on"""
        for pt in parts:
            check(lib.opusgpu_launch_celt_pvq(pt["hdr"][f].data_ptr(), pt["leaves"][f].data_ptr(), SP * T, pt["refs"][f].data_ptr(), pt["classes"][f],
                                              pt["work"].data_ptr(), None, pt["ps"]))
        for pt in parts:
            pt["evk"][f][0].record(pt["stream"])
            check(lib.opusgpu_launch_celt_spectrum(pt["hdr"][f].data_ptr(), SP * T, pt["work"].data_ptr(), None, pt["ps"]))
            pt["evk"][f][1].record(pt["stream"])
        for pt in parts:
            check(lib.opusgpu_launch_celt_tail(pt["hdr"][f].data_ptr(), SP * T, T, pt["work"].data_ptr(), pt["state"].data_ptr(),
                                               pt["pcm"].data_ptr(), stride, None, pt["ps"]))

    torch.cuda.synchronize(dev)
    for f in range(W):
        step(f)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for k in range(K):
        step(W + k)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dev)
    assert a.no_verify or bool(torch.isfinite(pcm).all())         # (profiling knobs that skip stages leave garbage: --no-verify)

    # ---- outside the timed region: the last step's PCM of a sample of streams (first, last, interior) against the CPU oracle,
    #      which decodes those streams from their first packet (all nsteps * T frames: the decoder state carries over)
    verified = None
    if not a.no_verify:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from gpu_util import oracle_check_last_step
        sample = sorted({0, 1, S // 3, S // 2, (2 * S) // 3 + 1, S - 1})
        rows = pcm.view(S, stride)[sample].cpu().numpy()
        worst = oracle_check_last_step({s: rows[i] for i, s in enumerate(sample)}, packets, sample, nframes, T, "bench last step")
        verified = {"ok": True, "streams": sample, "frames_each": T, "max_scaled_err": float(f"{worst:.3e}"),
                    "bar": "max abs err <= 1e-4 (scaled by the reference peak when that exceeds 1), SNR >= 90 dB vs the CPU oracle"}

    # ---- the same streams one frame per launch (frames_per_group = 1: what a caller that sends every packet as it arrives
    #      gets), outside the headline's timed region and reported beside it
    t1 = None
    if world == 1 and not a.no_t1 and P == 1:
        w1 = min(4, nframes // 4)
        n1 = min(24, nframes - w1)
        parser = BatchDecoder(S, 2, local_rank)
        hs1, _ = stepmod.build_celt_steps(parser, packets[:, :n1 + w1], 1, n1 + w1)
        parser.close()
        d1 = [dict(hdr=torch.from_numpy(st.hdr).to(dev), leaves=torch.from_numpy(st.leaves).to(dev), refs=torch.from_numpy(st.refs).to(dev),
                   classes=st.classes) for st in hs1]
        state1 = torch.zeros_like(state)
        pt = parts[0]

        def step1(f):
            d = d1[f]
            check(lib.opusgpu_launch_celt_pvq(d["hdr"].data_ptr(), d["leaves"].data_ptr(), S, d["refs"].data_ptr(), d["classes"],
                                              pt["work"].data_ptr(), None, pt["ps"]))
            check(lib.opusgpu_launch_celt_spectrum(d["hdr"].data_ptr(), S, pt["work"].data_ptr(), None, pt["ps"]))
            check(lib.opusgpu_launch_celt_tail(d["hdr"].data_ptr(), S, 1, pt["work"].data_ptr(), state1.data_ptr(), pcm.data_ptr(), stride,
                                               None, pt["ps"]))
        for f in range(w1):
            step1(f)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for f in range(n1):
            step1(w1 + f)
        torch.cuda.synchronize(dev)
        e1 = time.perf_counter() - t0
        t1 = {"value": round(S * n1 * FRAME_S / e1, 1), "unit": "audio-s/s", "frames_per_stream_per_step": 1, "steps": n1, "warmup": w1,
              "ms_per_step": round(e1 / n1 * 1e3, 4), "note": "same streams, one 20 ms frame per stream per launch (frames_per_group = 1)"}
        del d1, state1

    # ---- and as two sub-batches of streams per step, each with its own scratch on a HIP stream of its own (what --split 2
    #      times as the headline): the latency-bound kernels of one fill the gaps of the other's.  Reported beside the
    #      headline, which stays one launch per step so that its kernel times are those of the committed rocprof summary.
    ov = None
    if world == 1 and not a.no_overlap_leg and P == 1 and S % 2 == 0:
        no, wo = min(12, K), min(2, W)
        SH = S // 2
        halves = []
        for q in range(2):
            parser = BatchDecoder(SH, 2, local_rank)
            hs, _ = stepmod.build_celt_steps(parser, packets[q * SH:(q + 1) * SH], T, no + wo)
            parser.close()
            strm = torch.cuda.Stream(dev)
            halves.append(dict(d=[dict(hdr=torch.from_numpy(st.hdr).to(dev), leaves=torch.from_numpy(st.leaves).to(dev),
                                       refs=torch.from_numpy(st.refs).to(dev), classes=st.classes) for st in hs],
                               work=torch.zeros(lib.opusgpu_celt_work_bytes(SH * T), dtype=torch.uint8, device=dev),
                               state=torch.zeros(SH * SB, dtype=torch.uint8, device=dev), pcm=pcm[q * SH * stride:(q + 1) * SH * stride],
                               stream=strm, ps=C.c_void_p(strm.cuda_stream)))

        def step2(f):
            for h in halves:
                d = h["d"][f]
                check(lib.opusgpu_launch_celt_pvq(d["hdr"].data_ptr(), d["leaves"].data_ptr(), SH * T, d["refs"].data_ptr(), d["classes"],
                                                  h["work"].data_ptr(), None, h["ps"]))
            for h in halves:
                check(lib.opusgpu_launch_celt_spectrum(h["d"][f]["hdr"].data_ptr(), SH * T, h["work"].data_ptr(), None, h["ps"]))
            for h in halves:
                check(lib.opusgpu_launch_celt_tail(h["d"][f]["hdr"].data_ptr(), SH * T, T, h["work"].data_ptr(), h["state"].data_ptr(),
                                                   h["pcm"].data_ptr(), stride, None, h["ps"]))
        torch.cuda.synchronize(dev)
        for f in range(wo):
            step2(f)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for f in range(no):
            step2(wo + f)
        torch.cuda.synchronize(dev)
        eo = time.perf_counter() - t0
        ov = {"value": round(S * no * T * FRAME_S / eo, 1), "unit": "audio-s/s", "sub_batches": 2, "steps": no, "warmup": wo,
              "ms_per_step": round(eo / no * 1e3, 4),
              "note": "same workload, each step issued as two independent half-batches of streams on two HIP streams (bench.py --split 2)"}
        del halves

    # ---- and with consecutive steps pipelined over two HIP streams (the three-part launch API exists for this): K1 + plan of
    #      step n + 1 run beside the tail kernel of step n, on scratch of their own (two scratch blocks, alternating)
    pl = None
    if world == 1 and a.pipeline_leg and P == 1:
        pt = parts[0]
        works = [pt["work"], torch.zeros_like(pt["work"])]
        sA, sB = pt["stream"], torch.cuda.Stream(dev)
        if a.cu_mask:                                   # experiment: the two streams on disjoint sets of CUs
            nb = int(a.cu_mask.split(":")[0])           # "NB" : K1 + plan on NB CUs, the rest unmasked; "NB:rest": the others on the remaining CUs
            sB = cu_masked_stream(dev, range(nb))
            if a.cu_mask.endswith(":rest"):
                sA = cu_masked_stream(dev, range(nb, 256))
        pA, pB = C.c_void_p(sA.cuda_stream), C.c_void_p(sB.cuda_stream)
        state2 = torch.zeros_like(state)
        ev_plan = [torch.cuda.Event() for _ in range(nsteps)]
        ev_tail = [torch.cuda.Event() for _ in range(nsteps)]

        ev_k2 = [torch.cuda.Event() for _ in range(nsteps)]

        def front(f):                                   # K1 + plan of step f: beside the tail kernel of step f - 1, not beside its frame kernel
            if f >= 1:
                sB.wait_event(ev_k2[f - 1])             # (the frame kernel fills every CU by itself; this also orders f after tail(f - 2))
            check(lib.opusgpu_launch_celt_pvq(pt["hdr"][f].data_ptr(), pt["leaves"][f].data_ptr(), S * T, pt["refs"][f].data_ptr(), pt["classes"][f],
                                              works[f % 2].data_ptr(), None, pB))
            ev_plan[f].record(sB)

        def back(f):
            sA.wait_event(ev_plan[f])
            check(lib.opusgpu_launch_celt_spectrum(pt["hdr"][f].data_ptr(), S * T, works[f % 2].data_ptr(), None, pA))
            ev_k2[f].record(sA)
            check(lib.opusgpu_launch_celt_tail(pt["hdr"][f].data_ptr(), S * T, T, works[f % 2].data_ptr(), state2.data_ptr(),
                                               pcm.data_ptr(), stride, None, pA))
            ev_tail[f].record(sA)
        torch.cuda.synchronize(dev)
        front(0)
        for f in range(W):                               # (back before front: an event must be recorded before a wait on it is issued)
            back(f)
            if f + 1 < W:
                front(f + 1)
        torch.cuda.synchronize(dev)                      # the clock starts on an idle GPU with nothing issued: every kernel of the
        t0 = time.perf_counter()                         # timed steps runs inside the timed region
        check(lib.opusgpu_launch_celt_pvq(pt["hdr"][W].data_ptr(), pt["leaves"][W].data_ptr(), S * T, pt["refs"][W].data_ptr(), pt["classes"][W],
                                          works[W % 2].data_ptr(), None, pB))
        ev_plan[W].record(sB)
        for f in range(W, nsteps):
            back(f)
            if f + 1 < nsteps:
                front(f + 1)
        torch.cuda.synchronize(dev)
        ep = time.perf_counter() - t0
        okp = None
        if not a.no_verify:
            rows = pcm.view(S, stride)[sample].cpu().numpy()
            worstp = oracle_check_last_step({s: rows[i] for i, s in enumerate(sample)}, packets, sample, nframes, T, "bench last step (pipelined)")
            okp = float(f"{worstp:.3e}")
        pl = {"value": round(S * K * T * FRAME_S / ep, 1), "unit": "audio-s/s", "steps": K, "warmup": W, "ms_per_step": round(ep / K * 1e3, 4),
              "cu_mask": a.cu_mask or None,
              "verified_max_scaled_err": okp,
              "note": "same workload, consecutive steps pipelined over two HIP streams: K1 + plan of step n + 1 beside the tail kernel of "
                      "step n (two scratch blocks); every kernel of the timed steps inside the timed region.  Worth 3-6 %: every kernel "
                      "here is limited by registers / LDS per wavefront, so kernels that run side by side split the occupancy each needs"}

    # ---- end to end, at every --gpus N: packets in host memory -> opusgpu_batch_send (host parse on the rank's threads, pinned to
    #      the GPU's NUMA node) -> opusgpu_batch_decode (H2D + the same kernels), PCM left in HBM.  Same streams, same T frames
    #      per stream per send.  Every rank runs it at once, as in production: the figure shows what the host side sustains.
    e2e = None
    if not a.no_e2e and P == 1:
        ne, we = min(6, nsteps - 1), 1
        edec = BatchDecoder(S, 2, local_rank)
        pk_steps = [np.ascontiguousarray(packets[:, f * T:(f + 1) * T]) for f in range(ne + we)]
        for f in range(we):
            assert not edec.send_packed(pk_steps[f]).any()
            edec.decode()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for f in range(we, we + ne):
            errs = edec.send_packed(pk_steps[f])             # (waits for the previous step's H2D copies only: opusgpu_batch_send)
            edec.decode(sync=False)                          # asynchronous: the next step's parse runs beside this step's copies and kernels
        edec.sync()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        ee = sharding.max_over_ranks(time.perf_counter() - t0, dev)
        assert not errs.any()
        e2e = {"value": round(sharding.aggregate_audio_seconds_per_second(world, S, ne, T * FRAME_S, ee), 1), "unit": "audio-s/s",
               "steps": ne, "warmup": we, "ms_per_step": round(ee / ne * 1e3, 3), "parse_threads_per_rank": int(os.environ.get("OPUSGPU_THREADS", 0)) or min(16, os.cpu_count() or 1),
               "note": "packets in host memory -> opusgpu_batch_send (host parse) -> opusgpu_batch_decode (H2D + kernels), PCM left in HBM; "
                       "all ranks at once, max over ranks; bound by the host parser"}
        edec.close()

    avg_kernel_s = elapsed / K                               # the calls run back to back on one stream: per call = step time
    avg_alg = sum(alg_bytes[W:]) / K
    achieved = avg_alg / avg_kernel_s / 1e9
    k2_s = sum(e0.elapsed_time(e1) for pt in parts for e0, e1 in pt["evk"][W:]) / K / 1e3       # og_celt_kernel alone (summed over sub-batches)
    k2_bytes = S * T * K2_BYTES
    traffic = k2_traffic = c2_pmc = k2_valu = None
    if S == STREAMS_PER_GPU and T == 8:                          # HBM bytes / VALU issue per launch from the rocprofv3 --pmc passes: the default workload only
        traffic, c2_pmc = pmc_info("c2", avg_kernel_s)
        for k in (c2_pmc or {}).get("kernels", []):
            if k["kernel"] == "og_celt_kernel":
                k2_traffic, k2_valu = k.get("hbm_bytes"), k.get("valu_issue_frac")

    if rank == 0:
        out = {
            "metric": "decoded audio-sec/sec (48 kHz stereo CELT 20 ms, 4096 streams)",
            "value": round(sharding.aggregate_audio_seconds_per_second(world, S, K, T * FRAME_S, elapsed), 1),
            "unit": "audio-s/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "verified": bool(verified and verified["ok"]), "verify": verified,
            "config": {"workload": "BASELINE configs[1]: synthetic CELT-only FB 48 kHz stereo 20 ms frames (TOC 0xFC + "
                                   f"{PAYLOAD} random payload bytes), {S} streams per GPU, {T} consecutive frames per stream per step; "
                                   "device-only: side-info pre-parsed and resident in HBM, PCM stays in HBM",
                       "streams_per_gpu": S, "frame_ms": 20, "frames_per_stream_per_step": T, "payload_bytes": PAYLOAD, "sub_batches": P,
                       "parallelism": f"streams sharded over {world} GPU(s), one process per GPU, no collective in the data path; "
                                      "RCCL only for the barrier and the max of the elapsed times"},
            # the launch = every kernel of one opusgpu_launch_celt (K1, plan, og_celt_kernel, og_celt_tail_kernel),
            # priced with SURVEY 8(d)'s per-frame bytes; `dominant` = og_celt_kernel alone with the bytes it
            # moves by design (side-info + energies 1724, K1 row in 7680, sample row out 7680)
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": "one opusgpu_launch_celt call: og_pvq_all_kernel + og_plan_kernel + og_celt_kernel + og_celt_tail_kernel",
                         "avg_launch_us": round(avg_kernel_s * 1e6, 2),
                         "algorithmic_bytes_per_launch": int(avg_alg),
                         "dominant": {"kernel": "og_celt_kernel", "avg_launch_us": round(k2_s * 1e6, 2), "bytes_per_launch": int(k2_bytes),
                                      "achieved": round(k2_bytes / k2_s / 1e9, 2), "frac": round(k2_bytes / k2_s / 1e9 / HBM_PEAK_GBS, 5),
                                      "traffic": k2_traffic, "valu_issue_frac": k2_valu,
                                      "bound": "valu" if (k2_valu or 0) >= 0.5 else "hbm"}},
            "host_parse": {"value": round(S * nframes * FRAME_S / t_parse, 1), "unit": "audio-s/s",
                           "note": "packets -> side-info on the host threads (untimed in `value`)",
                           "mean_leaves_per_frame": round(sum(nleaves) / (S * nframes), 1)},
        }
        if c2_pmc:
            out["pmc"] = c2_pmc
            out["roofline"]["valu_issue_frac"] = c2_pmc.get("valu_issue_frac_of_step")
            out["roofline"]["limiter"] = ("VALU issue and latency, not HBM: see pmc.kernels (og_celt_kernel issues a VALU instruction in most "
                                          "SIMD cycles; K1, plan and tail are bound by dependent chains at their occupancy)")
        if t1:
            out["t1"] = t1
        if ov:
            out["two_sub_batches"] = ov
        if pl:
            out["pipelined"] = pl
        if e2e:
            out["end_to_end"] = e2e
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(packets)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
