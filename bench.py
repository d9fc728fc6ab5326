#!/usr/bin/env python3
"""bench.py -- decoded audio-seconds per second on BASELINE.json config 2:
synthetic CELT-only fullband 48 kHz stereo 20 ms frames, 4096 streams per MI355X (weak scaling over GPUs).

A step = one pass of the hot path over one batch: the next T consecutive 20 ms frames (default T = 8, 160 ms of
audio) of each of the 4096 streams through the C ABI (layer 3b): opusgpu_launch_celt_pvq (the K1 PVQ kernels over all
4096*T frames' leaves + the band-plan kernel) and opusgpu_launch_celt_frames (the K2+K3 kernel, one workgroup per stream
walking its T frames in time order), issued as one opusgpu_launch_celt call per step on one stream.  --pipeline issues
the two halves separately and lets the first half of step k+1 (which touches no decoder state) run on its own stream and
scratch buffer while the frame kernel of step k is still busy (measured: no gain on MI355X, the frame kernel's workgroups
hold all of the LDS, so the other kernels only get on a CU when it drains).  Inputs (side-info records, pre-parsed by the host layer) are resident in HBM before the timed region; PCM and
decoder state stay in HBM.  --frames-per-step 1 gives the one-frame-per-call figure.  Prints ONE JSON line on rank 0.
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="EXPERIMENTAL: the stateless chain (K1, plan, og_celt_kernel) of step k+1 on one stream, the tail kernel of step k on "
                         "another; 1-2 %% measured (DESIGN.md section 5): not used for any reported number")
    return ap.parse_args()


def cpu_baseline(packets, budget_s=12.0):
    """The CPU oracle (restatement of lu-zero/opus + RFC 6716 synthesis tail; the Rust crate cannot be built here),
    one stream per thread on the host cores, on a bounded sample of the same packets."""
    from concurrent.futures import ThreadPoolExecutor
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    cores = min(16, os.cpu_count() or 1)          # the GPU box grants 16 CPUs per GPU
    nstreams, nframes = packets.shape[0], packets.shape[1]
    sample_streams = min(nstreams, cores * 4)
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


def main():
    a = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    import torch
    import torch.distributed as dist
    from opus_amd import BatchDecoder, _abi as A, sharding, synth

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = A.load()

    S, K, W, T = a.streams, a.steps, a.warmup, a.frames_per_step
    nsteps = K + W
    nframes = nsteps * T
    ids = sharding.shard_stream_ids(rank, world, S)              # streams shard across ranks, no collective
    packets = synth.celt_fb_stereo_packets(ids, nframes, PAYLOAD)

    # ---- host half (untimed here; its rate is reported separately): packets -> side-info, copied to HBM.
    # Per step the records are laid out [stream][t] (a workgroup walks one stream's T frames in time order).
    parser = BatchDecoder(S, 2, local_rank)
    hdr_dev, leaf_dev, ref_dev, ref_classes, alg_bytes, nleaves = [], [], [], [], [], []
    t_parse = 0.0
    HS = C.sizeof(A.CeltHdr)
    o_lb, o_po = A.CeltHdr.leaf_begin.offset, A.CeltHdr.pcm_off.offset
    for k in range(nsteps):
        hb_all = np.zeros((S, T, HS), np.uint8)
        leaf_parts, leaf_off, pf_on = [], 0, 0
        for t in range(T):
            f = k * T + t
            pk = [packets[s, f].tobytes() for s in range(S)]
            t0 = time.perf_counter()
            errs = parser.send(pk)
            t_parse += time.perf_counter() - t0
            assert not any(errs)
            celt, leaves, _ = parser.sideinfo()
            assert len(celt) == S
            hb = np.frombuffer(celt, dtype=np.uint8).reshape(S, HS).copy()
            hb[:, o_lb:o_lb + 4].view(np.uint32)[:] += leaf_off          # leaf indices are relative to the step's array
            hb[:, o_po:o_po + 4].view(np.uint32)[:] = t * 960            # frame t lands at sample t*960 of the PCM row
            gq = hb[:, A.CeltHdr.pf_gain_q.offset:A.CeltHdr.pf_gain_q.offset + 3]
            pf_on += int((gq != 0).any(axis=1).sum())
            hb_all[:, t] = hb
            leaf_parts.append(np.frombuffer(leaves, dtype=np.uint8).copy())
            leaf_off += len(leaves)
        lb = np.concatenate(leaf_parts)
        hbc = np.ascontiguousarray(hb_all.reshape(S * T, HS))
        refs = (A.LeafRef * max(leaf_off, 1))()
        classes = (C.c_uint32 * 4)()
        nref = lib.opusgpu_sort_pvq_leaves(hbc.ctypes.data, S * T, lb.ctypes.data, refs, leaf_off, classes)
        alg_bytes.append(int(S * T * ALG_BYTES_BASE + pf_on * ALG_BYTES_PF))
        nleaves.append(leaf_off)
        hdr_dev.append(torch.from_numpy(hbc).to(dev))
        leaf_dev.append(torch.from_numpy(lb).to(dev))
        ref_dev.append(torch.from_numpy(np.frombuffer(refs, dtype=np.uint8)[: max(nref, 1) * 8].copy()).to(dev))
        ref_classes.append(classes)
    parser.close()

    state = torch.zeros(S * C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
    # K1 -> plan -> frame-kernel scratch (kept zero by the kernels); two of them so that consecutive steps can overlap
    works = [torch.zeros(lib.opusgpu_celt_work_bytes(S * T), dtype=torch.uint8, device=dev) for _ in range(2)]
    stride = T * 960 * 2
    pcm = torch.zeros(S * stride, dtype=torch.float32, device=dev)
    # the first half is short and on the critical path of the next frame kernel: its stream ranks first for free CU slots
    s_pvq, s_frm = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    p_pvq, p_frm = C.c_void_p(s_pvq.cuda_stream), C.c_void_p(s_frm.cuda_stream)
    ready = [torch.cuda.Event() for _ in range(nsteps)]
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]
    evk = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nsteps)]   # around og_celt_kernel

    def check(rc):
        if rc:
            raise RuntimeError(lib.opusgpu_last_error().decode())

    def step(f):
        """step f; og_celt_kernel is bracketed by evk[f] on the stream it runs on"""
        w = works[f % 2]
        if (not a.pipeline):
            # = opusgpu_launch_celt, issued as its three parts so that the dominant kernel can be bracketed on its stream
            check(lib.opusgpu_launch_celt_pvq(hdr_dev[f].data_ptr(), leaf_dev[f].data_ptr(), S * T, ref_dev[f].data_ptr(), ref_classes[f],
                                              w.data_ptr(), None, p_frm))
            evk[f][0].record(s_frm)
            check(lib.opusgpu_launch_celt_spectrum(hdr_dev[f].data_ptr(), S * T, w.data_ptr(), None, p_frm))
            evk[f][1].record(s_frm)
            check(lib.opusgpu_launch_celt_tail(hdr_dev[f].data_ptr(), S * T, T, w.data_ptr(), state.data_ptr(), pcm.data_ptr(), stride,
                                               None, p_frm))
            return
        # pipelined: the stateless chain K1 -> plan -> og_celt_kernel of step f on one stream, the tail kernels (which carry
        # the decoder state, in order) on another: tail(f) overlaps the stateless chain of step f + 1
        if f >= 2:
            s_pvq.wait_event(ev[f - 2][1])              # the scratch buffer is free once tail(f-2) has consumed it
        check(lib.opusgpu_launch_celt_pvq(hdr_dev[f].data_ptr(), leaf_dev[f].data_ptr(), S * T, ref_dev[f].data_ptr(), ref_classes[f],
                                          w.data_ptr(), None, p_pvq))
        evk[f][0].record(s_pvq)
        check(lib.opusgpu_launch_celt_spectrum(hdr_dev[f].data_ptr(), S * T, w.data_ptr(), None, p_pvq))
        evk[f][1].record(s_pvq)
        ready[f].record(s_pvq)
        s_frm.wait_event(ready[f])
        ev[f][0].record(s_frm)
        check(lib.opusgpu_launch_celt_tail(hdr_dev[f].data_ptr(), S * T, T, w.data_ptr(), state.data_ptr(), pcm.data_ptr(), stride,
                                           None, p_frm))
        ev[f][1].record(s_frm)

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
    assert bool(torch.isfinite(pcm).all())

    if a.pipeline:
        kernel_ms = [e0.elapsed_time(e1) for e0, e1 in ev[W:]]
        avg_kernel_s = sum(kernel_ms) / K / 1e3
    else:                                                    # the calls run back to back on one stream: per call = step time
        avg_kernel_s = elapsed / K
    avg_alg = sum(alg_bytes[W:]) / K
    achieved = avg_alg / avg_kernel_s / 1e9
    step_achieved = avg_alg / (elapsed / K) / 1e9           # the same bytes over the whole step (all kernels)
    k2_s = sum(e0.elapsed_time(e1) for e0, e1 in evk[W:]) / K / 1e3       # og_celt_kernel alone
    k2_bytes = S * T * K2_BYTES
    traffic = k2_traffic = None
    tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")      # HBM bytes per launch from the rocprofv3 --pmc passes
    if os.path.exists(tp) and S == STREAMS_PER_GPU and T == 8:   # collected on the default workload only
        try:
            tj = json.load(open(tp))
            traffic, k2_traffic = tj.get("launch_bytes"), tj.get("og_celt_kernel_bytes")
        except Exception:
            traffic = k2_traffic = None

    if rank == 0:
        out = {
            "metric": "decoded audio-sec/sec (48 kHz stereo CELT 20 ms, 4096 streams)",
            "value": round(sharding.aggregate_audio_seconds_per_second(world, S, K, T * FRAME_S, elapsed), 1),
            "unit": "audio-s/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(elapsed / K * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic CELT-only FB 48 kHz stereo 20 ms frames (TOC 0xFC + "
                                   f"{PAYLOAD} random payload bytes), {S} streams per GPU, {T} consecutive frames per stream per step",
                       "streams_per_gpu": S, "frame_ms": 20, "frames_per_stream_per_step": T, "payload_bytes": PAYLOAD,
                       "parallelism": f"streams sharded over {world} GPU(s), no collective",
                       "pipelined": bool(a.pipeline)},
            # the launch = every kernel of one opusgpu_launch_celt (K1, plan, og_celt_kernel, og_celt_tail_kernel),
            # priced with SURVEY 8(d)'s per-frame bytes; `dominant` = og_celt_kernel alone (52 % of the launch) with the bytes it
            # moves by design (side-info + energies 1724, K1 row in 7680, sample row out 7680)
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "kernel": ("one opusgpu_launch_celt call: og_pvq_all_kernel + og_plan_kernel + og_celt_kernel + og_celt_tail_kernel"
                                    if (not a.pipeline) else "og_celt_kernel + og_celt_tail_kernel; K1 + plan of the next step overlap them"),
                         "avg_launch_us": round(avg_kernel_s * 1e6, 2),
                         "algorithmic_bytes_per_launch": int(avg_alg),
                         "dominant": {"kernel": "og_celt_kernel", "avg_launch_us": round(k2_s * 1e6, 2), "bytes_per_launch": int(k2_bytes),
                                      "achieved": round(k2_bytes / k2_s / 1e9, 2), "frac": round(k2_bytes / k2_s / 1e9 / HBM_PEAK_GBS, 5),
                                      "traffic": k2_traffic},
                         "whole_step": {"achieved": round(step_achieved, 2), "frac": round(step_achieved / HBM_PEAK_GBS, 5)}},
            "host_parse": {"value": round(S * nframes * FRAME_S / t_parse, 1), "unit": "audio-s/s",
                           "note": "packets -> side-info on the host threads (untimed in `value`)",
                           "mean_leaves_per_frame": round(sum(nleaves) / (S * nframes), 1)},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(packets)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
