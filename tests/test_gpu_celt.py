"""GPU parity: the HIP CELT path, called through the C ABI, against the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from gpu_util import CeltLaunch, HostParser, check_pcm, oracle_check_last_step
from opus_amd import BatchDecoder, Decoder, _abi as A, synth

pytestmark = pytest.mark.gpu


def _decode_single(vid, npackets, channels=2):
    extra, pk = orc.load_vectors(ids=[vid], max_packets=npackets)[vid]
    if channels == 1:
        extra = extra[:9] + b"\x01" + extra[10:]
    d = Decoder(0)
    d.set_extradata(extra)
    d.configure()
    ref = orc.Decoder(channels)
    worst = 0.0
    for i, p in enumerate(pk):
        d.send_packet(p)
        worst = max(worst, check_pcm(ref.decode(p), d.receive_frame(), f"vector {vid} packet {i}"))
    return worst


@pytest.mark.parametrize("vid,n", [(11, 200), (1, 400), (7, 4186)])
def test_single_stream_vectors(vid, n):
    """reference call order set_extradata -> configure -> send_packet -> receive_frame (decoder.rs:217-231)"""
    _decode_single(vid, n)


def test_single_stream_mono_output():
    _decode_single(7, 1500, channels=1)
    _decode_single(1, 100, channels=1)


def test_batch_mixed_vectors():
    """48 streams on different CELT vectors / offsets, one packet per stream per call"""
    V = orc.load_vectors(ids=[1, 7, 11])
    ns, steps = 48, 30
    dec = BatchDecoder(ns, 2, 0)
    refs = [orc.Decoder(2) for _ in range(ns)]
    src = [V[(1, 7, 11)[s % 3]][1] for s in range(ns)]
    start = [(37 * s) % 300 for s in range(ns)]
    for t in range(steps):
        pk = [src[s][start[s] + t] for s in range(ns)]
        assert not any(dec.send(pk))
        dec.decode()
        for s in range(ns):
            check_pcm(refs[s].decode(pk[s]), dec.read(s), f"stream {s} step {t}")


def test_batch_subset_and_bad_packet():
    """a bad packet reports its own error and leaves the other streams alone; absent streams produce nothing"""
    V = orc.load_vectors(ids=[11], max_packets=8)[11][1]
    dec = BatchDecoder(4, 2, 0)
    refs = [orc.Decoder(2) for _ in range(4)]
    errs = dec.send([V[0], b"\x03\x00", V[0]], stream_ids=[0, 1, 3])       # (code 3 with a frame count of zero, packet.rs:145)
    assert errs == [0, A.INVALID_DATA, 0]
    dec.decode()
    check_pcm(refs[0].decode(V[0]), dec.read(0))
    check_pcm(refs[3].decode(V[0]), dec.read(3))
    assert dec.read(1).shape[0] == 0 and dec.read(2).shape[0] == 0
    assert not any(dec.send([V[1], V[0]], stream_ids=[0, 1]))
    dec.decode()
    check_pcm(refs[0].decode(V[1]), dec.read(0))
    check_pcm(refs[1].decode(V[0]), dec.read(1))


def test_send_while_decode_in_flight():
    """decode(sync=False) -> send -> sync: batch_send waits for the previous decode's H2D copies before it rewrites the
    pinned side-info, so back-to-back calls with no sync in between still decode every frame right.  Enough streams that
    the copies take a while (4096 x 668-byte headers + leaves)."""
    ns, nf = 4096, 6
    pk = synth.celt_fb_stereo_packets(range(ns), nf, 160)
    dec = BatchDecoder(ns, 2, 0)
    sample = [0, 5, 1000, 2047, 4095]
    refs = {s: orc.Decoder(2) for s in sample}
    lists = [[pk[s, f].tobytes() for s in range(ns)] for f in range(nf)]
    assert not any(dec.send(lists[0]))
    for f in range(nf):
        dec.decode(sync=False)
        if f + 1 < nf:
            assert not any(dec.send(lists[f + 1]))     # parses (and rewrites the staging arrays) while frame f is in flight
        for s in sample:                               # read() synchronises; frame f's PCM is still in the device rows
            check_pcm(refs[s].decode(lists[f][s]), dec.read(s), f"stream {s} frame {f}")


def test_flush_resets_stream():
    V = orc.load_vectors(ids=[11], max_packets=6)[11][1]
    dec = BatchDecoder(1, 2, 0)
    for p in V[:4]:
        dec.send([p]); dec.decode()
    dec.flush(0)
    ref = orc.Decoder(2)
    for p in V[:3]:
        dec.send([p]); dec.decode()
        check_pcm(ref.decode(p), dec.read(0), "after flush")


@pytest.mark.parametrize("payload", [160, 320])
def test_synthetic_c2(payload):
    """BASELINE config 2 shape (CELT FB stereo 20 ms, random payloads) on 64 streams x 12 frames"""
    ns, nf = 64, 12
    pk = synth.celt_fb_stereo_packets(range(ns), nf, payload)
    dec = BatchDecoder(ns, 2, 0)
    refs = [orc.Decoder(2) for _ in range(ns)]
    for f in range(nf):
        assert not any(dec.send([pk[s, f].tobytes() for s in range(ns)]))
        dec.decode()
        for s in range(ns):
            check_pcm(refs[s].decode(pk[s, f].tobytes()), dec.read(s), f"stream {s} frame {f}")


@pytest.mark.parametrize("payload", [8, 24, 80])
def test_low_rate_many_fill_leaves(payload):
    """starved frames: most bands carry no pulses, up to ~42 noise / folded leaves per frame over several dependency
    levels -- more than the 16 fill ops the spectrum kernel stages in LDS, so the rest are read from the plan in HBM"""
    ns, nf = 48, 30
    pk = synth.celt_fb_stereo_packets(range(ns), nf, payload)
    dec = BatchDecoder(ns, 2, 0)
    refs = [orc.Decoder(2) for _ in range(ns)]
    for f in range(nf):
        assert not any(dec.send([pk[s, f].tobytes() for s in range(ns)]))
        dec.decode()
        for s in range(ns):
            check_pcm(refs[s].decode(pk[s, f].tobytes()), dec.read(s), f"payload {payload} stream {s} frame {f}")


def test_stage_dumps_bit_exact_pulses():
    """raw launch (layer 3b) with stage dumps: PVQ pulse vectors and collapse masks bit-exact, normalised
    coefficients and MDCT bins within f32 rounding of the oracle's"""
    import torch
    lib = A.load()
    assert lib.opusgpu_init(0) == 0
    cases = []
    _, pk7 = orc.load_vectors(ids=[7])[7]
    cases += [bytes(p) for p in pk7[::23]]
    syn = synth.celt_fb_stereo_packets(range(6), 20, 200)
    cases += [syn[s, f].tobytes() for s in range(6) for f in range(20)]
    hp = HostParser(2)
    ref = orc.Decoder(2, trace=True)
    dev = torch.device("cuda:0")
    cl = CeltLaunch(1)
    state = torch.zeros(C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
    pcm = torch.zeros(5760 * 2, dtype=torch.float32, device=dev)
    for i, p in enumerate(cases):
        n, nc, nl, _ = hp.parse(p)
        assert n == 1 and nc == 1
        refpcm = ref.decode(p)
        t = ref.trace
        hp.celt[0].stream = 0
        hp.celt[0].pcm_off = 0
        up = cl.upload(np.frombuffer(hp.celt, dtype=np.uint8)[: C.sizeof(A.CeltHdr)],
                       np.frombuffer(hp.leaves, dtype=np.uint8)[: nl * 16], 1)
        pulses = torch.zeros((2, 960), dtype=torch.int16, device=dev)
        norm = torch.zeros((2, 960), dtype=torch.float32, device=dev)
        freq = torch.zeros((2, 960), dtype=torch.float32, device=dev)
        coll = torch.zeros((2, 21), dtype=torch.uint8, device=dev)
        dbg = A.CeltDebug(pulses.data_ptr(), norm.data_ptr(), freq.data_ptr(), coll.data_ptr())
        cl.launch(up, state, pcm, 5760 * 2, dbg)
        torch.cuda.synchronize()
        Cn = t.C
        assert np.array_equal(pulses.cpu().numpy()[:Cn], np.ctypeslib.as_array(t.pulses)[:Cn]), f"case {i}: pulses"
        assert np.array_equal(coll.cpu().numpy()[:Cn], np.ctypeslib.as_array(t.collapse)[:Cn]), f"case {i}: collapse masks"
        tx = np.ctypeslib.as_array(t.X)[:Cn]
        assert np.abs(norm.cpu().numpy()[:Cn] - tx).max() <= 1e-5, f"case {i}: normalised coefficients"
        N = 120 << t.lm                      # the oracle's trace holds N valid bins per channel
        tf = np.ctypeslib.as_array(t.freq)[:, :N]
        df = np.abs(freq.cpu().numpy()[:, :N] - tf)
        assert df.max() <= 1e-5 * max(1.0, np.abs(tf).max()), (
            f"case {i}: MDCT bins differ at {np.argwhere(df > 1e-5 * max(1.0, np.abs(tf).max()))[:8].tolist()} "
            f"lm {t.lm} transient {t.transient} C {t.C} anticollapse {t.anticollapse_on} silence {t.silence}")
        check_pcm(refpcm, pcm.cpu().numpy()[: refpcm.size].reshape(refpcm.shape), f"case {i}: pcm")
        # the IMDCT stage on its own against ground truth: the kernel's samples before the window (the frame's scratch row)
        # must be the O(N^2) f64 inverse MDCT of the kernel's own MDCT bins, block by block, >= 100 dB
        row = cl.work[: 1920 * 4].view(torch.float32).cpu().numpy().reshape(2, 960)
        gf = freq.cpu().numpy()
        B = (1 << t.lm) if t.transient else 1
        NB = N // B
        want = np.zeros((2, N), np.float64)
        L = orc.lib()
        for ch in range(2):
            for b in range(B):
                src = np.ascontiguousarray(gf[ch, :N])
                D = np.zeros(NB, np.float64)
                L.orc_imdct_raw_direct(src[b:].ctypes.data_as(C.c_void_p), B, NB, D.ctypes.data_as(C.c_void_p))
                want[ch, NB * b: NB * (b + 1)] = D
        if np.abs(want).max() > 0:
            err = ((row[:, :N] - want) ** 2).sum()
            snr = 10 * np.log10((want ** 2).sum() / max(err, 1e-300))
            assert snr >= 100.0, f"case {i}: IMDCT stage {snr:.1f} dB against the f64 direct transform (lm {t.lm}, blocks {B})"
        else:
            assert not row[:, :N].any()


def test_full_size_properties():
    """BASELINE size (4096 streams): duplicate streams decode bit-identically (stream independence, no cross-talk),
    a sample of streams matches the oracle, and the rest are finite"""
    import torch
    ns, nf = 4096, 3
    half = synth.celt_fb_stereo_packets(range(ns // 2), nf, 160)
    dec = BatchDecoder(ns, 2, 0)
    sample = list(range(0, ns // 2, 128))
    refs = {s: orc.Decoder(2) for s in sample}
    for f in range(nf):
        pk = [half[s % (ns // 2), f].tobytes() for s in range(ns)]
        assert not any(dec.send(pk))
        dec.decode()
        ptr, stride, nsamp = dec.device_pcm()
        assert (nsamp == 960).all()
        for s in sample:
            a, b = dec.read(s), dec.read(s + ns // 2)
            assert np.array_equal(a, b)
            check_pcm(refs[s].decode(pk[s]), a, f"stream {s} frame {f}")


def test_split_launches_match_single_call():
    """opusgpu_launch_celt_pvq / _spectrum / _tail issued separately, alternating two scratch buffers (the way a caller
    pipelines consecutive batches), give bit-identical PCM to the one-call launch.  Everything runs on one stream (the
    test is about the split, not about overlap)."""
    import torch
    lib = A.load()
    dev = torch.device("cuda:0")
    S, T, steps = 96, 4, 5
    pk = synth.celt_fb_stereo_packets(range(S), T * steps, 160)
    parser = BatchDecoder(S, 2, 0)
    HS = C.sizeof(A.CeltHdr)
    ups = []
    for k in range(steps):
        hb_all = np.zeros((S, T, HS), np.uint8)
        parts, off = [], 0
        for t in range(T):
            assert not any(parser.send([pk[s, k * T + t].tobytes() for s in range(S)]))
            celt, leaves, _ = parser.sideinfo()
            hb = np.frombuffer(celt, dtype=np.uint8).reshape(S, HS).copy()
            hb[:, A.CeltHdr.leaf_begin.offset:A.CeltHdr.leaf_begin.offset + 4].view(np.uint32)[:] += off
            hb[:, A.CeltHdr.pcm_off.offset:A.CeltHdr.pcm_off.offset + 4].view(np.uint32)[:] = t * 960
            hb_all[:, t] = hb
            parts.append(np.frombuffer(leaves, dtype=np.uint8).copy())
            off += len(leaves)
        ups.append((np.ascontiguousarray(hb_all.reshape(S * T, HS)), np.concatenate(parts)))
    stride = T * 960 * 2

    def run(split):
        cl = [CeltLaunch(S * T), CeltLaunch(S * T)]
        state = torch.zeros(S * C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
        pcm = torch.zeros((steps, S * stride), dtype=torch.float32, device=dev)
        s2 = torch.cuda.Stream(dev)
        p2 = C.c_void_p(s2.cuda_stream)
        keep = []
        torch.cuda.synchronize()
        for k, (hb, lb) in enumerate(ups):
            c = cl[k % 2]
            up = c.upload(hb, lb, S * T)
            keep.append(up)
            torch.cuda.synchronize()            # uploads ran on the default stream
            if not split:
                c.launch(up, state, pcm[k], stride, None, p2, frames_per_group=T)
                continue
            rc = lib.opusgpu_launch_celt_pvq(up["hdr"].data_ptr(), up["leaves"].data_ptr(), S * T, up["refs"].data_ptr(), up["classes"],
                                             c.work.data_ptr(), None, p2)
            assert rc == 0, lib.opusgpu_last_error()
            rc = lib.opusgpu_launch_celt_spectrum(up["hdr"].data_ptr(), S * T, c.work.data_ptr(), None, p2)
            assert rc == 0, lib.opusgpu_last_error()
            rc = lib.opusgpu_launch_celt_tail(up["hdr"].data_ptr(), S * T, T, c.work.data_ptr(), state.data_ptr(), pcm[k].data_ptr(),
                                              stride, None, p2)
            assert rc == 0, lib.opusgpu_last_error()
        torch.cuda.synchronize()
        return pcm.cpu().numpy()

    a, b = run(False), run(True)
    assert np.array_equal(a, b)
    ref = orc.Decoder(2)                                # and stream 0 against the oracle
    for k in range(steps):
        for t in range(T):
            r = ref.decode(pk[0, k * T + t].tobytes())
            check_pcm(r, a[k][t * 1920:(t + 1) * 1920].reshape(960, 2), f"step {k} frame {t}")


def test_bench_launch_shape_against_oracle():
    """bench.py's exact launch shape -- 4096 streams x 8 frames per launch (32768 frames), frames_per_group = 8 -- for three
    consecutive launches on the same decoder state, through the layer-3b entry points bench.py calls; the first, the last
    and interior streams against the CPU oracle after every launch, every sample finite.  (Round 1's oracle comparisons
    ran at frames_per_group = 1, or T = 4 with 96 streams; DESIGN.md section 5a has the fault this shape once showed.)"""
    import torch
    from opus_amd import steps as stepmod
    lib = A.load()
    dev = torch.device("cuda:0")
    S, T, nsteps = 4096, 8, 3
    pk = synth.celt_fb_stereo_packets(range(S), T * nsteps, 160)
    parser = BatchDecoder(S, 2, 0)
    host_steps, _ = stepmod.build_celt_steps(parser, pk, T, nsteps)
    parser.close()
    state = torch.zeros(S * C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
    work = torch.zeros(lib.opusgpu_celt_work_bytes(S * T), dtype=torch.uint8, device=dev)
    stride = T * 960 * 2
    pcm = torch.zeros(S * stride, dtype=torch.float32, device=dev)
    sample = [0, 1, 777, 2047, 2048, 3333, 4095]
    for k, st in enumerate(host_steps):
        hdr, leaves, refs = (torch.from_numpy(x).to(dev) for x in (st.hdr, st.leaves, st.refs))
        torch.cuda.synchronize()
        assert lib.opusgpu_launch_celt_pvq(hdr.data_ptr(), leaves.data_ptr(), S * T, refs.data_ptr(), st.classes, work.data_ptr(), None, None) == 0
        assert lib.opusgpu_launch_celt_spectrum(hdr.data_ptr(), S * T, work.data_ptr(), None, None) == 0
        assert lib.opusgpu_launch_celt_tail(hdr.data_ptr(), S * T, T, work.data_ptr(), state.data_ptr(), pcm.data_ptr(), stride, None, None) == 0
        torch.cuda.synchronize()
        assert bool(torch.isfinite(pcm).all()), f"launch {k}"
        rows = pcm.view(S, stride)[sample].cpu().numpy()
        oracle_check_last_step({s: rows[i] for i, s in enumerate(sample)}, pk, sample, (k + 1) * T, T, f"launch {k}")


@pytest.mark.parametrize("kind", ["celt", "mixed"])
def test_multi_packet_send(kind):
    """opusgpu_batch_send_multi: T consecutive packets per stream in one call.  CELT-only single-frame packets take the
    one-launch path (a tail workgroup walks a stream's T frames); anything else the round-by-round path.  Against the oracle
    and against the same packets sent one per call."""
    T, ns = 6, 24
    if kind == "celt":
        pk = synth.celt_fb_stereo_packets(range(ns), 3 * T, 160)
        rows = [[pk[s, f].tobytes() for f in range(3 * T)] for s in range(ns)]
    else:
        vec = orc.load_vectors(ids=[10, 12], max_packets=900)
        rows = [[bytes(p) for p in vec[10 if s % 2 else 12][1][100 + 11 * s: 100 + 11 * s + 3 * T]] for s in range(ns)]
        rows[3][4] = b""                               # a lost packet among them
    multi, single = BatchDecoder(ns, 2, 0), BatchDecoder(ns, 2, 0)
    refs = [orc.Decoder(2) for _ in range(ns)]
    for k in range(3):
        errs = multi.send_multi([r[k * T:(k + 1) * T] for r in rows])
        assert not any(any(e) for e in errs)
        multi.decode()
        outs = [[] for _ in range(ns)]
        for t in range(T):
            assert not any(single.send([r[k * T + t] for r in rows]))
            single.decode()
            for s in range(ns):
                outs[s].append(single.read(s))
        for s in range(ns):
            want = np.concatenate([refs[s].decode(p) for p in rows[s][k * T:(k + 1) * T]])
            got = multi.read(s)
            assert got.shape == want.shape, (s, k)
            check_pcm(want, got, f"{kind} stream {s} send {k}")
            check_pcm(np.concatenate(outs[s]), got, f"{kind} stream {s} send {k}: multi against one packet per call")


def test_multi_frame_packets_in_one_launch():
    """code 1-3 CELT packets (vectors 01 and 11 hold up to six frames per packet): streams in lockstep bring packets of the same
    frame count, which the batch engine issues as ONE launch with frames_per_group = frames per packet (x packets per stream
    with send_multi); streams out of step fall back to a round of launches per frame.  Both against the oracle."""
    ns = 6
    for vid in (1, 11):
        _, pk = orc.load_vectors(ids=[vid], max_packets=260)[vid]
        pk = [bytes(p) for p in pk]
        step = BatchDecoder(ns, 2, 0)                   # every stream at the same packet: grouped
        skew = BatchDecoder(ns, 2, 0)                   # stream s runs s packets ahead: mixed frame counts
        refs = [orc.Decoder(2) for _ in range(ns)]
        refk = [orc.Decoder(2) for _ in range(ns)]
        for i in range(200):
            assert not any(step.send([pk[i]] * ns))
            step.decode()
            want = refs[0].decode(pk[i])
            for s in range(ns):
                if s:
                    refs[s].decode(pk[i])
                got = step.read(s)
                assert got.shape == want.shape, (vid, i, s)
                check_pcm(want, got, f"vector {vid} packet {i} stream {s} (lockstep)")
            assert not any(skew.send([pk[i + s] for s in range(ns)]))
            skew.decode()
            for s in range(ns):
                w = refk[s].decode(pk[i + s])
                g = skew.read(s)
                assert g.shape == w.shape, (vid, i, s)
                check_pcm(w, g, f"vector {vid} packet {i + s} stream {s} (out of step)")
        # two packets per stream per call on top of several frames per packet
        multi = BatchDecoder(2, 2, 0)
        refm = [orc.Decoder(2), orc.Decoder(2)]
        for i in range(0, 120, 2):
            errs = multi.send_multi([[pk[i], pk[i + 1]], [pk[i], pk[i + 1]]])
            assert not any(any(e) for e in errs)
            multi.decode()
            for s in range(2):
                w = np.concatenate([refm[s].decode(pk[i]), refm[s].decode(pk[i + 1])])
                g = multi.read(s)
                assert g.shape == w.shape, (vid, i, s)
                check_pcm(w, g, f"vector {vid} packets {i}, {i + 1} stream {s} (send_multi)")


@pytest.mark.gpu
def test_uniform_send_finished_on_the_threads():
    """CELT-only sends of equal frame count are merged and sorted by the parse threads (engine.cpp merge_uniform): the
    records, PCM and error codes must be those of the general path -- here forced by one stream that brings a different
    frame count -- and of send_packed (pointer array computed from a numpy array); the K1 work list is checked inside
    the library on every send of the test session (OPUSGPU_CHECK_SORT, tests/conftest.py)."""
    assert os.environ.get("OPUSGPU_CHECK_SORT")
    ns, nf = 301, 5                                      # 301 streams: the workers' blocks differ in size
    pk = synth.celt_fb_stereo_packets(range(ns), nf, 160)
    a, b, c = BatchDecoder(ns, 2, 0), BatchDecoder(ns, 2, 0), BatchDecoder(ns + 1, 2, 0)
    two = bytes([0xFD]) + pk[0, 0, 1:].tobytes()         # code 1: two frames of 80 bytes in the extra stream
    for f in range(nf):
        rows = [pk[s, f].tobytes() for s in range(ns)]
        assert not any(a.send(rows))
        assert not b.send_packed(np.ascontiguousarray(pk[:, f])).any()
        assert not any(c.send(rows + [two]))
        sa, sb, sc = a.sideinfo(), b.sideinfo(), c.sideinfo()
        assert bytes(sa[0]) == bytes(sb[0]) and bytes(sa[1]) == bytes(sb[1])
        assert len(sc[0]) == ns + 2
        for d in (a, b, c):
            d.decode()
        for s in (0, 1, 150, 299, 300):
            ga, gb, gc = a.read(s), b.read(s), c.read(s)
            assert ga.shape == (960, 2) and np.array_equal(ga, gb) and np.array_equal(ga, gc), (f, s)
    # several packets per stream per call: [n, T, bytes]
    m1, m2 = BatchDecoder(ns, 2, 0), BatchDecoder(ns, 2, 0)
    e1 = m1.send_multi([[pk[s, f].tobytes() for f in range(4)] for s in range(ns)])
    e2 = m2.send_packed(np.ascontiguousarray(pk[:, :4]))
    assert not any(any(e) for e in e1) and not e2.any()
    assert bytes(m1.sideinfo()[0]) == bytes(m2.sideinfo()[0])
    m1.decode(), m2.decode()
    for s in (0, 77, 300):
        assert np.array_equal(m1.read(s), m2.read(s)) and m1.read(s).shape == (4 * 960, 2)
