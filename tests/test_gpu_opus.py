"""GPU parity over every mode: all 12 RFC 6716 vectors (SILK NB/MB/WB, hybrid SWB/FB, CELT, mode switching) through
the one-stream C ABI, SILK goldens through the raw launch, and a mixed batch.  Oracle = oracle/liborc.so."""
import ctypes as C

import numpy as np
import pytest

import orc
from gpu_util import HostParser, RecentPeak, check_pcm
from opus_amd import BatchDecoder, Decoder, _abi as A, parse_packet

pytestmark = pytest.mark.gpu
K = orc.kats()


@pytest.mark.parametrize("vid,n,exact", [(2, 1185, True), (3, 998, True), (4, 1265, True), (12, 1332, False), (5, 800, False),
                                         (6, 800, False), (8, 1247, False), (9, 700, False), (10, 900, False)])
def test_vectors_single_stream(vid, n, exact):
    extra, pk = orc.load_vectors(ids=[vid], max_packets=n)[vid]
    d = Decoder(0)
    d.set_extradata(extra)
    d.configure()
    ref = orc.Decoder(2)
    for i, p in enumerate(pk):
        d.send_packet(p)
        out, r = d.receive_frame(), ref.decode(p)
        if exact:                       # SILK-only: same operation order as the oracle -> identical bits
            assert np.array_equal(out, r), f"vector {vid} packet {i}: max diff {np.abs(out - r).max()}"
        else:
            check_pcm(r, out, f"vector {vid} packet {i}")


@pytest.mark.parametrize("name,nframes,fs", [("silk_midonly_to_stereo", 3, 160), ("silk_unmix", 1, 160)])
def test_reference_silk_goldens_on_gpu(name, nframes, fs):
    """silk/tests.rs:22-514, :516-845: the reference's own golden PCM, exact f32 equality, from the HIP kernels"""
    import torch
    lib = A.load()
    k = K[name]
    hp = HostParser(2)
    n, _, _, ns = hp.parse(bytes(k["packet"]))
    assert n == 1 and ns == nframes
    dev = torch.device("cuda:0")
    state = torch.zeros(lib.opusgpu_silk_state_bytes(1), dtype=torch.uint8, device=dev)
    pcm = torch.zeros(5760 * 2, dtype=torch.float32, device=dev)
    left, right = [], []
    for j in range(ns):
        hp.silk[j].stream, hp.silk[j].pcm_off = 0, 0
        rec = torch.frombuffer(bytearray(bytes(hp.silk[j])), dtype=torch.uint8).to(dev)
        internal = torch.zeros((2, 320), dtype=torch.float32, device=dev)
        rc = lib.opusgpu_launch_silk(rec.data_ptr(), 1, state.data_ptr(), 1, pcm.data_ptr(), 5760 * 2, internal.data_ptr(), None)
        assert rc == 0, lib.opusgpu_last_error()
        torch.cuda.synchronize()
        x = internal.cpu().numpy()
        left.append(x[0, :fs].copy())
        right.append(x[1, :fs].copy())
    assert np.array_equal(np.concatenate(right), np.array(k["right"], np.float32))
    assert np.array_equal(np.concatenate(left), np.array(k["left"], np.float32))


@pytest.mark.parametrize("vid", list(range(1, 13)))
def test_opus_compare_gpu_vs_oracle(vid):
    """the RFC's own conformance metric (tools/opus_compare.py; no `.dec` files exist here, so the oracle's PCM stands
    in as the reference): quality 100 = indistinguishable, >= 0 passes; the GPU path must be at 99.5 or better (isolated 1-LSB rounding flips of the 16-bit conversion cost about 0.1)"""
    import os
    import sys
    sys.path.insert(0, os.path.join(orc.ROOT, "tools"))
    import opus_compare as oc
    extra, pk = orc.load_vectors(ids=[vid], max_packets=150)[vid]
    d = Decoder(0)
    d.set_extradata(extra)
    d.configure()
    ref = orc.Decoder(2)
    out, exp = [], []
    for p in pk:
        d.send_packet(p)
        out.append(d.receive_frame())
        exp.append(ref.decode(p))
    out, exp = np.concatenate(out), np.concatenate(exp)
    q, err = oc.compare(oc.to_pcm16(exp), oc.to_pcm16(out), 48000, True)
    assert q >= 99.5, f"vector {vid}: opus_compare quality {q:.2f} (error {err:.4g})"


def test_container_to_pcm_with_pre_skip_and_gain():
    """file image -> demux -> decoder, with an OpusHead that asks for pre-skip and output gain (the reference reads both
    and ignores them, decoder.rs:137-150; RFC 7845 wants them applied)"""
    import struct
    from test_container import make_mka, make_ogg
    from opus_amd import Demuxer
    extra, pk = orc.load_vectors(ids=[10], max_packets=80)[10]
    head = extra[:10] + struct.pack("<HIh", 1000, 48000, -1536) + extra[18:]     # 1000 samples, -6 dB
    ref = orc.Decoder(2)
    exp = np.concatenate([ref.decode(p) for p in pk])[1000:] * np.float32(10.0 ** (-1536 / 5120.0))
    for image in (make_mka(head, pk, lacing="xiph"), make_ogg(head, pk)):
        dm = Demuxer(image)
        d = Decoder(0)
        d.set_extradata(dm.extradata)
        d.configure()
        out = []
        for p in dm:
            d.send_packet(p)
            out.append(d.receive_frame())
        out = np.concatenate(out)
        check_pcm(exp, out, "pre-skip + gain")
        d.flush()                                   # a flushed stream starts over: pre-skip applies again
        d.send_packet(pk[0])
        info = parse_packet(pk[0])
        assert d.receive_frame().shape[0] == max(0, info["frame_size"] * info["nframes"] - 1000)


def test_batch_all_modes():
    """BASELINE config 5 shape in small: the 12 vectors as independent streams of one batch, 60 packets each"""
    V = orc.load_vectors()
    ids = sorted(V)
    ns, steps = 2 * len(ids), 60
    dec = BatchDecoder(ns, 2, 0)
    refs = [orc.Decoder(2) for _ in range(ns)]
    start = [0 if s < len(ids) else 211 for s in range(ns)]
    for t in range(steps):
        pk = [V[ids[s % len(ids)]][1][start[s] + t] for s in range(ns)]
        assert not any(dec.send(pk))
        dec.decode()
        for s in range(ns):
            check_pcm(refs[s].decode(pk[s]), dec.read(s), f"stream {s} (vector {ids[s % len(ids)]}) step {t}")


def test_silk_wb_mono_batch_c3_shape():
    """BASELINE config 3 shape: SILK-only WB mono 20 ms packets of vectors 04/12 looped with per-stream offsets"""
    V = orc.load_vectors(ids=[4, 12])
    pool = [p for vid in (4, 12) for p in V[vid][1] if (p[0] >> 3) == 9 and not (p[0] & 4) and (p[0] & 3) == 0]
    assert len(pool) > 100
    ns, steps = 512, 8
    dec = BatchDecoder(ns, 1, 0)
    sample = list(range(0, ns, 37))
    refs = {s: orc.Decoder(1) for s in sample}
    for t in range(steps):
        pk = [pool[(s * 7 + t) % len(pool)] for s in range(ns)]
        assert not any(dec.send(pk))
        dec.decode()
        for s in sample:
            out, r = dec.read(s), refs[s].decode(pk[s])
            assert np.array_equal(out, r), f"stream {s} step {t}"


def test_silk_wb_mono_c3_full_size_16384_streams():
    """BASELINE config 3 at its full size: 16 384 SILK-only WB mono 20 ms streams in one batch, four steps; 64 sampled streams
    (spread over the wavefronts' eight channel slots) bit-exact against the oracle, and streams fed the same packets equal"""
    V = orc.load_vectors(ids=[4, 12])
    pool = [p for vid in (4, 12) for p in V[vid][1] if (p[0] >> 3) == 9 and not (p[0] & 4) and (p[0] & 3) == 0]
    ns, steps, period = 16384, 4, 2048
    dec = BatchDecoder(ns, 1, 0)
    sample = sorted(set(list(range(0, ns, 293)) + [1, 7, 8, ns - 1]))
    refs = {s: orc.Decoder(1) for s in sample}
    exact = total = 0
    for t in range(steps):
        pk = [pool[((s % period) * 7 + t) % len(pool)] for s in range(ns)]
        assert not any(dec.send(pk))
        dec.decode()
        for s in sample:
            out, r = dec.read(s), refs[s].decode(pk[s])
            exact += int(np.array_equal(out, r))
            total += 1
            check_pcm(r, out, f"stream {s} step {t}")     # (a SILK packet that ends in a redundant CELT frame is held to the CELT bar)
        for s in (0, 5, 1000, 2047):                      # same packets, other slots of other wavefronts
            assert np.array_equal(dec.read(s), dec.read(s + period)) and np.array_equal(dec.read(s), dec.read(s + 7 * period))
    assert exact >= 0.9 * total, (exact, total)             # plain SILK frames are bit-exact


@pytest.mark.parametrize("stereo", [0, 1])
def test_random_silk_payloads_with_lbrr_on_gpu(stereo):
    """any bytes are a valid SILK packet and about half carry LBRR (in-band FEC) flags, where the reference returns
    Unsupported (silk/decoder.rs:2070-2072): the LBRR frames are read past (RFC 6716 4.2.4-4.2.7; tests/test_lbrr.py pins
    the layout with re-encoded packets).  One-stream C ABI -> HIP kernels, bit-exact against the oracle, all 12 SILK configs."""
    L = orc.lib()
    seen = C.c_long.in_dll(L, "orc_silk_lbrr_packets")
    before = seen.value
    rng = np.random.default_rng(77 + stereo)
    extra, _ = orc.load_vectors(ids=[2], max_packets=1)[2]
    d = Decoder(0)
    d.set_extradata(extra)
    d.configure()
    ref = orc.Decoder(2)
    hp = HostParser(2)                  # the same parse on the side: tells which packets carry CELT records
    for i in range(96):
        config = i % 12
        n = int(rng.integers(20, 120)) * (1, 1, 2, 3)[config & 3]
        p = bytes([config << 3 | stereo << 2]) + rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        d.send_packet(p)
        out, r = d.receive_frame(), ref.decode(p)
        # most random payloads leave bits behind the SILK frames, i.e. carry a (random) redundant CELT frame: CELT bar there
        nfr, *_ = hp.parse(p)
        if not any(hp.fi[f].has_celt for f in range(nfr)):
            assert np.array_equal(out, r), f"packet {i} (config {config}): max diff {np.abs(out - r).max()}"
        else:
            check_pcm(r, out, f"packet {i} (config {config})")
    assert seen.value - before >= 20


@pytest.mark.parametrize("vid,npk", [(1, 200), (4, 200), (6, 200), (10, 1250), (12, 300)])
def test_lost_packets_and_redundant_frames_on_gpu(vid, npk):
    """packets reported lost (len 0) are concealed in the mode of the last packet -- SILK frames made up from the last frame's
    parameters, CELT noise under decaying band energies, both for hybrid; single and double losses and a burst of ten -- and the
    redundant CELT frames of the mode switches (vector 10: packets 965 ..; vector 12: 136 ..) are decoded and cross-faded (RFC
    6716 4.5.1).  Neither exists in the reference (packet.rs:215-225 panics, decoder.rs:84-115 drops): HIP path against the oracle."""
    extra, pk = orc.load_vectors(ids=[vid], max_packets=npk)[vid]
    d = Decoder(0)
    d.set_extradata(extra)
    d.configure()
    ref = orc.Decoder(2)
    for i, p in enumerate(pk):
        lost = i % 41 == 30 or i % 41 == 31 or 100 <= i < 110
        d.send_packet(b"" if lost else p)
        out, r = d.receive_frame(), ref.decode(b"" if lost else p)
        assert out.shape == r.shape, f"vector {vid} packet {i}"
        check_pcm(r, out, f"vector {vid} packet {i}{' (lost)' if lost else ''}")


def test_hybrid_to_silk_flush_behind_redundant_head_on_gpu():
    """a hybrid -> SILK switch that carries a CELT -> SILK redundant frame (crafted from vector 12's packets, see
    tests/test_emu_silk.py): redundant frame and overlap flush run in rounds of their own, one after the other, and the
    result is the oracle's -- several times, on fresh decoders: nothing may depend on workgroup timing"""
    from test_emu_silk import crafted_hybrid_to_silk_with_redundant_head
    seq = crafted_hybrid_to_silk_with_redundant_head()
    extra, _ = orc.load_vectors(ids=[12], max_packets=1)[12]
    for rep in range(4):
        d = Decoder(0)
        d.set_extradata(extra)
        d.configure()
        ref = orc.Decoder(2)
        for i, p in enumerate(seq):
            d.send_packet(p)
            check_pcm(ref.decode(p), d.receive_frame(), f"rep {rep} crafted packet {i}")


def test_mixed_modes_c4_shape_8192_streams():
    """BASELINE config 4 at full size through layer 2: 8192 streams on vectors 05 / 06 / 07 / 10 (hybrid SWB / FB, CELT of every
    frame size with transients, hybrid <-> CELT switching with redundant frames), one packet per stream per step.  Every step is
    the SILK launches by depth + ONE CELT launch whose tail walks the streams' runs (1 .. 8+ records each); a sample of streams
    is compared with the oracle at every step, the rest must be finite."""
    import torch
    V = orc.load_vectors(ids=[5, 6, 7, 10])
    vids = (5, 6, 7, 10)
    ns, steps = 8192, 6
    dec = BatchDecoder(ns, 2, 0)

    def packet(s, k):
        v = V[vids[s % 4]][1]
        return v[((s // 4) * 7919 + k) % len(v)]
    sample = sorted({0, 1, 2, 3, 1366, 2731, 4097, 6002, 8190, 8191})
    refs = {s: orc.Decoder(2) for s in sample}
    lens_seen = set()
    for k in range(steps):
        pk = [packet(s, k) for s in range(ns)]
        assert not any(dec.send(pk))
        pl = dec.plan()
        lens_seen |= set(np.diff(pl["run_off"]).tolist())
        dec.decode()
        for s in sample:
            check_pcm(refs[s].decode(pk[s]), dec.read(s), f"stream {s} (vector {vids[s % 4]}) step {k}")
    assert len(lens_seen) > 2, "runs of several lengths in one launch"


def test_all_vectors_c5_shape_6144_streams():
    """BASELINE config 5 at full size through layer 2: the twelve RFC vectors x 512 replicas, each stream from its own offset, one
    packet per stream per step -- SILK NB / MB / WB with 10-60 ms packets (depths 1-3), hybrid, CELT of every size, the mode
    switches of vectors 08-10 with their redundant frames, all in the same launches.  One stream of every vector and a few more are
    compared with the oracle at every step."""
    vids = tuple(range(1, 13))
    V = orc.load_vectors(ids=list(vids))
    ns, steps = 12 * 512, 8
    dec = BatchDecoder(ns, 2, 0)

    def packet(s, k):
        v = V[vids[s % 12]][1]
        return v[((s // 12) * 7919 + k) % len(v)]
    sample = sorted(set(range(12)) | {2047, 3073, 4100, 6142, 6143})
    refs = {s: orc.Decoder(2) for s in sample}
    recent = {s: RecentPeak() for s in sample}
    for k in range(steps):
        pk = [packet(s, k) for s in range(ns)]
        assert not any(dec.send(pk))
        dec.decode()
        for s in sample:
            want = refs[s].decode(pk[s])
            check_pcm(want, dec.read(s), f"stream {s} (vector {vids[s % 12]}) step {k}", recent[s].value())
            recent[s].push(want)
    dec.close()


@pytest.mark.parametrize("vid,npk", [(2, 300), (4, 400)])
def test_fec_recovery_on_gpu(vid, npk):
    """in-band FEC through the one-stream C ABI (opusgpu_decoder_send_packet_fec): every third packet is lost and recovered from
    a copy of its frames carried as LBRR frames by its successor (rebuilt with a range encoder, tests/test_lbrr.py); the HIP
    kernels must return the lost packet's PCM bit for bit (SILK), then the successor's.  The reference stops at the LBRR flag
    (silk/decoder.rs:2070-2072)."""
    from test_lbrr import traced_symbols, with_fec_copy
    extra, pk = orc.load_vectors(ids=[vid], max_packets=npk)[vid]
    pk = [bytes(p) for p in pk]
    d = Decoder(0)
    d.set_extradata(extra)
    d.configure()
    plain = orc.Decoder(2)
    recovered, i = 0, 0

    def feed(p, want, what):
        d.send_packet(p)
        out = d.receive_frame()
        assert np.array_equal(out, want), f"{what}: max diff {np.abs(out - want).max()}"
    while i < len(pk):
        want_i, syms_i = traced_symbols(plain, pk[i])
        if i + 1 < len(pk) and i % 3 == 1:
            want_n, syms_n = traced_symbols(plain, pk[i + 1])
            q = with_fec_copy(pk[i + 1], syms_n, pk[i], syms_i)
            if q is None:
                feed(pk[i], want_i, f"packet {i}")
                feed(pk[i + 1], want_n, f"packet {i + 1}")
            else:
                d.send_packet_fec(q)
                out = d.receive_frame()
                assert np.array_equal(out, want_i), f"vector {vid}: packet {i} recovered from the LBRR frames of {i + 1}"
                feed(q, want_n, f"packet {i + 1} behind its LBRR frames")
                recovered += 1
            i += 2
            continue
        feed(pk[i], want_i, f"packet {i}")
        i += 1
    assert recovered >= 20, recovered


def test_celt_loopback_on_gpu():
    """encode -> decode loopback (tests/test_loopback.py: a known signal encoded into CELT packets by the oracle's decoder run
    backwards over a textbook f64 MDCT): the HIP path must give the SOURCE back (SNR >= 20 dB at 64 kb/s per channel, the same
    alignment to the sample), beside matching the oracle"""
    import test_loopback as TL
    nframes = 24
    for tools, floor in (({}, 20.0), (dict(transient=True, joint=True, spread=2), 15.0),       # plain; short blocks + joint stereo + spreading;
                         (dict(postfilter=(37, 7, 1)), 19.0), (dict(postfilter=(733, 6, 0)), 18.0),     # the comb filter's register and LDS paths
                         (dict(tf=2, joint=True, spread=2), 15.0)):                                     # Haar / Hadamard band transforms
        src, packets, _ = TL.encode_stream(2, nframes, 320, **tools)
        d = Decoder(0)
        d.configure()
        ref = orc.Decoder(2)
        outs = []
        for i, p in enumerate(packets):
            d.send_packet(p)
            out = d.receive_frame()
            check_pcm(ref.decode(p), out, f"loopback frame {i}")
            outs.append(out)
        got = np.concatenate(outs)[2 * TL.N:]
        want = TL.aligned(src, nframes)
        assert TL.snr_db(want, got) >= floor, (tools, TL.snr_db(want, got))
        assert TL.snr_db(want[1:], got[:-1]) < TL.snr_db(want, got) - 3.0          # (a one-sample slip would show)
        d.close()



def fuzz_one_stream(channels, seed, npackets):
    """random packets under every TOC (losses and FEC recoveries strewn in) through the one-stream C ABI against the oracle
    (also run for many seeds by tools/fuzz_gpu.py).  A random packet is now and then one that the decoder must refuse (a hybrid
    frame whose redundancy length leaves no room for the frame itself: decoder.rs:100-104): then both refuse it, and both start
    again from a fresh state, since what a refused packet leaves behind is not defined."""
    from test_emu_silk import random_packet
    from opus_amd.decoder import OpusError
    rng = np.random.default_rng(seed)

    def fresh():
        d = Decoder(0)
        if channels == 1:
            d.set_extradata(b"OpusHead" + bytes([1, 1, 0, 0]) + (48000).to_bytes(4, "little") + bytes([0, 0, 0]))
        d.configure()
        return d, orc.Decoder(channels)
    d, ref = fresh()
    refused, recent = 0, RecentPeak()
    for i in range(npackets):
        config, stereo, code = int(rng.integers(0, 32)), int(rng.integers(0, 2)), int(rng.integers(0, 4))
        what = f"seed {seed} packet {i} (config {config}, stereo {stereo}, code {code})"
        p = random_packet(rng, config, stereo, code)
        try:
            if rng.integers(0, 25) == 0:
                p = b""                                 # lost
            elif rng.integers(0, 12) == 0:              # the previous one "was lost": FEC recovery from this one first
                d.send_packet_fec(p)
                r = ref.decode_fec(p)
                check_pcm(r, d.receive_frame(), what + ", FEC", recent.value())
                recent.push(r)
            d.send_packet(p)
        except OpusError:
            with pytest.raises(ValueError):             # the oracle refuses it as well (FEC recovery included)
                ref.decode_fec(p)
                ref.decode(p)
            refused += 1
            d.close()
            d, ref = fresh()
            recent = RecentPeak()
            continue
        out, r = d.receive_frame(), ref.decode(p)
        assert out.shape == r.shape, what
        check_pcm(r, out, what, recent.value())
        recent.push(r)
    d.close()
    assert refused <= npackets // 100, refused


@pytest.mark.parametrize("channels", [1, 2])
def test_random_packets_every_toc_on_gpu(channels):
    """tests/test_emu_silk.py::test_random_packets_every_toc through the one-stream C ABI and the HIP kernels: every TOC
    configuration, mono / stereo packets, all frame-count codes, random contents, one decoder (every mode switch), against the oracle"""
    fuzz_one_stream(channels, 200 + channels, 600)


def fuzz_batch_engine(seed, nsteps, ns=96, wide=False, channels=2):
    """the batch engine under the same fuzz (also run for many seeds by tools/fuzz_gpu.py; `wide`: sends of 1-8 packets per
    stream, and every third step the same CELT-only configuration and frame count for every stream -- the sends that become one
    launch with several frames per workgroup)"""
    from test_emu_silk import random_packet
    rng = np.random.default_rng(seed)
    dec = BatchDecoder(ns, channels, 0)
    refs = [orc.Decoder(channels) for _ in range(ns)]
    refused, recent = [0], [RecentPeak() for _ in range(ns)]

    def refuse(s, packets, fec=False):
        """the engine refused a packet of stream s (see fuzz_one_stream): so does the oracle; both start the stream afresh"""
        with pytest.raises(ValueError):
            for q in packets:
                refs[s].decode_fec(q) if fec else refs[s].decode(q)
        refs[s] = orc.Decoder(channels)
        recent[s] = RecentPeak()
        dec.flush(s)
        refused[0] += 1
    for step in range(nsteps):
        ids = [s for s in range(ns) if rng.integers(0, 5) != 0]
        T = 2 if step % 7 == 3 else 1
        if wide and step % 5 != 2:
            T = int(rng.choice([1, 1, 2, 3, 4, 8]))
        if wide and step % 3 == 0:
            config, code = int(rng.integers(16, 32)), int(rng.integers(0, 4))
            pk = [[random_packet(rng, config, int(rng.integers(0, 2)), code) for _ in range(T)] for _ in ids]
            if code == 3:                               # (code 3 draws its own frame count: keep the streams that drew the first one's)
                n0 = pk[0][0][1] & 63
                keep = [j for j, p in enumerate(pk) if all(q[1] & 63 == n0 for q in p)]
                ids, pk = [ids[j] for j in keep], [pk[j] for j in keep]
                if not ids:
                    continue
        else:
            pk = [[b"" if rng.integers(0, 25) == 0 else random_packet(rng, int(rng.integers(0, 32)), int(rng.integers(0, 2)), int(rng.integers(0, 4)))
                   for _ in range(T)] for _ in ids]
        if T == 1 and step % 5 == 2:                     # every packet of this step follows a lost one: recover first
            fec_ids = [s for s, p in zip(ids, pk) if p[0]]
            fec_pk = [p[0] for p in pk if p[0]]
            errs = dec.send(fec_pk, fec_ids, fec=True)
            dec.decode()
            for s, p, e in zip(fec_ids, fec_pk, errs):
                if e:
                    refuse(s, [p], fec=True)
                else:
                    want = refs[s].decode_fec(p)
                    check_pcm(want, dec.read(s), f"seed {seed} step {step} stream {s} FEC", recent[s].value())
                    recent[s].push(want)
        if T == 1:
            errs = [[e] for e in dec.send([p[0] for p in pk], ids)]
        else:
            errs = dec.send_multi(pk, ids)
        dec.decode()
        for s, p, e in zip(ids, pk, errs):
            if any(e):
                refuse(s, p)
                continue
            want = np.concatenate([refs[s].decode(q) for q in p])
            check_pcm(want, dec.read(s)[: len(want)], f"seed {seed} step {step} stream {s}", recent[s].value())
            recent[s].push(want)
    dec.close()
    assert refused[0] <= 1 + nsteps * ns // 100, refused


def test_batch_engine_random_packets_losses_fec_and_multi_sends():
    """the batch engine under the same fuzz: 96 streams, each step a random subset of them gets a random packet of any TOC (or a
    loss, or an FEC recovery send first), some steps as a two-packets-per-stream send -- runs of uneven length, SILK depths 0-3,
    every launch round -- every stream's PCM against its own oracle decoder after every step"""
    fuzz_batch_engine(77, 50)
