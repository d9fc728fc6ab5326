"""Multistream packets and channel mapping families 1 / 255 (RFC 7845 5.1.1), where the reference ends in unimplemented!()
(decoder.rs:151-157, :170-171).  The packets are assembled here from the reference's own vectors: one coupled (stereo)
stream and one mono stream (fed stereo-coded packets, which its mono decoder down-mixes) per packet, the first made self-delimiting (RFC 6716 Appendix B) by the writer below; the
expected output is the two streams decoded on their own by the oracle, dealt to the channels by the mapping table."""
import struct

import numpy as np
import pytest

import orc
from gpu_util import check_pcm
from opus_amd import Decoder, _abi as A, parse_packet


def frame_len_bytes(n):
    return bytes([n]) if n < 252 else bytes([252 + (n & 3), (n - (252 + (n & 3))) >> 2])


def self_delimit(pkt):
    """the ordinary packet pkt as a self-delimiting one (RFC 6716 Appendix B): the last frame's length is written as well"""
    p = parse_packet(pkt)
    toc, code = pkt[0], pkt[0] & 3
    frames = [pkt[o: o + n] for o, n in p["frames"]]
    if code == 0:
        return bytes([toc]) + frame_len_bytes(len(frames[0])) + frames[0]
    if code == 1:
        return bytes([toc]) + frame_len_bytes(len(frames[0])) + frames[0] + frames[1]
    if code == 2:
        return bytes([toc]) + frame_len_bytes(len(frames[0])) + frame_len_bytes(len(frames[1])) + frames[0] + frames[1]
    fc = pkt[1]
    head = bytes([toc, fc])
    pos = 2
    if fc & 0x40:                                       # padding length bytes stay where they are
        while True:
            b = pkt[pos]
            head += bytes([b])
            pos += 1
            if b != 255:
                break
    pad = pkt[len(pkt) - p["padding"]:] if p["padding"] else b""
    if fc & 0x80:
        lens = b"".join(frame_len_bytes(len(f)) for f in frames)
    else:
        lens = frame_len_bytes(len(frames[0]))
    return head + lens + b"".join(frames) + pad


def test_self_delimit_writer_lengths():
    assert frame_len_bytes(0) == b"\x00" and frame_len_bytes(251) == b"\xfb"
    for n in (252, 253, 255, 256, 1000, 1275):
        b = frame_len_bytes(n)
        assert len(b) == 2 and 252 <= b[0] and 4 * b[1] + b[0] == n


def opus_head(channels, family, streams, coupled, mapping):
    return b"OpusHead" + bytes([1, channels]) + struct.pack("<HIh", 0, 48000, 0) + bytes([family, streams, coupled]) + bytes(mapping)


def pairs(n):
    """(packet of the coupled stream, packet of the mono stream) of equal duration, both from vector 11 (stereo CELT, framing
    codes 0-3, 20-60 ms per packet): the mono stream's are stereo-coded, so its mono decoder has to down-mix"""
    v = [bytes(p) for p in orc.load_vectors(ids=[11])[11][1]]
    dur = [parse_packet(p)["nframes"] * parse_packet(p)["frame_size"] for p in v]
    out, used = [], set()
    for i in range(len(v)):
        j = next((j for j in range(len(v) // 2, len(v)) if j not in used and j != i and dur[j] == dur[i]), None)
        if j is None:
            continue
        used.add(j)
        out.append((v[i], v[j]))
        if len(out) == n:
            break
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("family,mapping", [(255, [0, 2, 1, 255]), (1, [0, 2, 1])])
def test_multistream_packets(family, mapping):
    pk = pairs(160)
    assert len(pk) >= 100 and len({p[0][0] & 3 for p in pk}) >= 2, "several framing codes among the coupled stream's packets"
    d = Decoder(0)
    d.set_extradata(opus_head(len(mapping), family, 2, 1, mapping))
    d.configure()
    ra, rb = orc.Decoder(2), orc.Decoder(1)
    for i, (pa, pb) in enumerate(pk):
        lost = i % 53 == 40
        d.send_packet(b"" if lost else self_delimit(pa) + pb)
        out = d.receive_frame()
        wa, wb = ra.decode(b"" if lost else pa), rb.decode(b"" if lost else pb)
        dec = np.concatenate([wa, wb], axis=1)          # the decoded channel set: coupled stream L, R, then the mono stream
        want = np.stack([dec[:, m] if m != 255 else np.zeros(len(dec), np.float32) for m in mapping], axis=1)
        assert out.shape == want.shape, f"packet {i}"
        check_pcm(want, out, f"packet {i}{' (lost)' if lost else ''}")


@pytest.mark.gpu
def test_multistream_flush_resets_every_stream():
    """flush() on a multistream decoder (a seek): every stream of both batches starts over -- the packets that follow decode
    exactly as on a fresh decoder, on the coupled stream and on the mono stream"""
    pk = pairs(40)
    mapping = [0, 2, 1]
    head = opus_head(3, 1, 2, 1, mapping)
    d, fresh = Decoder(0), Decoder(0)
    for x in (d, fresh):
        x.set_extradata(head)
        x.configure()
    for pa, pb in pk[:20]:                              # history of its own
        d.send_packet(self_delimit(pa) + pb)
        d.receive_frame()
    d.flush()
    for i, (pa, pb) in enumerate(pk[20:]):
        p = self_delimit(pa) + pb
        d.send_packet(p)
        fresh.send_packet(p)
        a, b = d.receive_frame(), fresh.receive_frame()
        assert np.array_equal(a, b), f"packet {i} after the flush: max diff {np.abs(a - b).max()}"


@pytest.mark.gpu
def test_one_stream_family_255_goes_through_the_mapping_table():
    """family 255 with ONE mono stream on two output channels, mapping {255, 0}: silence left, the stream right (such a header
    used to take the plain stereo path, which ignores the table)"""
    v = [bytes(p) for p in orc.load_vectors(ids=[11], max_packets=30)[11][1]]
    d = Decoder(0)
    d.set_extradata(opus_head(2, 255, 1, 0, [255, 0]))
    d.configure()
    ref = orc.Decoder(1)
    for i, p in enumerate(v):
        d.send_packet(p)
        out, w = d.receive_frame(), ref.decode(p)
        assert out.shape == (len(w), 2) and not out[:, 0].any(), f"packet {i}"
        check_pcm(w[:, 0], out[:, 1], f"packet {i}")


def test_family_1_with_more_than_8_channels_is_rejected():
    """RFC 7845 5.1.1.2: mapping family 1 is defined for 1..8 channels"""
    import ctypes as C
    lib = A.load()
    out = A.OpusHead()
    head = opus_head(9, 1, 9, 0, list(range(9)))
    assert lib.opusgpu_opus_head_parse(head, len(head), C.byref(out)) == A.CONFIG_INVALID
    head = opus_head(8, 1, 8, 0, list(range(8)))
    assert lib.opusgpu_opus_head_parse(head, len(head), C.byref(out)) == A.OK


def test_undelimit_round_trip():
    """host framing: self-delimiting packet -> ordinary packet gives back the original bytes and the right length"""
    import ctypes as C
    lib = A.load()
    for vid in (1, 10, 11):
        for p in orc.load_vectors(ids=[vid], max_packets=400)[vid][1]:
            p = bytes(p)
            sd = self_delimit(p) + b"\xAA\xBB\xCC"      # something behind it, as the next stream would be
            out = (C.c_uint8 * 2048)()
            used, n = C.c_int(), C.c_size_t()
            rc = lib.opusgpu_packet_undelimit(sd, len(sd), out, 2048, C.byref(n), C.byref(used))
            assert rc == 0 and used.value == len(sd) - 3
            got = bytes(out[: n.value])
            a, b = parse_packet(got), parse_packet(p)
            assert (a["nframes"], a["frame_size"], a["code"]) == (b["nframes"], b["frame_size"], b["code"])
            assert [got[o: o + n] for o, n in a["frames"]] == [p[o: o + n] for o, n in b["frames"]]


def fuzz_multistream(seed, npackets):
    """random multistream decoders (1-4 streams, any number of them coupled, mapping family 255 with a random table -- silent
    channels and channels fed twice included) on random packets: per packet one duration for all streams, every stream its own
    mode / bandwidth / stereo flag / framing code (0-2), all but the last self-delimited; losses strewn in.  Expected: each
    stream decoded on its own by the oracle, dealt to the channels by the table.  (Also run for many seeds by tools/fuzz_gpu.py.)"""
    from gpu_util import RecentPeak
    from opus_amd.decoder import OpusError
    from test_emu_silk import random_packet
    rng = np.random.default_rng(seed)
    done = 0
    while done < npackets:
        ns = int(rng.integers(1, 5))
        nc = int(rng.integers(0, ns + 1))
        nch = int(rng.integers(1, 9))
        mapping = [255 if rng.integers(0, 6) == 0 else int(rng.integers(0, ns + nc)) for _ in range(nch)]
        d = Decoder(0)
        d.set_extradata(opus_head(nch, 255, ns, nc, mapping))
        d.configure()
        refs = [orc.Decoder(2 if s < nc else 1) for s in range(ns)]
        recent = RecentPeak()
        for i in range(int(rng.integers(20, 60))):
            ms = [10, 20, 20, 20, 2.5, 5, 40, 60][int(rng.integers(0, 8))]
            code = int(rng.integers(0, 3))
            parts = []
            for s in range(ns):
                if ms in (10, 20):
                    k = 0 if ms == 10 else 1
                    config = int(rng.choice([0 + k, 4 + k, 8 + k, 12 + k, 14 + k, 18 + k, 22 + k, 26 + k, 30 + k]))
                elif ms in (2.5, 5):
                    config = int(rng.choice([16, 20, 24, 28])) + (0 if ms == 2.5 else 1)
                else:
                    config = int(rng.choice([0, 4, 8])) + (2 if ms == 40 else 3)
                parts.append(random_packet(rng, config, int(rng.integers(0, 2)), code))
            lost = rng.integers(0, 25) == 0
            pkt = b"" if lost else b"".join(self_delimit(p) for p in parts[:-1]) + parts[-1]
            what = f"seed {seed} decoder ({ns} streams, {nc} coupled, mapping {mapping}) packet {i}"
            try:
                d.send_packet(pkt)
            except OpusError:
                with pytest.raises(ValueError):         # (a hybrid frame refused midway: tests/test_gpu_opus.py::fuzz_one_stream)
                    for s in range(ns):
                        refs[s].decode(parts[s])
                break                                   # a fresh decoder set
            out = d.receive_frame()
            w = [refs[s].decode(b"" if lost else parts[s]) for s in range(ns)]
            assert len({len(x) for x in w}) == 1, what
            dec = np.concatenate(w, axis=1)
            want = np.stack([dec[:, m] if m != 255 else np.zeros(len(dec), np.float32) for m in mapping], axis=1)
            assert out.shape == want.shape, what
            check_pcm(want, out, what, recent.value())
            recent.push(dec)
            done += 1
        d.close()


@pytest.mark.gpu
def test_multistream_random_decoders_and_packets():
    fuzz_multistream(5, 400)
