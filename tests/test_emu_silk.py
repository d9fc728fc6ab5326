"""CPU CI of the SILK kernel logic + the whole packet pipeline (host parser -> SILK kernels -> CELT kernel) in the
g++ emulation build (tests/emu), against the oracle and the reference's golden PCM.  See test_emu_celt.py."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from gpu_util import HostParser, check_pcm
from opus_amd import _abi as A, build

K = orc.kats()


@pytest.fixture(scope="module")
def emu():
    lib = C.CDLL(build.build_emu())
    lib.emu_silk_state_floats.restype = C.c_size_t
    lib.emu_launch_silk.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.emu_launch_celt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


class EmuDecoder:
    """host parser + emulated kernels for one stream, in the engine's order (SILK frames, then the CELT part)"""

    def __init__(self, emu, channels):
        self.emu, self.ch = emu, channels
        self.hp = HostParser(channels)
        self.sst = np.zeros(emu.emu_silk_state_floats(1), np.float32)
        self.cst = A.CeltState()
        self.pcm = np.zeros(5760 * channels, np.float32)
        self.internal = np.zeros((2, 320), np.float32)

    def decode(self, p, fec=False):
        hp = self.hp
        n, nc, nl, ns = hp.parse(p, fec)
        self.nframes = n
        if n < 0:
            return n
        off = 0
        for f in range(n):
            fi = hp.fi[f]
            sub = 0
            for rnd in range(4):                        # the engine's launch rounds of one Opus frame (OG_ROUND_*)
                if rnd < fi.nsilk:
                    sf = hp.silk[fi.silk_index + rnd]
                    sf.stream, sf.pcm_off = 0, off + sub
                    sub += sf.f_size * (6, 4, 3)[sf.bandwidth]
                    self.emu.emu_launch_silk(C.byref(sf), 1, self.sst.ctypes.data, 1, self.pcm.ctypes.data, 5760 * self.ch,
                                             self.internal.ctypes.data)
                for c in range(fi.has_celt):            # CELT records: redundant frame in front, own CELT part, redundant frame behind
                    hd = hp.celt[fi.celt_index + c]
                    if hd.spread >> 6 != rnd:
                        continue
                    hd.stream, hd.pcm_off = 0, off + hd.pcm_off
                    self.emu.emu_launch_celt(C.byref(hd), hp.leaves, 1, 1, C.byref(self.cst), self.pcm.ctypes.data, 5760 * self.ch, None)
            off += fi.frame_size
        return self.pcm[: off * self.ch].reshape(off, self.ch).copy()


@pytest.mark.parametrize("name,nframes,fs", [("silk_midonly_to_stereo", 3, 160), ("silk_unmix", 1, 160)])
def test_reference_silk_goldens(emu, name, nframes, fs):
    """silk/tests.rs:22-514 and :516-845: exact f32 equality at the internal rate, through host parser + kernels"""
    k = K[name]
    d = EmuDecoder(emu, 2)
    hp = d.hp
    pkt = bytes(k["packet"])
    n, _, _, ns = hp.parse(pkt)
    assert n == 1 and ns == nframes
    left, right = [], []
    for j in range(ns):
        sf = hp.silk[j]
        sf.stream, sf.pcm_off = 0, 0
        emu.emu_launch_silk(C.byref(sf), 1, d.sst.ctypes.data, 1, d.pcm.ctypes.data, 5760 * 2, d.internal.ctypes.data)
        left.append(d.internal[0, :fs].copy())
        right.append(d.internal[1, :fs].copy())
    assert np.array_equal(np.concatenate(right), np.array(k["right"], np.float32))
    assert np.array_equal(np.concatenate(left), np.array(k["left"], np.float32))


@pytest.mark.parametrize("vid,channels", [(2, 2), (3, 1), (4, 2), (12, 2), (12, 1)])
def test_emu_silk_vectors_bit_exact(emu, vid, channels):
    """SILK-only vectors: the kernels keep the oracle's operation order, so 48 kHz PCM is bit-identical"""
    _, pk = orc.load_vectors(ids=[vid])[vid]
    d, ref = EmuDecoder(emu, channels), orc.Decoder(channels)
    for i, p in enumerate(pk):
        out, r = d.decode(p), ref.decode(p)
        fi = d.hp.fi[0]
        if fi.mode == 0 and not fi.has_celt:           # (a SILK-only packet with a redundant CELT frame is held to the CELT bar)
            assert np.array_equal(out, r), f"vector {vid} packet {i}"
        else:
            check_pcm(r, out, f"vector {vid} packet {i}")


@pytest.mark.parametrize("vid,n", [(5, 700), (6, 700), (8, 1247), (9, 600), (10, 900)])
def test_emu_hybrid_and_mode_switching(emu, vid, n):
    _, pk = orc.load_vectors(ids=[vid], max_packets=n)[vid]
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    for i, p in enumerate(pk):
        check_pcm(ref.decode(p), d.decode(p), f"vector {vid} packet {i}")


def crafted_hybrid_to_silk_with_redundant_head():
    """Packets of vector 12 in an order of our own: three hybrid packets, then a SILK packet that opens with a CELT -> SILK
    redundant frame -- a hybrid -> SILK switch WITH a redundant frame in front, which none of the 12 vectors holds: the
    overlap flush then follows the redundant frame in a round of its own and stores no PCM (stream_parser.cpp)."""
    _, pk = orc.load_vectors(ids=[12])[12]
    return [pk[386], pk[387], pk[388], pk[137], pk[138], pk[139]]


def test_emu_hybrid_to_silk_flush_behind_redundant_head(emu):
    seq = crafted_hybrid_to_silk_with_redundant_head()
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    for i, p in enumerate(seq):
        out, r = d.decode(p), ref.decode(p)
        if i == 3:                                      # one record per round: redundant frame in 0, mute flush in 1
            fi = d.hp.fi[0]
            tags = [(d.hp.celt[fi.celt_index + c].spread >> 6, (d.hp.celt[fi.celt_index + c].spread >> 4) & 3) for c in range(fi.has_celt)]
            assert tags == [(0, 1), (1, 3)], tags
        check_pcm(r, out, f"crafted packet {i}")


def test_lbrr_packets_decode(emu):
    """silk/decoder.rs:2070-2072 stops at an LBRR flag with Error::Unsupported; here the LBRR frames are read past (RFC 6716
    4.2.4-4.2.7, tests/test_lbrr.py): a packet whose LBRR flag is set decodes, and the same in the oracle"""
    hp = HostParser(2)
    d = orc.Decoder(2)
    for b0 in range(256):                               # SILK NB 20 ms mono; find first bytes with the VAD and LBRR bits set
        p = bytes([0x08, b0, 0x55, 0xAA, 0x0F, 0xF0, 0x33, 0xCC])
        n, *_ = hp.parse(p)
        assert n == 1
        assert d.decode(p).shape == (960, 2)


def random_packet(rng, config, stereo, code):
    """a packet with TOC (config, stereo, code) and random frame contents (the range decoder is total: any bytes decode)"""
    ms = (10, 20, 40, 60)[config & 3] if config < 12 else (10, 20)[config & 1] if config < 16 else (2.5, 5, 10, 20)[config & 3]
    size = lambda: int(rng.integers(2, max(4, int(ms * 6))))       # noqa: E731  (bytes per frame: up to ~48 kb/s)
    toc = bytes([config << 3 | stereo << 2 | code])
    body = lambda n: rng.integers(0, 256, n, dtype=np.uint8).tobytes()   # noqa: E731
    if code == 0:
        return toc + body(size())
    if code == 1:
        n = size()
        return toc + body(2 * n)
    if code == 2:
        n1 = min(size(), 251)
        return toc + bytes([n1]) + body(n1) + body(size())
    m = int(rng.integers(1, max(2, int(120 // ms) + 1)))                 # code 3: M frames, at most 120 ms
    m = min(m, 6)
    if rng.integers(0, 2):                                               # CBR
        n = size()
        return toc + bytes([m]) + body(m * n)
    sizes = [min(size(), 251) for _ in range(m)]                         # VBR: the first M - 1 sizes are coded
    return toc + bytes([0x80 | m]) + bytes(sizes[:-1]) + b"".join(body(n) for n in sizes)


@pytest.mark.parametrize("channels", [1, 2])
def test_random_packets_every_toc(emu, channels):
    """every configuration (SILK NB / MB / WB 10-60 ms, hybrid SWB / FB, CELT NB-FB 2.5-20 ms), mono and stereo packets, all four
    frame-count codes, random contents, in random order through ONE decoder -- so every mode switch occurs as well -- with lost
    packets and in-band FEC recoveries strewn in: host parser + kernel bodies against the oracle.  (Run over 10^5 packets while
    it was written; it found a scratch channel the oracle cleared between a packet's LBRR frames.)"""
    rng = np.random.default_rng(100 + channels)
    d, ref = EmuDecoder(emu, channels), orc.Decoder(channels)
    seen = set()
    for i in range(900):
        config, stereo, code = int(rng.integers(0, 32)), int(rng.integers(0, 2)), int(rng.integers(0, 4))
        p = random_packet(rng, config, stereo, code)
        if rng.integers(0, 25) == 0:
            p = b""                                     # a lost packet now and then: concealed in whatever mode came before
        elif rng.integers(0, 12) == 0:                  # ... or the previous one "was lost": in-band FEC recovery from this one first
            check_pcm(ref.decode_fec(p), d.decode(p, fec=True), f"packet {i} (config {config}, stereo {stereo}, code {code}), FEC")
        r = ref.decode(p)
        out = d.decode(p)
        if isinstance(r, int) or r is None:
            assert isinstance(out, int) and out < 0, f"packet {i}: oracle rejects, parser accepts"
            continue
        assert not isinstance(out, int), f"packet {i} (config {config}, code {code}): parser rejects ({out}), oracle accepts"
        check_pcm(r, out, f"packet {i} (config {config}, stereo {stereo}, code {code})")
        if p:
            seen.add((config, stereo))
    assert len(seen) == 64


# A hybrid packet (config 13, mono, code 3) that the decoder must refuse in the middle of a frame: the frame's redundancy length
# leaves no room for the frame itself (decoder.rs:100-104).  Found by the long fuzz runs (about one random hybrid packet in 20 000).
REFUSED_HYBRID_PACKET = bytes.fromhex(
    "6b844743321935fb956732ba1843e717e4a1b64205a825f34e31c89f1fe99c71d6e79323ca5d6c2b41e1a45479fd48ec1dac885c9a116f8c"
    "dbb554824e094b592274ed82d7421d1bddad9af52b201edaa25dd371e573a9a40f1c7a7bcf8365c861bee8ee612865d41748921f4afec4cc"
    "dbe0d47bf2fe90e2cca00316a2d9749c0856715bf02f1df9b89416392cf4f434dc024e29a1d7a4ae06ea47cd1a8c83b869750d60006c3e7f"
    "9c1acb7a125aaa6d84433191b399d9759446f2f862aebbd34f19a456e381fef8d16cfbece161be1f93a247f0f3f859283175df749ca5e09f")


def test_refused_packet_restarts_the_stream(emu):
    """stereo SILK, then a mono hybrid packet refused midway, then stereo SILK again: the host parser must not run ahead of the
    device state (the refused packet's records are dropped) -- the stream starts again, as the oracle does from a fresh decoder;
    the same with an explicit flush after the refusal.  (Before the fix the side channel kept its old history on the device: its
    reset record went out with the refused packet, and flush skipped the channel as never coded.)"""
    for explicit_flush in (False, True):
        rng = np.random.default_rng(31)
        d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
        for i in range(4):
            p = random_packet(rng, 9, 1, 0)
            check_pcm(ref.decode(p), d.decode(p), f"packet {i}")
        assert d.decode(REFUSED_HYBRID_PACKET) < 0
        with pytest.raises(ValueError):
            ref.decode(REFUSED_HYBRID_PACKET)
        if explicit_flush:
            d.hp.reset()
        ref = orc.Decoder(2)
        for i in range(4):
            p = random_packet(rng, 9, 1, int(rng.integers(0, 2)))
            check_pcm(ref.decode(p), d.decode(p), f"packet {i} after the refused one")


def test_hostile_packets_are_refused_or_decoded_alike():
    """tools/fuzz_hostile.py, short: random lengths, random TOCs, runs of 0x00 / 0xFF -- the host parser + kernel bodies and the
    oracle refuse the same packets and agree on the PCM of the others"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_hostile.py"), "3", "400"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "bad 0" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
