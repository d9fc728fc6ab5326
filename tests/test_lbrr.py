"""SILK packets that carry LBRR (in-band FEC) frames: RFC 6716 4.2.4-4.2.7.  The reference stops at them with
Error::Unsupported (silk/decoder.rs:2070-2072) and none of its 12 vectors holds one, so the expected results are made here:

* re-encoding: the oracle logs every symbol it decodes from a real SILK packet; a range ENCODER (RFC 6716 5.1, below) writes
  a new packet with the LBRR flags set and a copy of the regular frames' symbols in front of them as LBRR frames.  A decoder
  that reads past the LBRR frames correctly must produce exactly the PCM of the original packet; one that misreads their
  extent decodes garbage for the regular frames.  Run on the oracle and on the host parser + kernel bodies (g++ emulation).
* random payloads (the range decoder is total: any bytes are a valid packet, half of them with LBRR flags): host parser +
  kernels bit-exact against the oracle, all SILK configurations.
Parity unpinned by the reference (it holds neither code nor data for this); pinned by the round trip above."""
import ctypes as C

import numpy as np
import pytest

import orc
from opus_amd import _abi as A, build
from gpu_util import check_pcm
from test_emu_silk import EmuDecoder


class RangeEncoder:
    """RFC 6716 5.1 (ec_encode / ec_enc_done) on unbounded integers: `low` keeps every carried bit, so no carry buffering"""

    def __init__(self):
        self.low, self.rng, self.shifts = 0, 1 << 31, 0

    def encode(self, fl, fh, ft):
        assert 0 <= fl < fh <= ft
        r = self.rng // ft
        if fl > 0:
            self.low += self.rng - r * (ft - fl)
            self.rng = r * (fh - fl)
        else:
            self.rng -= r * (ft - fh)
        while self.rng <= (1 << 23):
            self.low <<= 8
            self.rng <<= 8
            self.shifts += 1

    def done(self):
        l = 32 - self.rng.bit_length()
        msk = ((1 << 31) - 1) >> l
        end = (self.low + msk) & ~msk
        if (end | msk) >= self.low + self.rng:
            l += 1
            msk >>= 1
            end = (self.low + msk) & ~msk
        nf = max(0, (l + 7) // 8)
        total = self.shifts + nf
        big = end << (8 * nf)
        out = bytes((big >> (23 + 8 * (total - j))) & 0xFF for j in range(total))
        assert (big >> (31 + 8 * total)) == 0, "carry out of the first byte"
        return out


def encode_symbols(syms):
    enc = RangeEncoder()
    for fl, fh, ft in syms:
        enc.encode(int(fl), int(fh), int(ft))
    return enc.done()


def test_range_encoder_round_trip():
    """the encoder above against the oracle's decoder: re-encoding a packet's own symbols gives a packet that decodes alike"""
    _, pk = orc.load_vectors(ids=[2], max_packets=40)[2]
    a, b = orc.Decoder(2), orc.Decoder(2)
    for p in pk:
        p = bytes(p)
        syms = traced_symbols(a, p)[1]
        q = p[:1] + encode_symbols([s for s in syms if s[1] != 0])
        assert np.array_equal(a.last, b.decode(q))


def traced_symbols(dec, pkt):
    """decode pkt with the oracle, returning (pcm, [(low, high, total) ...]) with (0, 0, tag) markers at SILK frame starts (1) and ends (2)"""
    L = orc.lib()
    buf = (C.c_uint32 * (3 * 8192))()
    L.orc_trace_set(buf, 8192)
    try:
        dec.last = dec.decode(pkt)
        n = L.orc_trace_count()
    finally:
        L.orc_trace_set(None, 0)
    assert n < 8192
    arr = np.ctypeslib.as_array(buf).reshape(-1, 3)[:n].copy()
    return dec.last, [tuple(int(v) for v in r) for r in arr]


LBRR40 = [0, 53, 106, 256]
LBRR60 = [0, 41, 61, 90, 131, 146, 174, 256]


def with_lbrr(pkt, syms):
    """pkt: a code-0 SILK-only packet whose VAD flags are all set; syms: its traced symbols.  Returns the packet with every
    regular frame's symbols repeated in front as an LBRR frame (RFC 6716 4.2.3-4.2.5), or None if the packet does not qualify."""
    toc = pkt[0]
    config, stereo, code = toc >> 3, (toc >> 2) & 1, toc & 3
    if config >= 12 or code != 0:
        return None
    frames = (1, 1, 2, 3)[config & 3]
    nch = 2 if stereo else 1
    pos = 0
    hdr = []
    for ch in range(nch):
        for _ in range(frames):
            fl, fh, ft = syms[pos]
            if not (ft == 2 and fl == 1):               # VAD flag clear: the frame uses the inactive type table, LBRR frames cannot
                return None
            hdr.append(syms[pos]); pos += 1
        if syms[pos][0] == 1:                           # has LBRR already
            return None
        hdr.append((1, 2, 2)); pos += 1                 # LBRR flag set
    # the rest: per 20 ms frame [stereo weights (5 symbols; no mid-only flag since the side VAD is set)] mid frame [side frame]
    body = syms[pos:]
    nstart = sum(1 for s in body if s[1] == 0 and s[2] == 1)          # markers (high = 0 is no symbol): 1 = a frame starts, 2 = it ends
    if nstart != frames * nch:
        return None                                     # a mid-only frame: not handled here
    last_end = max(k for k, s in enumerate(body) if s[1] == 0 and s[2] == 2)
    if any(s[1] != 0 for s in body[last_end + 1:]):
        return None                                     # symbols behind the last SILK frame (a redundancy frame's flags): not handled here
    groups = [[s for s in body if s[1] != 0]]
    # symbol value v (1 .. 2^frames - 1) has cumulative bounds [tab[v-1] .. tab[v]) in the {0, ...} tables above, with tab[0] = 0
    lbrr_sym = []
    if frames > 1:
        tab = LBRR40 if frames == 2 else LBRR60
        v = (1 << frames) - 1
        lbrr_sym = [(tab[v - 1], tab[v], 256)] * nch
    flat = [s for g in groups for s in g]
    return pkt[:1] + encode_symbols(hdr + lbrr_sym + flat + flat)


@pytest.fixture(scope="module")
def emu():
    lib = C.CDLL(build.build_emu())
    lib.emu_silk_state_floats.restype = C.c_size_t
    lib.emu_launch_silk.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.emu_launch_celt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


@pytest.mark.parametrize("vid,channels,npk", [(2, 2, 500), (3, 1, 400), (4, 2, 700), (12, 1, 400)])
def test_lbrr_frames_are_read_past(emu, vid, channels, npk):
    """packets rebuilt with LBRR copies of their frames decode to the PCM of the originals: oracle and host parser + kernels"""
    L = orc.lib()
    seen = C.c_long.in_dll(L, "orc_silk_lbrr_packets")
    _, pk = orc.load_vectors(ids=[vid], max_packets=npk)[vid]
    plain, fec, prod = orc.Decoder(channels), orc.Decoder(channels), EmuDecoder(emu, channels)
    before = seen.value
    made = {}
    for i, p in enumerate(pk):
        p = bytes(p)
        want, syms = traced_symbols(plain, p)
        q = with_lbrr(p, syms)
        if q is None:
            q = p
        else:
            made[p[0] >> 3] = made.get(p[0] >> 3, 0) + 1
        got = fec.decode(q)
        assert np.array_equal(got, want), f"vector {vid} packet {i}: oracle, LBRR copy in front"
        out = prod.decode(q)
        assert not isinstance(out, int), f"vector {vid} packet {i}: host parser returned {out}"
        if prod.hp.fi[0].mode == 0 and not prod.hp.fi[0].has_celt:      # SILK alone: bit-exact; a CELT part or redundant frame: tolerance
            assert np.array_equal(out, want), f"vector {vid} packet {i}: host parser + kernels, LBRR copy in front"
        else:
            check_pcm(want, out, f"vector {vid} packet {i}")
    assert sum(made.values()) >= 50 and seen.value - before >= sum(made.values()), made


@pytest.mark.parametrize("config", range(12))
@pytest.mark.parametrize("stereo", [0, 1])
def test_random_silk_payloads_with_lbrr(emu, config, stereo):
    """any bytes are a valid SILK packet; about half carry LBRR flags.  Host parser + kernels bit-exact against the oracle."""
    L = orc.lib()
    seen = C.c_long.in_dll(L, "orc_silk_lbrr_packets")
    before = seen.value
    rng = np.random.default_rng(1000 + 2 * config + stereo)
    ref, prod = orc.Decoder(2), EmuDecoder(emu, 2)
    for i in range(24):
        n = int(rng.integers(20, 120)) * (1, 1, 2, 3)[config & 3]
        p = bytes([config << 3 | stereo << 2]) + rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        want = ref.decode(p)
        out = prod.decode(p)
        assert not isinstance(out, int), f"packet {i}: host parser returned {out}"
        if not prod.hp.fi[0].has_celt:                  # (random SILK payloads mostly end in a -- random -- redundant CELT frame)
            assert np.array_equal(out, want), f"config {config} stereo {stereo} packet {i}"
        else:
            check_pcm(want, out, f"config {config} stereo {stereo} packet {i}")
    assert seen.value > before


# ---------------------------------------------------------------------------------------------------
# In-band FEC recovery: the LBRR frames decoded in place of a lost packet (decode_fec).  The reference returns Unsupported at the
# LBRR flag; the expected result is made by re-encoding: packet n + 1 is rebuilt with LBRR frames that are a copy of packet
# n's frames.  A decoder that lost packet n and recovers it from the rebuilt n + 1 must produce packet n's PCM bit for bit
# (same parameters into the same state), and then packet n + 1's.

def split_header(pkt, syms):
    """-> (header symbols [VAD flags + LBRR flag per channel], body symbols without markers, qualifies_as_lbrr_source) of a code-0
    SILK-only packet; None if the packet is something else or already carries LBRR frames"""
    toc = pkt[0]
    config, stereo, code = toc >> 3, (toc >> 2) & 1, toc & 3
    if config >= 12 or code != 0:
        return None
    frames = (1, 1, 2, 3)[config & 3]
    nch = 2 if stereo else 1
    pos, hdr, all_vad = 0, [], True
    for ch in range(nch):
        for _ in range(frames):
            all_vad = all_vad and syms[pos] == (1, 2, 2)
            hdr.append(syms[pos]); pos += 1
        if syms[pos][0] == 1:
            return None
        hdr.append(syms[pos]); pos += 1
    body = syms[pos:]
    nstart = sum(1 for s in body if s[1] == 0 and s[2] == 1)
    last_end = max(k for k, s in enumerate(body) if s[1] == 0 and s[2] == 2)
    clean = nstart == frames * nch and not any(s[1] != 0 for s in body[last_end + 1:])     # no mid-only frame, nothing behind the frames
    return hdr, [s for s in body if s[1] != 0], all_vad and clean, clean, frames, nch


def with_fec_copy(pkt_next, syms_next, pkt_prev, syms_prev):
    """packet n + 1 rebuilt with packet n's frames in front as LBRR frames, or None"""
    if pkt_prev[0] != pkt_next[0]:
        return None
    a, b = split_header(pkt_prev, syms_prev), split_header(pkt_next, syms_next)
    if a is None or b is None or not a[2] or not b[3]:
        return None
    hdr, body_next, _, _, frames, nch = b
    per = frames + 1
    hdr = list(hdr)
    for ch in range(nch):
        hdr[ch * per + frames] = (1, 2, 2)              # LBRR flag set
    lbrr_sym = []
    if frames > 1:
        tab = LBRR40 if frames == 2 else LBRR60
        v = (1 << frames) - 1
        lbrr_sym = [(tab[v - 1], tab[v], 256)] * nch
    return pkt_next[:1] + encode_symbols(hdr + lbrr_sym + a[1] + body_next)


@pytest.mark.parametrize("vid,channels,npk", [(2, 2, 500), (3, 1, 400), (4, 2, 700), (12, 1, 300)])
def test_fec_recovers_the_lost_packet_bit_exact(emu, vid, channels, npk):
    _, pk = orc.load_vectors(ids=[vid], max_packets=npk)[vid]
    pk = [bytes(p) for p in pk]
    plain, fec, prod = orc.Decoder(channels), orc.Decoder(channels), EmuDecoder(emu, channels)
    traced = [None] * len(pk)
    recovered, i = 0, 0
    while i < len(pk):
        want_i, syms_i = traced_symbols(plain, pk[i])
        q = None
        if i + 1 < len(pk) and i % 3 == 1:              # every third packet is "lost" where its successor can carry its copy
            # (the successor's symbols: traced on a throw-away copy of the decoder state is not available, so trace it for real)
            want_n, syms_n = traced_symbols(plain, pk[i + 1])
            q = with_fec_copy(pk[i + 1], syms_n, pk[i], syms_i)
            if q is None:                               # does not qualify: both packets arrive
                for p, w in ((pk[i], want_i), (pk[i + 1], want_n)):
                    assert np.array_equal(fec.decode(p), w)
                    out = prod.decode(p)
                    (np.testing.assert_array_equal if not prod.hp.fi[0].has_celt and prod.hp.fi[0].mode == 0 else check_pcm)(w, out)
                i += 2
                continue
            silk_only_before = prod.hp.fi[0].mode == 0 if i else True
            got = fec.decode_fec(q)
            assert np.array_equal(got, want_i), f"vector {vid}: oracle, packet {i} recovered from the LBRR frames of {i + 1}"
            assert np.array_equal(fec.decode(q), want_n), f"vector {vid}: oracle, packet {i + 1} behind its own LBRR frames"
            out = prod.decode(q, fec=True)
            assert not isinstance(out, int), f"vector {vid} packet {i}: host parser (FEC) returned {out}"
            if not any(prod.hp.fi[f].has_celt for f in range(prod.nframes)):
                assert np.array_equal(out, want_i), f"vector {vid}: host parser + kernels, packet {i} recovered from {i + 1}"
            else:
                check_pcm(want_i, out, f"vector {vid} packet {i} (FEC)")
            out = prod.decode(q)
            (np.testing.assert_array_equal if not prod.hp.fi[0].has_celt and prod.hp.fi[0].mode == 0 else check_pcm)(want_n, out)
            recovered += 1
            i += 2
            continue
        assert np.array_equal(fec.decode(pk[i]), want_i)
        out = prod.decode(pk[i])
        (np.testing.assert_array_equal if not prod.hp.fi[0].has_celt and prod.hp.fi[0].mode == 0 else check_pcm)(want_i, out)
        i += 1
    assert recovered >= 20, recovered


def test_fec_without_lbrr_frames_conceals(emu):
    """decode_fec on a packet that carries no LBRR frames: the lost span is concealed as by a lost packet of that duration;
    host parser + kernels against the oracle, and the stream goes on"""
    _, pk = orc.load_vectors(ids=[12], max_packets=200)[12]
    ref, prod = orc.Decoder(2), EmuDecoder(emu, 2)
    for i, p in enumerate(pk):
        p = bytes(p)
        if i % 7 == 3:
            w = ref.decode_fec(p)
            out = prod.decode(p, fec=True)
            assert out.shape == w.shape
            check_pcm(w, out, f"packet {i} (FEC, nothing to recover)")
        check_pcm(ref.decode(p), prod.decode(p), f"packet {i}")
