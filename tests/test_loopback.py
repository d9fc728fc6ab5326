"""Encode -> decode loopback: ground truth for the CELT path that is not the oracle's own output.

The reference is a decoder and holds no CELT encoder, no CELT golden PCM and no code behind the band decoding (celt/decoder.rs
ends at decode_bands), so everything after the PVQ is pinned here from the signal side: a known signal is ENCODED into CELT
packets, and whatever decodes them -- the oracle, the host parser + kernel bodies, the GPU -- must give the signal back.

The encoder is the oracle's decoder run backwards (oracle/orc_rangedec.h, forced mode): the decoder walks a frame exactly as it
would decode it and makes every bit-allocation decision itself, but each value it would read is ASKED of this file, which answers
from the textbook f64 MDCT of the source (tests/test_oracle_kats.py: mdct_forward) -- band energies, split angles, pulse vectors
(a greedy PVQ search) -- and the range ENcoder of tests/test_lbrr.py writes the answers into a packet.  A transform, window,
overlap, band layout, energy or sign convention that is wrong in the decoder cannot reproduce the source; the PVQ codebook, the
range coder and the allocation are exercised on bits that no decoder wrote.  Three checks per stream: the packet decodes to
exactly what the forced run produced (the packet says what was meant); decoded vs source SNR; product vs oracle."""
import ctypes as C

import numpy as np
import pytest

import orc
from gpu_util import check_pcm
from test_lbrr import RangeEncoder
from test_oracle_kats import mdct_forward

L = orc.lib()
HOOK = C.CFUNCTYPE(C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int))
L.orc_set_encoder.argtypes = [HOOK]
L.orc_icwrs.restype = C.c_uint32
N = 960
EBANDS = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100]      # RFC 6716 table 55 (x 8 bins at 20 ms)
EMEANS = [6.4375, 6.25, 5.75, 5.3125, 5.0625, 4.8125, 4.5, 4.375, 4.875, 4.6875, 4.5625, 4.4375, 4.875, 4.625, 4.3125, 4.5,
          4.375, 4.625, 4.75, 4.4375, 3.75]                                                       # RFC 6716 4.3.2.1 (quant_bands.c eMeans)
PREEMPH = 0.85000610


def pvq_search(t, k):
    """the pulse vector y (integers, sum |y| = k) closest in direction to t: projection, then greedy pulses"""
    a = np.abs(t)
    s = a.sum()
    if s < 1e-30:
        y = np.zeros(len(t), np.int64)
        y[0] = k
        return y
    y = np.floor(k * a / s).astype(np.int64)
    xy, yy = float(a @ y), float(y @ y)
    for _ in range(k - int(y.sum())):
        gain = (xy + a) ** 2 / (yy + 2 * y + 1)
        j = int(np.argmax(gain))
        xy += a[j]
        yy += 2 * y[j] + 1
        y[j] += 1
    return np.where(t < 0, -y, y)


def _rot1(x, stride, c, s):
    n = len(x)
    for i in range(n - stride):
        x1, x2 = x[i], x[i + stride]
        x[i + stride] = c * x2 + s * x1
        x[i] = c * x1 - s * x2
    for i in range(n - 2 * stride - 1, -1, -1):
        x1, x2 = x[i], x[i + stride]
        x[i + stride] = c * x2 + s * x1
        x[i] = c * x1 - s * x2


def exp_rotation_forward(x, blocks, k, spread):
    """RFC 6716 vq.c exp_rotation with dir = +1 -- what an ENCODER applies to a band before the pulse search (the decoder applies
    dir = -1 to what it decoded); written from the RFC text, f64, in place"""
    n = len(x)
    if 2 * k >= n or spread == 0:
        return
    gain = n / (n + (15, 10, 5)[spread - 1] * k)
    theta = 0.5 * gain * gain
    c, s = np.cos(0.5 * np.pi * theta), np.cos(0.5 * np.pi * (1.0 - theta))
    stride2 = 0
    if n >= 8 * blocks:
        stride2 = 1
        while (stride2 * stride2 + stride2) * blocks + (blocks >> 2) < n:
            stride2 += 1
    m = n // blocks
    for i in range(blocks):
        v = x[i * m:(i + 1) * m]
        _rot1(v, 1, c, -s)
        if stride2:
            _rot1(v, stride2, s, -c)


ORDERY = {2: [1, 0], 4: [3, 0, 2, 1], 8: [7, 0, 4, 3, 6, 1, 5, 2], 16: [15, 0, 8, 7, 12, 3, 11, 4, 14, 1, 9, 6, 13, 2, 10, 5]}      # RFC 6716 bands.c ordery_table


def haar1(x, n0, stride):
    """RFC 6716 bands.c haar1, in place: pairs (2j, 2j + 1) of each of `stride` interleaved sequences -> (sum, difference) / sqrt 2"""
    v = x[:n0 * stride].reshape(n0 // 2, 2, stride)
    a, b = v[:, 0, :].copy(), v[:, 1, :].copy()
    v[:, 0, :], v[:, 1, :] = (a + b) * np.sqrt(0.5), (a - b) * np.sqrt(0.5)


def band_to_coding_domain(x, tf_change):
    """what an encoder does to a long-block band before it codes it when the band's time-frequency change is negative (RFC 6716
    bands.c quant_band): a Haar level per step (more time resolution), then the Hadamard-ordered de-interleave of the resulting
    `blocks` sequences; the decoder undoes both after decoding.  -> the vector as coded"""
    x = x.copy()
    n_b, blocks = len(x), 1
    while n_b % 2 == 0 and tf_change < 0:
        haar1(x, n_b, blocks)
        blocks, n_b, tf_change = blocks * 2, n_b // 2, tf_change + 1
    if blocks > 1:
        out = np.empty_like(x)
        for i in range(blocks):
            out[ORDERY[blocks][i] * n_b:(ORDERY[blocks][i] + 1) * n_b] = x[i::blocks]
        x = out
    return x


class Encoder:
    """one stream: frames of n (120 | 240 | 480 | 960) samples per channel in, CELT-only FB packets of `nbytes` payload bytes out"""

    def __init__(self, channels, nbytes, n=N, transient=False, joint=False, spread=0, postfilter=None, hybrid=False, tf=0):
        """transient: every frame as short blocks (n / 120 transforms of 120 bins, interleaved); joint: the two channels' bands
        coded as mid / side with a stereo angle (else one by one: dual stereo); spread: the spreading rotation's strength 0 .. 3"""
        self.cc, self.nbytes, self.n, self.m = channels, nbytes, n, n // 120
        self.transient, self.joint, self.spread = transient, joint and channels == 2, spread
        # postfilter = (period, gain index, tap set): the decoder then runs its comb filter y[n] = x[n] + g sum taps y[n - T +- k]
        # (RFC 6716 4.3.7.1) on what it decodes, so the encoder codes x[n] = y[n] - g sum taps y[n - T +- k] of the signal it wants
        self.pf = postfilter
        # tf = 1 | 2: long blocks with every band's time-frequency change at -2 | -3 (20 ms frames: RFC 6716 table 60)
        self.tf = tf
        assert not (tf and (transient or n != 960))
        self.pre_all = [np.zeros(1100) for _ in range(channels)]       # the pre-emphasised signal so far (behind 1100 zeros)
        self.dec = orc.Decoder(channels)                   # the decoder that is run forced: its state IS the encoder's state
        self.toc = bytes([(28 + {120: 0, 240: 1, 480: 2, 960: 3}[n]) << 3 | (4 if channels == 2 else 0)])     # CELT-only FB, code 0
        if hybrid:                                         # hybrid FB 20 ms: a SILK WB frame in front (every symbol 0: an inactive
            assert n == 960                                # frame at the lowest gain, next to silence), CELT from band 17 (8 kHz) up
            self.toc = bytes([15 << 3 | (4 if channels == 2 else 0)])
        self.win = np.ctypeslib.as_array(C.cast(L.orc_celt_window(), C.POINTER(C.c_float)), (120,)).astype(np.float64)
        self.mem = np.zeros(channels)
        self.hist = np.zeros((channels, n))                # the previous block of pre-emphasised samples
        self.hook = HOOK(self._ask)

    def _ask(self, kind, band, ch, a, b, c, f, y):
        if kind == 1:                                      # coarse energy: the integer that brings the prediction closest
            want = int(np.round(self.E[ch][band] - f))
            return max(-1, min(1, want)) if a == 1 else min(0, max(-1, want)) if a == 2 else want
        if kind == 2:                                      # fine energy: the decoder adds (q + 1/2) / 2^bits - 1/2
            return int(min((1 << a) - 1, max(0, np.floor((self.E[ch][band] - f + 0.5) * (1 << a)))))
        if kind == 3:
            return int(self.E[ch][band] > f)
        if kind == 4:                                      # split angle between the two halves of a vector, in qn steps
            mid, side = np.linalg.norm(self.T[ch][a:a + b]), np.linalg.norm(self.T[ch][a + b:a + 2 * b])
            return int(min(c, max(0, np.round(np.arctan2(side, mid) * 2 / np.pi * c))))
        if kind == 5:                                      # f = blocks + 16 * spread
            t = self.T[ch][a:a + b].copy()
            exp_rotation_forward(t, int(f) & 15, c, int(f) >> 4)
            v = pvq_search(t, c)
            for i in range(b):
                y[i] = int(v[i])
            return 0
        if kind == 6:
            return int(self.T[ch][a] < 0)
        if kind == 7:                                      # jointly coded band: mid = l + r, side = r - l of the unit-norm bands
            l, r = self.T[0][a:a + b].copy(), self.T[1][a:a + b].copy()
            l, r = l / max(1e-30, np.linalg.norm(l)), r / max(1e-30, np.linalg.norm(r))
            self.T[0][a:a + b], self.T[1][a:a + b] = l + r, r - l
            return int(min(c, max(0, np.round(np.arctan2(np.linalg.norm(r - l), np.linalg.norm(l + r)) * 2 / np.pi * c))))
        if kind == 8:                                      # two-bin joint band (RFC 6716 bands.c quant_band_stereo, N == 2): the coded
            x2, y2 = (self.T[1][a:a + 2], self.T[0][a:a + 2]) if c else (self.T[0][a:a + 2], self.T[1][a:a + 2])      # vector and the other one,
            return int(x2[0] * y2[1] - x2[1] * y2[0] < 0)  # which the decoder rebuilds as +-(-x2[1], x2[0])
        raise AssertionError(kind)

    def encode(self, frame):
        """frame: (n, channels) in [-1, 1) -> (packet, the PCM the forced run produced)"""
        self.F, self.E, self.T = [], [], []                # spectrum, band energies, the spectrum in the order / form it is coded in
        for ch in range(self.cc):
            x = frame[:, ch].astype(np.float64) * 32768.0
            pre = x - PREEMPH * np.concatenate([[self.mem[ch]], x[:-1]])
            self.mem[ch] = x[-1]
            if self.pf:
                T_, gq, ts = self.pf
                g = 0.09375 * (gq + 1)
                t0, t1, t2 = ((0.3066406250, 0.2170410156, 0.1296386719), (0.4638671875, 0.2680664062, 0.0), (0.7998046875, 0.1000976562, 0.0))[ts]
                y = np.concatenate([self.pre_all[ch], pre])
                self.pre_all[ch] = y
                i = np.arange(len(y) - self.n, len(y))
                pre = y[i] - g * (t0 * y[i - T_] + t1 * (y[i - T_ - 1] + y[i - T_ + 1]) + t2 * (y[i - T_ - 2] + y[i - T_ + 2]))
            blk = np.concatenate([self.hist[ch], pre])
            self.hist[ch] = pre
            T = np.zeros(self.n)
            if self.transient:
                # short blocks: M transforms of 120 bins over the long window's support, bin i of block j at i M + j; a band is
                # coded block by block (decoded[j N0 + i] = band[i M + j]: RFC 6716 bands.c deinterleave_hadamard, plain form)
                M, s0 = self.m, (self.n - 120) // 2
                F = np.zeros(self.n)
                for j in range(M):
                    F[j::M] = mdct_forward(blk[s0 + 120 * j:s0 + 120 * j + 240], self.win) * (2.0 / 120)
                for i in range(21):
                    lo, hi = M * EBANDS[i], M * EBANDS[i + 1]
                    T[lo:hi] = F[lo:hi].reshape(-1, M).T.reshape(-1)
            else:
                F = mdct_forward(blk, self.win) * (2.0 / self.n)
                T[:] = F
                if self.tf:
                    for i in range(21):
                        lo, hi = self.m * EBANDS[i], self.m * EBANDS[i + 1]
                        T[lo:hi] = band_to_coding_domain(F[lo:hi], -1 - self.tf)
            self.F.append(F)
            self.T.append(T)
            self.E.append([np.log2(max(1e-9, np.linalg.norm(F[self.m * EBANDS[i]:self.m * EBANDS[i + 1]]))) - EMEANS[i] for i in range(21)])
        buf = (C.c_uint32 * (3 * 16384))()
        L.orc_trace_set(buf, 16384)
        L.orc_set_encoder_opts(int(self.transient), int(not self.joint), self.spread)
        L.orc_set_encoder_postfilter(*(self.pf or (0, 0, 0)))
        L.orc_set_encoder_tf(self.tf)
        L.orc_set_encoder(self.hook)
        try:
            pcm = self.dec.decode(self.toc + bytes(self.nbytes))
            n = L.orc_trace_count()
        finally:
            L.orc_set_encoder(HOOK(0))
            L.orc_set_encoder_opts(0, 1, 0)
            L.orc_set_encoder_postfilter(0, 0, 0)
            L.orc_set_encoder_tf(0)
            L.orc_trace_set(None, 0)
        assert n < 16384
        syms = np.ctypeslib.as_array(buf).reshape(-1, 3)[:n]
        enc, raw, nraw = RangeEncoder(), 0, 0
        for lo, hi, tot in syms:
            lo, hi, tot = int(lo), int(hi), int(tot)
            if lo == 0xFFFFFFFF:                           # raw bits: read from the END of the frame, least significant first
                raw |= hi << nraw
                nraw += tot
            elif hi:
                enc.encode(lo, hi, tot)
        front = enc.done()
        out = bytearray(self.nbytes)
        out[:len(front)] = front
        back = raw.to_bytes((nraw + 7) // 8, "little")
        assert len(front) + len(back) <= self.nbytes + 1, (len(front), len(back))
        for i, v in enumerate(back):
            assert out[self.nbytes - 1 - i] & v == 0
            out[self.nbytes - 1 - i] |= v
        return self.toc + bytes(out), pcm


def source(channels, nframes, seed=5, n=N, high=False):
    """tones (different ones per channel), a slow sweep and a little noise; peak below 0.6.  high: everything between 9 and 19 kHz"""
    rng = np.random.default_rng(seed)
    t = np.arange((nframes + 1) * n) / 48000.0
    cols = []
    for ch in range(channels):
        f = [220.0, 1318.5, 4186.0] if ch == 0 else [329.6, 2093.0, 6271.9]
        if high:
            f = [9531.0, 12345.0, 15101.0] if ch == 0 else [10007.0, 13999.0, 17203.0]
        s = sum(0.15 / (1 + i) * np.sin(2 * np.pi * fi * t + i + ch) for i, fi in enumerate(f))
        s = s + 0.08 * np.sin(2 * np.pi * ((10000.0 if high else 500.0) + 900.0 * t) * t) + (0.0 if high else 0.002) * rng.standard_normal(len(t))
        cols.append(s * np.minimum(1.0, t / 0.02))
    return np.stack(cols, axis=1)


def encode_stream(channels, nframes, nbytes, n=N, **tools):
    src = source(channels, nframes, n=n, high=bool(tools.get("hybrid")))
    enc = Encoder(channels, nbytes, n, **tools)
    packets, forced = [], []
    for b in range(nframes):
        p, pcm = enc.encode(src[b * n:(b + 1) * n])
        packets.append(p)
        forced.append(pcm)
    return src, packets, forced


def snr_db(ref, got):
    return 10 * np.log10(float(np.sum(ref.astype(np.float64) ** 2)) / max(1e-30, float(np.sum((ref - got).astype(np.float64) ** 2))))


def aligned(src, nframes, n=N):
    """decoded sample j of frame b is source sample (b - 1) n + n / 2 - 60 + j: the frame the encoder's block b ENDS with is the one
    its window is centred on (tests/test_oracle_kats.py), and block b = [previous frame | this frame]"""
    lag = n // 2 - 60 - n
    return src[2 * n + lag:nframes * n + lag]


def test_icwrs_inverts_cwrsi():
    rng = np.random.default_rng(1)
    for n, k in [(2, 1), (3, 7), (8, 4), (16, 9), (24, 3), (48, 5), (96, 2), (176, 2), (11, 40)]:
        v = int(L.orc_pvq_v(n, k))
        for idx in [0, v - 1] + [int(rng.integers(0, v)) for _ in range(20)]:
            y = (C.c_int * n)()
            L.orc_cwrsi(n, k, C.c_uint32(idx), y)
            assert sum(abs(t) for t in y) == k and int(L.orc_icwrs(n, y)) == idx


@pytest.mark.parametrize("channels,nbytes,floor_db", [(1, 160, 20.0), (2, 320, 20.0), (2, 640, 26.0), (2, 120, 12.0)])
def test_celt_loopback_oracle(channels, nbytes, floor_db):
    nframes = 30
    src, packets, forced = encode_stream(channels, nframes, nbytes)
    dec = orc.Decoder(channels)
    out = [dec.decode(p) for p in packets]
    for b in range(nframes):                               # the packets carry exactly the values the forced run was given
        assert np.array_equal(out[b], forced[b]), f"frame {b}"
    got = np.concatenate(out)[2 * N:]
    ref = aligned(src, nframes)
    s = snr_db(ref, got.reshape(ref.shape))
    assert s >= floor_db, f"decoded vs source: {s:.1f} dB"
    if channels == 2:                                      # the channels carry different tones: no leakage from one into the other
        assert snr_db(ref[:, 0], got[:, 1]) < 3.0


@pytest.mark.parametrize("n,nbytes,floor_db,joint", [(480, 160, 18.0, False), (240, 80, 14.0, False), (120, 48, 10.0, False),
                                                     (480, 160, 15.0, True), (240, 80, 15.0, True), (120, 48, 15.0, True)])
def test_celt_loopback_short_frames(request, n, nbytes, floor_db, joint):
    """10 / 5 / 2.5 ms frames (the 240 / 120 / 60-point transforms, one-bin bands with their sign bits at 2.5 ms): oracle and
    kernel bodies return the source at the rate's SNR, aligned to the sample.  joint: the bands coded as mid / side -- at 5 and
    2.5 ms that includes the two-bin bands, whose side is rebuilt from the mid by a rotation and a sign bit"""
    from test_emu_silk import EmuDecoder
    emu = request.getfixturevalue("emu")
    nframes = 4800 // n * 4
    src, packets, forced = encode_stream(2, nframes, nbytes, n, joint=joint)
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    outs = []
    for i, p in enumerate(packets):
        want = ref.decode(p)
        assert np.array_equal(want, forced[i]), f"frame {i}"
        check_pcm(want, d.decode(p), f"frame {i}")
        outs.append(want)
    got, want = np.concatenate(outs)[2 * n:], aligned(src, nframes, n)
    s = snr_db(want, got)
    assert s >= floor_db, f"decoded vs source: {s:.1f} dB"
    assert snr_db(want[1:], got[:-1]) < s - 3.0


@pytest.mark.parametrize("tools,nbytes,floor_db", [(dict(spread=3), 120, 11.0), (dict(joint=True), 320, 16.0), (dict(transient=True), 320, 16.0),
                                                    (dict(transient=True, joint=True, spread=2), 320, 15.0),
                                                    (dict(postfilter=(15, 5, 2)), 320, 19.0), (dict(postfilter=(37, 7, 1)), 320, 19.0),
                                                    (dict(postfilter=(100, 3, 0)), 320, 19.0), (dict(postfilter=(733, 6, 0)), 320, 18.0),
                                                    (dict(postfilter=(1022, 2, 1), joint=True, spread=2), 320, 15.0),
                                                    (dict(tf=1), 320, 18.0), (dict(tf=2), 320, 17.0), (dict(tf=1, joint=True, spread=2), 320, 16.0)])
def test_celt_loopback_coding_tools(request, tools, nbytes, floor_db):
    """the same loopback with the tools every real stream uses, each done on the encoder side from the RFC's text in f64:
    the spreading rotation (forward before the pulse search; at 24 kb/s per channel leaving it out costs 10 dB), jointly coded
    stereo (mid = l + r, side = r - l of the unit-norm bands and their angle: the decoder's stereo_merge must undo it), short
    blocks (eight 120-bin transforms interleaved, bands coded block by block), all three together; and the pitch post-filter:
    the encoder codes x[n] = y[n] - g sum taps y[n - T +- k], so only a decoder whose comb filter has that period, those taps
    in that order and that gain (periods 15 .. 1022: the register path, the LDS steps, the history reach) returns y; and the
    time-frequency change of long blocks (-2 / -3: Haar levels and the Hadamard-ordered de-interleave, which the decoder undoes)"""
    from test_emu_silk import EmuDecoder
    emu = request.getfixturevalue("emu")
    nframes = 20
    src, packets, forced = encode_stream(2, nframes, nbytes, **tools)
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    outs = []
    for i, p in enumerate(packets):
        want = ref.decode(p)
        assert np.array_equal(want, forced[i]), f"frame {i}"
        check_pcm(want, d.decode(p), f"frame {i}")
        outs.append(want)
    got, want = np.concatenate(outs)[2 * N:], aligned(src, nframes)
    s = snr_db(want, got)
    assert s >= floor_db, f"decoded vs source: {s:.1f} dB"
    assert snr_db(want[:, 0], got[:, 1]) < 3.0             # the channels carry different tones


@pytest.mark.parametrize("channels", [1, 2])
def test_hybrid_loopback(request, channels):
    """a hybrid FB packet: the SILK part coded as an inactive frame at the lowest gain (every symbol 0: next to silence), the CELT
    part, from band 17 (8 kHz) up, coding a source that lives between 9 and 19 kHz.  Oracle and kernel bodies return the source
    (the SILK part, the resampler and the accumulation onto the SILK output must not disturb it): the hybrid start band, the
    bit budget behind the SILK part and the band layout, from the signal side."""
    from test_emu_silk import EmuDecoder
    emu = request.getfixturevalue("emu")
    nframes = 20
    src, packets, forced = encode_stream(channels, nframes, 200 * channels, hybrid=True)
    d, ref = EmuDecoder(emu, channels), orc.Decoder(channels)
    outs = []
    for i, p in enumerate(packets):
        want = ref.decode(p)
        assert np.array_equal(want, forced[i]), f"frame {i}"
        check_pcm(want, d.decode(p), f"frame {i}")
        assert d.hp.fi[0].mode == 1 and d.hp.fi[0].nsilk == 1 and d.hp.fi[0].has_celt == 1
        outs.append(want)
    got = np.concatenate(outs)[2 * N:]
    want = aligned(src, nframes).reshape(got.shape)
    assert snr_db(want, got) >= 25.0, snr_db(want, got)


def test_celt_loopback_product(request):
    """the same packets through the host parser + the kernel bodies (g++ build): against the oracle, and against the source"""
    from test_emu_silk import EmuDecoder
    emu = request.getfixturevalue("emu")
    nframes = 20
    src, packets, _ = encode_stream(2, nframes, 320)
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    outs = []
    for i, p in enumerate(packets):
        out = d.decode(p)
        check_pcm(ref.decode(p), out, f"frame {i}")
        outs.append(out)
    got = np.concatenate(outs)[2 * N:]
    assert snr_db(aligned(src, nframes), got) >= 20.0


from test_emu_silk import emu  # noqa: E402,F401  (fixture)
