"""The oracle against every known-answer vector the reference's own unit tests hold."""
import ctypes as C

import numpy as np
import pytest

import orc

L = orc.lib()
K = orc.kats()


def test_reverse_bitread():  # entropy.rs:82-133
    k = K["reverse_bitread"]
    buf = bytes(k["buf"])
    w = (C.c_int * len(k["reads"]))(*[r[0] for r in k["reads"]])
    out = (C.c_uint32 * len(k["reads"]))()
    L.orc_kat_reverse_bits(buf, len(buf), w, len(k["reads"]), out)
    assert list(out) == [r[1] for r in k["reads"]]


def test_decode_laplace():  # entropy.rs:135-257
    k = K["decode_laplace"]
    buf = bytes(k["buf"])
    n = len(k["triples"])
    sym = (C.c_int * n)(*[t[1] for t in k["triples"]])
    dec = (C.c_int * n)(*[t[2] for t in k["triples"]])
    out = (C.c_int * n)()
    L.orc_kat_laplace(buf, len(buf), sym, dec, n, out)
    assert list(out) == [t[0] for t in k["triples"]]


def test_cwrsi():  # celt/decoder.rs:2414-2434
    for k in K["cwrsi"]:
        y = (C.c_int * k["n"])()
        assert L.orc_cwrsi(k["n"], k["k"], k["i"], y) == k["norm"]
        assert list(y) == k["y"]


def test_cwrsi_exhaustive_small():
    """every index of small (N,K) codebooks decodes to a distinct vector with sum|y| = K"""
    for n in range(2, 7):
        for k in range(1, 6):
            v = L.orc_pvq_v(n, k)
            seen = set()
            y = (C.c_int * n)()
            for i in range(v):
                norm = L.orc_cwrsi(n, k, i, y)
                t = tuple(y)
                assert sum(abs(a) for a in t) == k and norm == sum(a * a for a in t)
                seen.add(t)
            assert len(seen) == v


def test_extract_collapse_mask():  # celt/decoder.rs:2398-2411
    for k in K["extract_collapse_mask"]:
        y = (C.c_int * len(k["y"]))(*k["y"])
        assert L.orc_extract_collapse_mask(y, len(k["y"]), k["b"]) == k["mask"]


def test_stereo_merge():  # celt/decoder.rs:2330-2396
    k = K["stereo_merge"]
    x = np.array(k["x"], np.float32)
    y = np.array(k["y"], np.float32)
    L.orc_stereo_merge(x.ctypes.data, y.ctypes.data, C.c_float(k["mid"]), k["n"])
    assert np.array_equal(x, np.array(k["ox"], np.float32))
    assert np.array_equal(y, np.array(k["oy"], np.float32))


def _haar_naive(a, n0, stride):  # celt/decoder.rs:2436-2449 (the reference test's own naive form)
    a = a.copy()
    r = np.float32(0.70710678118654752440)
    for i in range(stride):
        for j in range(n0 // 2):
            x0, x1 = a[stride * 2 * j + i], a[stride * (2 * j + 1) + i]
            a[stride * 2 * j + i] = np.float32(x0 + x1) * r
            a[stride * (2 * j + 1) + i] = np.float32(x0 - x1) * r
    return a


def test_haar1():  # celt/decoder.rs:2451-2481
    for name in ("haar1_32_1", "haar1_16_2"):
        k = K[name]
        a = np.array(k["a"], np.float32)
        b = _haar_naive(a, k["n0"], k["stride"])
        L.orc_haar1(a.ctypes.data_as(C.c_void_p), k["n0"], k["stride"])
        assert np.array_equal(a, b)


def test_fft5():  # celt/imdct15.rs:213-251: radix-5 butterfly on a[0], a[3], .. of a[v] = (v, -v), v < 15
    """The reference's `fft5` test asserts exact f32 equality with its own butterfly's operation order; that function is an
    orphan (IMDCT15 is never constructed outside its tests) and the oracle's DFT is a different
    factorisation, so the bar here is f32 rounding (2 ulp of the largest output), not bit equality."""
    k = K["fft5"]
    a = np.array([[v, -v] for v in range(15)], np.float32)
    out = np.zeros((5, 2), np.float32)
    L.orc_fft(a.ctypes.data_as(C.c_void_p), k["stride"], 5, out.ctypes.data_as(C.c_void_p))
    # the reference's butterfly uses the e^{+2 pi i jk/5} kernel (its IMDCT is built on an inverse FFT), the oracle's DFT the
    # forward one: output k of one is output (5 - k) mod 5 of the other
    want = np.array(k["out"], np.float32)[[0, 4, 3, 2, 1]]
    assert np.abs(out - want).max() <= 2 * np.spacing(np.float32(30.0)), (out, want)
    # and the sizes the IMDCT uses, against numpy's f64 FFT
    rng = np.random.default_rng(5)
    for n in (15, 60, 120, 240, 480):
        x = rng.standard_normal((n, 2)).astype(np.float32)
        y = np.zeros((n, 2), np.float32)
        L.orc_fft(x.ctypes.data_as(C.c_void_p), 1, n, y.ctypes.data_as(C.c_void_p))
        ref = np.fft.fft(x[:, 0].astype(np.float64) + 1j * x[:, 1].astype(np.float64))
        err = np.abs((y[:, 0] + 1j * y[:, 1]) - ref).max()
        assert err <= 2e-5 * np.abs(ref).max(), (n, err)


def _snr(ref, x):
    ref = np.asarray(ref, np.float64)
    return 10 * np.log10((ref ** 2).sum() / max(((np.asarray(x, np.float64) - ref) ** 2).sum(), 1e-300))


@pytest.mark.parametrize("n2", [120, 240, 480, 960])
def test_imdct_fast_against_f64_direct(n2):
    """the oracle's f32 FFT-based IMDCT (pre-twiddle, n2/2-point FFT, post-twiddle) against the O(N^2)
    f64 evaluation of the transform, >= 100 dB, at every CELT block size, contiguous and (short blocks) interleaved input"""
    rng = np.random.default_rng(n2)
    for stride in (1, 8 if n2 == 120 else 2):
        x = rng.standard_normal(n2 * stride).astype(np.float32)
        fast = np.zeros(n2, np.float32)
        direct = np.zeros(n2, np.float64)
        L.orc_imdct_raw(x.ctypes.data_as(C.c_void_p), stride, n2, fast.ctypes.data_as(C.c_void_p))
        L.orc_imdct_raw_direct(x.ctypes.data_as(C.c_void_p), stride, n2, direct.ctypes.data_as(C.c_void_p))
        assert _snr(direct, fast) >= 100.0, (n2, stride, _snr(direct, fast))


def mdct_forward(frame2n, window120):
    """The textbook MDCT of a block of 2N samples under the CELT low-overlap window (RFC 6716 4.3.7: flat in the middle, the
    120-sample power-complementary window at both ends, zero outside), written from the definition in f64 -- independent of
    every line of oracle/ and opus_amd/:  X[k] = sum_n w[n] x[n] cos(pi/N (n + 1/2 + N/2)(k + 1/2))."""
    N = len(frame2n) // 2
    w = np.zeros(2 * N)
    pad = (N - 120) // 2
    w[pad:pad + 120] = window120
    w[pad + 120:2 * N - pad - 120] = 1.0
    w[2 * N - pad - 120:2 * N - pad] = window120[::-1]
    n = np.arange(2 * N)[:, None]
    k = np.arange(N)[None, :]
    return (w * frame2n) @ np.cos(np.pi / N * (n + 0.5 + N / 2) * (k + 0.5))


def tdac_overlap_add(prev_tail, D, window120):
    """what the decoder does with the raw IMDCT output D[0..N) of one block (RFC 6716 mdct.c clt_mdct_backward's mirror step +
    the overlap-add): -> (N output samples, the 60-sample tail the next block starts from)"""
    N = len(D)
    buf = np.concatenate([prev_tail, D])                # [60 tail of the previous block | N raw samples]
    w = window120
    for i in range(60):
        x1, x2 = buf[119 - i], buf[i]
        buf[i] = w[119 - i] * x2 - w[i] * x1
        buf[119 - i] = w[i] * x2 + w[119 - i] * x1
    return buf[:N], buf[N:N + 60]


@pytest.mark.parametrize("N", [120, 240, 480, 960])
def test_imdct_and_window_reconstruct_a_textbook_mdct(N):
    """Ground truth that is not the author's second copy: a known signal goes through the MDCT as the textbook defines it
    (f64, above), and the oracle's IMDCT + the decoder's TDAC mirror / overlap-add must give the signal back (time-domain
    aliasing cancels only with the right transform phase, sign, time order and window) -- up to the codec's fixed scale 2 / N
    ... which the denormalisation absorbs: measured here and required to be exactly N / 2 within f32 rounding."""
    rng = np.random.default_rng(N)
    win = np.ctypeslib.as_array(C.cast(L.orc_celt_window(), C.POINTER(C.c_float)), (120,)).astype(np.float64)
    assert np.allclose(win ** 2 + win[::-1] ** 2, 1.0, atol=1e-6)        # power-complementary
    nblk = 6
    t = np.arange((nblk + 1) * N)
    sig = np.sin(2 * np.pi * 440 / 48000 * t) + 0.3 * np.sin(2 * np.pi * 3111 / 48000 * t + 1.0) + 0.05 * rng.standard_normal(len(t))
    # block b covers samples [b N - pad - 120 .. ) with its 2N window; the decoder's output block b is delayed accordingly
    tail = np.zeros(60)
    out = []
    for b in range(nblk):
        X = mdct_forward(sig[b * N:(b + 2) * N], win).astype(np.float32)
        D = np.zeros(N, np.float32)
        L.orc_imdct_raw(X.ctypes.data_as(C.c_void_p), 1, N, D.ctypes.data_as(C.c_void_p))
        y, tail = tdac_overlap_add(tail, D.astype(np.float64), win)
        out.append(y)
    out = np.concatenate(out)
    # The decoder keeps the N distinct values of a block's 2N-sample IMDCT output (the rest are their mirror images), which start
    # in the middle of the rising window: output sample j of block b is signal sample b N + N / 2 - 60 + j.
    lag = N // 2 - 60
    ref = sig[lag + N:lag + nblk * N]                   # skip the first block (no overlap partner)
    got = out[N:nblk * N]
    scale = (ref @ got) / (ref @ ref)
    assert abs(scale / (N / 2) - 1) < 1e-5, (scale, N / 2)
    assert _snr(ref * scale, got) >= 100.0, _snr(ref * scale, got)


def test_lsf_to_lpc():  # silk/tests.rs:847-870
    k = K["lsf_to_lpc"]
    nlsf = (C.c_int16 * 10)(*k["nlsf"])
    lpc = np.zeros(10, np.float32)
    L.orc_lsf_to_lpc(nlsf, 10, lpc.ctypes.data_as(C.c_void_p))
    assert np.array_equal(lpc, np.array(k["lpc"], np.float32))


def _silk_packet(pkt, stereo_out=True):
    """silk/tests.rs:5-20: Packet::from_slice + Silk::setup + decode per frame"""
    p = orc.Packet()
    raw = bytes(pkt)
    assert L.orc_packet_parse(raw, len(raw), C.byref(p)) == 0
    assert p.mode == 0
    s = L.orc_silk_create(int(stereo_out))
    left = np.zeros(960, np.float32)
    right = np.zeros(960, np.float32)
    n = 0
    rd = C.create_string_buffer(256)
    for f in range(p.nframes):
        frame = raw[p.off[f]:p.off[f] + p.len[f]]
        L.orc_kat_rd_init(rd, frame, len(frame))
        n = L.orc_silk_decode(C.c_void_p(s), rd, p.bandwidth, p.frame_size // 48, p.stereo,
                              left.ctypes.data_as(C.c_void_p), right.ctypes.data_as(C.c_void_p))
        assert n > 0
    L.orc_silk_destroy(s)
    return left[:n], right[:n]


def test_silk_midonly_to_stereo():  # silk/tests.rs:22-514, exact f32
    k = K["silk_midonly_to_stereo"]
    l, r = _silk_packet(k["packet"])
    assert np.array_equal(r, np.array(k["right"], np.float32))
    assert np.array_equal(l, np.array(k["left"], np.float32))


def test_silk_unmix():  # silk/tests.rs:516-845, exact f32
    k = K["silk_unmix"]
    l, r = _silk_packet(k["packet"])
    assert np.array_equal(r, np.array(k["right"], np.float32))
    assert np.array_equal(l, np.array(k["left"], np.float32))


@pytest.mark.parametrize("rate_khz", [8, 12, 16])
def test_resampler_from_the_signal_side(rate_khz):
    """The 8 / 12 / 16 -> 48 kHz resampler is the build's own stage (the reference has none; RFC 6716 leaves it open), so nothing
    of the reference pins it.  From the signal side: a tone below the pass-band edge comes out as the same tone at 48 kHz, unit
    gain, 36 output samples late (the 73-tap filter's group delay, the same at every rate) at >= 60 dB; DC has unit gain; a tone in
    the transition band's far end is attenuated; block boundaries leave no trace (one call vs many)."""
    Lr = 48 // rate_khz
    fs = rate_khz * 1000.0
    def run(x, hist, chunk):
        out = []
        for o in range(0, len(x), chunk):
            seg = np.ascontiguousarray(x[o:o + chunk], np.float32)
            y = np.zeros(len(seg) * Lr, np.float32)
            L.orc_resample_to_48k(seg.ctypes.data_as(C.c_void_p), len(seg), rate_khz, hist.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p))
            out.append(y)
        return np.concatenate(out)
    n = 3200
    t_in, t_out = np.arange(n) / fs, (np.arange(n * Lr) - 36) / 48000.0
    for f, floor, droop in ((200.0, 60.0, 0.999), (1000.0, 50.0, 0.99), (0.35 * fs, 40.0, 0.85)):
        x = 0.5 * np.sin(2 * np.pi * f * t_in + 0.3)
        y = run(x, np.zeros(32, np.float32), 160)
        want = 0.5 * np.sin(2 * np.pi * f * t_out + 0.3)
        g = float(want[400:] @ y[400:]) / float(want[400:] @ want[400:])          # the filter's gain at f: 73 taps at 48 kHz droop towards the edge
        assert droop <= g <= 1.001, (f, g)
        assert _snr(g * want[400:], y[400:]) >= floor, (f, _snr(g * want[400:], y[400:]))     # same tone, same phase: 36 samples late
        assert np.array_equal(y, run(x, np.zeros(32, np.float32), 320))          # block size leaves no trace
    dc = run(np.full(800, 0.25), np.zeros(32, np.float32), 160)
    assert np.abs(dc[200:] - 0.25).max() < 2e-4
    img = run(0.5 * np.sin(2 * np.pi * 0.499 * fs * t_in), np.zeros(32, np.float32), 160)      # at the band edge: mostly gone
    assert np.sqrt(np.mean(img[400:] ** 2)) < 0.5 * 0.5 / np.sqrt(2)
