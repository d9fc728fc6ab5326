#!/usr/bin/env python3
"""opus_compare -- the RFC 6716 conformance metric (test infrastructure, numpy).

A restatement of the quality measure of RFC 6716's opus_compare tool (the RFC's conformance criterion, section 6.1:
a decoder passes a test vector when this tool reports a quality >= 0 against the vector's `.dec` reference PCM).
lu-zero/opus has no counterpart (its assets hold the 12 `.mka` bitstreams but no `.dec` files), so in
this container there is no expected output to run it against: **parity unpinned** -- the algorithm below follows the
published tool from memory of its text (480-sample raised-cosine windows every 120 samples, 21 Bark-like bands over
240 bins of 100 Hz, spectral + temporal masking, log-spectral distortion averaged to the 16th power).  It is used
(a) by tests/ as one more GPU-vs-oracle check in the codec's own perceptual terms, and (b) from the command line on
`.dec` files when a user has them:

    python tools/opus_compare.py [-s] [-r 48000|24000|16000|12000|8000] reference.dec decoded.pcm

Both files are raw 16-bit little-endian PCM; the reference is 48 kHz stereo; `-s` compares in stereo (otherwise both
are down-mixed / taken as mono); `-r` is the sampling rate of the decoded file.
"""
import sys

import numpy as np

NBANDS, NFREQS = 21, 240
WIN, STEP = 480, 120
BANDS = np.array([0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 68, 80, 96, 120, 156, 200])


def _power_spectrum(x, nframes, win, step, downsample, nfreq):
    """x: [samples, channels] -> ps [nframes, nfreq, channels]: |DFT_win(window * frame)|^2 * downsample^2 + 1e5"""
    w = (0.5 - 0.5 * np.cos(2 * np.pi / (win - 1) * np.arange(win))).astype(np.float64)
    idx = np.arange(nframes)[:, None] * step + np.arange(win)[None, :]
    fr = x[idx] * w[None, :, None]                       # [nframes, win, ch]
    spec = np.fft.rfft(fr, axis=1)[:, :nfreq] * downsample
    return spec.real ** 2 + spec.imag ** 2 + 100000.0


def compare(ref, dec, rate=48000, stereo=True):
    """ref: 48 kHz stereo reference [n, 2]; dec: decoded [m, channels] at `rate`; both in 16-bit sample units.
    Returns (quality, err): quality >= 0 passes (100 = identical)."""
    ref = np.asarray(ref, np.float64).reshape(-1, 2)
    if rate not in (8000, 12000, 16000, 24000, 48000):
        raise ValueError("sampling rate must be 8000, 12000, 16000, 24000 or 48000")
    downsample = 48000 // rate
    ybands = {48000: NBANDS, 24000: 19, 16000: 17, 12000: 15, 8000: 13}[rate]
    yfreqs = NFREQS // downsample
    nch = 2 if stereo else 1
    x = ref if stereo else 0.5 * (ref[:, :1] + ref[:, 1:])
    y = np.asarray(dec, np.float64).reshape(-1, nch)
    if x.shape[0] != y.shape[0] * downsample:
        raise ValueError(f"sample counts do not match ({x.shape[0]} != {y.shape[0]} * {downsample})")
    if x.shape[0] < WIN:
        raise ValueError(f"insufficient data ({x.shape[0]} < {WIN} samples)")
    nframes = (x.shape[0] - WIN + STEP) // STEP
    X = _power_spectrum(x, nframes, WIN, STEP, 1, BANDS[NBANDS])
    Y = _power_spectrum(y, nframes, WIN // downsample, STEP // downsample, downsample, BANDS[ybands])
    # per-band mean energy of the reference
    xb = np.stack([X[:, BANDS[b]:BANDS[b + 1]].mean(axis=1) for b in range(NBANDS)], axis=1)     # [nframes, NBANDS, ch]
    for b in range(1, NBANDS):                           # frequency masking low -> high: 10 dB/Bark
        xb[:, b] += 0.1 * xb[:, b - 1]
    for b in range(NBANDS - 2, -1, -1):                  # high -> low: 15 dB/Bark
        xb[:, b] += 0.03 * xb[:, b + 1]
    for t in range(nframes):
        if t > 0:                                        # temporal masking: -3 dB / 2.5 ms (recursive over frames)
            xb[t] += 0.5 * xb[t - 1]
        if nch == 2:                                     # some cross-talk
            l, r = xb[t, :, 0].copy(), xb[t, :, 1].copy()
            xb[t, :, 0] += 0.01 * r
            xb[t, :, 1] += 0.01 * l
    nf = BANDS[ybands]
    band_of = np.repeat(np.arange(NBANDS), np.diff(BANDS))[:nf]
    mask = 0.1 * xb[:, band_of]                          # [nframes, nf, ch]
    Xm = X[:, :nf] + mask
    Ym = Y[:, :nf] + mask
    Xm[1:] += Xm[:-1].copy()                             # average of consecutive frames
    Ym[1:] += Ym[:-1].copy()
    max_compare = BANDS[NBANDS] if rate == 48000 else BANDS[ybands] if rate == 12000 else BANDS[ybands] - 3
    re = Ym / Xm
    im = re - np.log(re) - 1.0
    im[:, 79:82] *= 0.1                                  # less sensitive around the SILK/CELT cross-over
    im[:, 80:81] *= 0.1
    im[:, max_compare:] = 0.0
    Ef = np.zeros(nframes)
    for b in range(ybands):
        Eb = im[:, BANDS[b]:BANDS[b + 1]].sum(axis=(1, 2)) / ((BANDS[b + 1] - BANDS[b]) * nch)
        Ef += Eb * Eb
    Ef /= NBANDS
    Ef *= Ef
    err = float(((Ef * Ef).sum() / nframes) ** (1.0 / 16))
    q = 100.0 * (1.0 - 0.5 * np.log(1.0 + err) / np.log(1.13))
    return float(q), err


def to_pcm16(pcm_f32):
    """float PCM in [-1, 1) -> 16-bit sample units, rounded and saturated like a .dec writer"""
    return np.clip(np.rint(np.asarray(pcm_f32, np.float64) * 32768.0), -32768, 32767)


def main(argv):
    stereo, rate = False, 48000
    args = list(argv)
    while args and args[0].startswith("-"):
        o = args.pop(0)
        if o == "-s":
            stereo = True
        elif o == "-r":
            rate = int(args.pop(0))
        else:
            print(__doc__)
            return 2
    if len(args) != 2:
        print(__doc__)
        return 2
    ref = np.fromfile(args[0], dtype="<i2")
    dec = np.fromfile(args[1], dtype="<i2")
    q, err = compare(ref, dec, rate, stereo)
    if q < 0:
        print(f"Test vector FAILS\nInternal weighted error is {err:f}")
        return 1
    print(f"Test vector PASSES\nOpus quality metric: {q:.1f} % (internal weighted error is {err:f})")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
