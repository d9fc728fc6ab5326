"""tools/opus_compare.py (the RFC 6716 conformance metric, restated; parity unpinned: the reference's assets hold no
`.dec` reference PCM) -- behaviour checks on the oracle's output, and the oracle against itself through a lossy copy."""
import os
import sys

import numpy as np
import pytest

import orc

sys.path.insert(0, os.path.join(orc.ROOT, "tools"))
import opus_compare as oc  # noqa: E402


def _oracle_pcm(vid, n):
    _, pk = orc.load_vectors(ids=[vid], max_packets=n)[vid]
    d = orc.Decoder(2)
    return np.concatenate([d.decode(p) for p in pk])


@pytest.fixture(scope="module")
def pcm11():
    return _oracle_pcm(11, 100)


def test_identical_is_100(pcm11):
    x = oc.to_pcm16(pcm11)
    q, err = oc.compare(x, x, 48000, True)
    assert q == 100.0 and err == 0.0
    q, _ = oc.compare(x, 0.5 * (x[:, :1] + x[:, 1:]), 48000, False)
    assert q == 100.0


def test_quality_drops_with_distortion(pcm11):
    x = oc.to_pcm16(pcm11)
    rng = np.random.default_rng(1)
    q_prev = 100.0
    for sigma in (30.0, 300.0, 3000.0):                 # white noise at rising level (16-bit units)
        q, _ = oc.compare(x, x + rng.normal(0, sigma, x.shape), 48000, True)
        assert q < q_prev
        q_prev = q
    assert q_prev < 0                                   # the loudest one fails the vector
    q, _ = oc.compare(x, np.zeros_like(x), 48000, True)
    assert q < 0                                        # silence instead of the signal fails
    q, _ = oc.compare(x, x[:, ::-1], 48000, True)       # swapped channels on a stereo recording
    assert q < 100.0


def test_rounding_noise_passes(pcm11):
    """f32 vs 16-bit rounding (what a .dec comparison sees from a conforming float decoder) stays near 100"""
    x = pcm11.astype(np.float64) * 32768.0
    q, _ = oc.compare(oc.to_pcm16(pcm11), x, 48000, True)
    assert q > 95.0


def test_lower_rates_and_errors(pcm11):
    x = oc.to_pcm16(pcm11)
    for rate in (24000, 16000, 12000, 8000):
        ds = 48000 // rate
        n = x.shape[0] // (ds * 120) * ds * 120
        # crude decimation of a band-limited copy: enough to exercise the rate paths (quality is not asserted high)
        spec = np.fft.rfft(x[:n], axis=0)
        spec[int(0.45 * n / ds):] = 0
        y = np.fft.irfft(spec, n, axis=0)[::ds]
        q, err = oc.compare(x[:n], y, rate, True)
        assert np.isfinite(q) and err >= 0
    with pytest.raises(ValueError):
        oc.compare(x, x[:-2], 48000, True)
    with pytest.raises(ValueError):
        oc.compare(x[:100], x[:100], 48000, True)
    with pytest.raises(ValueError):
        oc.compare(x, x, 44100, True)
