"""Synthetic Opus streams of BASELINE.json's shapes.

C2: CELT-only fullband stereo 20 ms: packet = TOC 0xFC (config 31, stereo, code 0; packet.rs:219-221) +
P payload bytes, i.i.d. uniform from xoshiro256** seeded with splitmix64(0x4F505553 ^ stream_id).  Every byte
string is a decodable CELT frame (the range decoder is total) and uniform bits yield symbols distributed
per the codec's own probability models.
"""
import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
        return x, z ^ (z >> np.uint64(31))


def _rotl(x, k):
    return ((x << np.uint64(k)) | (x >> np.uint64(64 - k))) & _M


class Xoshiro256ss:
    """xoshiro256** vectorised over streams: one generator per stream id."""

    def __init__(self, stream_ids):
        x = np.uint64(0x4F505553) ^ np.asarray(stream_ids, dtype=np.uint64)
        s = []
        for _ in range(4):
            x, z = _splitmix64(x)
            s.append(z)
        self.s = s

    def next(self):
        s0, s1, s2, s3 = self.s
        with np.errstate(over="ignore"):
            r = (_rotl((s1 * np.uint64(5)) & _M, 7) * np.uint64(9)) & _M
            t = (s1 << np.uint64(17)) & _M
            s2 = s2 ^ s0
            s3 = s3 ^ s1
            s1 = s1 ^ s2
            s0 = s0 ^ s3
            s2 = s2 ^ t
            s3 = _rotl(s3, 45)
        self.s = [s0, s1, s2, s3]
        return r


def celt_fb_stereo_packets(stream_ids, nframes, payload=160, toc=0xFC):
    """-> uint8 array [nstreams, nframes, 1 + payload]"""
    ids = np.asarray(stream_ids, dtype=np.uint64)
    g = Xoshiro256ss(ids)
    words = (payload + 7) // 8
    out = np.empty((len(ids), nframes, 1 + payload), np.uint8)
    out[:, :, 0] = toc
    for f in range(nframes):
        w = np.stack([g.next() for _ in range(words)], axis=1)            # [streams, words] uint64
        b = w.view(np.uint8).reshape(len(ids), words * 8)                 # little-endian bytes
        out[:, f, 1:] = b[:, :payload]
    return out
