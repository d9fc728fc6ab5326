"""RFC 6716 4.5.1 redundant frames (5 ms of CELT bridging a CELT <-> SILK / hybrid switch), which the reference parses and
drops (decoder.rs:84-115).  No expected PCM exists for them (no .dec files: parity unpinned), so the oracle is checked
through what the frames are for -- the switch points of the RFC vectors must come out smooth -- and the product against the
oracle at those points."""
import ctypes as C

import numpy as np
import pytest

import orc
from gpu_util import check_pcm
from test_emu_silk import EmuDecoder, emu  # noqa: F401  (fixture)


def _decode(vid, npk, drop):
    L = orc.lib()
    cnt = (C.c_long * 2).in_dll(L, "orc_redundant_frames")
    _, pk = orc.load_vectors(ids=[vid], max_packets=npk)[vid]
    L.orc_set_drop_redundancy(drop)
    try:
        d = orc.Decoder(2)
        outs, events = [], []
        for i, p in enumerate(pk):
            b = (cnt[0], cnt[1])
            outs.append(d.decode(bytes(p)))
            if cnt[0] > b[0]:
                events.append((i, "silk_to_celt"))
            if cnt[1] > b[1]:
                events.append((i, "celt_to_silk"))
    finally:
        L.orc_set_drop_redundancy(0)
    return outs, events


def _roughness(outs, i, kind):
    """energy of the second difference over the 5 ms either side of the switch the redundant frame of packet i bridges"""
    j = i + 1 if kind == "silk_to_celt" else i
    x = np.concatenate([outs[j - 1][-240:], outs[j][:480]])[:, 0].astype(np.float64)
    return float((np.diff(x, 2) ** 2).sum())


def test_vectors_hold_redundant_frames():
    L = orc.lib()
    cnt = (C.c_long * 2).in_dll(L, "orc_redundant_frames")
    got = {}
    for vid in (8, 9, 10, 12):
        _, ev = _decode(vid, None, 0)
        got[vid] = (sum(k == "silk_to_celt" for _, k in ev), sum(k == "celt_to_silk" for _, k in ev))
    assert got == {8: (1, 0), 9: (1, 0), 10: (8, 8), 12: (6, 6)}, got


def test_redundant_frames_smooth_the_mode_switches():
    """vector 10 switches between hybrid and CELT 16 times, each bridged by a redundant frame: with the frames decoded and
    cross-faded the switch is two to three orders of magnitude smoother than with the frames dropped"""
    with_r, ev = _decode(10, None, 0)
    dropped, _ = _decode(10, None, 1)
    assert len(ev) == 16
    ratios = []
    for i, kind in ev:
        a, b = _roughness(with_r, i, kind), _roughness(dropped, i, kind)
        assert a < b, (i, kind, a, b)
        ratios.append(b / a)
    assert np.median(ratios) > 100, ratios
    # ... and as smooth as the vector is away from any switch
    calm = np.median([_roughness(with_r, i, "celt_to_silk") for i in range(100, 900, 50)])
    assert max(_roughness(with_r, i, k) for i, k in ev) < 50 * calm


def test_product_matches_oracle_across_redundant_frames(emu):  # noqa: F811
    """host parser + kernel bodies (g++ emulation) against the oracle over the switch-heavy stretch of vector 10 (packets
    940 .. 1250 hold six redundant frames, both directions) -- decoded from the start: the state carries"""
    _, pk = orc.load_vectors(ids=[10], max_packets=1250)[10]
    d, ref = EmuDecoder(emu, 2), orc.Decoder(2)
    nred = 0
    for i, p in enumerate(pk):
        want = ref.decode(p)
        out = d.decode(p)
        nred += sum(d.hp.fi[f].has_celt > 1 for f in range(d.nframes))
        if i >= 900:
            check_pcm(want, out, f"vector 10 packet {i}")
    assert nred >= 6
