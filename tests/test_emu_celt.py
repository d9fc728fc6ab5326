"""CPU CI of the CELT kernel logic: tests/emu builds the kernel body (opus_amd/csrc/hip/celt_frame.h) with g++, lanes
run one after the other (hip/wg.h), and the result is compared with the oracle.  A fixture for boxes without a GPU;
the parity tests proper are tests/test_gpu_*.py, which run the hipcc-built kernel through the C ABI."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from gpu_util import HostParser, check_pcm
from opus_amd import _abi as A, build, synth


@pytest.fixture(scope="module")
def emu():
    lib = C.CDLL(build.build_emu())
    lib.emu_launch_celt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


def _run(emu, packets, channels):
    hp = HostParser(channels)
    ref = orc.Decoder(channels, trace=True)
    st = A.CeltState()
    pcm = np.zeros(5760 * channels, np.float32)
    for i, p in enumerate(packets):
        n, nc, nl, _ = hp.parse(p)
        assert n > 0 and nc == n
        r = ref.decode(p)
        off = 0
        for f in range(n):
            hp.celt[f].stream = 0
            hp.celt[f].pcm_off = off
            off += hp.fi[f].frame_size
        pulses = np.zeros((n, 2, 960), np.int16)
        dbg = A.CeltDebug(pulses.ctypes.data, None, None, None)
        rc = emu.emu_launch_celt(hp.celt, hp.leaves, n, 1, C.byref(st), pcm.ctypes.data, 5760 * channels, C.byref(dbg))
        assert rc == 0, f"packet {i}: the two forms of the band plan differ on {rc} frame(s)"
        check_pcm(r, pcm[: off * channels].reshape(off, channels), f"packet {i}")
        t = ref.trace
        assert np.array_equal(pulses[n - 1][: t.C], np.ctypeslib.as_array(t.pulses)[: t.C]), f"packet {i}: PVQ pulses"


@pytest.mark.parametrize("vid,n,ch", [(11, 120, 2), (1, 300, 2), (7, 4186, 2), (7, 1200, 1)])
def test_emu_vectors(emu, vid, n, ch):
    _, pk = orc.load_vectors(ids=[vid], max_packets=n)[vid]
    _run(emu, pk, ch)


def _run_grouped(emu, packets, channels, per_group, reset_at=()):
    """`per_group` packets (each of one or more frames, sizes may differ) go through one launch as ONE tail group: the run
    keeps the stream's state in LDS across its frames.  reset_at: packet indices in front of which the stream is flushed
    (a reset flag inside a run when the index is not a multiple of per_group)."""
    hp = HostParser(channels)
    ref = orc.Decoder(channels)
    st = A.CeltState()
    HS = C.sizeof(A.CeltHdr)
    for g0 in range(0, len(packets), per_group):
        hdrs, leaves, want, off = [], [], [], 0
        for i in range(g0, min(g0 + per_group, len(packets))):
            if i in reset_at:
                hp.lib.opusgpu_parser_reset(hp.h)
                ref = orc.Decoder(channels)
            n, nc, nl, _ = hp.parse(packets[i])
            assert n > 0 and nc == n
            want.append(ref.decode(packets[i]))
            base = sum(len(x) for x in leaves) // 16
            for f in range(n):
                hp.celt[f].stream = 0
                hp.celt[f].pcm_off = off
                hp.celt[f].leaf_begin += base
                off += hp.fi[f].frame_size
                hdrs.append(bytes(C.string_at(C.addressof(hp.celt[f]), HS)))
            leaves.append(bytes(C.string_at(C.addressof(hp.leaves), nl * 16)))
        hb, lb = b"".join(hdrs), b"".join(leaves) or bytes(16)
        pcm = np.zeros(off * channels + 16, np.float32)
        rc = emu.emu_launch_celt(hb, lb, len(hdrs), len(hdrs), C.byref(st), pcm.ctypes.data, off * channels + 16, None)
        assert rc == 0, f"group at packet {g0}: the two forms of the band plan differ on {rc} frame(s)"
        r = np.concatenate(want)
        check_pcm(r, pcm[: off * channels].reshape(off, channels), f"group at packet {g0}")


@pytest.mark.parametrize("vid,n,ch,per_group", [(11, 96, 2, 8), (7, 2400, 2, 5), (7, 900, 1, 3), (1, 120, 2, 2)])
def test_emu_runs_keep_state_in_lds(emu, vid, n, ch, per_group):
    """several packets per tail group (vector 07: frame sizes 2.5 .. 20 ms change from packet to packet, transients,
    post-filter on and off): ring history, IMDCT tail and de-emphasis memory carried in LDS across the run's frames,
    written back once, picked up by the next launch"""
    _, pk = orc.load_vectors(ids=[vid], max_packets=n)[vid]
    _run_grouped(emu, pk, ch, per_group)


def test_emu_reset_inside_a_run(emu):
    pk = synth.celt_fb_stereo_packets(range(2), 24, 160)
    for s in range(2):
        _run_grouped(emu, [pk[s, f].tobytes() for f in range(24)], 2, 6, reset_at=(3, 8, 12, 17))
    _, pk7 = orc.load_vectors(ids=[7], max_packets=600)[7]
    _run_grouped(emu, pk7, 2, 4, reset_at=tuple(range(5, 600, 13)))


@pytest.fixture(scope="module")
def emu_small():
    """same kernel body with room for 4 fill ops in LDS: the rest are read from the plan in global memory"""
    lib = C.CDLL(build.build_emu("_ops4", ["OG_LDS_OPS=4"]))
    lib.emu_launch_celt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return lib


@pytest.mark.parametrize("payload", [8, 24, 80])
def test_emu_low_rate_many_fill_leaves(emu, emu_small, payload):
    """starved frames: most bands carry no pulses, up to 42 noise / folded leaves per frame over several levels"""
    pk = synth.celt_fb_stereo_packets(range(3), 30, payload)
    for lib in (emu, emu_small):
        for s in range(3):
            _run(lib, [pk[s, f].tobytes() for f in range(30)], 2)


@pytest.mark.parametrize("payload", [160, 320])
def test_emu_synthetic(emu, payload):
    pk = synth.celt_fb_stereo_packets(range(6), 25, payload)
    for s in range(6):
        _run(emu, [pk[s, f].tobytes() for f in range(25)], 2)


@pytest.mark.parametrize("channels", [1, 2])
def test_emu_synthetic_every_celt_config(emu, channels):
    """random payloads under every CELT-only TOC: NB / WB / SWB / FB x 2.5 / 5 / 10 / 20 ms, mono and stereo packets, two rates,
    into a mono and a stereo decoder.  (The RFC vectors and the 20 ms synthetic streams leave corners dark: a dual-stereo band of
    one bin at 2.5 ms put its second sign on the wrong channel until tests/test_loopback.py found it.)"""
    for config in range(16, 32):
        lm = config & 3
        for stereo in (0, 1):
            for rate in (1, 3):
                payload = max(6, (10 << lm) * rate * (1 + stereo))
                pk = synth.celt_fb_stereo_packets([1000 * config + 10 * stereo + rate], 10, payload, toc=config << 3 | stereo << 2)
                _run(emu, [pk[0, f].tobytes() for f in range(10)], channels)


def test_kernel_cwrsi_kats_and_exhaustive(emu):
    """the kernel's branch-uniform cwrsi (celt_pvq.h) against the reference's known answers (celt/decoder.rs:2414-2434),
    the oracle on every index of small codebooks, and on random indices of large ones (incl. the largest, V < 2^32)"""
    L = orc.lib()
    emu.emu_cwrsi.restype = C.c_uint32
    emu.emu_cwrsi.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_int]

    def both(n, k, i, blocks=1):
        ry = (C.c_int * n)()
        rnorm = L.orc_cwrsi(n, k, i, ry)
        for probes in (2, 4, 0, 3, 22):                 # og_cwrsi; og_cwrsi_wide<4> (the kernel's for N <= 8); og_cwrsi_sparse (N > 24); wide<3>, wide<2>
            y, cm = (C.c_int32 * n)(), C.c_int32()
            norm = emu.emu_cwrsi(n, k, i, blocks, y, C.byref(cm), probes)
            assert norm == rnorm and list(y) == list(ry), (probes, n, k, i, list(y), list(ry))
            if blocks > 1:
                assert cm.value == L.orc_extract_collapse_mask(ry, n, blocks), (probes, n, k, i)
        return list(y), norm

    for kat in orc.kats()["cwrsi"]:
        y, norm = both(kat["n"], kat["k"], kat["i"])
        assert y == kat["y"] and norm == kat["norm"]
    for n in range(2, 8):
        for k in range(1, 7):
            for i in range(L.orc_pvq_v(n, k)):
                both(n, k, i, 2 if n % 2 == 0 else 1)
    rng = np.random.default_rng(7)
    row = [0, 176, 351, 525, 698, 870, 1041, 1131, 1178, 1207, 1226, 1240, 1248, 1254, 1257, 1272 - 15]   # celt/decoder.rs:410-413

    def in_table(a, b):                                 # is U(a, b) tabulated, i.e. does the codebook fit 32 bits
        lo, hi = min(a, b), max(a, b)
        return lo <= 14 and row[lo] + hi < row[lo + 1] + lo + 1

    big = []
    for n in (2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 16, 18, 22, 24, 32, 36, 44, 48, 64, 72, 88, 96, 144, 176):
        kmax = max(k for k in range(1, 177) if in_table(n, k) and in_table(n, k + 1))
        big += [(n, kmax), (n, max(1, kmax // 2)), (n, 1)]
    tested = 0
    for n, k in big:
        tested += 1
        v = L.orc_pvq_v(n, k)
        for i in [0, 1, v - 1, v - 2, v // 2] + [int(t) for t in rng.integers(0, v, 60)]:
            both(n, k, i, 8 if n % 8 == 0 else 1)
    assert tested >= 8
