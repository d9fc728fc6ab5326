"""Host layer (C++ parser behind the C ABI) against the CPU oracle: framing, side-info records, final range.
No GPU: these call only the opusgpu_parse_* / opusgpu_packet_parse entry points."""
import ctypes as C
import os

import numpy as np
import pytest

import orc
from gpu_util import HostParser
from opus_amd import _abi as A, parse_packet, synth

EB = [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 34, 40, 48, 60, 78, 100]


def test_abi_exports_every_declared_symbol():
    lib = A.load()
    for name in A.PROTOTYPES:
        assert hasattr(lib, name), name
    # include/opusgpu.h and the ctypes mirror agree on the record layouts
    assert lib.opusgpu_sizeof_celt_hdr() == C.sizeof(A.CeltHdr)
    assert lib.opusgpu_sizeof_celt_leaf() == C.sizeof(A.CeltLeaf) == 16
    assert lib.opusgpu_sizeof_celt_state() == C.sizeof(A.CeltState)
    assert lib.opusgpu_sizeof_silk_frame() == C.sizeof(A.SilkFrame)
    assert lib.opusgpu_strerror(A.INVALID_DATA) == b"invalid data"


def test_describe_matches_the_reference_descriptor():
    """Descriptor::describe of OPUS_DESCR (decoder.rs:190-197): codec, name and mime are the reference's strings"""
    d = A.load().opusgpu_describe().contents
    assert (d.codec, d.name, d.mime) == (b"opus", b"opus", b"audio/OPUS")
    assert d.desc and b"opus" in d.desc


def test_static_library_carries_the_same_entry_points():
    """libopusgpu.a (the static flavour north_star names) defines every symbol the header declares"""
    import os
    import subprocess
    a = os.path.join(A.ROOT, "opus_amd", "libopusgpu.a")
    assert os.path.exists(a), "build() archives the objects next to the shared library"
    out = subprocess.run(["nm", "--defined-only", "-g", a], capture_output=True, text=True).stdout
    defined = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(A.PROTOTYPES) <= defined, set(A.PROTOTYPES) - defined


def test_header_declares_what_the_mirror_binds():
    import os
    import re
    hdr = open(os.path.join(A.ROOT, "include", "opusgpu.h")).read()
    declared = set(re.findall(r"\b(opusgpu_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(A.PROTOTYPES), declared ^ set(A.PROTOTYPES)


def _toc(config, stereo, code):
    return bytes([(config << 3) | (stereo << 2) | code])


def test_packet_framing_codes():
    # code 0
    p = parse_packet(_toc(31, 1, 0) + bytes(10))
    assert (p["mode"], p["bandwidth"], p["frame_size"], p["stereo"], p["frames"]) == (2, 4, 960, 1, [(1, 10)])
    # code 1: two equal halves (the reference yields one chunk, packet.rs:133: RFC 6716 3.2.3 is authoritative)
    p = parse_packet(_toc(9, 0, 1) + bytes(8))
    assert p["frames"] == [(1, 4), (5, 4)] and p["mode"] == 0 and p["bandwidth"] == 2 and p["frame_size"] == 960
    with pytest.raises(Exception):
        parse_packet(_toc(9, 0, 1) + bytes(7))
    # code 2: first length coded, 1-byte and 2-byte forms
    p = parse_packet(_toc(13, 0, 2) + bytes([3]) + bytes(9))
    assert p["frames"] == [(2, 3), (5, 6)] and p["mode"] == 1 and p["frame_size"] == 960
    p = parse_packet(_toc(13, 0, 2) + bytes([252, 1]) + bytes(300))
    assert p["frames"] == [(3, 256), (259, 44)]
    # code 3 CBR with padding
    p = parse_packet(_toc(16, 0, 3) + bytes([0x40 | 3, 2]) + bytes(9) + bytes(2))
    assert p["frames"] == [(3, 3), (6, 3), (9, 3)] and p["padding"] == 2 and p["frame_size"] == 120
    # code 3 VBR: count-1 coded lengths, last frame = remainder (the reference drops it, packet.rs:177-188)
    p = parse_packet(_toc(16, 0, 3) + bytes([0x80 | 3, 2, 4]) + bytes(2 + 4 + 5))
    assert p["frames"] == [(4, 2), (6, 4), (10, 5)]


@pytest.mark.parametrize("pkt", [b"", _toc(16, 0, 3), _toc(16, 0, 3) + bytes([0]), _toc(3, 0, 3) + bytes([3]) + bytes(12),
                                 _toc(16, 0, 2) + bytes([200]) + bytes(5), _toc(31, 0, 0) + bytes(1276)])
def test_packet_framing_rejects_malformed(pkt):
    """the reference panics on an empty packet (packet.rs:215-225); here every malformed packet is InvalidData"""
    lib = A.load()
    p = A.Packet()
    assert lib.opusgpu_packet_parse(pkt, len(pkt), C.byref(p)) == A.INVALID_DATA


def test_framing_matches_oracle_on_all_vectors():
    L = orc.lib()
    for vid, (_, pk) in orc.load_vectors().items():
        for p in pk[:400]:
            a = parse_packet(p)
            o = orc.Packet()
            assert L.orc_packet_parse(bytes(p), len(p), C.byref(o)) == 0
            assert (a["mode"], a["bandwidth"], a["frame_size"], a["stereo"], a["code"], a["nframes"]) == (
                o.mode, o.bandwidth, o.frame_size, o.stereo, o.code, o.nframes)
            assert a["frames"] == [(o.off[i], o.len[i]) for i in range(o.nframes)]


def test_range_decoder_never_overruns_on_the_vectors():
    """every frame of every packet of the 12 RFC vectors: the host parser ends inside the frame's bit budget, and so does
    the oracle (which raises on an overrun).  tf / allocation / theta / bits2pulses have no reference KAT; a
    misread symbol would desynchronise the decoder and overrun some frame of these 1.1e4 packets almost surely."""
    frames = 0
    for vid, (_, pk) in sorted(orc.load_vectors().items()):
        hp = HostParser(2)
        for i, p in enumerate(pk):
            n, _, _, _ = hp.parse(p)
            assert n > 0, (vid, i, n)
            for f in range(n):
                # SILK-only frames are exempt: the encoder strips their trailing zero bytes (RFC 6716 4.1.1: the decoder
                # reads zeros past the end), so their tell may legitimately pass the stored length
                if hp.fi[f].has_celt:
                    assert hp.fi[f].tell_bits <= hp.fi[f].budget_bits, (vid, i, f, hp.fi[f].tell_bits, hp.fi[f].budget_bits)
            frames += n
        # the oracle over the whole vector (state carries across packets): any overrun raises
        d = orc.Decoder(2)
        for p in pk:
            d.decode(p)
    assert frames > 11000


def _compare_frame(hd, leaves, fi, t):
    assert (hd.lm, hd.start, hd.end) == (t.lm, t.start, t.end)
    assert bool(hd.flags & A.F_TRANSIENT) == bool(t.transient) and bool(hd.flags & A.F_SILENCE) == bool(t.silence)
    assert bool(hd.flags & A.F_ANTICOLLAPSE) == bool(t.anticollapse_on)
    assert (hd.spread & 3, hd.intensity, hd.seed, fi.final_range) == (t.spread, t.intensity, t.seed_in, t.seed_out)
    assert (hd.pf_period[2], hd.pf_tapset[2], hd.pf_gain_q[2]) == (t.pf_period, t.pf_tapset, t.pf_gain_q + 1 if t.pf_on else 0)
    for c in range(t.C):
        assert np.array_equal(np.array(hd.energy[c][:]), np.array(t.energy[c][:]))       # bit-exact f32
    M = 1 << t.lm
    for i in range(t.start, t.end):
        assert hd.band[i].tf_change == t.tf_change[i]
        lb = t.band_lowband[i]
        assert hd.band[i].lowband == (0xFF if lb < 0 else lb >> t.lm)
    assert hd.nleaves == t.nleaves
    # the host groups each band's leaves (X buffer then Y, fill leaves first, decode order inside each run)
    want = sorted((t.leaf[j] for j in range(t.nleaves)),
                  key=lambda ol: (ol.band, 2 * bool(ol.ch) + (0 if (ol.k == 0 and ol.kind != orc_LEAF_N1) else 1)))
    for j, ol in enumerate(want):
        l = leaves[hd.leaf_begin + j]
        assert (l.n, l.k, l.index, bool(l.meta & A.L_CH_Y), l.off, 1 << (l.meta & A.L_LOGB_MASK), l.band,
                bool(l.meta & A.L_N1)) == (ol.n, ol.k, ol.idx, bool(ol.ch), ol.off - M * EB[ol.band], ol.blocks, ol.band,
                                           ol.kind == orc_LEAF_N1)
        assert np.float32(l.gain) == np.float32(ol.gain)
    for i in range(t.start, t.end):
        bd = hd.band[i]
        grp = [l for l in want if l.band == i]
        assert bd.nx + bd.ny == len(grp) and bd.nx == sum(1 for l in grp if not l.ch)
        assert bd.fx == sum(1 for l in grp if not l.ch and l.k == 0 and l.kind != orc_LEAF_N1)
        assert bd.fy == sum(1 for l in grp if l.ch and l.k == 0 and l.kind != orc_LEAF_N1)


orc_LEAF_N1 = 4


@pytest.mark.parametrize("vid,step", [(1, 3), (7, 2), (11, 1)])
def test_celt_sideinfo_matches_oracle_trace(vid, step):
    """every parsed symbol of the CELT frame (energies, tf, allocation-derived leaves: N, K, PVQ index, gain) and the
    range decoder's final state equal the oracle's, on the reference's own CELT vectors"""
    _, pk = orc.load_vectors(ids=[vid])[vid]
    hp = HostParser(2)
    d = orc.Decoder(2, trace=True)
    for i, p in enumerate(pk):
        n, nc, nl, _ = hp.parse(p)
        d.decode(p)
        assert n > 0 and nc == n
        if i % step == 0:
            _compare_frame(hp.celt[n - 1], hp.leaves, hp.fi[n - 1], d.trace)      # the trace holds the packet's last frame


def test_celt_sideinfo_synthetic_c2():
    pk = synth.celt_fb_stereo_packets(range(8), 30, 160)
    for s in range(8):
        hp = HostParser(2)
        d = orc.Decoder(2, trace=True)
        for f in range(30):
            p = pk[s, f].tobytes()
            n, nc, nl, _ = hp.parse(p)
            d.decode(p)
            assert n == 1
            _compare_frame(hp.celt[0], hp.leaves, hp.fi[0], d.trace)


def test_synth_generator_is_deterministic_per_stream():
    a = synth.celt_fb_stereo_packets([5, 9], 3, 160)
    b = synth.celt_fb_stereo_packets([9], 3, 160)
    assert np.array_equal(a[1], b[0]) and not np.array_equal(a[0], a[1])
    assert (a[:, :, 0] == 0xFC).all() and a.shape == (2, 3, 161)


def test_sort_pvq_leaves_orders_by_size_then_pulses():
    """opusgpu_sort_pvq_leaves (host helper of the K1 launch): every PVQ leaf (K > 0, N > 1) exactly once, and the zero-pulse
    leaves that share a (band, buffer) with one (K1 stores their zeros), key (N, min(K, 63)) non-increasing, class counts =
    leaves with N > 64 / > 24 / > 8 / the rest"""
    lib = A.load()
    hp = HostParser(2)
    pk = synth.celt_fb_stereo_packets(range(4), 6, 160)
    N1 = 64                                              # OG_L_N1 (include/opusgpu_sideinfo.h)
    for s in range(4):
        for f in range(6):
            n, nc, nl, _ = hp.parse(pk[s, f].tobytes())
            assert n == 1 and nc == 1
            refs = (A.LeafRef * max(nl, 1))()
            classes = (C.c_uint32 * 4)()
            got = lib.opusgpu_sort_pvq_leaves(hp.celt, 1, hp.leaves, refs, nl, classes)
            leaves = [hp.leaves[i] for i in range(nl)]
            def shares_pvq(lf):
                bd = hp.celt[0].band[lf.band]
                return bd.ny > bd.fy if lf.meta & A.L_CH_Y else bd.nx > bd.fx
            want = [i for i, lf in enumerate(leaves) if not (lf.meta & N1) and (lf.k or shares_pvq(lf))]
            assert any(leaves[i].k == 0 for i in want) or s + f > 0
            assert got == len(want) and sorted(refs[i].leaf for i in range(got)) == want
            keys = [(leaves[refs[i].leaf].n, min(leaves[refs[i].leaf].k, 63)) for i in range(got)]
            assert keys == sorted(keys, reverse=True)
            ns = [leaves[i].n for i in want]
            assert list(classes) == [sum(x > 64 for x in ns), sum(24 < x <= 64 for x in ns), sum(8 < x <= 24 for x in ns), sum(x <= 8 for x in ns)]
            assert all((refs[i].frame & 0x3FFFFFFF) == 0 for i in range(got))
            # a buffer that is too small is refused, not overrun
            if got:
                assert lib.opusgpu_sort_pvq_leaves(hp.celt, 1, hp.leaves, refs, got - 1, classes) == 0


def test_tell_frac_thresholds_match_the_squaring_loop():
    """the host range decoder's tell_frac (host/rangedec.hpp) counts thresholds instead of running the reference's three
    squarings (entropy.rs:380-392): the recurrence must be monotonic over all 32768 arguments and step exactly at the
    constants in the header"""
    import re

    def frac(r):
        l = 0
        for _ in range(3):
            r = (r * r) >> 15
            b = r >> 16
            l = 2 * l + b
            r >>= b
        return l

    steps, prev = [], 0
    for r in range(1 << 15, 1 << 16):
        f = frac(r)
        assert f in (prev, prev + 1)
        if f != prev:
            steps.append(r)
        prev = f
    src = open(os.path.join(os.path.dirname(__file__), "..", "opus_amd", "csrc", "host", "rangedec.hpp")).read()
    body = src[src.index("int tell_frac() const"):src.index("void to_end()")]
    assert [int(x) for x in re.findall(r"r >= (\d+)u", body)] == steps and len(steps) == 7


def test_host_parser_under_sanitizers_on_arbitrary_bytes(tmp_path):
    """tools/parse_fuzz_asan.cpp: the host parser built with AddressSanitizer + UBSan (bounds, signed overflow) fed packets nobody
    would send -- random lengths, random TOCs, runs of 0x00 / 0xFF, losses, FEC calls: no report.  (Seed 1 is the run that found
    the NLSF stabilisation's i16 overflow -- a negative cosine-table index, decoder.rs:189 / :335 -- and the wrapping difference
    in the LPC stability test.)"""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "opus_amd", "csrc", "host")
    exe = str(tmp_path / "parse_fuzz_asan")
    srcs = [os.path.join(host, f + ".cpp") for f in ("celt_parse", "packet", "stream_parser", "silk_parse", "silk_tables", "celt_tables", "capi_parse")]
    cmd = ["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=shift", "-fno-sanitize-recover=all", "-std=c++17",
           "-fno-strict-aliasing", "-o", exe, os.path.join(root, "tools", "parse_fuzz_asan.cpp")] + srcs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "1", "100000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
