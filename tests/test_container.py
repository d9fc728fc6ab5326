"""Container layer (include/opusgpu.h: opusgpu_opus_head_parse, opusgpu_demux_*): Matroska and Ogg files are written
here from the packets of the RFC vectors (tests/golden/opus_vectors.bin) and must come back packet for packet.  When
the reference's own `.mka` assets are present (this container only, never on the GPU box) they are demuxed too."""
import os
import struct

import pytest

import orc
from opus_amd import Demuxer, OpusError, _abi as A, parse_opus_head


# ------------------------------------------------------------------ writers (test-side, independent of the demuxer)
def _vint(n, width=None):
    w = width or next(k for k in range(1, 9) if n < (1 << (7 * k)) - 1)
    return ((1 << (7 * w)) | n).to_bytes(w, "big")


def _el(eid, payload):
    return bytes.fromhex(eid) + _vint(len(payload)) + payload


def _uint(v):
    return v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")


def _xiph_sizes(sizes):
    out = b""
    for s in sizes:
        out += b"\xff" * (s // 255) + bytes([s % 255])
    return out


def _ebml_sizes(sizes):
    out = _vint(sizes[0])
    for a, b in zip(sizes, sizes[1:]):
        d = b - a
        w = next(k for k in range(1, 9) if -((1 << (7 * k - 1)) - 1) <= d <= (1 << (7 * k - 1)) - 1)
        out += _vint(d + (1 << (7 * w - 1)) - 1, w)
    return out


def make_mka(extra, packets, lacing=None, unknown_sizes=False, other_track=False, block_groups=False):
    """lacing: None, "xiph", "ebml" or "fixed" (frames per block = 3)"""
    head = _el("1A45DFA3", _el("4282", b"matroska"))
    info = _el("1549A966", _el("2AD7B1", _uint(1000000)))
    entries = b""
    if other_track:
        entries += _el("AE", _el("D7", _uint(1)) + _el("86", b"A_VORBIS") + _el("63A2", b"junk"))
    entries += _el("AE", _el("D7", _uint(2)) + _el("86", b"A_OPUS") + _el("63A2", extra))
    tracks = _el("1654AE6B", entries)
    clusters = b""
    per = 3 if lacing else 1
    groups = [packets[i:i + per] for i in range(0, len(packets), per)]
    for ci in range(0, len(groups), 20):
        body = _el("E7", _uint(ci * 60))
        for gi, g in enumerate(groups[ci:ci + 20]):
            flags, lace = 0, b""
            if lacing and len(g) > 1:
                sizes = [len(x) for x in g]
                if lacing == "fixed" and len(set(sizes)) == 1:
                    flags, lace = 0x04, bytes([len(g) - 1])
                elif lacing == "ebml":
                    flags, lace = 0x06, bytes([len(g) - 1]) + _ebml_sizes(sizes[:-1])
                else:
                    flags, lace = 0x02, bytes([len(g) - 1]) + _xiph_sizes(sizes[:-1])
            blk = _vint(2) + struct.pack(">h", gi * 20) + bytes([flags | (0x80 if not block_groups else 0)]) + lace + b"".join(g)
            if other_track and gi % 4 == 0:
                body += _el("A3", _vint(1) + struct.pack(">h", gi * 20) + b"\x80" + b"not opus")
            body += _el("A0", _el("A1", blk)) if block_groups else _el("A3", blk)
        if unknown_sizes:
            clusters += bytes.fromhex("1F43B675") + b"\x01\xff\xff\xff\xff\xff\xff\xff" + body
        else:
            clusters += _el("1F43B675", body)
    seg_body = _el("114D9B74", b"") + info + tracks + _el("EC", b"\0" * 17) + clusters
    if unknown_sizes:
        return head + bytes.fromhex("18538067") + b"\x01\xff\xff\xff\xff\xff\xff\xff" + seg_body
    return head + _el("18538067", seg_body)


def _ogg_crc(b):
    crc = 0
    for v in b:
        crc ^= v << 24
        for _ in range(8):
            crc = ((crc << 1) ^ 0x04C11DB7) & 0xFFFFFFFF if crc & 0x80000000 else (crc << 1) & 0xFFFFFFFF
    return crc


def make_ogg(extra, packets, serial=0x1234, per_page=4, max_segs=255):
    """pages of whole packets where they fit; a packet longer than a page's segment table continues on the next page"""
    pages, seq, granule = [], 0, 0

    def page(segs, body, htype, gp):
        nonlocal seq
        h = b"OggS\0" + bytes([htype]) + struct.pack("<q", gp) + struct.pack("<II", serial, seq) + b"\0\0\0\0" + bytes([len(segs)]) + bytes(segs)
        raw = bytearray(h + body)
        raw[22:26] = struct.pack("<I", _ogg_crc(raw))
        pages.append(bytes(raw))
        seq += 1

    def lace(n):
        return [255] * (n // 255) + [n % 255]
    page(lace(len(extra)), extra, 2, 0)
    tags = b"OpusTags" + struct.pack("<I", 4) + b"test" + struct.pack("<I", 0)
    page(lace(len(tags)), tags, 0, 0)
    segs, body, cont = [], b"", False
    for i, p in enumerate(packets):
        granule += 960
        ls = lace(len(p))
        q = 0
        for s in ls:
            if len(segs) == max_segs:
                page(segs, body, 1 if cont else 0, -1 if not any(x < 255 for x in segs) else granule - 960)
                cont = q > 0                       # the packet in flight continues (unless it starts the new page)
                segs, body = [], b""
            segs.append(s)
            body += p[q:q + s]
            q += s
        if (i + 1) % per_page == 0 or i + 1 == len(packets):
            page(segs, body, (1 if cont else 0) | (4 if i + 1 == len(packets) else 0), granule)
            segs, body, cont = [], b"", False
    return b"".join(pages)


@pytest.fixture(scope="module")
def vec():
    return orc.load_vectors(ids=[1, 8, 10], max_packets=120)


# ------------------------------------------------------------------ OpusHead
def test_opus_head_of_the_vectors(vec):
    for vid, (extra, _) in vec.items():
        h = parse_opus_head(extra)
        assert h == dict(version=1, channels=2, pre_skip=0, input_sample_rate=44100, output_gain_q8=0, mapping_family=0,
                         stream_count=1, coupled_count=1, mapping=[0, 1])


def test_opus_head_fields_and_errors():
    base = b"OpusHead" + bytes([1, 1]) + struct.pack("<HIh", 312, 48000, -768) + b"\0"
    h = parse_opus_head(base)
    assert (h["channels"], h["pre_skip"], h["output_gain_q8"], h["coupled_count"], h["mapping"]) == (1, 312, -768, 0, [0])
    fam1 = b"OpusHead" + bytes([1, 6]) + struct.pack("<HIh", 0, 48000, 0) + bytes([1, 4, 2, 0, 4, 1, 2, 3, 5])
    h = parse_opus_head(fam1)
    assert (h["stream_count"], h["coupled_count"], h["mapping"]) == (4, 2, [0, 4, 1, 2, 3, 5])
    for bad, code in [(b"OpusHeaX" + base[8:], A.CONFIG_INVALID), (base[:18], A.CONFIG_INVALID),
                      (base[:9] + b"\0" + base[10:], A.CONFIG_INVALID),               # zero channels
                      (base[:9] + b"\3" + base[10:], A.CONFIG_INVALID),               # family 0 with three channels
                      (fam1[:24], A.CONFIG_INCOMPLETE),                               # decoder.rs:167
                      (fam1[:19] + bytes([2, 3]) + fam1[21:], A.CONFIG_INVALID),      # more coupled than streams
                      (fam1[:21] + bytes([9]) + fam1[22:], A.CONFIG_INVALID),         # mapping index out of range
                      (base[:8] + b"\x11" + base[9:], A.UNSUPPORTED)]:
        with pytest.raises(OpusError) as e:
            parse_opus_head(bad)
        assert e.value.code == code


# ------------------------------------------------------------------ Matroska
@pytest.mark.parametrize("kw", [dict(), dict(lacing="xiph"), dict(lacing="ebml"), dict(lacing="fixed"), dict(unknown_sizes=True),
                                dict(other_track=True), dict(block_groups=True, lacing="xiph")])
def test_matroska_roundtrip(vec, kw):
    for vid, (extra, pk) in vec.items():
        if kw.get("lacing") == "fixed":
            pk = [p for p in pk if len(p) == len(pk[0])] * 3      # equal-size frames only
        d = Demuxer(make_mka(extra, pk, **kw))
        assert d.container == "matroska" and d.extradata == extra
        assert len(d) == len(pk)
        assert list(d) == pk
        assert d.packet(0)[1] == 0 and d.packet(len(pk) - 1)[1] >= 0


def test_matroska_malformed(vec):
    extra, pk = vec[1]
    good = make_mka(extra, pk[:10])
    Demuxer(good).close()
    for bad in (good[:40], good[:-7], b"\x1a\x45\xdf\xa3" + b"\xff" * 40, good.replace(b"A_OPUS", b"A_OPUZ"), b"RIFFxxxxWAVE"):
        with pytest.raises(OpusError) as e:
            Demuxer(bad)
        assert e.value.code == A.INVALID_DATA


def test_matroska_nesting_is_bounded(vec):
    """master elements are entered only below their proper parent, so a crafted file cannot drive the walker deep:
    100 000 nested BlockGroups (2 bytes a level: 0xA0 + unknown... here 0xA0 with a size that covers the rest) in a
    Cluster are malformed, not a stack overflow; the same for Segments in Segments and Clusters in Clusters"""
    extra, pk = vec[1]
    head = _el("1A45DFA3", _el("4282", b"matroska"))
    tracks = _el("1654AE6B", _el("AE", _el("D7", _uint(2)) + _el("86", b"A_OPUS") + _el("63A2", extra)))
    depth = 100000
    nest = b""
    for _ in range(depth):                                    # innermost first; sizes grow, so widths are minimal
        nest = b"\xa0" + _vint(len(nest)) + nest
    for bad in (head + _el("18538067", tracks + _el("1F43B675", nest)),
                head + _el("18538067", tracks + _el("1F43B675", _el("1F43B675", _el("1F43B675", b"")))),
                head + _el("18538067", _el("18538067", tracks))):
        with pytest.raises(OpusError) as e:
            Demuxer(bad)
        assert e.value.code == A.INVALID_DATA
    # many unknown-size Clusters in a row are walked iteratively (each ends where the next one starts)
    blk = _vint(2) + struct.pack(">h", 0) + b"\x80" + pk[0]
    many = b"".join(bytes.fromhex("1F43B675") + b"\x01\xff\xff\xff\xff\xff\xff\xff" + _el("E7", _uint(i)) + _el("A3", blk)
                    for i in range(20000))
    d = Demuxer(head + bytes.fromhex("18538067") + b"\x01\xff\xff\xff\xff\xff\xff\xff" + tracks + many)
    assert len(d) == 20000 and d.packet(19999)[0] == pk[0]


def test_matroska_track_needs_a_usable_opus_head(vec):
    """an A_OPUS track with an empty or short CodecPrivate, or a family != 0 header cut short, does not open"""
    extra, pk = vec[1]
    fam1_short = extra[:18] + b"\x01"                           # mapping family 1 without its stream counts / mapping
    for bad_extra in (b"", extra[:12], fam1_short):
        with pytest.raises(OpusError) as e:
            Demuxer(make_mka(bad_extra, pk[:3]))
        assert e.value.code == A.INVALID_DATA


# ------------------------------------------------------------------ Ogg
@pytest.mark.parametrize("per_page,max_segs", [(4, 255), (1, 255), (7, 3)])
def test_ogg_roundtrip(vec, per_page, max_segs):
    for vid, (extra, pk) in vec.items():
        d = Demuxer(make_ogg(extra, pk, per_page=per_page, max_segs=max_segs))
        assert d.container == "ogg" and d.extradata == extra
        assert list(d) == pk
        assert d.packet(len(pk) - 1)[1] == 960 * len(pk)         # last packet ends its page: granule position


def test_ogg_crc_and_foreign_stream(vec):
    extra, pk = vec[10]
    good = make_ogg(extra, pk[:30])
    other = make_ogg(b"NotOpus!" + extra[8:], [b"x" * 10] * 5, serial=77)
    # a foreign logical stream multiplexed in front is ignored
    assert list(Demuxer(other + good)) == pk[:30]
    bad = bytearray(good)
    bad[len(good) // 2] ^= 0x40
    with pytest.raises(OpusError):
        Demuxer(bytes(bad))
    with pytest.raises(OpusError):
        Demuxer(good[:-5])
    with pytest.raises(OpusError):
        Demuxer(other)                                            # no Opus stream at all


# ------------------------------------------------------------------ the reference's own assets, when present
@pytest.mark.parametrize("vid", list(range(1, 13)))
def test_reference_assets(vid):
    path = f"/root/reference/assets/testvector{vid:02d}.mka"
    if not os.path.exists(path):
        pytest.skip("reference assets not present (GPU box)")
    extra, pk = orc.load_vectors(ids=[vid])[vid]
    d = Demuxer(open(path, "rb").read())
    assert d.extradata == extra
    assert list(d) == pk


def test_demuxers_under_sanitizers_on_mutated_images(tmp_path):
    """tools/demux_fuzz_asan.cpp: the container layer built with AddressSanitizer + UBSan, fed Matroska and Ogg images written here
    and then mutated (bytes overwritten, runs of 0x00 / 0xFF, truncations, pieces copied or inserted): opened or refused, every
    packet handed back read to its last byte, no report.  (While it was written: 4 x 10^6 images; it found a shift by a width
    that a failed EBML lacing size had not set.)"""
    import random
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    rnd = random.Random(1)
    extra = b"OpusHead" + bytes([1, 2, 0, 0]) + (48000).to_bytes(4, "little") + bytes([0, 0, 0])
    pk = [bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 300))) for _ in range(40)]
    images = {"a.mka": make_mka(extra, pk), "b.mka": make_mka(extra, pk, lacing="ebml"), "c.mka": make_mka(extra, pk, lacing="xiph"),
              "d.mka": make_mka(extra, pk, unknown_sizes=True, other_track=True, block_groups=True),
              "e.ogg": make_ogg(extra, pk), "f.ogg": make_ogg(extra, pk, per_page=1, max_segs=3)}
    files = []
    for name, img in images.items():
        files.append(str(tmp_path / name))
        open(files[-1], "wb").write(img)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "demux_fuzz_asan")
    r = subprocess.run(["g++", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-std=c++17", "-o", exe,
                        os.path.join(root, "tools", "demux_fuzz_asan.cpp"), os.path.join(root, "opus_amd", "csrc", "host", "container.cpp")],
                       capture_output=True, text=True)
    if r.returncode and "sanitize" in r.stderr and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, "1", "100000"] + files, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
    opened = int(r.stdout.split("demux_fuzz_asan:")[1].split()[0])
    assert opened > 5000, r.stdout                      # (the mutations leave a fair share of the images readable)
