#!/usr/bin/env python3
"""Builds the committed golden fixtures from the reference tree (run in the build container only).

 * kats.json          - the known-answer vectors that the reference's own unit tests hold
                        (src/entropy.rs:82-257, src/celt/decoder.rs:2330-2481,
                        src/celt/imdct15.rs:213-251, src/silk/tests.rs:22-870), transcribed as data.
 * opus_vectors.bin   - the Opus packets + OpusHead of assets/testvector01..12.mka, demuxed from
                        Matroska (SimpleBlocks, no lacing) into a flat container:
                          magic "OPVEC1\\0\\0", u32 nvec, then per vector:
                          u32 id, u32 extradata_len, extradata, u32 npackets, npackets x (u16 len, bytes)

Nothing else reads /root/reference: tests, smoke() and bench.py only load these two files.
"""
import json
import re
import struct
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
HERE = Path(__file__).resolve().parent


# ------------------------------------------------------------------ KATs
def nums(txt):
    return [float(x) if ("." in x or "e" in x) else int(x)
            for x in re.findall(r"-?\d+\.\d+(?:e-?\d+)?|-?\d+", txt)]


def between(txt, start, end, pos=0):
    a = txt.index(start, pos)
    b = txt.index(end, a)
    return txt[a + len(start):b], b


def extract_kats():
    k = {}
    ent = (REF / "src/entropy.rs").read_text()
    body, p = between(ent, "fn reverse_bitread()", "fn decode_laplace()")
    buf, _ = between(body, "let buf = &[", "];")
    reads = re.findall(r"get_bits_32\((\d+)\), (\d+)\)", body)
    k["reverse_bitread"] = {"buf": nums(buf), "reads": [[int(a), int(b)] for a, b in reads]}
    body, _ = between(ent, "fn decode_laplace()", "/// Opus Range Decoder")
    buf, q = between(body, "let buf = [", "];")
    val, _ = between(body, "let val = [", "];", q)
    triples = re.findall(r"\((-?\d+), (\d+), (\d+)\)", val)
    k["decode_laplace"] = {"buf": nums(buf), "triples": [[int(a), int(b), int(c)] for a, b, c in triples]}

    celt = (REF / "src/celt/decoder.rs").read_text()
    tests = celt[celt.index("mod test {"):]
    body, _ = between(tests, "fn stereo_merge()", "fn extract_collapse_mask()")
    x, p = between(body, "let mut x = [", "];")
    y, p = between(body, "let mut y = [", "];", p)
    ox, p = between(body, "let ox = [", "];", p)
    oy, p = between(body, "let oy = [", "];", p)
    mid = float(re.search(r"let mid = ([\d.]+)f32", body).group(1))
    k["stereo_merge"] = {"x": nums(x), "y": nums(y), "ox": nums(ox), "oy": nums(oy), "mid": mid, "n": 96}
    k["extract_collapse_mask"] = [
        {"y": [0, 0, 1, -1, 4, 8, -4, 4], "b": 8, "mask": 252},
        {"y": [1, -2, 0, -2, 0, 2, 0, 0, 0, 1, 1, 0, 1, 1, -1, 0], "b": 4, "mask": 15},
    ]
    body, _ = between(tests, "fn cwrsi()", "fn haar1(")
    assert "68441748" in body and "66182001" in body
    k["cwrsi"] = [
        {"n": 8, "k": 22, "i": 68441748, "y": [0, 0, -1, -1, 4, 8, -4, 4], "norm": 114},
        {"n": 8, "k": 22, "i": 66182001, "y": [0, 0, 4, -11, -1, 1, -2, -3], "norm": 152},
    ]
    body, _ = between(tests, "fn haar1_32_1()", "fn haar1_16_2()")
    a, _ = between(body, "let mut a = [", "];")
    k["haar1_32_1"] = {"a": nums(a), "n0": 32, "stride": 1}
    body = tests[tests.index("fn haar1_16_2()"):]
    a, _ = between(body, "let mut a = [", "];")
    k["haar1_16_2"] = {"a": nums(a), "n0": 16, "stride": 2}

    im = (REF / "src/celt/imdct15.rs").read_text()
    body = im[im.index("fn fft5()"):]
    ref, _ = between(body, "let reference = [", "];")
    v = nums(ref)
    k["fft5"] = {"stride": 3, "out": [[v[i], v[i + 1]] for i in range(0, 10, 2)]}

    st = (REF / "src/silk/tests.rs").read_text()
    body, _ = between(st, "fn decode_midonly_to_stereo()", "fn decode_unmix()")
    pkt, p = between(body, "let in_slice = &[", "];")
    out, _ = between(body, "let outbuf = vec![", "];", p)
    k["silk_midonly_to_stereo"] = {"packet": nums(pkt), "right": nums(out), "left": nums(out)}
    body, _ = between(st, "fn decode_unmix()", "fn lsf_to_lpc()")
    pkt, p = between(body, "let in_slice = &[", "];")
    r, p = between(body, "let right = vec![", "];", p)
    l, p = between(body, "let left = vec![", "];", p)
    k["silk_unmix"] = {"packet": nums(pkt), "right": nums(r), "left": nums(l)}
    body = st[st.index("fn lsf_to_lpc()"):]
    lsf, p = between(body, "let lsf = vec![", "];")
    ref, _ = between(body, "let reference = [", "];", p)
    k["lsf_to_lpc"] = {"nlsf": nums(lsf.replace("i16", "")), "lpc": nums(ref)}
    return k


# ------------------------------------------------------------------ Matroska (EBML) demux
def vint(b, p, keep_marker=False):
    first = b[p]
    n = 1
    mask = 0x80
    while n <= 8 and not (first & mask):
        n += 1
        mask >>= 1
    v = first if keep_marker else first & (mask - 1)
    for i in range(1, n):
        v = (v << 8) | b[p + i]
    return v, p + n


MASTERS = {0x18538067, 0x1F43B675, 0x1654AE6B, 0xAE, 0xA0}


def demux(path):
    b = path.read_bytes()
    packets, extradata, codec = [], None, None

    def walk(p, end):
        nonlocal extradata, codec
        while p < end:
            eid, p = vint(b, p, keep_marker=True)
            p0 = p
            size, p = vint(b, p)
            unknown = size == (1 << (7 * (p - p0))) - 1   # all value bits set = unknown size
            e = end if unknown else min(end, p + size)
            if eid in MASTERS:
                walk(p, e)
            elif eid == 0x63A2:
                extradata = b[p:e]
            elif eid == 0x86:
                codec = b[p:e].decode()
            elif eid in (0xA3, 0xA1):
                _, q = vint(b, p)          # track number
                q += 3                     # timecode + flags
                assert (b[q - 1] & 0x06) == 0, "laced block"
                packets.append(b[q:e])
            p = e

    walk(0, len(b))
    assert codec == "A_OPUS", codec
    return extradata, packets


def main():
    kats = extract_kats()
    (HERE / "kats.json").write_text(json.dumps(kats))
    out = bytearray(b"OPVEC1\0\0")
    vecs = sorted((REF / "assets").glob("testvector*.mka"))
    out += struct.pack("<I", len(vecs))
    for v in vecs:
        vid = int(re.search(r"(\d+)", v.stem).group(1))
        extra, pk = demux(v)
        out += struct.pack("<II", vid, len(extra)) + extra + struct.pack("<I", len(pk))
        for d in pk:
            out += struct.pack("<H", len(d)) + d
        print(v.name, "extradata", len(extra), "packets", len(pk), "bytes", sum(map(len, pk)))
    (HERE / "opus_vectors.bin").write_bytes(out)
    print("kats:", {k: (len(v) if isinstance(v, list) else list(v)[:3]) for k, v in kats.items()})


if __name__ == "__main__":
    main()
