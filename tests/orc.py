"""ctypes binding of the CPU oracle (oracle/liborc.so).  Test infrastructure only."""
import ctypes as C
import json
import os
import struct
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
_LIB = os.path.join(ROOT, "oracle", "liborc.so")

MAX_LEAVES = 1024


class Leaf(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("band", "ch", "off", "n", "k", "blocks", "kind")] + [
        ("idx", C.c_uint32), ("gain", C.c_float)]


class CeltTrace(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("lm", "start", "end", "C", "silence", "transient", "intra", "spread",
                                  "pf_on", "pf_period", "pf_gain_q", "pf_tapset",
                                  "alloc_trim", "intensity", "dual_stereo", "codedbands",
                                  "anticollapse_rsv", "anticollapse_on", "tell_end")]
        + [(n, C.c_uint32) for n in ("seed_in", "seed_bands", "seed_out")]
        + [(n, C.c_int32 * 21) for n in ("tf_change", "pulses_alloc", "fine_bits", "fine_prio", "boost",
                                         "band_b", "band_lowband", "itheta", "inv")]
        + [("energy", (C.c_float * 21) * 2), ("collapse", (C.c_uint8 * 21) * 2), ("pad_", C.c_uint8 * 2),
           ("nleaves", C.c_int32), ("leaf", Leaf * MAX_LEAVES),
           ("pulses", (C.c_int16 * 960) * 2), ("X", (C.c_float * 960) * 2), ("freq", (C.c_float * 960) * 2)])


class Packet(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("mode", "bandwidth", "frame_size", "stereo", "code", "nframes", "padding")] + [
        ("off", C.c_int * 48), ("len", C.c_int * 48)]


def lib():
    if not os.path.exists(_LIB):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    L = C.CDLL(_LIB)
    L.orc_cwrsi.restype = C.c_uint32
    L.orc_pvq_v.restype = C.c_uint32
    L.orc_celt_window.restype = C.POINTER(C.c_float)
    L.orc_resample_taps.restype = C.POINTER(C.c_float)
    for f in ("orc_decoder_create", "orc_silk_create", "orc_celt_create"):
        getattr(L, f).restype = C.c_void_p
    L.orc_decoder_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_decoder_decode_fec.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int]
    L.orc_decoder_destroy.argtypes = [C.c_void_p]
    L.orc_decoder_set_celt_trace.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_silk_destroy.argtypes = [C.c_void_p]
    L.orc_celt_destroy.argtypes = [C.c_void_p]
    L.orc_stereo_merge.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int]
    return L


def kats():
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        return json.load(f)


def load_vectors(ids=None, max_packets=None):
    """-> {id: (extradata bytes, [packet bytes, ...])} from tests/golden/opus_vectors.bin"""
    b = open(os.path.join(GOLDEN, "opus_vectors.bin"), "rb").read()
    assert b[:8] == b"OPVEC1\0\0"
    (nvec,) = struct.unpack_from("<I", b, 8)
    p, out = 12, {}
    for _ in range(nvec):
        vid, el = struct.unpack_from("<II", b, p); p += 8
        extra = b[p:p + el]; p += el
        (npk,) = struct.unpack_from("<I", b, p); p += 4
        pk = []
        for _ in range(npk):
            (l,) = struct.unpack_from("<H", b, p); p += 2
            pk.append(b[p:p + l]); p += l
        if ids is None or vid in ids:
            out[vid] = (extra, pk[:max_packets] if max_packets else pk)
    return out


class Decoder:
    """whole-packet oracle decoder -> (nsamples, channels) float32 @48 kHz"""

    def __init__(self, channels=2, trace=False):
        self.L = lib()
        self.ch = channels
        self.h = self.L.orc_decoder_create(channels)
        self.trace = CeltTrace() if trace else None
        if trace:
            self.L.orc_decoder_set_celt_trace(self.h, C.byref(self.trace))
        self.buf = np.zeros(5760 * channels, np.float32)

    def decode(self, pkt):
        n = self.L.orc_decoder_decode(self.h, bytes(pkt), len(pkt), self.buf.ctypes.data, 5760)
        if n < 0:
            raise ValueError(f"oracle decode error {n}")
        return self.buf[: n * self.ch].reshape(n, self.ch).copy()

    def decode_fec(self, pkt):
        """pkt follows a lost packet: the lost packet's audio recovered from pkt's LBRR frames (in-band FEC)"""
        n = self.L.orc_decoder_decode_fec(self.h, bytes(pkt), len(pkt), self.buf.ctypes.data, 5760)
        if n < 0:
            raise ValueError(f"oracle FEC decode error {n}")
        return self.buf[: n * self.ch].reshape(n, self.ch).copy()

    def __del__(self):
        try:
            self.L.orc_decoder_destroy(self.h)
        except Exception:
            pass
