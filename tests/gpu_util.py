"""Helpers for the GPU parity tests: device buffers come from torch (plumbing only)."""
import ctypes as C

import numpy as np

from opus_amd import _abi as A


def snr_db(ref, out):
    ref = ref.astype(np.float64)
    err = out.astype(np.float64) - ref
    return 10 * np.log10(max((ref ** 2).sum(), 1e-30) / max((err ** 2).sum(), 1e-30))


def check_pcm(ref, out, what=""):
    """BASELINE.json tolerance: max abs error <= 1e-4 full-scale and SNR >= 90 dB vs the oracle.
    Synthetic random-bit streams decode to amplitudes far above full scale (their band energies random-walk), so the
    absolute bound scales with the reference peak when that exceeds 1."""
    assert ref.shape == out.shape, (what, ref.shape, out.shape)
    scale = max(1.0, float(np.abs(ref).max()))
    e = float(np.abs(out - ref).max()) / scale
    assert e <= 1e-4, f"{what}: max abs error {e:.3e} (scaled) > 1e-4"
    if float((ref.astype(np.float64) ** 2).sum()) > 1e-12:
        s = snr_db(ref, out)
        assert s >= 90.0, f"{what}: SNR {s:.1f} dB < 90"
    return e


class HostParser:
    """opusgpu_parser_* wrapper: packets -> side-info arrays (ctypes)."""

    def __init__(self, channels=2, max_leaves=1 << 16):
        self.lib = A.load()
        self.h = C.c_void_p()
        assert self.lib.opusgpu_parser_create(C.byref(self.h), channels) == 0
        self.fi = (A.FrameInfo * 48)()
        self.celt = (A.CeltHdr * 48)()
        self.leaves = (A.CeltLeaf * max_leaves)()
        self.silk = (A.SilkFrame * 16)()
        self.max_leaves = max_leaves

    def parse(self, pkt):
        nc, nl, ns = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        b = bytes(pkt)
        n = self.lib.opusgpu_parser_parse(self.h, b, len(b), self.fi, self.celt, 48, C.byref(nc), self.leaves, self.max_leaves,
                                          C.byref(nl), self.silk, 16, C.byref(ns))
        return n, nc.value, nl.value, ns.value

    def __del__(self):
        try:
            self.lib.opusgpu_parser_destroy(self.h)
        except Exception:
            pass
