"""Helpers for the GPU parity tests: device buffers come from torch (plumbing only)."""
import ctypes as C

import numpy as np

from opus_amd import _abi as A


def snr_db(ref, out):
    ref = ref.astype(np.float64)
    err = out.astype(np.float64) - ref
    return 10 * np.log10(max((ref ** 2).sum(), 1e-30) / max((err ** 2).sum(), 1e-30))


def check_pcm(ref, out, what="", context_peak=0.0):
    """BASELINE.json tolerance: max abs error <= 1e-4 full-scale and SNR >= 90 dB vs the oracle.
    Synthetic random-bit streams decode to amplitudes far above full scale (their band energies random-walk), so the
    absolute bound scales with the reference peak when that exceeds 1.
    context_peak (the fuzz runs pass it): the peak of the stream's last few blocks.  float32 rounding is relative to the
    largest values that went through the overlap, the comb filter's 1024-sample history and the de-emphasis memory, so a quiet
    block behind a loud one carries the loud one's rounding: the absolute bound then scales with that peak, and the SNR is asked
    only of blocks whose level is within 40 dB of it (found by the long GPU fuzz: a concealed frame at rms 0.1 whose first 240
    samples are the comb filter's echo of audio 60 dB louder)."""
    assert ref.shape == out.shape, (what, ref.shape, out.shape)
    scale = max(1.0, float(np.abs(ref).max()) if ref.size else 0.0, float(context_peak))
    e = float(np.abs(out - ref).max()) / scale if ref.size else 0.0
    assert e <= 1e-4, f"{what}: max abs error {e:.3e} (scaled) > 1e-4"
    # (the SNR is asked of audio above -120 dB full scale: below that -- the last ripples of an overlap fading out, 6e-7 at the
    # peak in the case the long GPU fuzz found -- it measures float32 rounding against nothing; the absolute bound still holds)
    ms = float((ref.astype(np.float64) ** 2).mean()) if ref.size else 0.0
    if ms > 1e-12 and ms ** 0.5 >= 1e-2 * float(context_peak):
        s = snr_db(ref, out)
        assert s >= 90.0, f"{what}: SNR {s:.1f} dB < 90"
    return e


class RecentPeak:
    """peak of a stream's last few reference blocks (check_pcm's context_peak)"""

    def __init__(self, depth=4):
        self.depth, self.peaks = depth, []

    def value(self):
        return max(self.peaks, default=0.0)

    def push(self, ref):
        self.peaks = (self.peaks + [float(np.abs(ref).max()) if ref.size else 0.0])[-self.depth:]


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

    def parse(self, pkt, fec=False):
        nc, nl, ns = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        b = bytes(pkt)
        fn = self.lib.opusgpu_parser_parse_fec if fec else self.lib.opusgpu_parser_parse
        n = fn(self.h, b, len(b), self.fi, self.celt, 48, C.byref(nc), self.leaves, self.max_leaves, C.byref(nl), self.silk, 16, C.byref(ns))
        return n, nc.value, nl.value, ns.value

    def reset(self):
        self.lib.opusgpu_parser_reset(self.h)

    def __del__(self):
        try:
            self.lib.opusgpu_parser_destroy(self.h)
        except Exception:
            pass


class CeltLaunch:
    """Raw layer-3b CELT launch on torch-owned device memory: sorts the PVQ leaves, uploads, launches K1 + K2/K3."""

    def __init__(self, nframes_max, device="cuda:0"):
        import torch
        self.torch = torch
        self.lib = A.load()
        self.dev = torch.device(device)
        self.work = torch.zeros(self.lib.opusgpu_celt_work_bytes(nframes_max), dtype=torch.uint8, device=self.dev)
        self.keep = []

    def upload(self, hdr_bytes, leaf_bytes, nframes):
        """hdr_bytes / leaf_bytes: numpy uint8 arrays (host side-info).  -> dict of device tensors + class counts"""
        torch = self.torch
        nleaves = leaf_bytes.size // 16
        refs = (A.LeafRef * max(nleaves, 1))()
        classes = (C.c_uint32 * 4)()
        hb = np.ascontiguousarray(hdr_bytes)
        lb = np.ascontiguousarray(leaf_bytes) if nleaves else np.zeros(16, np.uint8)
        n = self.lib.opusgpu_sort_pvq_leaves(hb.ctypes.data, nframes, lb.ctypes.data, refs, max(nleaves, 1), classes)
        rb = np.frombuffer(refs, dtype=np.uint8)[: max(n, 1) * 8].copy()
        return dict(hdr=torch.from_numpy(hb.copy()).to(self.dev), leaves=torch.from_numpy(lb.copy()).to(self.dev),
                    refs=torch.from_numpy(rb).to(self.dev), classes=classes, nframes=nframes)

    def launch(self, up, state, pcm, stride, dbg=None, stream=None, frames_per_group=1):
        rc = self.lib.opusgpu_launch_celt(up["hdr"].data_ptr(), up["leaves"].data_ptr(), up["nframes"], frames_per_group,
                                          up["refs"].data_ptr(), up["classes"], self.work.data_ptr(), state.data_ptr(),
                                          pcm.data_ptr(), stride, C.byref(dbg) if dbg is not None else None, stream)
        assert rc == 0, self.lib.opusgpu_last_error()


def oracle_check_last_step(pcm_rows, packets, streams, nframes_done, T, what="step"):
    """The last T frames of `streams` (rows of the last step's PCM, numpy [S, T*960*2]) against the CPU oracle, which
    decodes each of those streams from its first packet (decoder state carries across steps).  -> worst scaled error"""
    import orc
    worst = 0.0
    for s in streams:
        ref = orc.Decoder(2)
        for f in range(nframes_done):
            r = ref.decode(packets[s, f].tobytes())
            t = f - (nframes_done - T)
            if t >= 0:
                out = pcm_rows[s][t * 1920:(t + 1) * 1920].reshape(960, 2)
                worst = max(worst, check_pcm(r, out, f"{what}: stream {s} frame {f}"))
    return worst
