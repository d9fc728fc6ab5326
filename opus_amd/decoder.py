"""Host-side mirror of the reference decoder interface (lu-zero/opus src/decoder.rs) over the C ABI."""
import ctypes as C

import numpy as np

from . import _abi as A


class OpusError(Exception):
    """av_codec::error::Error as the reference uses it (InvalidData, ConfigurationInvalid, ...)."""

    def __init__(self, code, where=""):
        lib = A.load()
        msg = lib.opusgpu_strerror(code).decode()
        if code == A.DEVICE:
            msg += ": " + lib.opusgpu_last_error().decode()
        super().__init__(f"{where}: {msg}" if where else msg)
        self.code = code


def _chk(rc, where):
    if rc < 0:
        raise OpusError(rc, where)
    return rc


def parse_packet(pkt):
    """Packet::from_slice (packet.rs:202-317) -> dict."""
    p = A.Packet()
    b = bytes(pkt)
    _chk(A.load().opusgpu_packet_parse(b, len(b), C.byref(p)), "packet_parse")
    return dict(mode=p.mode, bandwidth=p.bandwidth, frame_size=p.frame_size, stereo=p.stereo, code=p.code,
                nframes=p.nframes, padding=p.padding, frames=[(p.off[i], p.len[i]) for i in range(p.nframes)])


def parse_opus_head(extra):
    """OpusHead -> dict (what Dec::configure reads, decoder.rs:133-182)."""
    h = A.OpusHead()
    b = bytes(extra)
    _chk(A.load().opusgpu_opus_head_parse(b, len(b), C.byref(h)), "opus_head_parse")
    return dict(version=h.version, channels=h.channels, pre_skip=h.pre_skip, input_sample_rate=h.input_sample_rate,
                output_gain_q8=h.output_gain_q8, mapping_family=h.mapping_family, stream_count=h.stream_count,
                coupled_count=h.coupled_count, mapping=list(h.mapping[: h.channels]))


class Demuxer:
    """Packets of the Opus track of a Matroska or Ogg file image (stands in for the matroska / av-format crates of the
    reference's tests, decoder.rs:202-232)."""

    def __init__(self, data):
        self._lib = A.load()
        self._h = C.c_void_p()
        b = bytes(data)
        _chk(self._lib.opusgpu_demux_open(C.byref(self._h), b, len(b)), "demux_open")
        cont, ex, el, n = C.c_int(), C.c_void_p(), C.c_size_t(), C.c_uint32()
        _chk(self._lib.opusgpu_demux_info(self._h, C.byref(cont), C.byref(ex), C.byref(el), C.byref(n)), "demux_info")
        self.container = "ogg" if cont.value else "matroska"
        self.extradata = C.string_at(ex, el.value)
        self.npackets = n.value

    def packet(self, i):
        """-> (bytes, pts)"""
        p, l, pts = C.c_void_p(), C.c_size_t(), C.c_int64()
        _chk(self._lib.opusgpu_demux_packet(self._h, i, C.byref(p), C.byref(l), C.byref(pts)), "demux_packet")
        return C.string_at(p, l.value), pts.value

    def __len__(self):
        return self.npackets

    def __iter__(self):
        for i in range(self.npackets):
            yield self.packet(i)[0]

    def close(self):
        if self._h:
            self._lib.opusgpu_demux_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """One stream: set_extradata -> configure -> (send_packet -> receive_frame)* -> flush (decoder.rs:42-188)."""

    def __init__(self, device=0):
        self._lib = A.load()
        self._h = C.c_void_p()
        _chk(self._lib.opusgpu_decoder_create(C.byref(self._h), device), "decoder_create")
        self._buf = np.zeros(5760 * 8, np.float32)      # up to 8 channels (mapping family 1); grown on demand for family 255

    def set_extradata(self, extra):
        b = bytes(extra)
        if len(b) > 9 and 5760 * b[9] > self._buf.size:
            self._buf = np.zeros(5760 * b[9], np.float32)
        _chk(self._lib.opusgpu_decoder_set_extradata(self._h, b, len(b)), "set_extradata")

    def configure(self):
        _chk(self._lib.opusgpu_decoder_configure(self._h), "configure")

    def send_packet(self, pkt):
        b = bytes(pkt)
        _chk(self._lib.opusgpu_decoder_send_packet(self._h, b, len(b)), "send_packet")

    def send_packet_fec(self, pkt):
        """pkt = the packet FOLLOWING a lost one: queues the lost packet's audio recovered from pkt's LBRR frames (in-band FEC);
        send pkt itself with send_packet() afterwards."""
        b = bytes(pkt)
        _chk(self._lib.opusgpu_decoder_send_packet_fec(self._h, b, len(b)), "send_packet_fec")

    def receive_frame(self):
        n = C.c_size_t()
        ch = C.c_int()
        _chk(self._lib.opusgpu_decoder_receive_frame(self._h, self._buf.ctypes.data, self._buf.size, C.byref(n), C.byref(ch)),
             "receive_frame")
        return self._buf[: n.value * ch.value].reshape(n.value, ch.value).copy()

    def flush(self):
        _chk(self._lib.opusgpu_decoder_flush(self._h), "flush")

    def close(self):
        if self._h:
            self._lib.opusgpu_decoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchDecoder:
    """`nstreams` independent decoders on one GPU; one packet per stream per send()."""

    def __init__(self, nstreams, channels=2, device=0):
        self._lib = A.load()
        self.nstreams, self.channels = nstreams, channels
        self._h = C.c_void_p()
        _chk(self._lib.opusgpu_batch_create(C.byref(self._h), nstreams, channels, device), "batch_create")

    def send(self, packets, stream_ids=None, fec=False):
        """packets: list of bytes; returns per-packet error codes (0 = ok).  fec: the packets are the successors of lost packets,
        whose audio is recovered from their LBRR frames (opusgpu_batch_send_fec)."""
        n = len(packets)
        keep = [bytes(p) for p in packets]
        arr = (C.c_char_p * n)(*keep)
        lens = (C.c_uint32 * n)(*[len(p) for p in keep])
        ids = (C.c_uint32 * n)(*stream_ids) if stream_ids is not None else None
        errs = (C.c_int32 * n)()
        _chk((self._lib.opusgpu_batch_send_fec if fec else self._lib.opusgpu_batch_send)(self._h, arr, lens, ids, n, errs), "batch_send")
        return list(errs)

    def send_packed(self, packets, lens=None):
        """packets: C-contiguous uint8 array [n, bytes] (or [n, T, bytes]: T consecutive packets per stream), one row per
        packet; lens: bytes used of each row (default: all).  The same call as send() / send_multi() without a Python
        object per packet: the pointer array is computed from the array's address.  Returns the error codes as an array."""
        import numpy as np
        assert packets.dtype == np.uint8 and packets.flags.c_contiguous and packets.ndim in (2, 3)
        n, T = packets.shape[0], (packets.shape[1] if packets.ndim == 3 else 1)
        width = packets.shape[-1]
        ptrs = packets.ctypes.data + np.arange(n * T, dtype=np.uint64) * np.uint64(width)
        ln = np.full(n * T, width, np.uint32) if lens is None else np.ascontiguousarray(lens, np.uint32).reshape(n * T)
        errs = np.zeros(n * T, np.int32)
        pp = C.cast(ptrs.ctypes.data, C.POINTER(C.c_char_p))
        lp, ep = ln.ctypes.data_as(C.POINTER(C.c_uint32)), errs.ctypes.data_as(C.POINTER(C.c_int32))
        if packets.ndim == 3:
            _chk(self._lib.opusgpu_batch_send_multi(self._h, pp, lp, None, n, T, ep), "batch_send_multi")
            self._cap = max(getattr(self, "_cap", 5760), 5760 * T)
        else:
            _chk(self._lib.opusgpu_batch_send(self._h, pp, lp, None, n, ep), "batch_send")
        return errs

    def send_multi(self, packets, stream_ids=None):
        """packets: list (one entry per stream) of equally long lists of consecutive packets; returns per-stream lists of
        error codes.  The stream's PCM row then holds the packets' output back to back."""
        n, T = len(packets), len(packets[0])
        assert all(len(p) == T for p in packets)
        keep = [bytes(p) for row in packets for p in row]
        arr = (C.c_char_p * (n * T))(*keep)
        lens = (C.c_uint32 * (n * T))(*[len(p) for p in keep])
        ids = (C.c_uint32 * n)(*stream_ids) if stream_ids is not None else None
        errs = (C.c_int32 * (n * T))()
        _chk(self._lib.opusgpu_batch_send_multi(self._h, arr, lens, ids, n, T, errs), "batch_send_multi")
        self._cap = max(getattr(self, "_cap", 5760), 5760 * T)
        return [list(errs[i * T:(i + 1) * T]) for i in range(n)]

    def decode(self, sync=True):
        _chk(self._lib.opusgpu_batch_decode(self._h), "batch_decode")
        if sync:
            _chk(self._lib.opusgpu_batch_sync(self._h), "batch_sync")

    def sync(self):
        _chk(self._lib.opusgpu_batch_sync(self._h), "batch_sync")

    def read(self, stream):
        buf = np.zeros(getattr(self, "_cap", 5760) * self.channels, np.float32)
        n = C.c_size_t()
        _chk(self._lib.opusgpu_batch_read(self._h, stream, buf.ctypes.data, buf.size, C.byref(n)), "batch_read")
        return buf[: n.value * self.channels].reshape(n.value, self.channels).copy()

    def device_pcm(self):
        """-> (device pointer, row stride in floats, nsamples per stream as numpy array)"""
        p = C.c_void_p()
        stride = C.c_size_t()
        ns = C.POINTER(C.c_uint32)()
        _chk(self._lib.opusgpu_batch_pcm(self._h, C.byref(p), C.byref(stride), C.byref(ns)), "batch_pcm")
        return p.value, stride.value, np.ctypeslib.as_array(ns, shape=(self.nstreams,)).copy()

    def sideinfo(self):
        """Side-info of the last send(): (celt hdr array, leaf array, silk frame array) as ctypes arrays (copies)."""
        ce, lv, sk = C.POINTER(A.CeltHdr)(), C.POINTER(A.CeltLeaf)(), C.POINTER(A.SilkFrame)()
        nc, nl, ns = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _chk(self._lib.opusgpu_batch_sideinfo(self._h, C.byref(ce), C.byref(nc), C.byref(lv), C.byref(nl), C.byref(sk), C.byref(ns)),
             "batch_sideinfo")

        def take(ptr, n, T):
            out = (T * n.value)()
            if n.value:
                C.memmove(out, ptr, n.value * C.sizeof(T))
            return out
        return take(ce, nc, A.CeltHdr), take(lv, nl, A.CeltLeaf), take(sk, ns, A.SilkFrame)

    def plan(self):
        """Launch plan of the last send(): dict of numpy copies -- refs (uint8 view of OgLeafRef[]), class_counts, run_off,
        run_len, silk_depth_off (see OgBatchPlan in include/opusgpu.h)."""
        import numpy as np
        pl = A.BatchPlan()
        _chk(self._lib.opusgpu_batch_plan(self._h, C.byref(pl)), "batch_plan")
        refs = np.frombuffer(C.string_at(pl.refs, pl.nrefs * 8), np.uint8).copy() if pl.nrefs else np.zeros(8, np.uint8)
        return dict(refs=refs, nrefs=pl.nrefs, class_counts=(C.c_uint32 * 4)(*pl.class_counts),
                    run_off=np.ctypeslib.as_array(pl.run_off, (pl.nruns + 1,)).copy() if pl.nruns else np.zeros(1, np.uint32),
                    nruns=pl.nruns, run_len=pl.run_len,
                    silk_depth_off=np.ctypeslib.as_array(pl.silk_depth_off, (pl.ndepths + 1,)).copy() if pl.ndepths else np.zeros(1, np.uint32))

    def flush(self, stream):
        _chk(self._lib.opusgpu_batch_flush(self._h, stream), "batch_flush")

    def close(self):
        if self._h:
            self._lib.opusgpu_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
