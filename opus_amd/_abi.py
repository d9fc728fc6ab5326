"""ctypes mirror of include/opusgpu.h and include/opusgpu_sideinfo.h (struct layouts + prototypes)."""
import ctypes as C
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("OPUSGPU_LIB") or os.path.join(ROOT, "opus_amd", "libopusgpu.so")     # (OPUSGPU_LIB: tools/build_variant.py flavours)

OK, INVALID_DATA, CONFIG_INVALID, CONFIG_INCOMPLETE, UNSUPPORTED, MORE_DATA_NEEDED, DEVICE, BAD_ARG = 0, -1, -2, -3, -4, -5, -6, -7

F_TRANSIENT, F_SILENCE, F_STEREO, F_ANTICOLLAPSE, F_RESET, F_OUT_STEREO, F_ACCUM = 1, 2, 4, 8, 16, 32, 64
B_JOINT, B_INV, B_N2SWAP, B_N2NEG, B_DUAL, B_MIDFILL0, B_MERGE_NORM, B_LAST = 1, 2, 4, 8, 16, 32, 64, 128
L_LOGB_MASK, L_FILL_DEAD, L_CH_Y, L_N1, L_HAS_LOWBAND = 7, 8, 32, 64, 128


class CeltBand(C.Structure):
    _fields_ = [("imid", C.c_uint16), ("iside", C.c_uint16), ("flags", C.c_uint16), ("leaf0", C.c_uint16),
                ("nx", C.c_uint8), ("ny", C.c_uint8), ("fx", C.c_uint8), ("fy", C.c_uint8),
                ("lowband", C.c_uint8), ("tf_change", C.c_int8), ("fold_lo", C.c_uint8), ("fold_hi", C.c_uint8)]


class CeltLeaf(C.Structure):
    _fields_ = [("index", C.c_uint32), ("gain", C.c_float), ("off", C.c_uint8), ("n", C.c_uint8), ("k", C.c_uint8),
                ("meta", C.c_uint8), ("cmshift", C.c_uint8), ("band", C.c_uint8), ("pos", C.c_uint16)]


class LeafRef(C.Structure):
    _fields_ = [("leaf", C.c_uint32), ("frame", C.c_uint32)]


class CeltHdr(C.Structure):
    _fields_ = [("stream", C.c_uint32), ("pcm_off", C.c_uint32), ("leaf_begin", C.c_uint32), ("seed", C.c_uint32),
                ("nleaves", C.c_uint16), ("lm", C.c_uint8), ("flags", C.c_uint8), ("start", C.c_uint8),
                ("end", C.c_uint8), ("spread", C.c_uint8), ("intensity", C.c_uint8),
                ("pf_period", C.c_uint16 * 3), ("pf_gain_q", C.c_uint8 * 3), ("pf_tapset", C.c_uint8 * 3),
                ("pulses_depth", C.c_uint16 * 21), ("energy", (C.c_float * 21) * 2), ("prev_min", (C.c_float * 21) * 2),
                ("band", CeltBand * 21)]


class SilkChan(C.Structure):
    _fields_ = [("flags", C.c_uint8), ("order", C.c_uint8), ("subframes", C.c_uint8), ("sf_size", C.c_uint8),
                ("ltpscale", C.c_float), ("gain", C.c_float * 4), ("pitch_lag", C.c_int16 * 4),
                ("ltp_taps", (C.c_float * 5) * 4), ("lpc", C.c_float * 16), ("lpc_interp", C.c_float * 16),
                ("excitation", C.c_int32 * 320)]


class SilkFrame(C.Structure):
    _fields_ = [("flags", C.c_uint8), ("bandwidth", C.c_uint8), ("f_size", C.c_uint16), ("w0", C.c_float),
                ("w1", C.c_float), ("prev_w0", C.c_float), ("prev_w1", C.c_float), ("stream", C.c_uint32), ("pcm_off", C.c_uint32), ("pad_", C.c_uint32), ("ch", SilkChan * 2)]


class FrameInfo(C.Structure):
    _fields_ = [("mode", C.c_uint8), ("bandwidth", C.c_uint8), ("stereo", C.c_uint8), ("nsilk", C.c_uint8),
                ("frame_size", C.c_uint16), ("has_celt", C.c_uint16), ("celt_index", C.c_uint32),
                ("silk_index", C.c_uint32), ("final_range", C.c_uint32), ("tell_bits", C.c_uint32), ("budget_bits", C.c_uint32)]


class Packet(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("mode", "bandwidth", "frame_size", "stereo", "code", "nframes", "padding")] + [
        ("off", C.c_int32 * 48), ("len", C.c_int32 * 48)]


class CeltState(C.Structure):
    _fields_ = [("hist", (C.c_float * 1024) * 2), ("overlap", (C.c_float * 60) * 2), ("deemph", C.c_float * 2),
                ("pos", C.c_uint32), ("pad_", C.c_uint32)]


class BatchPlan(C.Structure):
    _fields_ = [("refs", C.c_void_p), ("nrefs", C.c_uint32), ("class_counts", C.c_uint32 * 4), ("run_off", C.POINTER(C.c_uint32)),
                ("nruns", C.c_uint32), ("run_len", C.c_uint32), ("silk_depth_off", C.POINTER(C.c_uint32)), ("ndepths", C.c_uint32)]


class OpusHead(C.Structure):
    _fields_ = [("version", C.c_uint8), ("channels", C.c_uint8), ("pre_skip", C.c_uint16), ("input_sample_rate", C.c_uint32),
                ("output_gain_q8", C.c_int16), ("mapping_family", C.c_uint8), ("stream_count", C.c_uint8),
                ("coupled_count", C.c_uint8), ("mapping", C.c_uint8 * 255)]


class CeltDebug(C.Structure):
    _fields_ = [("pulses", C.c_void_p), ("norm", C.c_void_p), ("freq", C.c_void_p), ("collapse", C.c_void_p), ("cycles", C.c_void_p)]


class Descr(C.Structure):
    _fields_ = [("codec", C.c_char_p), ("name", C.c_char_p), ("desc", C.c_char_p), ("mime", C.c_char_p)]


# every symbol include/opusgpu.h declares: name -> (restype, argtypes or None)
_P = C.c_void_p
PROTOTYPES = {
    "opusgpu_version": (C.c_char_p, []),
    "opusgpu_last_error": (C.c_char_p, []),
    "opusgpu_strerror": (C.c_char_p, [C.c_int]),
    "opusgpu_describe": (C.POINTER(Descr), []),
    "opusgpu_decoder_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "opusgpu_decoder_destroy": (None, [_P]),
    "opusgpu_decoder_set_extradata": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "opusgpu_decoder_configure": (C.c_int, [_P]),
    "opusgpu_decoder_send_packet": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "opusgpu_decoder_send_packet_fec": (C.c_int, [_P, C.c_char_p, C.c_size_t]),
    "opusgpu_decoder_receive_frame": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "opusgpu_decoder_flush": (C.c_int, [_P]),
    "opusgpu_opus_head_parse": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(OpusHead)]),
    "opusgpu_demux_open": (C.c_int, [C.POINTER(_P), C.c_char_p, C.c_size_t]),
    "opusgpu_demux_close": (None, [_P]),
    "opusgpu_demux_info": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "opusgpu_demux_packet": (C.c_int, [_P, C.c_uint32, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.c_int64)]),
    "opusgpu_batch_create": (C.c_int, [C.POINTER(_P), C.c_uint32, C.c_int, C.c_int]),
    "opusgpu_batch_destroy": (None, [_P]),
    "opusgpu_batch_send": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int32)]),
    "opusgpu_batch_send_fec": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_int32)]),
    "opusgpu_batch_send_multi": (C.c_int, [_P, C.POINTER(C.c_char_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32,
                                           C.POINTER(C.c_int32)]),
    "opusgpu_batch_decode": (C.c_int, [_P]),
    "opusgpu_batch_sync": (C.c_int, [_P]),
    "opusgpu_batch_pcm": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint32))]),
    "opusgpu_batch_read": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_size_t)]),
    "opusgpu_batch_flush": (C.c_int, [_P, C.c_uint32]),
    "opusgpu_batch_sideinfo": (C.c_int, [_P, C.POINTER(C.POINTER(CeltHdr)), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(CeltLeaf)),
                                        C.POINTER(C.c_uint32), C.POINTER(C.POINTER(SilkFrame)), C.POINTER(C.c_uint32)]),
    "opusgpu_batch_plan": (C.c_int, [_P, C.POINTER(BatchPlan)]),
    "opusgpu_parser_create": (C.c_int, [C.POINTER(_P), C.c_int]),
    "opusgpu_parser_destroy": (None, [_P]),
    "opusgpu_parser_reset": (None, [_P]),
    "opusgpu_parser_parse": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(FrameInfo), C.POINTER(CeltHdr), C.c_uint32,
                                      C.POINTER(C.c_uint32), C.POINTER(CeltLeaf), C.c_uint32, C.POINTER(C.c_uint32),
                                      C.POINTER(SilkFrame), C.c_uint32, C.POINTER(C.c_uint32)]),
    "opusgpu_parser_parse_fec": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(FrameInfo), C.POINTER(CeltHdr), C.c_uint32,
                                          C.POINTER(C.c_uint32), C.POINTER(CeltLeaf), C.c_uint32, C.POINTER(C.c_uint32),
                                          C.POINTER(SilkFrame), C.c_uint32, C.POINTER(C.c_uint32)]),
    "opusgpu_packet_parse": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(Packet)]),
    "opusgpu_packet_undelimit": (C.c_int, [C.c_char_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "opusgpu_init": (C.c_int, [C.c_int]),
    "opusgpu_launch_celt": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint32), _P, _P, _P, C.c_size_t, C.POINTER(CeltDebug), _P]),
    "opusgpu_launch_celt_pvq": (C.c_int, [_P, _P, C.c_uint32, _P, C.POINTER(C.c_uint32), _P, C.POINTER(CeltDebug), _P]),
    "opusgpu_launch_celt_frames": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P, _P, C.c_size_t, C.POINTER(CeltDebug), _P]),
    "opusgpu_launch_celt_spectrum": (C.c_int, [_P, C.c_uint32, _P, C.POINTER(CeltDebug), _P]),
    "opusgpu_launch_celt_tail": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, _P, _P, C.c_size_t, C.POINTER(CeltDebug), _P]),
    "opusgpu_launch_celt_tail_runs": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P, _P, _P, C.c_size_t, C.POINTER(CeltDebug), _P]),
    "opusgpu_launch_celt_runs": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.POINTER(C.c_uint32), _P, _P, _P, C.c_size_t,
                                          C.POINTER(CeltDebug), _P]),
    "opusgpu_noalloc_max_frames": (C.c_uint32, []),
    "opusgpu_launch_celt_noalloc": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, _P, C.POINTER(C.c_uint32), _P, _P, C.c_size_t, _P]),
    "opusgpu_sort_pvq_leaves": (C.c_uint32, [_P, C.c_uint32, _P, _P, C.c_uint32, C.POINTER(C.c_uint32)]),
    "opusgpu_celt_work_bytes": (C.c_size_t, [C.c_uint32]),
    "opusgpu_silk_state_bytes": (C.c_size_t, [C.c_uint32]),
    "opusgpu_launch_silk": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P, C.c_size_t, _P, _P]),
    "opusgpu_sizeof_celt_state": (C.c_size_t, []),
    "opusgpu_sizeof_celt_hdr": (C.c_size_t, []),
    "opusgpu_sizeof_celt_leaf": (C.c_size_t, []),
    "opusgpu_sizeof_silk_frame": (C.c_size_t, []),
}

_lib = None


def load(path=None):
    """Loads libopusgpu.so (the hipcc-built product library).  Fails loudly if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:
        # torch bundles its own libamdhip64.so.7; whichever HIP runtime is loaded first serves the whole process.
        # Loading torch's first keeps one runtime when a caller also holds torch device tensors (tests, bench.py).
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib
