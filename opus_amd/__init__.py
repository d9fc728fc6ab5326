"""opus_amd: batched Opus decode engine for MI355X (gfx950).

Python is only the harness here: `Decoder` / `BatchDecoder` mirror the reference's rust-av Decoder surface
(lu-zero/opus src/decoder.rs:42-188) over the C ABI of include/opusgpu.h.  All work happens in
opus_amd/libopusgpu.so (C++ host parser + HIP kernels); there is no Python or CPU fallback.
"""
from .decoder import BatchDecoder, Decoder, Demuxer, OpusError, parse_opus_head, parse_packet  # noqa: F401
