#!/usr/bin/env python3
"""Decode the Opus track of a Matroska (.mka/.webm) or Ogg (.opus) file on the GPU to raw 16-bit PCM (48 kHz, the
header's channel count) -- the format RFC 6716's `.dec` reference files use, so that

    python tools/decode_file.py testvector01.mka out.pcm && python tools/opus_compare.py -s testvector01.dec out.pcm

is the conformance run for a vector whose `.dec` file a user supplies (none ship with lu-zero/opus).
Call order is the reference's: demux -> set_extradata -> configure -> (send_packet -> receive_frame)* (decoder.rs:202-232)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opus_amd import Decoder, Demuxer  # noqa: E402


def decode_file(path, device=0):
    dm = Demuxer(open(path, "rb").read())
    d = Decoder(device)
    d.set_extradata(dm.extradata)
    d.configure()
    out = []
    for pkt in dm:
        d.send_packet(pkt)
        out.append(d.receive_frame())
    return np.concatenate(out) if out else np.zeros((0, 2), np.float32)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        print(__doc__)
        sys.exit(2)
    pcm = decode_file(sys.argv[1])
    np.clip(np.rint(pcm * 32768.0), -32768, 32767).astype("<i2").tofile(sys.argv[2])
    print(f"{pcm.shape[0]} samples x {pcm.shape[1]} channels -> {sys.argv[2]}")
