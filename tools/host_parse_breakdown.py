"""Where a send's host time goes (GPU box: the batch's side-info arrays are pinned memory).  OPUSGPU_TIMING=1 makes the
library print its own phases; this script adds the Python marshalling around the C call.
   python tools/host_parse_breakdown.py [streams] [sends]"""
import ctypes as C
import os
import sys
import time

os.environ["OPUSGPU_TIMING"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opus_amd import synth                                  # noqa: E402
from opus_amd.decoder import BatchDecoder                   # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
pk = synth.celt_fb_stereo_packets(range(S), K, 160)
dec = BatchDecoder(S, 2, 0)
for k in range(K):
    t0 = time.perf_counter()
    keep = [pk[s, k].tobytes() for s in range(S)]
    t1 = time.perf_counter()
    arr = (C.c_char_p * S)(*keep)
    lens = (C.c_uint32 * S)(*[len(p) for p in keep])
    errs = (C.c_int32 * S)()
    t2 = time.perf_counter()
    rc = dec._lib.opusgpu_batch_send(dec._h, arr, lens, None, S, errs)
    t3 = time.perf_counter()
    assert rc == 0 and not any(errs)
    print(f"send {k}: tobytes {1e3 * (t1 - t0):.2f} ms, ctypes arrays {1e3 * (t2 - t1):.2f} ms, C call {1e3 * (t3 - t2):.2f} ms "
          f"-> {S * 0.02 / (t3 - t2):.0f} audio-s/s in the C call", flush=True)
dec.close()
