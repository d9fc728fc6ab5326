import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from gpu_util import HostParser
from opus_amd import Decoder
extra, pk = orc.load_vectors(ids=[11], max_packets=120)[11]
d = Decoder(0); d.set_extradata(extra); d.configure()
ref = orc.Decoder(2)
hp = HostParser(2)
bad = 0
for i, p in enumerate(pk):
    d.send_packet(p)
    out = d.receive_frame()
    r = ref.decode(p)
    n, nc, nl, _ = hp.parse(p)
    e = np.abs(out - r)
    if e.max() > 1e-4 * max(1, np.abs(r).max()):
        hd = hp.celt[0]
        idx = np.argwhere(e > 1e-4)
        print("packet", i, "nframes", n, "err", e.max(), "first bad sample", idx[0], "last", idx[-1], "count", len(idx),
              "lm", hd.lm, "flags", hd.flags, "pf", list(hd.pf_period), list(hd.pf_gain_q), list(hd.pf_tapset))
        bad += 1
        if bad > 6: break
print("done", bad)
