#!/usr/bin/env python3
"""Per-stage shader-clock profile of the fused CELT kernel on the C2 workload (diagnostic; uses OgCeltDebug.cycles)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opus_amd import BatchDecoder, _abi as A, synth  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import CeltLaunch  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NF = 6
lib = A.load()
dev = torch.device("cuda:0")
pk = synth.celt_fb_stereo_packets(range(S), NF, 160)
parser = BatchDecoder(S, 2, 0)
cl = CeltLaunch(S)
state = torch.zeros(S * C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
pcm = torch.zeros(S * 1920, dtype=torch.float32, device=dev)
names = ["init", "pvq", "bands", "anticollapse", "denorm", "imdct", "row store", "| tail: load+tdac", "comb", "state+deemph", "pcm"]
tot = np.zeros(11)
for f in range(NF):
    assert not any(parser.send([pk[s, f].tobytes() for s in range(S)]))
    celt, leaves, _ = parser.sideinfo()
    up = cl.upload(np.frombuffer(celt, dtype=np.uint8), np.frombuffer(leaves, dtype=np.uint8), S)
    cyc_all = torch.zeros(S * 64 + 177 * 4096 * 4, dtype=torch.int64, device=dev)
    cyc = cyc_all[: S * 64].view(S, 64)
    dbg = A.CeltDebug(None, None, None, None, cyc_all.data_ptr())
    cl.launch(up, state, pcm, 1920, dbg)
    torch.cuda.synchronize()
    c = cyc.cpu().numpy()
    if f >= 1:
        # og_celt_kernel stamps 0..5, 11, 13; og_celt_tail_kernel stamps 12, 7, 20, 21, 8 (another kernel: its own time base)
        da = np.diff(c[:, [0, 1, 2, 3, 4, 5, 11, 13]], axis=1).astype(np.float64)
        dt = np.diff(c[:, [12, 7, 20, 21, 8]], axis=1).astype(np.float64)
        d = np.concatenate([da, dt], axis=1)
        # stamps 2/3 are skipped on silent frames; ignore rows with non-monotonic stamps
        ok = (d >= 0).all(axis=1)
        tot += d[ok].mean(axis=0)
        bb = np.concatenate([c[:, 2:3], c[:, 16:20]], axis=1)          # band phase: set-up, pass A, pass B, pass C
        bd = np.diff(bb, axis=1).astype(np.float64)
        okb = ok & (bd >= 0).all(axis=1)
        band_tot = band_tot + bd[okb].mean(axis=0) if "band_tot" in dir() else bd[okb].mean(axis=0)
        tail = np.stack([c[:, 7], c[:, 20], c[:, 21], c[:, 8]], axis=1)      # comb | de-emphasis | stores
        td = np.diff(tail, axis=1).astype(np.float64)
        okt = ok & (td >= 0).all(axis=1) & (c[:, 20] > 0)
        tail_tot = tail_tot + td[okt].mean(axis=0) if "tail_tot" in dir() else td[okt].mean(axis=0)
        span = c[:, 13].max() - c[:, 0].min()
        span_t = c[:, 8].max() - c[:, 12].min()
        print("denorm apply region clk", (c[:, 10] - c[:, 9]).mean())
for n, v in zip(names, tot / (NF - 1)):
    print(f"{n:14s} {v:12.0f} clk  {100 * v / tot.sum() * (NF - 1):5.1f}%")
print("mean per-frame total", tot.sum() / (NF - 1), "clk; launch spans (last frame): spectrum kernel", span, "clk, tail kernel", span_t, "clk")
print("tail (comb filter, de-emphasis, PCM + state stores) mean clk:", " ".join(f"{v:.0f}" for v in tail_tot / (NF - 1)))
print("band phase (-, level 0, fill levels, merges) mean clk:", " ".join(f"{v:.0f}" for v in band_tot / (NF - 1)))
k1 = cyc_all[S * 64:].view(177, 4096, 4).cpu().numpy()
for nmax in (176, 64, 24, 8):
    blk = k1[nmax]
    used = blk[:, 0] != 0
    if used.any():
        d = np.diff(blk[used].astype(np.float64), axis=1)
        print(f"K1<{nmax}>: {used.sum()} workgroups; lane-0 mean clk cwrsi {d[:,0].mean():.0f} scale+rotate {d[:,1].mean():.0f} store {d[:,2].mean():.0f}; max total {(blk[used][:,3]-blk[used][:,0]).max()}")
