#!/usr/bin/env python3
"""Shader-clock stamps of og_celt_tail_kernel at the bench's launch shape (4096 streams x 8 frames per launch): per frame of a run,
the time from stage to stage (stamps 12 -> 7 -> 20 -> 21 -> 8: row + TDAC | comb filter | slide + de-emphasis | PCM) and the run's span."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from opus_amd import BatchDecoder, _abi as A, steps as stepmod, synth  # noqa: E402

S, T, NS = 4096, 8, 3
lib = A.load()
dev = torch.device("cuda:0")
pk = synth.celt_fb_stereo_packets(range(S), NS * T, 160)
parser = BatchDecoder(S, 2, 0)
hs, _ = stepmod.build_celt_steps(parser, pk, T, NS)
parser.close()
work = torch.zeros(lib.opusgpu_celt_work_bytes(S * T), dtype=torch.uint8, device=dev)
state = torch.zeros(S * C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
pcm = torch.zeros(S * T * 1920, dtype=torch.float32, device=dev)
for k, st in enumerate(hs):
    hdr, leaves, refs = (torch.from_numpy(x).to(dev) for x in (st.hdr, st.leaves, st.refs))
    cyc_all = torch.zeros(S * T * 64 + 177 * 4096 * 4, dtype=torch.int64, device=dev)
    dbg = A.CeltDebug(None, None, None, None, cyc_all.data_ptr())
    rc = lib.opusgpu_launch_celt(hdr.data_ptr(), leaves.data_ptr(), S * T, T, refs.data_ptr(), st.classes, work.data_ptr(), state.data_ptr(),
                                 pcm.data_ptr(), T * 1920, C.byref(dbg), None)
    assert rc == 0
    torch.cuda.synchronize()
    c = cyc_all[: S * T * 64].view(S, T, 64).cpu().numpy().astype(np.int64)
    if k == 0:
        continue
    st_ = c[:, :, [12, 7, 20, 21, 8]]
    d = np.diff(st_, axis=2)
    pf = (np.frombuffer(st.hdr, np.uint8).reshape(S, T, -1)[:, :, A.CeltHdr.pf_gain_q.offset:A.CeltHdr.pf_gain_q.offset + 3] != 0).any(axis=2)
    print(f"step {k}: per frame index t (mean clk over streams): row+tdac | comb | slide+deemph | pcm | frame total | gap to next frame")
    for t in range(T):
        gap = (c[:, t + 1, 12] - c[:, t, 8]).mean() if t + 1 < T else float("nan")
        print(f"  t={t}  " + "  ".join(f"{v:8.0f}" for v in d[:, t].mean(axis=0)) + f"  {(st_[:, t, 4] - st_[:, t, 0]).mean():8.0f}  {gap:8.0f}")
    print(f"  comb stage, frames with the post-filter on ({pf.mean():.2f} of them): {d[:, :, 1][pf].mean():.0f} clk; off: {d[:, :, 1][~pf].mean():.0f} clk")
    hb = np.frombuffer(st.hdr, np.uint8).reshape(S, T, -1)
    per = hb[:, :, A.CeltHdr.pf_period.offset:A.CeltHdr.pf_period.offset + 6].copy().view(np.uint16).reshape(S, T, 3).astype(np.int64)
    gq = hb[:, :, A.CeltHdr.pf_gain_q.offset:A.CeltHdr.pf_gain_q.offset + 3]
    on = gq != 0
    comb = d[:, :, 1]
    for name, m in (("only the old set", on[:, :, 0] & ~on[:, :, 1] & ~on[:, :, 2]), ("only the new set", ~on[:, :, 0] & ~on[:, :, 1] & on[:, :, 2]),
                    ("old + current", on[:, :, 0] & on[:, :, 1] & ~on[:, :, 2]), ("current + new", ~on[:, :, 0] & on[:, :, 1] & on[:, :, 2]),
                    ("all three", on.all(axis=2))):
        if m.any():
            print(f"    comb, {name:18s} {m.mean():.3f} of the frames: {comb[m].mean():6.0f} clk")
    m3 = on[:, :, 2]
    pn = np.maximum(per[:, :, 2], 15)
    for lo, hi in ((15, 32), (32, 64), (64, 130), (130, 258), (258, 1100)):
        mm = m3 & (pn >= lo) & (pn < hi)
        if mm.any():
            print(f"    comb, new set on with period in [{lo}, {hi}): {mm.mean():.3f} of the frames: {comb[mm].mean():6.0f} clk")
    run = c[:, T - 1, 8] - c[:, 0, 12]
    print(f"  run span (first stamp of frame 0 to last of frame {T - 1}): mean {run.mean():.0f} clk, max {run.max()}; kernel span {c[:, :, 8].max() - c[:, 0, 12].min()} clk")
