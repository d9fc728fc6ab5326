"""Side-info of a CELT "step" for the raw layer-3b launches: the next T consecutive frames of each of S streams, laid out
[stream][t] (the tail kernel walks one stream's T frames in time order), with the K1 work list.  bench.py and the
parity tests at the bench's launch shape build their inputs here, so they launch exactly the same records."""
import ctypes as C
import time

import numpy as np

from . import _abi as A


class CeltStep:
    """host arrays of one step: hdr [S*T, sizeof(OgCeltHdr)] u8, leaves u8, refs u8, classes (4 x u32), counters"""
    __slots__ = ("hdr", "leaves", "refs", "classes", "nframes", "nleaves", "pf_on")


def build_celt_steps(parser, packets, T, nsteps, first_frame=0):
    """parser: a BatchDecoder over the S streams (host half only is used); packets: uint8 [S, >= first_frame + nsteps*T, bytes].
    -> (list of CeltStep, seconds spent in the host parser)"""
    lib = A.load()
    S = packets.shape[0]
    HS = C.sizeof(A.CeltHdr)
    o_lb, o_po, o_gq = A.CeltHdr.leaf_begin.offset, A.CeltHdr.pcm_off.offset, A.CeltHdr.pf_gain_q.offset
    steps, t_parse = [], 0.0
    for k in range(nsteps):
        hb_all = np.zeros((S, T, HS), np.uint8)
        leaf_parts, leaf_off, pf_on = [], 0, 0
        for t in range(T):
            f = first_frame + k * T + t
            pk = np.ascontiguousarray(packets[:, f])
            t0 = time.perf_counter()
            errs = parser.send_packed(pk)                    # the timed region is the C call (opusgpu_batch_send) and its pointer array
            t_parse += time.perf_counter() - t0
            assert not errs.any(), errs
            celt, leaves, _ = parser.sideinfo()
            assert len(celt) == S
            hb = np.frombuffer(celt, dtype=np.uint8).reshape(S, HS).copy()
            hb[:, o_lb:o_lb + 4].view(np.uint32)[:] += leaf_off          # leaf indices are relative to the step's array
            hb[:, o_po:o_po + 4].view(np.uint32)[:] = t * 960            # frame t lands at sample t*960 of the PCM row
            pf_on += int((hb[:, o_gq:o_gq + 3] != 0).any(axis=1).sum())
            hb_all[:, t] = hb
            leaf_parts.append(np.frombuffer(leaves, dtype=np.uint8).copy())
            leaf_off += len(leaves)
        st = CeltStep()
        st.leaves = np.concatenate(leaf_parts) if leaf_off else np.zeros(16, np.uint8)
        st.hdr = np.ascontiguousarray(hb_all.reshape(S * T, HS))
        refs = (A.LeafRef * max(leaf_off, 1))()
        st.classes = (C.c_uint32 * 4)()
        nref = lib.opusgpu_sort_pvq_leaves(st.hdr.ctypes.data, S * T, st.leaves.ctypes.data, refs, max(leaf_off, 1), st.classes)
        st.refs = np.frombuffer(refs, dtype=np.uint8)[: max(nref, 1) * 8].copy()
        st.nframes, st.nleaves, st.pf_on = S * T, leaf_off, pf_on
        steps.append(st)
    return steps, t_parse
