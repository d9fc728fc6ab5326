"""opusgpu_launch_celt_noalloc (the entry point that allocates nothing: its scratch is a static __device__ array inside the
library) against opusgpu_launch_celt on the same inputs: bit-identical PCM and decoder state, on two calls in a row, the
second with a smaller input (the array keeps the first call's contents).  Through the C ABI."""
import ctypes as C

import numpy as np
import pytest

from opus_amd import BatchDecoder, _abi as A, steps as stepmod, synth

pytestmark = pytest.mark.gpu


def _run(noalloc, host_steps, shapes):
    import torch
    lib = A.load()
    dev = torch.device("cuda:0")
    outs = []
    smax = max(S for S, _ in shapes)
    state = torch.zeros(smax * C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
    for st, (S, T) in zip(host_steps, shapes):
        hdr, leaves, refs = (torch.from_numpy(x).to(dev) for x in (st.hdr, st.leaves, st.refs))
        stride = T * 960 * 2
        pcm = torch.zeros(S * stride, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        if noalloc:
            rc = lib.opusgpu_launch_celt_noalloc(hdr.data_ptr(), leaves.data_ptr(), S * T, T, refs.data_ptr(), st.classes,
                                                 state.data_ptr(), pcm.data_ptr(), stride, None)
        else:
            work = torch.zeros(lib.opusgpu_celt_work_bytes(S * T), dtype=torch.uint8, device=dev)
            rc = lib.opusgpu_launch_celt(hdr.data_ptr(), leaves.data_ptr(), S * T, T, refs.data_ptr(), st.classes, work.data_ptr(),
                                         state.data_ptr(), pcm.data_ptr(), stride, None, None)
        assert rc == 0, lib.opusgpu_last_error()
        torch.cuda.synchronize()
        outs.append(pcm.cpu().numpy())
    return outs, state.cpu().numpy()


def test_noalloc_entry_matches_launch_celt():
    lib = A.load()
    assert lib.opusgpu_init(0) == 0
    maxf = lib.opusgpu_noalloc_max_frames()
    assert maxf == 256
    shapes = [(64, 4), (24, 2)]                          # 256 frames (the maximum), then 48: the second call is smaller
    host_steps = []
    first = 0
    for S, T in shapes:
        pk = synth.celt_fb_stereo_packets(range(S), first + T, 160)
        parser = BatchDecoder(S, 2, 0)
        if first:                                       # streams 0..S-1 continue where the first call left them
            stepmod.build_celt_steps(parser, pk, first, 1)
        st, _ = stepmod.build_celt_steps(parser, pk, T, 1, first_frame=first)
        parser.close()
        host_steps.append(st[0])
        first += T
    a, sa = _run(False, host_steps, shapes)
    b, sb = _run(True, host_steps, shapes)
    for k in range(len(shapes)):
        assert np.isfinite(a[k]).all() and np.abs(a[k]).max() > 0
        assert np.array_equal(a[k], b[k]), f"call {k}"
    assert np.array_equal(sa, sb)


def test_noalloc_entry_rejects_oversize_launch():
    lib = A.load()
    assert lib.opusgpu_init(0) == 0
    rc = lib.opusgpu_launch_celt_noalloc(None, None, lib.opusgpu_noalloc_max_frames() + 1, 1, None, (C.c_uint32 * 4)(), None, None, 0, None)
    assert rc == A.BAD_ARG
    assert b"exceed the static scratch" in lib.opusgpu_last_error()
