import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import orc
from gpu_util import HostParser, CeltLaunch
from opus_amd import _abi as A, build
emu = C.CDLL(os.path.join(ROOT, "tests", "emu", "libemu.so"))
emu.emu_launch_celt.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
extra, pk = orc.load_vectors(ids=[11], max_packets=100)[11]
hp = HostParser(2)
dev = torch.device("cuda:0")
cl = CeltLaunch(1)
cl.work[:] = 0
state = torch.zeros(C.sizeof(A.CeltState), dtype=torch.uint8, device=dev)
est = A.CeltState()
pcm = torch.zeros(5760 * 2, dtype=torch.float32, device=dev)
epcm = np.zeros(5760 * 2, np.float32)
HS = C.sizeof(A.CeltHdr)
bad = 0
for i, p in enumerate(pk):
    n, nc, nl, _ = hp.parse(p)
    off = 0
    for f in range(n):
        hp.celt[f].stream = 0; hp.celt[f].pcm_off = off; off += hp.fi[f].frame_size
    en = np.zeros((n, 2, 960), np.float32); ef = np.zeros((n, 2, 960), np.float32)
    edbg = A.CeltDebug(None, en.ctypes.data, ef.ctypes.data, None)
    emu.emu_launch_celt(hp.celt, hp.leaves, n, 1, C.byref(est), epcm.ctypes.data, 5760 * 2, C.byref(edbg))
    lb = np.frombuffer(hp.leaves, dtype=np.uint8)[: nl * 16]
    for f in range(n):
        hb = np.frombuffer(hp.celt, dtype=np.uint8)[f * HS:(f + 1) * HS]
        up = cl.upload(hb, lb, 1)
        norm = torch.zeros((2, 960), dtype=torch.float32, device=dev)
        freq = torch.zeros((2, 960), dtype=torch.float32, device=dev)
        dbg = A.CeltDebug(None, norm.data_ptr(), freq.data_ptr(), None)
        lib = cl.lib
        assert lib.opusgpu_launch_celt_pvq(up["hdr"].data_ptr(), up["leaves"].data_ptr(), 1, up["refs"].data_ptr(), up["classes"], cl.work.data_ptr(), None, None) == 0
        torch.cuda.synchronize()
        row = cl.work[:7680].cpu().numpy().view(np.float32).reshape(2, 960).copy()
        assert lib.opusgpu_launch_celt_frames(up["hdr"].data_ptr(), 1, 1, cl.work.data_ptr(), state.data_ptr(), pcm.data_ptr(), 5760 * 2, C.byref(dbg), None) == 0
        torch.cuda.synchronize()
        dn = np.abs(norm.cpu().numpy() - en[f]).max(); df = np.abs(freq.cpu().numpy() - ef[f]).max()
        o = hp.celt[f].pcm_off
        dp = np.abs(pcm.cpu().numpy()[o * 2:(o + 960) * 2] - epcm[o * 2:(o + 960) * 2]).max()
        only = os.environ.get('DBG_ONLY')
        if (only and f'{i}:{f}' in only.split(',')) or (not only and (dn > 1e-5 or dp > 1e-3)):
            nn = norm.cpu().numpy(); w = np.argwhere(np.abs(nn - en[f]) > 1e-5)
            print("packet", i, "frame", f, "of", n, "norm diff", dn, "freq diff", df, "pcm diff", dp, "first", w[:3].tolist(), "n", len(w))
            bad += 1
            hd = hp.celt[f]
            for b in range(3):
                bd = hd.band[b]
                print("  band", b, "nx ny fx fy", bd.nx, bd.ny, bd.fx, bd.fy, "flags", hex(bd.flags), "leaf0", bd.leaf0)
                for l in range(bd.nx + bd.ny):
                    lf = hp.leaves[hd.leaf_begin + bd.leaf0 + l]
                    print("     leaf n k meta pos", lf.n, lf.k, hex(lf.meta), lf.pos, "idx", lf.index)
            print("  classes", list(up["classes"]), "nrefs", up["refs"].numel() // 8)
            print("  row ch0[0:16]", row[0, :16]); print("  row ch1[0:16]", row[1, :16]); print("  row ch1[16:24]", row[1,16:24]); print("  gpu norm ch1 16:24", norm.cpu().numpy()[1,16:24])
            print("  gpu norm ch1", nn[1, :16]);
            print("  gpu norm ch0", nn[0, :16]); print("  emu norm ch0", en[f][0, :16])
    if bad > 8 and not os.environ.get('DBG_ONLY'): break
print("done", bad)
