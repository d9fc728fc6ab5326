"""The N>1 path on CPU: world_size 2 over gloo.  Each rank takes its shard of the streams, runs the host half
(synthetic packets -> side-info through the C ABI parser, no GPU) and the ranks agree on coverage and timing."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

S = 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from gpu_util import HostParser
    from opus_amd import sharding, synth
    ids = sharding.shard_stream_ids(rank, world, S)
    pk = synth.celt_fb_stereo_packets(ids, 3, 160)
    leaves = 0
    for s in range(S):
        hp = HostParser(2)
        for f in range(3):
            n, nc, nl, _ = hp.parse(pk[s, f].tobytes())
            assert n == 1 and nc == 1
            leaves += nl
    gathered = [torch.zeros(S, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(ids.astype(np.int64)))
    allids = torch.cat(gathered).numpy()
    assert sorted(allids.tolist()) == list(range(world * S))            # disjoint, complete, contiguous blocks
    dist.barrier()
    slow = sharding.max_over_ranks(1.0 + rank)                           # every rank sees the slowest rank's time
    assert slow == float(world)
    v = sharding.aggregate_audio_seconds_per_second(world, S, 3, 0.02, slow)
    assert abs(v - world * S * 3 * 0.02 / world) < 1e-12
    tot = torch.tensor([leaves], dtype=torch.int64)
    dist.all_reduce(tot)
    if rank == 0:
        out.put((int(tot.item()), allids.tolist()))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    total_leaves, ids = q.get()
    assert total_leaves > 0 and len(ids) == 2 * S


def test_shards_generate_the_same_streams_as_one_rank():
    from opus_amd import sharding, synth
    whole = synth.celt_fb_stereo_packets(np.arange(2 * S, dtype=np.uint64), 2, 160)
    parts = [synth.celt_fb_stereo_packets(sharding.shard_stream_ids(r, 2, S), 2, 160) for r in range(2)]
    assert np.array_equal(np.concatenate(parts), whole)
