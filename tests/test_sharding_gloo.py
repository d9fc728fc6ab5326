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


def test_bench_entry_point_spawns_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run starts two ranks itself (before any GPU call) and rank 0
    prints one line with n_gpus 2; here on the host half only (--parse-only over gloo: no GPU in this container)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--parse-only", "--backend", "gloo",
                        "--streams", "6", "--steps", "2", "--frames-per-step", "2"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["shards"] == [[0, 5], [6, 11]]               # contiguous, disjoint blocks of streams
    assert out["config"]["leaves"] > 0 and out["value"] > 0


def test_bench_rejects_mismatched_world_size():
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--parse-only"], capture_output=True, text=True,
                       timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
