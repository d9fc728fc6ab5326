"""Stream sharding across GPUs: rank r of N owns streams [r*S, (r+1)*S).  No data-path collective:
the only cross-rank steps are a barrier and a max over the ranks' elapsed times, so any torch.distributed backend
(RCCL on the GPUs, gloo in CPU tests) serves."""
import numpy as np


def shard_stream_ids(rank, world, streams_per_gpu):
    return np.arange(streams_per_gpu, dtype=np.uint64) + np.uint64(rank * streams_per_gpu)


def max_over_ranks(value, device=None):
    """max of a python float over all ranks (identity when torch.distributed is not initialised)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_audio_seconds_per_second(world, streams_per_gpu, steps, frame_s, elapsed_max):
    """whole-job throughput: units all ranks processed / slowest rank's time"""
    return world * streams_per_gpu * steps * frame_s / elapsed_max
