"""Multi-GPU plumbing of the benchmark: one process per GPU, independent streams, no data-path collective.

The path shards by stream: rank r owns GPU r and converts its own frames.  The only
communication is control: a start/stop barrier and the MAX of the per-rank wall time, so that
value = (units processed by ALL ranks) / (slowest rank's time).  Backend "nccl" (= RCCL) on GPUs, "gloo" in
the CPU tests.
"""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend):
    """Initialise torch.distributed from the torchrun environment when WORLD_SIZE > 1."""
    rank, local, world = env_rank()
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if not dist.is_initialized():
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value, world, device="cpu"):
    if world <= 1:
        return float(value)
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, world, device="cpu"):
    """every rank's value, in rank order (per-rank min / max of the wall time for the detail file)"""
    if world <= 1:
        return [float(value)]
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    return [float(t.item()) for t in allv]


def shard_streams(n_streams, rank, world):
    """Round-robin assignment of independent stream ids to ranks (stream s -> rank s % world)."""
    return [s for s in range(n_streams) if s % world == rank]


def aggregate_throughput(units_per_rank, wall_seconds, world, device="cpu"):
    """Whole-job rate: every rank processed `units_per_rank`; time is the slowest rank's."""
    t = max_over_ranks(wall_seconds, world, device)
    return world * units_per_rank / t, t


def finalize(world):
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
