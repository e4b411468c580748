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


_backend = [None]


def backend():
    """the torch.distributed backend init() brought up ("nccl" = RCCL, "gloo"), None for a single rank"""
    return _backend[0]


def init(backend, fallback=None):
    """Initialise torch.distributed from the torchrun environment when WORLD_SIZE > 1.  Returns (rank, local_rank, world); the
    backend in use is backend().  `fallback`: tried when `backend` cannot be brought up — the control plane here is a barrier and one MAX, it must never
    be what takes an 8-GPU run down (RCCL needs dmabuf IPC, HSA_ENABLE_IPC_MODE_LEGACY=0, and refuses two ranks on one device).
    Every rank takes the same decision: a rank whose first attempt failed and a rank whose attempt succeeded cannot be told apart
    from inside, so the first backend is PROBED with a one-element all-reduce under a short timeout before it is trusted."""
    rank, local, world = env_rank()
    if world <= 1:
        return rank, local, world
    import datetime
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if dist.is_initialized():
        _backend[0] = dist.get_backend()
        return rank, local, world
    tries = [backend] + ([fallback] if fallback and fallback != backend else [])
    last = None
    for i, b in enumerate(tries):
        try:
            # (a second attempt rendezvouses at the same address: torch prefixes every process group's keys with its own count, and
            # under torchrun the store belongs to the agent, which listens on MASTER_PORT only)
            dist.init_process_group(b, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120 if b == "nccl" else 600))
            t = torch.ones(1, device="cuda" if b == "nccl" else "cpu")
            dist.all_reduce(t)
            if int(t.item()) != world:
                raise RuntimeError("control-plane probe returned %r for %d ranks" % (t.item(), world))
            _backend[0] = b
            return rank, local, world
        except Exception as e:                                   # noqa: BLE001 - any failure of the first backend falls through
            last = e
            try:
                if dist.is_initialized():
                    dist.destroy_process_group()
            except Exception:                                    # noqa: BLE001
                pass
            if rank == 0:
                print("gmat_amd.dist: backend %s unavailable (%s)%s" % (b, str(e).splitlines()[0][:200], ", trying " + tries[i + 1] if i + 1 < len(tries) else ""),
                      flush=True)
    raise RuntimeError("no torch.distributed backend could be initialised: %r" % (last,))


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
