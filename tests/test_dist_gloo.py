"""The N>1 plumbing (one process per GPU, independent streams, barrier + MAX-over-ranks timing) on
world_size-2 gloo, CPU only."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from gmat_amd import dist as gd
    rank, local, world = gd.init("gloo")
    assert world == 2
    mine = gd.shard_streams(8, rank, world)
    gd.barrier(world)
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))            # rank 1 is the slow one
    wall = time.perf_counter() - t0
    value, t = gd.aggregate_throughput(1000.0, wall, world)
    gd.finalize(world)
    if rank == 0:
        print(json.dumps({"value": value, "t": t, "streams0": mine, "wall0": wall}))
""") % ROOT


def test_two_rank_aggregate_is_sum_over_slowest(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29617", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["streams0"] == [0, 2, 4, 6]
    assert d["t"] >= 0.1 and d["t"] >= d["wall0"]                 # MAX over ranks: the slow rank's time
    assert abs(d["value"] - 2 * 1000.0 / d["t"]) < 1e-6            # both ranks' units over that time


def test_single_process_path():
    sys.path.insert(0, ROOT)
    from gmat_amd import dist as gd
    assert gd.shard_streams(5, 0, 1) == [0, 1, 2, 3, 4]
    v, t = gd.aggregate_throughput(10.0, 2.0, 1)
    assert v == 5.0 and t == 2.0


def test_bench_self_launches_two_ranks_through_the_library():
    """`python bench.py --gpus 2` from a plain shell becomes two ranks (torch.distributed.run on 127.0.0.1); each rank
    drives the LIBRARY (here the CPU-emulated build, --dry, gloo): its own context, frames, streams and pinned-ring
    pipeline — the multi-GPU path of BASELINE configs[4] minus the GPUs."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    line = lines[-1]                                   # the contract: the LAST stdout line is the compact JSON object
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    det = {}
    for l in lines[:-1]:
        if l.startswith("BENCH_DETAIL "):
            det.update(json.loads(l[len("BENCH_DETAIL "):]))
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["scaling"] == "weak"
    assert d["roofline"]["kernel"] in ("scale_yuv2s_kernel", "scale_yuv2s_blk_kernel")     # (a dry run's launches are two frames: the block form)
    assert det["host_pipeline"]["ranks"] == 2 and det["host_pipeline"]["value"] > 0
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["value"] > 0     # rank 0 at any N
    assert det["cpu_configs0"]["cores"] == 1
    # each rank timed its own steps on its own (emulated) device; rank order, MAX is what `value` uses
    assert len(det["per_rank"]["wall_s"]) == 2 and det["per_rank"]["max"] >= det["per_rank"]["min"] > 0
    # one device per rank (gmat_set_device(LOCAL_RANK); the emulated build has two), each rank bound to ITS device's host cores
    assert det["per_rank"]["device"] == [0, 1]
    assert len(det["per_rank"]["numa_node"]) == 2 and all(c >= 0 for c in det["per_rank"]["cpus_bound"])
    assert det["host_pipeline"]["pageable_fill_one_thread"]["GBps"] > 0
    # value = all ranks' pixels over the slowest rank's time
    px = 128 * 32 * d["config"]["frames_per_step"] * d["steps"] * 2
    assert abs(d["value"] - px / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e9) < 2e-3


def test_control_plane_falls_back_when_the_first_backend_cannot_start(tmp_path):
    """gmat_amd.dist.init(backend, fallback=...): configs[4]'s control plane is a barrier and one MAX — if RCCL cannot be brought up
    (here: "nccl" in a GPU-less container) every rank falls back to gloo instead of taking the run down, and backend() says which."""
    import json
    import textwrap
    script = tmp_path / "worker.py"
    script.write_text(textwrap.dedent("""
        import json, sys
        sys.path.insert(0, %r)
        from gmat_amd import dist as gd
        rank, local, world = gd.init("nccl", fallback="gloo")
        v = gd.max_over_ranks(float(rank + 1), world)
        gd.finalize(world)
        if rank == 0:
            print(json.dumps({"backend": gd.backend(), "max": v, "world": world}))
    """) % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d == {"backend": "gloo", "max": 2.0, "world": 2}


def test_same_device_rehearsal_flag_runs_two_ranks():
    """`bench.py --gpus 2 --same-device` (the 8-GPU launch rehearsed on a 1-GPU box: every rank on device 0, control plane gloo);
    here on the emulated build: the flag parses, the ranks rendezvous on gloo and the last line says what it was"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry", "--same-device", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--no-pipeline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.splitlines()[-1])
    assert d["n_gpus"] == 2 and "control plane gloo" in d["config"]["parallelism"] and "DEVICE 0" in d["config"]["parallelism"]
