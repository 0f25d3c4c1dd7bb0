"""Multi-rank plumbing on CPU: world_size 2 over gloo (the GPU path uses the same code with nccl = RCCL)."""
import hashlib
import os
import socket

import pytest


def test_partitions():
    from mpcith_kyber_kosk_amd import sharding as s
    assert s.proof_partition(46, 1) == [(0, 46)]
    assert s.proof_partition(47, 4) == [(0, 12), (12, 12), (24, 12), (36, 11)]
    assert sum(c for _, c in s.proof_partition(4096, 8)) == 4096
    assert s.lanes_to_proofs(65536) == 46 and s.lanes_to_proofs(2 ** 20) == 722
    assert s.aligned_partition(2 ** 20, 8) == (91, 728)
    for total in (1, 7, 46, 722):
        for w in (1, 2, 3, 8):
            parts = s.proof_partition(total, w)
            assert parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            assert parts[-1][0] + parts[-1][1] == total


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from mpcith_kyber_kosk_amd import sharding as s
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per, padded = s.aligned_partition(5 * 1454, world)     # 5 proofs over 2 ranks -> 3 per rank, 6 padded
    units = per * 1454
    first = rank * units
    # each rank "commits" its own lanes: digest = sha3_256(global lane id)
    loc = torch.empty((units, 32), dtype=torch.uint8)
    for i in range(0, units, 97):
        loc[i] = torch.frombuffer(bytearray(hashlib.sha3_256(str(first + i).encode()).digest()), dtype=torch.uint8)
    table = s.allgather_digest_table(loc, world, dist)
    ok = table.shape == (padded * 1454, 32)
    for gl in range(0, padded * 1454, 97 * 13):
        r, i = divmod(gl, units)
        if i % 97 == 0:
            ok &= bytes(table[gl].tolist()) == hashlib.sha3_256(str(gl).encode()).digest()
    # BASELINE configs[3]: the per-GPU digest tables of one commitment round, [91 proofs][1454][32] per rank, gathered the way
    # bench.py --config 4 gathers them from HBM (all_gather_into_tensor into [world * 91][1454][32], rank-major)
    tab = torch.full((91, 1454, 32), rank + 1, dtype=torch.uint8)
    tab[rank * 7, 11, 5] = 200 + rank
    cat = torch.empty((world * 91, 1454, 32), dtype=torch.uint8)   # concatenation form: valid for gloo and RCCL alike
    dist.all_gather_into_tensor(cat, tab)
    out = cat.view(world, 91, 1454, 32)
    for r in range(world):
        ok &= int(out[r, 0, 0, 0]) == r + 1 and int(out[r, r * 7, 11, 5]) == 200 + r
    flat = s.allgather_digest_table(tab.reshape(-1, 32), world, dist)
    ok &= flat.shape == (world * 91 * 1454, 32) and torch.equal(flat.reshape(world, 91, 1454, 32), out)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok &= float(t.item()) == float(world)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_allgather_world2_gloo():
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_host_budget_per_rank():
    """bench.py's host-thread budget: plenty of cores -> the configuration's threads and spinning waits; an 8-rank node with
    few cores per rank (LOCAL_WORLD_SIZE=8) -> sleeping waits and smaller pools, never fewer than three threads per slot."""
    import bench
    assert bench.host_budget(256, 1, 6, 6) == (6, False)
    assert bench.host_budget(256, 8, 4, 6) == (6, False)      # 32 cores per rank >= 4 x 7
    assert bench.host_budget(128, 8, 6, 6) == (5, False)      # 16 cores per rank, 6 slots: 2 * 16 // 6 = 5; napping waits from 12 cores up
    assert bench.host_budget(64, 8, 6, 6) == (3, True)        # 8 cores per rank: the floor of three
    assert bench.host_budget(8, 8, 4, 8) == (2, True)         # one core per rank: the floor of two
    assert bench.host_budget(1, 1, 1, 6) == (2, True)
    # the default configuration since round 5: three cohorts of six callers with three workers each (18 per merged run)
    assert bench.CONFIGS[3]["slots"] == 18 and bench.CONFIGS[3]["combine"] == 6 and bench.CONFIGS[3]["threads"] == 3
    assert bench.host_budget(256, 1, 3, 18) == (18, False)    # one rank on a whole host: 3 per caller, spinning waits
    assert bench.host_budget(256, 8, 3, 18) == (18, False)    # a rank of eight on 32 cores: 18 // 6 = 3 per caller, the default waits
    assert bench.host_budget(16, 1, 3, 18) == (10, False)     # ONE rank on a 16-core quota (the driver's N = 1 run): as round 5's line
    assert bench.host_budget(64, 8, 3, 18) == (5, True)       # 8 cores per rank: 2 * 8 // 3 = 5 per cohort -> the floor of three per caller
    assert bench.threads_per_caller(5, True, 6, 8) == 3
    # the driver's container (round 5): 256 hardware threads visible, a CPU quota of 16 cores, eight ranks -> two cores per rank:
    # sleeping waits and two workers per caller (what the quota-blind budget got wrong: it assumed 32 cores per rank and spun)
    assert bench.host_budget(16, 8, 3, 18) == (2, True)
    assert bench.threads_per_caller(2, True, 6, 16 // 8) == 2
    assert bench.threads_per_caller(18, False, 6, 256) == 3
    assert bench.threads_per_caller(18, True, 6, 32) == 3
    assert bench.threads_per_caller(10, False, 6, 16) == 3


def test_usable_host_cores_honours_the_cgroup_quota(monkeypatch):
    """usable cores = min(scheduler affinity, cgroup CPU quota in whole cores); no quota -> the affinity"""
    import bench
    aff = len(os.sched_getaffinity(0))
    monkeypatch.setattr(bench, "cgroup_cpu", lambda: (16.0, 0, 0.0))
    assert bench.usable_host_cores() == min(aff, 16)
    monkeypatch.setattr(bench, "cgroup_cpu", lambda: (2.5, 0, 0.0))
    assert bench.usable_host_cores() == min(aff, 2)
    monkeypatch.setattr(bench, "cgroup_cpu", lambda: (0.5, 0, 0.0))
    assert bench.usable_host_cores() == 1
    monkeypatch.setattr(bench, "cgroup_cpu", lambda: (None, None, None))
    assert bench.usable_host_cores() == aff


def test_cgroup_cpu_reads_or_declines():
    """bench.py's cgroup_cpu: (quota in cores or None, throttled periods or None, throttled seconds or None), never an exception --
    the line's `cgroup_cpu` object says whether the container's CPU quota stalled the run"""
    import bench
    q, n, t = bench.cgroup_cpu()
    assert q is None or q > 0
    assert (n is None) == (t is None) and (n is None or (n >= 0 and t >= 0))


@pytest.mark.gpu
def test_config4_two_ranks_on_one_gpu_rehearsal():
    """BASELINE.json configs[3]'s multi-rank control flow with world = 2 on ONE GPU: static step dealing, the round hook's
    all-gather per commitment round, pending.wait() before the verifier reuses the tables -- host-staged over gloo
    (KOSK_BENCH_REHEARSE=1), since RCCL cannot put two ranks on one device.  bench.py itself checks that every rank's block of
    the gathered tables equals that rank's resident table."""
    import json
    import subprocess
    import sys
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, KOSK_BENCH_REHEARSE="1", LOCAL_WORLD_SIZE="8")  # 8: also drives the scarce-cores branch of host_budget
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--config", "4", "--steps", "6", "--warmup", "2",
           "--slots", "3", "--no-cpu-baseline", "--no-kernels"]
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["kyber_k"] == 4 and line["config"]["proofs_per_gpu"] == 91
    g = line["digest_allgather"]
    assert g["own_block_matches_resident_table"] and g["peer_blocks_match_their_resident_tables"]
    assert g["gathered_shape"] == [2 * 91, 1454, 32]
    # two gathers (Tcomm, view) per step on every slot, the same count on every slot (static dealing)
    assert all(n == 2 * s for n, s in zip(g["gathers_issued_per_slot"], g["steps_per_slot"])) and len(set(g["steps_per_slot"])) == 1
    assert line["value"] > 0
    # round 5: handles with a round hook merge like any other (every member's hook fires from the merged run with its own block of
    # the tables): the three slots of a rank are one cohort, and their calls really ran merged
    assert line["config"]["handles_per_cohort"] == 3 and line["combining"]["mean_callers_per_run"] > 1.5, line["combining"]
    assert line["rccl"]["world"] == 2 and [d["rank"] for d in line["rccl"]["devices"]] == [0, 1], line["rccl"]


@pytest.mark.gpu
def test_config3_two_ranks_on_one_gpu_rehearsal():
    """The line the driver's scaling run produces (bench.py --gpus N, default configuration, by-proof sharding, no data-path
    collective) with world = 2 on one GPU over gloo: barrier + MAX-reduction of the window, the result gather of one digest per
    rank, value = proofs of BOTH ranks over the slower rank's window, exactly one JSON line on rank 0's stdout."""
    import json
    import subprocess
    import sys
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a GPU")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, KOSK_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--slots", "3", "--combine", "3"]  # one cohort of three per rank
    r = subprocess.run(cmd, cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 12 and line["warmup"] == 3 and line["scaling"] == "weak"
    assert line["config"]["baseline_config"] == "configs[2]" and line["config"]["proofs_per_gpu"] == 46
    # whole-job value: both ranks' proofs over the MAX window
    assert abs(line["value"] - 2 * 12 * 46 / (line["ms_per_step"] * 12 / 1e3)) < 1e-6 * line["value"]
    assert "drop_in" not in line and "cpu_baseline" not in line   # N = 1 extras only
    assert line["roofline"]["frac"] > 0
