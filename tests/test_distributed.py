"""N > 1 path on CPU: two gloo processes exercise the sharding, the max-over-ranks timing
reduction and the photometric->geometric map exchange (no GPU compute)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from colmap_amd import distributed as D


def test_shard_problems_partition():
    for n in (0, 1, 7, 100):
        for w in (1, 2, 3, 8):
            shards = [D.shard_problems(n, r, w) for r in range(w)]
            flat = sorted(i for s in shards for i in s)
            assert flat == list(range(n))
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    with pytest.raises(ValueError):
        D.shard_problems(4, 2, 2)


def test_rank_windows_are_disjoint_in_references():
    w0 = D.rank_window(0, 24, 10)
    w1 = D.rank_window(1, 24, 10)
    assert w0 == (-10, 44) and w1 == (14, 44)
    refs0 = set(range(w0[0] + 10, w0[0] + 10 + 24))
    refs1 = set(range(w1[0] + 10, w1[0] + 10 + 24))
    assert not (refs0 & refs1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = D.shard_problems(5, rank, world)
        elapsed = 1.0 + rank  # rank 1 is the slow one
        t = D.max_over_ranks(elapsed)
        counts = D.gather_counts(len(mine))
        local = {i: torch.full((2, 3), float(i)) for i in mine}
        merged = D.exchange_maps(local)
        # the packed exchange of the geometric pass: ragged shapes, one rank may hold nothing
        ragged = {i: torch.arange(4 * (2 + i) * 3, dtype=torch.float32).view(4, 2 + i, 3) + 100 * i
                  for i in (mine if rank == 0 else [])}
        packed = D.exchange_maps_device(ragged, torch.device("cpu"))
        info = D.last_exchange_info()
        ok = sorted(packed) == [0, 2, 4] and all(
            torch.equal(packed[i], torch.arange(4 * (2 + i) * 3, dtype=torch.float32).view(4, 2 + i, 3) + 100 * i) for i in packed)
        dist.barrier()
        q.put((rank, mine, t, counts, sorted(merged), [float(merged[k][0, 0]) for k in sorted(merged)], ok, info))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_roundtrip():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, t0, c0, keys0, vals0, ok0, info0), (r1, mine1, t1, c1, keys1, vals1, ok1, info1) = res
    assert ok0 and ok1 and info0["transport"] == info1["transport"] == "gloo-staged" and info0["images"] == 3
    assert info0["bytes"] == 4 * sum(4 * (2 + i) * 3 for i in (0, 2, 4))
    assert mine0 == [0, 2, 4] and mine1 == [1, 3]
    assert t0 == t1 == 2.0                       # MAX over ranks
    assert c0 == c1 == [3, 2]                    # units per rank -> aggregate = 5
    assert keys0 == keys1 == [0, 1, 2, 3, 4]     # every rank holds every image's maps
    assert vals0 == vals1 == [0.0, 1.0, 2.0, 3.0, 4.0]


def test_ba_image_sharding_partitions_the_observations():
    """ba_shard_num_observations (host-only C ABI helper): shards by pose index partition the
    active observation set; constant-everything observations belong to no shard."""
    from colmap_amd import estimators as est, scene
    d = scene.synthesize_flat(9, 60, 4, seed=2)
    fp = est.FlatProblem.from_arrays(d)
    est.fix_gauge_two_cams(fp)
    total = est.shard_num_observations(fp, 0, 1)
    assert total == len(fp.obs_pose)
    for world in (2, 3, 8):
        parts = [est.shard_num_observations(fp, r, world) for r in range(world)]
        assert sum(parts) == total
        want = [int(np.sum(fp.obs_pose % world == r)) for r in range(world)]
        assert parts == want
        by_point = [est.shard_num_observations(fp, r, world, est.SHARD_BY_POINT) for r in range(world)]
        assert sum(by_point) == total
        assert by_point == [int(np.sum(fp.obs_point % world == r)) for r in range(world)]
    # an observation whose pose, intrinsics and point are all constant is not part of the program
    fp.pose_const[:] = 1
    fp.cam_const[:] = 1
    fp.point_const[:5] = 1
    n_dead = int(np.sum(np.isin(fp.obs_point, np.arange(5))))
    assert est.shard_num_observations(fp, 0, 1) == total - n_dead


import numpy as np  # noqa: E402


def test_bench_self_spawn_command_and_world_check(monkeypatch, tmp_path):
    """bench.py --gpus N without a launcher re-executes itself under torch.distributed.run (spawn_ranks): the command
    it builds -- one rank per GPU, rendezvous on 127.0.0.1, its own flags passed through, dmabuf IPC for RCCL -- and
    the WORLD_SIZE check of the contract, executed here without a GPU. The command is then really run with a stand-in
    script on the gloo-capable CPU: two ranks come up and see WORLD_SIZE = 2."""
    import importlib.util, os, subprocess, sys, socket
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    cmd, env = bench.spawn_command(4, ["--gpus", "4", "--steps", "3"], 29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5] == os.path.join(root, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    bench.check_world(4, 4)
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.check_world(4, 2)
    # the same command line, executed for real with a stand-in for the script
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\nopen(os.path.join(os.path.dirname(__file__), 'rank' + os.environ['RANK']), 'w')"
                      ".write(os.environ['WORLD_SIZE'] + ' ' + ' '.join(sys.argv[1:]))\n")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd, env = bench.spawn_command(2, ["--gpus", "2"], port, script=str(script))
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    for r in (0, 1):
        assert (tmp_path / f"rank{r}").read_text() == "2 --gpus 2"
