#!/usr/bin/env python
"""Sharded bundle adjustment over N GPUs of one node (BASELINE.json config[4] shape):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      scripts/ba_multi_gpu.py --frames 5000 --points 2000000 --track 10 --mixed 1
One process per GPU; observations sharded by image; RCCL all-reduce over xGMI inside ba_solve_sharded
(E^T x and J_c^T v per PCG iteration). Prints LM-iterations/s (max over ranks) on rank 0."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from colmap_amd import estimators as est, scene

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=5000); ap.add_argument("--points", type=int, default=2000000)
ap.add_argument("--track", type=int, default=10); ap.add_argument("--mixed", type=int, default=1)
ap.add_argument("--iters", type=int, default=10); ap.add_argument("--backend", default="rccl")
a = ap.parse_args()
rank, local, world = int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only (id broadcast)
torch.cuda.set_device(local)
d = scene.synthesize_flat(a.frames, a.points, a.track, seed=42, mixed_models=bool(a.mixed),
                          noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
fp = est.FlatProblem.from_arrays(d); est.fix_gauge_two_cams(fp)
comm = est.Communicator(a.backend, gpu_index=local)
est.solve_flat(fp.copy(), est.SolverOptions(max_num_iterations=2), gpu_index=local, comm=comm)  # warm-up
dist.barrier()
s = est.solve_flat(fp, est.SolverOptions(max_num_iterations=a.iters), gpu_index=local, comm=comm)
t = torch.tensor([s.lm_seconds], dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    print(f"{world} GPU(s): {s.num_iterations} LM iterations, {s.total_linear_iterations} PCG iterations, "
          f"cost {s.initial_cost:.6e} -> {s.final_cost:.6e}, {s.num_iterations / float(t):.2f} LM-iterations/s "
          f"({est.shard_num_observations(fp, 0, world)} of {len(fp.obs_pose)} observations on rank 0)")
comm.close(); dist.barrier(); dist.destroy_process_group()
