"""One short torch-free GPU call covering what changed after the round's GPU budget was (almost) spent.

    python scripts/final_gpu_check.py make    # here: writes scripts/tmp/final_cases.pkl (inputs + the oracle's PatchMatch answer)
    python scripts/final_gpu_check.py run     # on the GPU box: stage by stage into gpurun_out/final_gpu_check.log

Stages: (A) PatchMatch smoke, bit for bit (the library rebuilt with the inline-assembly helpers in their own header);
(B) fusion pool sizes 1 / 3 / default on a scene whose passes are cut; (C) BA: exact tier with the formation's overflow flag,
PCG tier, and LM-iterations/s of both tiers at BASELINE config[3]'s own size. TEST / MEASUREMENT INFRASTRUCTURE."""
import ctypes as C
import os
import pickle
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
PKL = os.path.join(ROOT, "scripts", "tmp", "final_cases.pkl")


def make():
    import pm_oracle
    import test_fusion as TF
    from colmap_amd import fusion, synthetic as syn
    from pm_common import scene, oracle_inputs
    pm_oracle.build()
    views = scene(4, 64, 48)
    ref, src = 1, [0, 2, 3]
    dmin, dmax = syn.depth_range(views, ref)
    kw = dict(depth_min=dmin, depth_max=dmax, geom_consistency=0, filter=1, num_iterations=2)
    o = pm_oracle.default_options(order=1, **kw)
    want = pm_oracle.run(o, oracle_inputs(views), ref, src, want_cost=True)
    pm_case = dict(views=[(v.K, v.R, v.T, v.gray) for v in views], ref=ref, src=src, kw=kw,
                   want={k: want[k] for k in ("depth", "normal", "cost")})
    rng = np.random.default_rng(1)
    images = TF._images(scene(4, 24, 160))
    for im in images:
        im.depth_map = (im.depth_map * (1 + 0.01 * rng.standard_normal(im.depth_map.shape))).astype(np.float32)
    os.makedirs(os.path.dirname(PKL), exist_ok=True)
    with open(PKL, "wb") as fh:
        pickle.dump(dict(pm=pm_case, fusion_images=images), fh)
    print(PKL, os.path.getsize(PKL) / 1e6, "MB")


def run():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    log = open(os.path.join(ROOT, "gpurun_out", "final_gpu_check.log"), "w")

    def say(*a):
        s = " ".join(str(x) for x in a)
        print(s, flush=True)
        log.write(s + "\n")
        log.flush()
        os.fsync(log.fileno())

    def stage(name, fn):
        t = time.time()
        try:
            fn()
            say(f"[{name}] done in {time.time() - t:.1f}s")
        except Exception as e:
            say(f"[{name}] FAILED {type(e).__name__}: {e}")

    with open(PKL, "rb") as fh:
        cases = pickle.load(fh)

    def pm_stage():
        from colmap_amd import mvs
        c = cases["pm"]
        kw = dict(c["kw"])
        opt = mvs.PatchMatchOptions(gpu_index="0", depth_min=kw["depth_min"], depth_max=kw["depth_max"], sigma_spatial=5.0,
                                    geom_consistency=False, filter=True, num_iterations=kw["num_iterations"])
        images = [mvs.Image(K, R, T, g) for (K, R, T, g) in c["views"]]
        pm = mvs.PatchMatch(opt, mvs.PatchMatch.Problem(c["ref"], c["src"], images))
        pm.Run()
        eq = {k: bool(np.array_equal(v, {"depth": pm.GetDepthMap, "normal": pm.GetNormalMap, "cost": pm.GetCostMap}[k]()))
              for k, v in c["want"].items()}
        say("PatchMatch 4 x 64x48, 8 sweeps + filter, kernel", pm.GetSweepKernelName(), "bit-exact vs oracle:", eq)

    def fusion_stage():
        import fusion_oracle
        from colmap_amd import fusion
        images = cases["fusion_images"]
        overlap = [[j for j in range(4) if j != i] for i in range(4)]
        for nt in (1, 3, -1):
            opt = fusion.StereoFusionOptions(num_threads=nt, min_num_pixels=2, max_reproj_error=3.0, max_depth_error=0.05,
                                             max_normal_error=30.0)
            want = fusion_oracle.fuse(opt, images, overlap, mode=1)
            got = fusion.fuse(opt, images, overlap)
            same = (len(got.xyz) == len(want.xyz) and np.array_equal(got.xyz, want.xyz) and np.array_equal(got.normal, want.normal)
                    and np.array_equal(got.rgb, want.rgb) and all(np.array_equal(a, b) for a, b in zip(got.visibility, want.visibility)))
            extra = ""
            if nt == 1:
                w0 = fusion_oracle.fuse(opt, images, overlap, mode=0)
                extra = f" == row-major mode 0: {len(w0.xyz) == len(got.xyz) and np.array_equal(w0.xyz, got.xyz)}"
            say(f"fusion num_threads={nt}: equal={same} points={len(got.xyz)}{extra}")

    def ba_stage():
        import ba_oracle
        from colmap_amd import estimators as est, scene

        def flat(frames, points, track, seed, mixed=False):
            d = scene.synthesize_flat(frames, points, track, seed=seed, mixed_models=mixed,
                                      noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
            fp = est.FlatProblem.from_arrays(d)
            est.fix_gauge_two_cams(fp)
            return fp

        for name, fp, kw in (("dense tier 10 x 250 mixed", flat(10, 250, 5, 61, True),
                              dict(max_num_iterations=30, gradient_tolerance=1e-8, function_tolerance=1e-12,
                                   linear_solver_type=est.SOLVER_DENSE_SCHUR)),
                             ("sparse tier 120 x 3000", flat(120, 3000, 6, 120), dict(max_num_iterations=4, linear_solver_type=est.SOLVER_AUTO)),
                             ("PCG tier 40 x 2000 mixed", flat(40, 2000, 8, 40, True), dict(max_num_iterations=12))):
            so = est.SolverOptions(**kw)
            a, b = fp.copy(), fp.copy()
            want = est.solve_flat(a, so, solve_fn=ba_oracle.solve_fn)
            got = est.solve_flat(b, so, gpu_index=0)
            n = min(len(want.log_cost), len(got.log_cost))
            rel = float(np.max(np.abs(got.log_cost[:n] - want.log_cost[:n]) / want.log_cost[:n])) if n else -1.0
            say(f"BA {name}: iters {got.num_iterations}/{want.num_iterations} pcg {got.total_linear_iterations}/{want.total_linear_iterations} "
                f"max rel cost diff {rel:.2e} params {float(np.abs(b.poses - a.poses).max()):.2e} tier {got.linear_solver_used}")
        t = time.time()
        fp = flat(1000, 200000, 10, 42)
        say(f"BA-1 generated in {time.time() - t:.1f}s")
        for name, kw in (("PCG", dict(max_num_iterations=10)), ("PCG", dict(max_num_iterations=10)),
                         ("SPARSE_SCHUR", dict(max_num_iterations=4, linear_solver_type=est.SOLVER_SPARSE_SCHUR))):
            b = fp.copy()
            t = time.time()
            s = est.solve_flat(b, est.SolverOptions(**kw), gpu_index=0)
            say(f"BA-1 {name}: {s.num_iterations} LM iterations in {s.lm_seconds:.4f}s = {s.num_iterations / max(s.lm_seconds, 1e-9):.1f} LM-it/s "
                f"(wall {time.time() - t:.2f}s, pcg {s.total_linear_iterations}, factor {s.factor_seconds:.4f}s, cost {s.initial_cost:.6e} -> {s.final_cost:.6e})")

    stage("A PatchMatch", pm_stage)
    stage("B fusion", fusion_stage)
    stage("C bundle adjustment", ba_stage)
    say("done")


if __name__ == "__main__":
    make() if sys.argv[1:] == ["make"] else run()
