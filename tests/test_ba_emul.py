"""colmap_amd/csrc/ba_kernels.hip + ba_schur_explicit.hip -- the UNMODIFIED product sources, host loop and kernels --
executed on the CPU against the fp64 checker (oracle/ba_oracle.c), by running parity tests of tests/test_ba_gpu.py
with the library swapped for a CPU build of the same files.

tests/hip_emul/ is a HIP stand-in for exactly this purpose (the lanes of a workgroup as fibers, cross-lane
primitives and __syncthreads as barriers, v_mfma_f64_16x16x4_f64 with the hardware's operand / result layout,
streams synchronous): test infrastructure, never loaded by the product, whose library is built by hipcc and has
no CPU path. What these tests pin without a GPU: the linearisation of every camera-model family, the c-order /
p-order layouts and the tiled point passes, the wave-per-chunk reductions, the Schur-Jacobi blocks on the (emulated)
matrix cores, the pipelined PCG with its device-side stopping test, the LM loop, both exact tiers (explicit reduced
camera system with fixed-point accumulation, blocked Cholesky), priors, rigs, robust losses, the adapters. Sizes are
the small ones of the GPU suite (a lane is a fiber here: ~1 us per cross-lane primitive and lane). The GPU tests
run the same functions through the hipcc build."""
import ctypes as C
import os
import subprocess

import pytest

import test_ba_gpu as G
from colmap_amd import estimators as est

_HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hip_emul")
_CSRC = os.path.join(os.path.dirname(_HERE), "..", "colmap_amd", "csrc")
_LIB = None


def _emul_lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libba_emul.so")
        deps = [os.path.join(_CSRC, f) for f in ("ba_kernels.hip", "ba_schur_explicit.hip", "ba_schur_explicit.h")]
        deps += [os.path.join(_HERE, "hip", "hip_runtime.h"), os.path.join(_HERE, "rccl", "rccl.h"),
                 os.path.join(_HERE, "build_ba.sh"), os.path.join(os.path.dirname(_HERE), "..", "include", "colmap_amd_ba.h")]
        if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
            subprocess.check_call(["sh", os.path.join(_HERE, "build_ba.sh")])
        _LIB = C.CDLL(path)
        _LIB.ba_last_error.restype = C.c_char_p
    return _LIB


@pytest.fixture(autouse=True)
def emulated_library(monkeypatch):
    """est.solve_flat(..., gpu_index=0) -- what the GPU tests call -- reaches ba_solve of the CPU build."""
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") and "HIP_EMUL_CXX" not in os.environ:
        pytest.skip("the stand-in is built with ROCm's clang++ as host compiler")
    monkeypatch.setattr(est, "lib", _emul_lib)


def test_emulated_library_is_the_one_under_test():
    L = est.lib()
    assert L is _emul_lib() and os.path.basename(L._name) == "libba_emul.so"
    est._check_abi(L)


@pytest.mark.parametrize("frames,points,track,mixed", [(6, 40, 4, True), (12, 300, 5, False)])
def test_solution_matches_oracle(frames, points, track, mixed):
    G.test_solution_matches_oracle(frames, points, track, mixed)


def test_constant_blocks_gauges_and_partial_problems():
    G.test_constant_blocks_are_untouched_bitwise()
    G.test_shared_intrinsics_and_three_point_gauge()
    G.test_heavy_blocks_reduce_their_chunks_first()
    G.test_pair_terms_per_incidence_equal_per_observation()
    G.test_only_points_variable_and_only_cameras_variable()
    G.test_error_behaviour()
    G.test_iteration_callback_stops_with_the_last_accepted_state()


def test_reference_backend_cases_and_golden_fixture():
    G.test_backend_interface_reference_cases()
    G.test_against_committed_golden_fixture()
    G.test_reference_pose_prior_backend_case()


def test_exact_tiers():
    """DENSE_SCHUR / AUTO against the checker's exact tier; explicit formation against operator products."""
    G.test_dense_schur_tier_matches_oracle()
    G.test_exact_tier_explicit_formation_equals_operator_products()
    G.test_exact_tier_pair_major_formation_equals_point_major(60, 3000, 6, True)


def test_blocked_cholesky_beyond_one_panel():
    """n_c = 700: eleven 64-wide panels, three outer panels of 256, the two-stream lookahead order; the panel and
    trailing-update kernels on the emulated matrix cores (four waves per workgroup)."""
    fp = G._flat(120, 1500, 6, seed=120)
    assert est.fix_gauge_two_cams(fp)
    (a, want), (b, got) = G._both(fp, max_num_iterations=3, linear_solver_type=est.SOLVER_AUTO)
    assert got.linear_solver_used == want.linear_solver_used == est.SOLVER_SPARSE_SCHUR
    assert (got.log_linear_iters[:got.num_iterations] == 1).all()
    import numpy as np
    np.testing.assert_allclose(got.log_cost, want.log_cost, rtol=1e-7)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)


@pytest.mark.parametrize("model", [9, 14, 17])
def test_camera_model_families(model):
    """RADIAL_FISHEYE, SIMPLE_FISHEYE (no distortion parameter), EQUIRECTANGULAR; SIMPLE_RADIAL / PINHOLE / OPENCV
    variants run in the mixed problems above."""
    params = {9: (900.0, 512.0, 384.0, 0.03, -0.004), 14: (900.0, 512.0, 384.0), 17: (1024.0, 768.0)}[model]
    G._model_matches_oracle(model, params)


def test_rigs_priors_and_robust_loss():
    G.test_rig_frames_match_oracle()
    G.test_constant_rig_from_world_rotation_matches_oracle()
    G.test_pose_prior_adjuster_on_rigs_matches_oracle()
    G.test_robust_losses_match_oracle(est.LossFunctionType.SOFT_L1, 1.0)


# ------------------------------------------------------------------------------------------------
# the explicit formation called directly (ba_schur_explicit.h is an internal C++ interface: mangled names)
# ------------------------------------------------------------------------------------------------

class _FormArgs(C.Structure):
    _fields_ = [("n_obs", C.c_int), ("n_points", C.c_int), ("n_c", C.c_int), ("n_poses", C.c_int), ("kd", C.c_int)] + \
               [(n, C.c_void_p) for n in ("Jpose", "Jcam", "Jsens", "Jpt", "Cinv", "a2c", "pt_ptr", "pt_off", "a_pose", "a_cam",
                                          "a_pt", "pairs", "a_sensor", "pose_off", "pose_dim", "cam_off", "cam_dim", "sens_off")] + \
               [("fixed_point", C.c_bool), ("bad", C.c_void_p)]


class _Workspace(C.Structure):
    _fields_ = [("Linv", C.c_void_p), ("tmp", C.c_void_p), ("info", C.c_void_p), ("st2", C.c_void_p),
                ("ev_panel", C.c_void_p), ("ev_u2", C.c_void_p), ("min_rows128", C.c_int)]


class _PairLists(C.Structure):
    _fields_ = [("inc", C.c_void_p), ("n_inc", C.c_longlong), ("rec", C.c_void_p), ("rec_doubles", C.c_size_t)]


def _explicit_entry_points(pairs=False):
    L = _emul_lib()
    names = subprocess.run(["nm", "-D", "--defined-only", L._name], capture_output=True, text=True, check=True).stdout.split()
    pick = lambda key: getattr(L, next(n for n in names if key in n))
    if pairs:
        build, free = pick("ba_explicit16build_pair_listsE"), pick("ba_explicit15free_pair_listsE")
        build.restype = C.c_bool
        return build, free
    return pick("ba_explicit4formE"), pick("ba_explicit6finishE"), pick("ba_explicit12factor_solveE")


@pytest.mark.parametrize("scale,expect_bad", [(0.1, False), (100.0, True)])
def test_fixed_point_formation_and_its_overflow_flag(scale, expect_bad):
    """form_kernel<.., FIXED> on two observations of one point in two pose blocks against numpy:
    S = sum_ab J_a^T (delta_ab I - E_a C^-1 E_b^T) J_b, accumulated in 2^-60 fixed point. With columns scaled as Jacobi
    scaling leaves them (|term| < 1) the matrix is exact to the quantum; a term the fixed point cannot hold raises
    FormArgs::bad, finish() poisons S[0][0] and the factorisation answers NaN (the LM loop rejects such a step) --
    the integer conversion alone would have produced a finite, wrong matrix."""
    import numpy as np
    form, finish, factor_solve = _explicit_entry_points()
    rng = np.random.default_rng(5)
    N, n_c = 2, 12
    Jpose = (scale * rng.uniform(-1, 1, (12, N)))          # c-order planes [2 * 6][N]
    Jpt = (scale * rng.uniform(-1, 1, (6, N)))             # p-order planes [2 * 3][N]
    E = [np.array([[Jpt[r * 3 + m, a] for m in range(3)] for r in range(2)]) for a in range(N)]
    Cinv = np.linalg.inv(sum(e.T @ e for e in E) + 0.5 * scale * scale * np.eye(3))
    ints = lambda v: np.ascontiguousarray(v, np.int32)
    arrs = dict(Jpose=np.ascontiguousarray(Jpose), Jcam=np.zeros((8, N)), Jpt=np.ascontiguousarray(Jpt),
                Cinv=np.ascontiguousarray(Cinv.reshape(1, 9)), a2c=ints([0, 1]), pt_ptr=ints([0, 2]), pt_off=ints([0]),
                a_pose=ints([0, 1]), a_cam=ints([0, 0]), pose_off=ints([0, 6]), pose_dim=ints([6, 6]), cam_off=ints([-1]),
                cam_dim=ints([0]))
    bad = np.zeros(1, np.int32)
    fa = _FormArgs(n_obs=N, n_points=1, n_c=n_c, kd=4, fixed_point=True, bad=bad.ctypes.data)
    for k, v in arrs.items():
        setattr(fa, k, v.ctypes.data)
    S = np.full((n_c + 1, n_c), 7.0)  # (n_c + 1 rows: factor_solve's buffer)
    form(C.byref(fa), S.ctypes.data_as(C.c_void_p), None)
    finish(S.ctypes.data_as(C.c_void_p), C.c_int(n_c), C.c_bool(True), bad.ctypes.data_as(C.c_void_p), None)
    J = [np.array([[Jpose[r * 6 + d, a] for d in range(6)] for r in range(2)]) for a in range(N)]
    want = np.zeros((n_c, n_c))
    for a in range(N):
        for b in range(N):
            M = (np.eye(2) if a == b else 0.0) - E[a] @ Cinv @ E[b].T
            want[6 * a:6 * a + 6, 6 * b:6 * b + 6] += J[a].T @ M @ J[b]
    assert bool(bad[0]) == expect_bad
    if not expect_bad:
        low = np.tril_indices(n_c)
        assert np.abs(want).max() < 1.0
        np.testing.assert_allclose(S[low], want[low], rtol=0, atol=144 * 2.0 ** -60 + 1e-17)
        return
    assert np.isnan(S[0, 0])
    x, rhs = np.zeros(n_c), np.ones(n_c)
    linv, tmp, info = np.zeros(64 * 64), np.zeros(n_c), np.zeros(1, np.int32)
    ws = _Workspace(Linv=linv.ctypes.data, tmp=tmp.ctypes.data, info=info.ctypes.data, min_rows128=12 * 128)
    factor_solve(S.ctypes.data_as(C.c_void_p), C.c_int(n_c), rhs.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p),
                 C.byref(ws), None, None, None, None)
    assert info[0] == 1 and np.isnan(x).all()


def _random_formation_problem(seed, n_poses, n_points, shared_cams, rigs):
    """A random linearisation in the layouts FormArgs describes (c-order planes, p-order point columns, p-order
    topology) with constant poses / cameras / points, tracks of 1-7 observations, optionally cameras shared between
    images and rig frames (several images per pose block, each with its own sensor_from_rig block)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    n_cams = 2 if shared_cams else n_poses
    n_sens = 3 if rigs else 0
    obs = []  # (point, pose, cam, sensor)
    for j in range(n_points):
        t = int(rng.integers(1, 8))
        images = set()
        while len(images) < t:
            pose = int(rng.integers(n_poses))
            sens = int(rng.integers(n_sens)) if rigs else -1
            images.add((pose, sens))
        for pose, sens in sorted(images):
            obs.append((j, pose, pose % n_cams, sens))
    N = len(obs)
    perm = rng.permutation(N)  # p-order slot a -> c-order slot
    off = 0
    pose_off, pose_dim = [], []
    for i in range(n_poses):
        d = [6, 6, 6, 5, 0][int(rng.integers(5))]
        pose_dim.append(d); pose_off.append(off if d else -1); off += d
    cam_off, cam_dim = [], []
    for i in range(n_cams):
        d = [2, 3, 0][int(rng.integers(3))]
        cam_dim.append(d); cam_off.append(off if d else -1); off += d
    sens_off = []
    for i in range(n_sens):
        v = bool(rng.integers(2))
        sens_off.append(off if v else -1); off += 6 if v else 0
    n_c = off
    pt_off = [(-1 if rng.integers(5) == 0 else 3 * j) for j in range(n_points)]
    scale = 0.08
    Jpose, Jcam, Jsens = (scale * rng.uniform(-1, 1, (12, N)) for _ in range(3))
    Jcam = scale * rng.uniform(-1, 1, (8, N))
    Jpt = scale * rng.uniform(-1, 1, (6, N))
    pt_ptr = np.zeros(n_points + 1, np.int32)
    for j, *_ in obs:
        pt_ptr[j + 1] += 1
    pt_ptr = np.cumsum(pt_ptr).astype(np.int32)
    Cinv = np.zeros((n_points, 9))
    want = np.zeros((n_c, n_c))
    for j in range(n_points):
        sl = range(pt_ptr[j], pt_ptr[j + 1])
        E = {a: Jpt[:, a].reshape(2, 3) for a in sl}
        Ci = np.linalg.inv(sum(E[a].T @ E[a] for a in sl) + 0.01 * np.eye(3))
        Cinv[j] = Ci.reshape(9)
        cols = {}
        for a in sl:
            _, pose, cam, sens = obs[a]
            c = perm[a]
            J, idx = [], []
            if pose_off[pose] >= 0:
                for d in range(pose_dim[pose]):
                    J.append((Jpose[d, c], Jpose[6 + d, c])); idx.append(pose_off[pose] + d)
            if cam_off[cam] >= 0:
                for d in range(cam_dim[cam]):
                    J.append((Jcam[d, c], Jcam[4 + d, c])); idx.append(cam_off[cam] + d)
            if sens >= 0 and sens_off[sens] >= 0:
                for d in range(6):
                    J.append((Jsens[d, c], Jsens[6 + d, c])); idx.append(sens_off[sens] + d)
            cols[a] = (np.array(J).reshape(-1, 2).T, idx)
        for a in sl:
            for b in sl:
                if pt_off[j] < 0 and a != b:
                    continue
                M = (np.eye(2) if a == b else 0.0) - (E[a] @ Ci @ E[b].T if pt_off[j] >= 0 else 0.0)
                (Ja, ia), (Jb, ib) = cols[a], cols[b]
                if ia and ib:
                    want[np.ix_(ia, ib)] += Ja.T @ M @ Jb
    ints = lambda v: np.ascontiguousarray(v, np.int32)
    arrs = dict(Jpose=np.ascontiguousarray(Jpose), Jcam=np.ascontiguousarray(Jcam), Jpt=np.ascontiguousarray(Jpt),
                Cinv=np.ascontiguousarray(Cinv), a2c=ints(perm), pt_ptr=ints(pt_ptr), pt_off=ints(pt_off),
                a_pose=ints([o[1] for o in obs]), a_cam=ints([o[2] for o in obs]), a_pt=ints([o[0] for o in obs]),
                pose_off=ints(pose_off), pose_dim=ints(pose_dim), cam_off=ints(cam_off), cam_dim=ints(cam_dim))
    if rigs:
        arrs.update(Jsens=np.ascontiguousarray(Jsens), a_sensor=ints([o[3] for o in obs]), sens_off=ints(sens_off))
    return dict(N=N, n_points=n_points, n_c=n_c, n_poses=n_poses, arrs=arrs, want=want)


@pytest.mark.parametrize("shared_cams,rigs,fixed", [(False, False, True), (True, False, True), (True, True, True),
                                                    (True, True, False)])
def test_pair_major_formation_against_numpy_and_the_point_major_kernel(shared_cams, rigs, fixed):
    """form() both ways on a random linearisation -- per-image cameras; cameras shared between images (their blocks are
    reached from many pairs of images, and a pair of observations of one shared block meets its diagonal twice); rig
    frames (several images per pose block: runs of one pair of pose blocks change their targets) -- with constant poses,
    cameras, sensors and points and 5-wide pose blocks: the pair-major formation (records, incidence lists sorted by
    pose pair, one wave per 64 incidences) and the point-major kernel (one atomic per term) against numpy, fixed-point
    and fp64 accumulation."""
    import numpy as np
    form, finish, _ = _explicit_entry_points()
    build, free = _explicit_entry_points(pairs=True)
    P = _random_formation_problem(11 + 2 * shared_cams + rigs, n_poses=9, n_points=60, shared_cams=shared_cams, rigs=rigs)
    bad = np.zeros(1, np.int32)
    fa = _FormArgs(n_obs=P["N"], n_points=P["n_points"], n_c=P["n_c"], n_poses=P["n_poses"], kd=4, fixed_point=fixed,
                   bad=bad.ctypes.data)
    for k, v in P["arrs"].items():
        setattr(fa, k, v.ctypes.data)
    n_c, low = P["n_c"], np.tril_indices(P["n_c"])
    assert np.abs(P["want"]).max() < 1.0
    got = {}
    for which in ("points", "pairs"):
        pl = _PairLists()
        if which == "pairs":
            assert build(C.byref(fa), C.byref(pl), None) and pl.n_inc >= P["N"]
            fa.pairs = C.addressof(pl)
        S = np.full((n_c, n_c), 7.0)
        form(C.byref(fa), S.ctypes.data_as(C.c_void_p), None)
        finish(S.ctypes.data_as(C.c_void_p), C.c_int(n_c), C.c_bool(fixed), bad.ctypes.data_as(C.c_void_p), None)
        fa.pairs = None
        free(C.byref(pl))
        assert bad[0] == 0 and pl.inc is None
        np.testing.assert_allclose(S[low], P["want"][low], rtol=0, atol=2e-15)
        got[which] = S[low]
    np.testing.assert_allclose(got["pairs"], got["points"], rtol=0, atol=1e-15)


def _factor_solve_directly(A, rhs, min_rows128):
    import numpy as np
    _, _, factor_solve = _explicit_entry_points()
    n = A.shape[0]
    S = np.full((n + 1, n), 1e300)  # the upper triangle is never read; row n: the right-hand side rides along
    S[np.tril_indices(n)] = A[np.tril_indices(n)]
    x = np.zeros(n)
    linv, tmp, info = np.zeros(((n + 63) // 64) * 64 * 64), np.zeros(n), np.zeros(1, np.int32)
    ws = _Workspace(Linv=linv.ctypes.data, tmp=tmp.ctypes.data, info=info.ctypes.data, min_rows128=min_rows128)
    factor_solve(S.ctypes.data_as(C.c_void_p), C.c_int(n), np.ascontiguousarray(rhs).ctypes.data_as(C.c_void_p),
                 x.ctypes.data_as(C.c_void_p), C.byref(ws), None, None, None, None)
    return S, linv.reshape(-1, 64, 64), x, int(info[0])


@pytest.mark.parametrize("n,min_rows128", [(45, 12 * 128), (64, 12 * 128), (333, 12 * 128), (900, 256)])
def test_blocked_cholesky_directly_against_numpy(n, min_rows128):
    """factor_solve on a random SPD matrix: the factor, the stored inverses of its diagonal blocks and the solution
    against numpy. n = 45 / 64: one (short / full) diagonal block -- chol_diag_kernel alone (blocked over 16 x 16 tiles,
    the inverse of the triangle riding along); 333: six panels, a ragged last block, two outer
    panels; 900 with the tile threshold lowered: the 128 x 128 trailing update (the columns right of the next outer panel)
    with its register prefetch, including diagonal tiles and a ragged edge."""
    import numpy as np
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n + 8))
    A = B @ B.T / n + 0.5 * np.eye(n)
    rhs = rng.standard_normal(n)
    S, linv, x, info = _factor_solve_directly(A, rhs, min_rows128)
    assert info == 0
    L = np.linalg.cholesky(A)
    low = np.tril_indices(n)
    np.testing.assert_allclose(S[low], L[low], rtol=0, atol=1e-12)
    for k in range((n + 63) // 64):
        kb = min(64, n - 64 * k)
        blk = L[64 * k:64 * k + kb, 64 * k:64 * k + kb]
        np.testing.assert_allclose(linv[k][:kb, :kb], np.linalg.inv(blk), rtol=0, atol=1e-11)
        assert (linv[k][kb:, :] == 0).all() and (linv[k][:, kb:] == 0).all()
        assert (np.triu(linv[k][:kb, :kb], 1) == 0).all()
    np.testing.assert_allclose(x, np.linalg.solve(A, rhs), rtol=0, atol=1e-10)


def test_blocked_cholesky_reports_a_failed_pivot():
    import numpy as np
    n = 100
    A = np.eye(n)
    A[70, 70] = -1.0
    _, _, x, info = _factor_solve_directly(A, np.ones(n), 12 * 128)
    assert info == 1 and np.isnan(x).all()


# ------------------------------------------------------------------------------------------------
# sharded solves: N > 1 ranks of the HIP solver on the CPU (gloo + the sum-over-ranks callback)
# ------------------------------------------------------------------------------------------------

def _emul_sharded_worker(*args, **kw):
    """Child process of the sharded GPU tests (spawned: a fresh interpreter): the same worker, its library swapped."""
    from colmap_amd import estimators as est_child
    est_child.lib = _emul_lib
    import test_ba_gpu as G_child
    G_child._sharded_worker(*args, **kw)


@pytest.fixture
def sharded_workers_on_the_stand_in(monkeypatch):
    monkeypatch.setattr(G, "_sharded_worker", _emul_sharded_worker)


def test_two_rank_point_sharded_solve(sharded_workers_on_the_stand_in):
    """ba_solve_sharded, observations sharded by point, two ranks: every rank linearises its own observations, one
    fused all-reduce per linearisation and one per implicit product; ranks agree bitwise and with the single-rank solve."""
    G.test_two_rank_sharded_solve_matches_single_gpu(est.SHARD_BY_POINT, False)


def test_three_rank_image_sharded_solve_with_shared_intrinsics(sharded_workers_on_the_stand_in):
    """Image sharding over three ranks with intrinsics blocks shared across ranks: the all-reduced incidence products
    (ba_inc_* kernels) give the sharded solve the single-rank Schur-Jacobi preconditioner -- the same CG iteration counts."""
    G.test_three_rank_sharded_solve_with_shared_intrinsics(est.SHARD_BY_IMAGE)


def test_two_rank_sharded_exact_tier(sharded_workers_on_the_stand_in):
    """DENSE_SCHUR sharded by point: the ranks' explicitly formed partial systems are summed (fixed point) and factored."""
    G.test_two_rank_sharded_exact_tier(est.SHARD_BY_POINT)


# ------------------------------------------------------------------------------------------------
# COLMAP_AMD_TEST_SLOW=1: the remaining comparisons of the GPU suite that take 15-100 s each on the stand-in
# ------------------------------------------------------------------------------------------------

_slow = pytest.mark.skipif(os.environ.get("COLMAP_AMD_TEST_SLOW", "0") == "0", reason="COLMAP_AMD_TEST_SLOW=1 runs the long stand-in cases")


@_slow
def test_slow_kernel_variants():
    G.test_split_linearisation_is_bit_identical(False)
    G.test_split_linearisation_is_bit_identical("three")
    for m in ("SIMPLE_RADIAL", "PINHOLE", "SIMPLE_PINHOLE"):
        G.test_plain_linearisation_is_bit_identical(m)
    G.test_fused_pcg_kernel_matches_separate_kernels()
    G.test_run_to_run_determinism()
    G.test_tracks_longer_than_a_tile()


@_slow
def test_slow_models_priors_and_operator_precision():
    G.test_variable_sensor_from_rig_matches_oracle()
    G.test_radial_model_matches_oracle()
    G.test_opencv_model_matches_oracle()
    G.test_position_priors_match_oracle(est.LossFunctionType.TRIVIAL)
    G.test_position_priors_match_oracle(est.LossFunctionType.CAUCHY)
    G.test_robust_losses_match_oracle(est.LossFunctionType.CAUCHY, 1.0)
    G.test_robust_losses_match_oracle(est.LossFunctionType.HUBER, 2.0)
    G.test_fp32_operator_reaches_the_fp64_solution(40, 2000, 8, True)
    G.test_solution_matches_oracle(40, 2000, 8, True)
