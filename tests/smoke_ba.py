"""smoke(): one small bundle-adjustment solve on cuda:0 through the C ABI, checked against the
fp64 CPU oracle (test infrastructure; imported only from __graft_entry__.smoke(); lives under tests/ because it uses the oracle)."""
import numpy as np


def run():
    import ba_oracle
    from colmap_amd import estimators as est, scene
    d = scene.synthesize_flat(12, 300, 5, seed=3, mixed_models=True,
                              noise=scene.SyntheticNoiseOptions(0.01, 1.0, 0.05, 1.0))
    fp = est.FlatProblem.from_arrays(d)
    assert est.fix_gauge_two_cams(fp)
    so = est.SolverOptions(gradient_tolerance=1e-10, max_num_iterations=200)
    a, b = fp.copy(), fp.copy()
    want = est.solve_flat(a, so, solve_fn=ba_oracle.solve_fn)
    got = est.solve_flat(b, so, gpu_index=0)
    assert got.num_residuals == want.num_residuals
    assert abs(got.final_cost - want.final_cost) <= 1e-8 * want.final_cost, (got.final_cost, want.final_cost)
    np.testing.assert_allclose(b.points, a.points, atol=1e-6)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)
    print(f"smoke: BA HIP == oracle (final cost {got.final_cost:.9e}, rel diff "
          f"{abs(got.final_cost - want.final_cost) / want.final_cost:.1e}) on 12 cameras x 300 points")
    # the exact tier (explicit reduced camera system, pair-major formation, blocked Cholesky) on the same problem
    se = est.SolverOptions(gradient_tolerance=1e-8, function_tolerance=1e-12, max_num_iterations=30,
                           linear_solver_type=est.SOLVER_DENSE_SCHUR)
    a, b = fp.copy(), fp.copy()
    want = est.solve_flat(a, se, solve_fn=ba_oracle.solve_fn)
    got = est.solve_flat(b, se, gpu_index=0)
    assert got.linear_solver_used == est.SOLVER_DENSE_SCHUR
    assert abs(got.final_cost - want.final_cost) <= 1e-8 * want.final_cost, (got.final_cost, want.final_cost)
    np.testing.assert_allclose(b.poses, a.poses, atol=1e-6)
    print(f"smoke: BA exact tier HIP == oracle (final cost {got.final_cost:.9e}, {got.num_iterations} exact Newton steps)")
