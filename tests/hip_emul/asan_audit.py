"""Memory-safety audit of the product's kernels and host code: the CPU stand-in builds compiled with AddressSanitizer
(asan_audit.sh) run comparisons of the GPU suite -- every global / LDS access of the kernels is then a checked access
(device allocations are heap blocks with red zones, LDS arrays are globals with red zones). On a GPU an out-of-bounds read
mostly goes unnoticed; here it stops the run. TEST INFRASTRUCTURE ONLY (tests/hip_emul/README.md).

    sh tests/hip_emul/asan_audit.sh            # builds into a temporary directory, runs the four audits, prints a summary
"""
import sys, os, time, ctypes as C, inspect
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')]
import numpy as np
import pm_oracle; pm_oracle.build()
from colmap_amd import mvs, estimators as est, fusion, _lib
D = os.environ['HIP_EMUL_ASAN_DIR'].rstrip('/') + '/'
class Lib:
    def __init__(s, libs): s._l = libs
    def __getattr__(s, n):
        for L in s._l:
            try: return getattr(L, n)
            except AttributeError: pass
        raise AttributeError(n)
which = sys.argv[1]
fus = 'libfusion_small_asan.so' if which == 'fusion_small' else 'libfusion_asan.so'
lib = Lib([C.CDLL(D+'libpm_asan.so'), C.CDLL(D+'libba_asan.so'), C.CDLL(D+fus)])
lib.pm_last_error.restype = C.c_char_p; lib.pm_device_count.restype = C.c_int; lib.ba_last_error.restype = C.c_char_p
lib.fusion_last_error.restype = C.c_char_p; lib.fusion_num_points.restype = C.c_size_t
for m in (_lib, est, fusion, mvs): m.lib = lambda: lib
def run(fn, **kw):
    t = time.time()
    sig = inspect.signature(fn)
    if 'pm_oracle' in sig.parameters: kw['pm_oracle'] = pm_oracle
    fn(**kw); print(fn.__name__, kw.get('name', ''), {k:v for k,v in kw.items() if k not in ('pm_oracle',)}, f"ok {time.time()-t:.1f}s", flush=True)
if which.startswith('fusion'):
    import test_fusion as T
    if which == 'fusion':
        for name in sorted(T._CASES): run(T.test_hip_fusion_equals_parallel_oracle, name=name)
        run(T.test_hip_fusion_options_masks_and_bounding_box); run(T.test_reference_integration_case_hip)
        for nt in (1, 3, -1): run(T.test_hip_fusion_pool_sizes_and_cut_passes, num_threads=nt)
    else:
        import test_fusion_emul as TE
        from colmap_amd import fusion as F
        import fusion_oracle
        class EP(F._HipEntryPoints): pass
        for shape in [(5, 64, 48, 0.003), (4, 24, 160, 0.01)]:
            n, w, h, sigma = shape
            images, overlap = TE._noisy(n, w, h, sigma), TE._overlap(n)
            opt = F.StereoFusionOptions(max_num_pixels=1000, **TE._LOOSE)
            want = fusion_oracle.fuse(opt, images, overlap, mode=1)
            assert TE._same(F.fuse(opt, images, overlap), want); print('small', shape, 'ok', flush=True)
elif which == 'ba':
    import test_ba_gpu as T
    run(T.test_solution_matches_oracle, frames=6, points=40, track=4, mixed=True)
    run(T.test_solution_matches_oracle, frames=12, points=300, track=5, mixed=False)
    run(T.test_constant_blocks_are_untouched_bitwise); run(T.test_shared_intrinsics_and_three_point_gauge)
    run(T.test_heavy_blocks_reduce_their_chunks_first)   # heavy-block pre-reduction + p-order W buffers
    run(T.test_pair_terms_per_incidence_equal_per_observation)   # pair terms per (point, block) incidence
    run(T.test_only_points_variable_and_only_cameras_variable); run(T.test_backend_interface_reference_cases)
    run(T.test_dense_schur_tier_matches_oracle); run(T.test_exact_tier_explicit_formation_equals_operator_products)
    run(T.test_rig_frames_match_oracle); run(T.test_pose_prior_adjuster_on_rigs_matches_oracle)
    run(T.test_reference_pose_prior_backend_case); run(T.test_constant_rig_from_world_rotation_matches_oracle)
    run(T._model_matches_oracle, model=17, params=(1024.0, 768.0))
    run(T.test_exact_tier_pair_major_formation_equals_point_major, frames=60, points=3000, track=6, shared=True)
    import test_ba_emul as TE
    TE.est.lib = lambda: lib
    TE.test_blocked_cholesky_beyond_one_panel(); print('blocked cholesky ok', flush=True)
    # the internal C++ interface called directly: both formations, the factorisation with the right-hand side's row
    TE._LIB = C.CDLL(D + 'libba_asan.so')
    for shared_cams, rigs, fixed in ((False, False, True), (True, False, True), (True, True, True), (True, True, False)):
        TE.test_pair_major_formation_against_numpy_and_the_point_major_kernel(shared_cams, rigs, fixed)
        print('formation', shared_cams, rigs, fixed, 'ok', flush=True)
    for n, m128 in ((45, 12 * 128), (64, 12 * 128), (333, 12 * 128), (900, 256)):
        TE.test_blocked_cholesky_directly_against_numpy(n, m128); print('factor_solve', n, 'ok', flush=True)
    TE.test_blocked_cholesky_reports_a_failed_pivot(); print('failed pivot ok', flush=True)
    for scale, expect_bad in ((0.1, False), (100.0, True)):
        TE.test_fixed_point_formation_and_its_overflow_flag(scale, expect_bad); print('fixed point', scale, 'ok', flush=True)
elif which == 'pm':
    import test_pm_emul as T
    run(T.test_initial_state_cost_pose_tables_and_reference_filter)
    for k in (1, 2, 3, 4): run(T.test_each_sweep_direction, nsweeps=k)
    run(T.test_geometric_consistency_pass_and_both_filters)
    run(T.test_generic_kernel_other_windows, radius=2, step=1); run(T.test_generic_kernel_other_windows, radius=5, step=2)
    run(T.test_baseline_source_count_s20_m15); run(T.test_batched_run_equals_single_runs)
    import test_pm_gpu as G
    run(G.test_group_shapes_do_not_change_results, cols=3, threads=64)
    run(G.test_sources_larger_than_reference_slot); run(G.test_error_behaviour)
    run(G.test_cached_images_in_distant_slabs_are_rehomed)   # slab allocator's test mode: cached images re-homed
    class _Req:   # (the tests' `request` fixture: only addfinalizer is used)
        def __init__(s): s.f = []
        def addfinalizer(s, f): s.f.append(f)
    for geom in (0, 1):   # two waves per column group (pm_sweep_pair_kernel): helper wave, mailbox words, barriers
        rq = _Req(); run(T.test_two_waves_per_column_pair_kernel, request=rq, geom=geom); [f() for f in rq.f]
print('AUDIT DONE', which)
