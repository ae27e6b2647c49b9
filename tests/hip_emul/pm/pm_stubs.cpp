// TEST INFRASTRUCTURE ONLY (tests/hip_emul): the array the kernels' `extern __shared__ char smem[]` refers to --
// dynamic LDS, the whole 160 KB of a CU.
namespace colmap_amd {
alignas(16) thread_local char smem[160 * 1024];
}
