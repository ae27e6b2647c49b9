// TEST INFRASTRUCTURE ONLY (tests/hip_emul): the array the kernels' `extern __shared__ char smem[]` refers to --
// dynamic LDS, the whole 160 KB of a CU. It registers itself with the stand-in so that HIP_EMUL_POISON can refill it
// before every workgroup.
#include <hip/hip_runtime.h>
namespace colmap_amd {
alignas(16) thread_local char smem[160 * 1024];
}
namespace {
struct Register {
  Register() {
    hip_emul::dynamic_lds_base() = reinterpret_cast<unsigned char*>(colmap_amd::smem);
    hip_emul::dynamic_lds_size() = sizeof(colmap_amd::smem);
  }
} g_register;
}
