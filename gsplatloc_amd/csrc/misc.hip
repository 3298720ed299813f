// Library identification and status strings.
#include "gsloc_common.h"

extern "C" const char* gsl_version(void) { return "gsloc_hip 0.1.0 gfx950"; }

extern "C" const char* gsl_status_string(int status) {
  switch (status) {
    case GSL_OK: return "ok";
    case GSL_ERR_BAD_ARG: return "bad argument";
    case GSL_ERR_WORKSPACE: return "workspace too small";
    case GSL_ERR_HIP: return "HIP launch error";
    default: return "unknown status";
  }
}

// Zeroing of scratch arrays by a kernel of the library instead of hipMemsetAsync: a captured call sequence then
// holds kernel nodes only.  (A memset node in a captured iteration made ROCm 7.2 fault -- "write access to a
// read-only page" -- as soon as anything else was issued on the stream between two replays)
namespace gsl {
__global__ __launch_bounds__(256) void k_zero_u32(uint32_t* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0u;
}
int zero_u32(void* p, size_t n_dwords, hipStream_t st) {
  if (n_dwords == 0) return GSL_OK;
  hipLaunchKernelGGL(k_zero_u32, dim3((unsigned)((n_dwords + 255) / 256)), dim3(256), 0, st, (uint32_t*)p, n_dwords);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}

// Diagnostics: fills the LDS of every CU with a bit pattern (gsl_dev_poison_lds).  The compositing kernels let a lane
// without a candidate read SOME staged slot and multiply it by an exact zero, so every slot they can read must have been
// written in the same batch: a test runs them after LDS has been filled with NaNs and expects bit-identical results.
__global__ __launch_bounds__(1024) void k_poison_lds(uint32_t pattern, uint32_t* __restrict__ sink) {
  extern __shared__ uint32_t lds[];
  const int n = 65536 / 4;
  for (int i = threadIdx.x; i < n; i += 1024) lds[i] = pattern;
  __syncthreads();
  if (sink && lds[(threadIdx.x * 7) % n] != pattern) sink[0] = 1;  // (keeps the stores alive)
}
}  // namespace gsl

extern "C" int gsl_dev_poison_lds(uint32_t pattern, void* stream) {
  // 64 KB per workgroup, 2048 workgroups: two per CU at a time, so every CU's LDS is written a few times over
  hipLaunchKernelGGL(gsl::k_poison_lds, dim3(2048), dim3(1024), 65536, (hipStream_t)stream, pattern, (uint32_t*)nullptr);
  GSL_CHECK_LAUNCH();
  return GSL_OK;
}
