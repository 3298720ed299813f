// LDS atomic throughput microbenchmark (dev tool): ns per wave-instruction per CU for ds_add_f32 / ds_add_u32
// with G lanes sharing each address.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, int share) {
  __shared__ float buf[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0.f;
  __syncthreads();
  int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int idx = wv * 1024 + (lane / share) * 11;  // `share` lanes per address, pitch 11 words
  float v = 1.0f + lane;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      if (KIND == 0) atomicAdd(&buf[idx + r], v);
      if (KIND == 1) atomicAdd((unsigned*)&buf[idx + r], 1u);
      if (KIND == 2) buf[idx + r] = v;                 // plain store for comparison
      if (KIND == 3) { float o = atomicAdd(&buf[idx + r], v); v += o * 1e-30f; }  // returning
    }
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = buf[threadIdx.x] + v;
}
template <int KIND> void run(const char* name, float* d, int share) {
  int iters = 2000, grid = 256 * 4;
  k<KIND><<<grid, 256>>>(d, 10, share); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<KIND><<<grid, 256>>>(d, iters, share); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double winstr_per_cu = (double)iters * 10 * (grid * 4.0) / 256.0;
  printf("%-22s share=%2d  %.3f ms -> %.2f ns per wave-instr per CU\n", name, share, ms, ms * 1e6 / winstr_per_cu);
}
int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
  for (int s : {1, 2, 4, 8, 16, 64}) { run<0>("ds_add_f32", d, s); run<1>("ds_add_u32", d, s); run<3>("ds_add_rtn_f32", d, s); }
  run<2>("ds_write_b32", d, 1);
}
