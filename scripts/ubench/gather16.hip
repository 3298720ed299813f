// FETCH_SIZE calibration for 16-byte gathers (dev tool).  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`:
//   k_stream : every lane reads 16 contiguous bytes of a 1 GiB buffer once (wide coalesced streaming read)
//   k_gather : every lane reads one 16-byte record at a pseudo-random index of the same buffer (one per 64-byte line at most)
//   k_gather3: three 16-byte records at the same random index of three 256 MiB arrays (the Q0/Q1/Q2 pattern)
// The known byte counts are printed; compare with FETCH_SIZE (KiB) per dispatch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_stream(const float4* __restrict__ b, size_t n, float* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  float4 v = b[i % n];
  if (v.x == 123.456f) out[0] = v.y;
}
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__global__ void k_gather(const float4* __restrict__ b, size_t n, float* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float4 v = b[(size_t)hash32(i) % n];
  if (v.x == 123.456f) out[0] = v.y;
}
__global__ void k_gather3(const float4* __restrict__ b, size_t n3, float* out) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  size_t g = (size_t)hash32(i) % n3;
  float4 v0 = b[g], v1 = b[n3 + g], v2 = b[2 * n3 + g];
  if (v0.x + v1.x + v2.x == 123.456f) out[0] = v0.y;
}
int main() {
  size_t n = (size_t)1 << 26;  // 64 Mi float4 = 1 GiB (4x the Infinity Cache)
  float4* b; float* out;
  hipMalloc(&b, n * sizeof(float4)); hipMalloc(&out, 64);
  hipMemset(b, 0, n * sizeof(float4));
  size_t lanes = (size_t)1 << 24;  // 16 Mi lanes
  for (int rep = 0; rep < 2; ++rep) {
    k_stream<<<lanes / 256, 256>>>(b, n, out);
    k_gather<<<lanes / 256, 256>>>(b, n, out);
    k_gather3<<<lanes / 256, 256>>>(b, n / 4, out);
  }
  hipDeviceSynchronize();
  printf("k_stream : %zu lanes x 16 B = %.1f MiB useful, all of it fetched once\n", lanes, lanes * 16.0 / 1048576.0);
  printf("k_gather : %zu lanes x 16 B = %.1f MiB useful; one 64-B line per lane = %.1f MiB, one 128-B line = %.1f MiB\n", lanes,
         lanes * 16.0 / 1048576.0, lanes * 64.0 / 1048576.0, lanes * 128.0 / 1048576.0);
  printf("k_gather3: %zu lanes x 48 B = %.1f MiB useful; three 64-B lines = %.1f MiB, three 128-B lines = %.1f MiB\n", lanes,
         lanes * 48.0 / 1048576.0, lanes * 192.0 / 1048576.0, lanes * 384.0 / 1048576.0);
  return 0;
}
