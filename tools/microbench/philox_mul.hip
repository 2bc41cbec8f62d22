// Microbenchmark: Philox4x32-10 with v_mul_hi_u32 + v_mul_lo_u32 (compiler default) against one
// v_mad_u64_u32 per 32x32->64 product.  Build: hipcc --offload-arch=gfx950 -O3 philox_mul.hip -o philox_mul
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>
__device__ __forceinline__ void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  if (MODE == 0) {
    hi = __umulhi(a, b);
    lo = a * b;
  } else {
    unsigned long long p, cc;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(cc) : "v"(a), "v"(b));
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
}

template <int MODE>
__device__ __forceinline__ void philox(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    mulhilo<MODE>(0xD2511F53u, c[0], hi0, lo0);
    mulhilo<MODE>(0xCD9E8D57u, c[2], hi1, lo1);
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

template <int MODE>
__global__ __launch_bounds__(1024) void bench(uint32_t* out, int iters) {
  uint32_t c[4] = {threadIdx.x, blockIdx.x, 0u, 7u};
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {  // independent calls (counter mode), like the jitter of the flush
    uint32_t d[4] = {c[0] + (uint32_t)i, c[1], c[2], c[3]};
    philox<MODE>(d, 1u, 2u);
    acc ^= d[0] ^ d[1] ^ d[2] ^ d[3];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  uint32_t* d;
  const int blocks = 256 * 4, threads = 1024, iters = 2000;
  hipMalloc(&d, sizeof(uint32_t) * blocks * threads);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  uint32_t h[2][4];
  for (int mode = 0; mode < 2; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) bench<0><<<blocks, threads>>>(d, iters); else bench<1><<<blocks, threads>>>(d, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double calls = (double)blocks * threads * iters;
      printf("mode %d (%s): %.3f ms, %.1f G philox/s\n", mode, mode ? "v_mad_u64_u32" : "mul_hi+mul_lo", ms, calls / ms * 1e-6);
    }
    hipMemcpy(h[mode], d, sizeof h[mode], hipMemcpyDeviceToHost);
  }
  printf("results %s\n", (h[0][0] == h[1][0] && h[0][3] == h[1][3]) ? "identical" : "DIFFER");
  return 0;
}
