// Stress test: do two workgroups that share a CU keep their LDS hash tables apart?
// Every workgroup repeats the same deterministic insert pattern (CAS claim + u64 add + u32 max,
// like scatter.hip) into its own LDS table and checks occupancy and charge sums against the first
// repetition.  Run with 1 and with 2 workgroups per CU:  ./lds_coresidency <wgs_per_cu>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int THREADS = 512;
constexpr int SLOTS = 4096;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;

struct __align__(16) Shared {
  unsigned long long chg[SLOTS];
  uint32_t keys[SLOTS];
  unsigned int occupied, n_claimed;
  unsigned long long sum;
  char pad[81328 - SLOTS * 12 - 16];  // same LDS footprint as the scatter variant: 2 workgroups fill a CU
};

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

constexpr int NREG = 88;  // per-thread register state that must survive: fills the 128-VGPR budget

__global__ __launch_bounds__(THREADS, 4) void stress(unsigned long long* errors, int reps, int inserts,
                                                    const int16_t* __restrict__ table, uint32_t table_n) {
  __shared__ Shared sh;
  unsigned int ref_occ = 0;
  unsigned long long ref_sum = 0;
  uint32_t r[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = threadIdx.x * 977u + blockIdx.x * 131u + (uint32_t)i;
  for (int rep = 0; rep < reps; ++rep) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = r[i] * 1664525u + 1013904223u + (uint32_t)i;
    for (int i = threadIdx.x; i < SLOTS; i += THREADS) { sh.keys[i] = EMPTY; sh.chg[i] = 0ull; }
    if (threadIdx.x == 0) { sh.occupied = 0; sh.n_claimed = 0; sh.sum = 0; }
    __syncthreads();
    {  // ten scattered 2-byte gathers in flight from a table that does not fit the L1, like the pad LUT
      int got[10];
      uint32_t idx[10];
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        idx[g] = mix(blockIdx.x * 2654435761u + threadIdx.x * 40503u + (uint32_t)(rep * 10 + g)) % table_n;
        got[g] = (int)*reinterpret_cast<const int16_t*>(reinterpret_cast<const char*>(table) + 2u * idx[g]);
      }
      asm volatile("" : "+v"(got[0]), "+v"(got[1]), "+v"(got[2]), "+v"(got[3]), "+v"(got[4]), "+v"(got[5]), "+v"(got[6]),
                   "+v"(got[7]), "+v"(got[8]), "+v"(got[9]));
      unsigned int wrong = 0;
#pragma unroll
      for (int g = 0; g < 10; ++g) wrong += got[g] != (int)(int16_t)(mix(idx[g]) & 0x7fffu);
      if (wrong) atomicAdd(errors + 2, (unsigned long long)wrong);
    }
    for (int k = 0; k < inserts; ++k) {
      // ~1500 distinct 24-bit keys per workgroup, each inserted ~3.4 times
      const uint32_t key = mix(blockIdx.x * 7919u + (uint32_t)((threadIdx.x * inserts + k) % 1500)) & 0xFFFFFFu;
      const uint32_t want = key | ((uint32_t)(k & 3) << 24);
      uint32_t b = (key * 2654435761u) >> 22;  // 1024 buckets of 4
      for (;;) {
        const uint4 k4 = *reinterpret_cast<const uint4*>(&sh.keys[b * 4]);
        const uint32_t w[4] = {k4.x, k4.y, k4.z, k4.w};
        int hit = -1, free_slot = -1;
        for (int s = 3; s >= 0; --s) {
          if ((w[s] & 0xFFFFFFu) == key && w[s] != EMPTY) hit = s;
          if (w[s] == EMPTY) free_slot = s;
        }
        if (hit < 0 && free_slot >= 0) {
          const uint32_t old = atomicCAS(&sh.keys[b * 4 + free_slot], EMPTY, want);
          if (old == EMPTY) { hit = free_slot; atomicAdd(&sh.n_claimed, 1u); }
          else if ((old & 0xFFFFFFu) == key) hit = free_slot;
          else continue;
        }
        if (hit >= 0) {
          atomicMax(&sh.keys[b * 4 + hit], want);
          atomicAdd(&sh.chg[b * 4 + hit], (unsigned long long)(key & 1023u) + 1ull);
          break;
        }
        b = (b + 1) & 1023u;
      }
    }
    __syncthreads();
    unsigned int occ = 0;
    unsigned long long sum = 0;
    for (int i = threadIdx.x; i < SLOTS; i += THREADS)
      if (sh.keys[i] != EMPTY) { occ++; sum += sh.chg[i] + (sh.keys[i] >> 24); }
    atomicAdd(&sh.occupied, occ);
    atomicAdd(&sh.sum, sum);
    __syncthreads();
    if (threadIdx.x == 0) {
      if (rep == 0) { ref_occ = sh.occupied; ref_sum = sh.sum; }
      if (sh.occupied != ref_occ || sh.sum != ref_sum || sh.occupied != sh.n_claimed) atomicAdd(errors, 1ull);
    }
    __syncthreads();
  }
  // the register state must equal its recomputation
  unsigned int bad = 0;
#pragma unroll 1
  for (int i = 0; i < NREG; ++i) {
    uint32_t e = threadIdx.x * 977u + blockIdx.x * 131u + (uint32_t)i;
    for (int rep = 0; rep < reps; ++rep) e = e * 1664525u + 1013904223u + (uint32_t)i;
    uint32_t got = 0;
#pragma unroll
    for (int k = 0; k < NREG; ++k) got = k == i ? r[k] : got;
    bad += got != e;
  }
  if (bad) atomicAdd(errors + 1, (unsigned long long)bad);
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 2;
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  unsigned long long* d_err;
  hipMalloc(&d_err, 3 * sizeof(unsigned long long));
  hipMemset(d_err, 0, 3 * sizeof(unsigned long long));
  const uint32_t table_n = 313600;  // 560 x 560 int16, the size of the pad LUT
  int16_t* h_table = (int16_t*)malloc(table_n * sizeof(int16_t));
  for (uint32_t i = 0; i < table_n; ++i) {
    uint32_t x = i;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    h_table[i] = (int16_t)(x & 0x7fffu);
  }
  int16_t* d_table;
  hipMalloc(&d_table, table_n * sizeof(int16_t));
  hipMemcpy(d_table, h_table, table_n * sizeof(int16_t), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  stress<<<cus * per_cu, THREADS>>>(d_err, 4000, 10, d_table, table_n);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long err2[3] = {0, 0, 0};
  hipMemcpy(err2, d_err, sizeof err2, hipMemcpyDeviceToHost);
  const unsigned long long err = err2[0];
  printf("%d workgroups per CU (%d workgroups, %zu B LDS each): %.1f ms, %llu repetitions disagreed with their first, %llu corrupted registers, %llu wrong gathers\n",
         per_cu, cus * per_cu, sizeof(Shared), ms, err, err2[1], err2[2]);
  return 0;
}
