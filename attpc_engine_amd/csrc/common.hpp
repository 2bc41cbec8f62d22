// common.hpp -- shared device helpers of the AT-TPC HIP engine (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attpc_engine.h"

namespace attpc {

// detector/constants.py:23-35 -- scipy CODATA doubles, checked in tests/test_host_api.py
constexpr double MEV_2_JOULE = 1.6021766339999998e-13;
constexpr double MEV_2_KG = 1.7826619216278976e-30;
constexpr double C_LIGHT = 299792458.0;
constexpr double E_CHARGE = 1.602176634e-19;
constexpr double KE_LIMIT = 1e-6;     // detector/solver.py:14
constexpr double RHO_MAX = 0.292;     // detector/solver.py:240
constexpr double PI = 3.141592653589793;
constexpr double TWO_PI = 2.0 * PI;

// RNG domains of the counter word (see DESIGN.md "Random streams")
constexpr uint32_t DOMAIN_KIN = 0u;       // index = attempt * 64 + slot
constexpr uint32_t DOMAIN_FANO0 = 1u;     // + row of the nucleus; index = sample >> 1
constexpr uint32_t DOMAIN_JITTER = 0x100u; // folded into the key word of jitter_uniform() (Philox2x32-7)
constexpr uint32_t DOMAIN_MC = 0x200u;     // + entry number of the event; index = primary electron
constexpr uint32_t KIN_SLOTS = 64u;

// Philox4x32-10 (Salmon et al. SC'11), the algorithm of rocRAND's default generator, written
// out so that the counter layout (event id, draw index, domain) is ours and the plain-C
// oracle reproduces the identical stream.
// 32 x 32 -> 64 bit product in ONE v_mad_u64_u32: the compiler emits v_mul_hi_u32 + v_mul_lo_u32
// for __umulhi(a, b) and a * b, which makes Philox 1.37x slower on gfx950 (tools/microbench).
__device__ __forceinline__ void mul_hi_lo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  unsigned long long prod, carry;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(prod), "=s"(carry) : "v"(a), "v"(b));
  hi = (uint32_t)(prod >> 32);
  lo = (uint32_t)prod;
}

template <int ROUNDS>
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    mul_hi_lo(c0, 0xD2511F53u, hi0, lo0);
    mul_hi_lo(c2, 0xCD9E8D57u, hi1, lo1);
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// two uniforms in [0,1) for (seed, event, index, domain), Philox4x32-10 (the time-bucket jitter has a generator of
// its own, jitter_uniform() below)
template <int ROUNDS = 10>
__device__ __forceinline__ void rng_pair(uint64_t seed, uint64_t event, uint32_t index, uint32_t domain,
                                         double& ua, double& ub) {
  uint32_t r[4];
  philox4x32<ROUNDS>((uint32_t)event, (uint32_t)(event >> 32), index, domain, (uint32_t)seed,
                     (uint32_t)(seed >> 32), r);
  ua = u53(r[0], r[1]);
  ub = u53(r[2], r[3]);
}

// The time-bucket jitter of a cloud point (simulator.py:108: tb += U[0, 1)), one draw per point and the most numerous
// random numbers of the path, needs ONE uniform: Philox2x32-7 (Salmon et al., the 64-bit member of the family, 7 rounds
// = the fewest that pass BigCrush in the paper; pinned by the Random123 known-answer vectors) -- one 32 x 32 -> 64
// multiply per round where Philox4x32 has two.
//   counter = (event[31:0], event[39:32] << 24 | key), key = tb << 14 | pad (24 bits);
//   key word = seed[31:0] ^ rotl(seed[63:32], 13) ^ DOMAIN_JITTER;  U = u53(out0, out1).
// A pure function of (seed, global event id, time bucket, pad), like every other draw; event ids 2^40 apart share
// their jitter streams (nothing else).  The plain-C oracle and the host expansion of the 8-byte transfer record
// (unpack_host.cpp) compute the identical value.
__host__ __device__ __forceinline__ uint32_t jitter_key_word(uint32_t seed_lo, uint32_t seed_hi) {
  return seed_lo ^ ((seed_hi << 13) | (seed_hi >> 19)) ^ DOMAIN_JITTER;
}
// (`k` = jitter_key_word(seed): a per-launch constant the flush loop keeps in one scalar register)
__device__ __forceinline__ double jitter_uniform_k(uint32_t k, uint32_t ev_lo, uint32_t ev_hi, uint32_t key) {
  uint32_t c0 = ev_lo, c1 = (ev_hi << 24) | key;
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    uint32_t hi, lo;
    mul_hi_lo(c0, 0xD256D193u, hi, lo);
    c0 = hi ^ k ^ c1;
    c1 = lo;
    k += 0x9E3779B9u;
  }
  return u53(c0, c1);
}
__device__ __forceinline__ double jitter_uniform(uint32_t seed_lo, uint32_t seed_hi, uint32_t ev_lo, uint32_t ev_hi, uint32_t key) {
  return jitter_uniform_k(jitter_key_word(seed_lo, seed_hi), ev_lo, ev_hi, key);
}

// Box-Muller, cosine branch, on (1 - ua) in (0, 1]
__device__ __forceinline__ double normal_from(double ua, double ub) {
  return sqrt(-2.0 * log(1.0 - ua)) * cos(TWO_PI * ub);
}

// stopping-power table on the binade grid (include/attpc_engine.h): pure bit arithmetic on
// the f64 -- exponent selects the binade, the top 5 mantissa bits the sub-bin, the remaining
// 47 bits are the interpolation weight.
template <typename TablePtr>
__device__ __forceinline__ double dedx_lookup(TablePtr tab, double ke) {
  constexpr double E_LO = 9.313225746154785e-10;  // 2^-30
  constexpr double E_HI = 16384.0;                // 2^14
  if (!(ke >= E_LO)) return tab[0];
  if (ke >= E_HI) return tab[ATTPC_DEDX_NODES - 1];
  const uint64_t bits = (uint64_t)__double_as_longlong(ke);
  const int e = (int)((bits >> 52) & 0x7ff) - 1023;
  const int j = (int)((bits >> 47) & 31);
  const double t = (double)(bits & ((1ull << 47) - 1)) * (1.0 / 140737488355328.0);
  const int i = (e - ATTPC_DEDX_EMIN) * ATTPC_DEDX_SUB + j;
  const double lo = tab[i], hi = tab[i + 1];
  return lo + t * (hi - lo);
}

// Workgroup barrier with an explicit wait for this wave's own LDS operations.  hipcc places
// `s_waitcnt lgkmcnt(0)` in front of most `s_barrier`s by itself, but not in front of one at the
// head of a loop whose back edge carries a `ds_write` (scatter.hip's batch loop: thread 0's write of
// the next batch index could still be queued when the barrier released the other waves, which then
// read the OLD index and worked on another event than wave 0).  Never seen with one workgroup per CU;
// about once per 20 000 windows when two workgroups share a CU and contend for the LDS.
__device__ __forceinline__ void block_sync() {
#ifndef ATTPC_BARE_BARRIER  // (defined only by tests/test_isa_barriers.py to show what the checker catches)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __syncthreads();
}

// ---- device-side views of the configuration (pointers are device pointers) ----
struct DetDev {
  double length, efield, bfield, density, diffusion, fano_factor, w_value;
  double inv_dv;          // (windows_edge - micromegas_edge) / length   [time buckets / m]
  double dv;              // drift velocity, m / time bucket (parameters.py:172-174)
  double mm_edge;
  int64_t mpgd_gain;
  const int16_t* pad_lut; // folded whole-mm LUT, padded: pad_lut[ix * (lut_n + 1) + iy],
                          // row / column lut_n = -1 (off the pad plane)
  int32_t lut_n, lut_lo;
  int32_t n_species, ode_substeps;
  const double* dedx;     // [n_species][ATTPC_DEDX_NODES]
  double mass[ATTPC_MAX_SPECIES];
  int32_t Z[ATTPC_MAX_SPECIES];
  double longitudinal_diffusion;            // extension, 0 = reference behaviour
  double long_weights[ATTPC_LONG_STEPS];
  int32_t mc_diffusion;                     // extension: per-electron Monte-Carlo transverse diffusion
  int32_t mpgd_gain32;
  double path_step;                         // extension: track sample every path_step metres, 0 = time grid
};

// track sample arena: blocks of ARENA_BLK samples, each sample = (x, y, time bucket, electrons)
constexpr int ARENA_BLK = 128;
constexpr int MAX_BLOCKS_PER_TRACK = (ATTPC_TIME_SAMPLES + ARENA_BLK - 1) / ARENA_BLK;  // 79

struct TrackBuffers {
  double* arena;          // [arena_blocks][ARENA_BLK][4]
  int32_t* block_table;   // [n_tracks][MAX_BLOCKS_PER_TRACK]
  int32_t* counts;        // [n_tracks] samples with >= 1 electron
  int32_t* n_steps;       // [n_tracks] ODE rows recorded (reference track length)
  uint32_t* ctrl;         // [0] next track, [1] next arena block, [2] arena overflow flag, [3] events at the sample
                          // limit (count_status_kernel), [4] tracks ended by the sample cap (path-length step)
  uint32_t arena_blocks;
};

struct Segment {          // one flushed window of one event
  int32_t event;          // chunk-local event
  int32_t count;
  int64_t offset;         // row offset into the chunk's points/labels
  int64_t ev_offset;      // rows of the same event flushed before this window (its place in the event's CSR range)
};

// A single time bucket of an event that holds more keys than the scatter kernel's LDS table is left out
// there, recorded as a LoneBucket and scattered by lone_bucket_kernel (lone.hip) right behind it: inside
// one bucket the key is the pad, so that kernel's table is direct mapped over the 14-bit pad range of the
// key -- no hashing and no capacity limit (the reference's dict has none, simulator.py:93-101).
constexpr int LONE_PADS = 1 << 14;
constexpr int LONE_WORKGROUPS = 32;
struct LoneBucket {
  uint32_t event;   // launch-local event
  uint32_t tb;      // time bucket
};

struct CloudBuffers {
  double* points;         // [capacity][3]
  int64_t* labels;        // [capacity]
  Segment* segments;
  unsigned long long* ctrl;  // [0] row cursor, [1] segment cursor, [2] charge sum, [3] key sum,
                             // [4] failed events, [5] overflow windows retried, [6] out-of-capacity flag,
                             // [7] samples, [29] lone-bucket windows (global-memory table)
  uint32_t* ev_rows;         // [n_events] cloud rows of every event of the launch
  LoneBucket* lone_list;     // [lone_capacity] time buckets left to lone_bucket_kernel, count in ctrl[29]
  uint32_t lone_capacity;
  unsigned long long* lone_chg;  // [LONE_WORKGROUPS][LONE_PADS]      lone_bucket_kernel's tables, all 0 between uses
  uint32_t* lone_mask;           // [LONE_WORKGROUPS][LONE_PADS / 4]
  int64_t capacity;
  int64_t seg_capacity;
};

}  // namespace attpc
