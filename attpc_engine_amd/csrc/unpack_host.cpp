// unpack_host.cpp -- see unpack_host.hpp.  Worker threads write disjoint row ranges of the caller's arrays.
#include "unpack_host.hpp"

#include <algorithm>
#include <thread>
#include <vector>

#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
#define ATTPC_PLAIN_STORES 1  // sanitizer builds: ordinary stores, which the instrumentation sees
#elif !defined(__clang__) && defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace attpc {
namespace {

// streaming (non-temporal) stores of a double / a 64-bit integer: the shipped library is built by hipcc (clang); the
// sanitizer builds of the tests by g++
#if defined(ATTPC_PLAIN_STORES)
inline void nt_store(double v, double* p) { *p = v; }
inline void nt_store(long long v, long long* p) { *p = v; }
#elif defined(__clang__)
inline void nt_store(double v, double* p) { __builtin_nontemporal_store(v, p); }
inline void nt_store(long long v, long long* p) { __builtin_nontemporal_store(v, p); }
#elif defined(__SSE2__)
inline void nt_store(double v, double* p) { _mm_stream_si64(reinterpret_cast<long long*>(p), *reinterpret_cast<long long*>(&v)); }
inline void nt_store(long long v, long long* p) { _mm_stream_si64(p, v); }
#else
inline void nt_store(double v, double* p) { *p = v; }
inline void nt_store(long long v, long long* p) { *p = v; }
#endif

void unpack_slice(const PackedRow* src, int64_t lo, int64_t hi, double* points, int64_t* labels) {
  // streaming (non-temporal) stores: the expanded rows are written once and read by somebody else later, so
  // they should not be read into this core's cache first (half the memory traffic of ordinary stores)
  for (int64_t r = lo; r < hi; ++r) {
    const unsigned long long b = src[r].bits;
    nt_store((double)((b >> PACK_CHARGE_BITS) & ((1ull << PACK_PAD_BITS) - 1)), &points[3 * r]);
    nt_store(src[r].tb, &points[3 * r + 1]);
    nt_store((double)(b & ((1ull << PACK_CHARGE_BITS) - 1)), &points[3 * r + 2]);
    nt_store((long long)(b >> (PACK_CHARGE_BITS + PACK_PAD_BITS)), reinterpret_cast<long long*>(&labels[r]));
  }
}

void unpack_spyral_slice(const SpyralPacked* src, int64_t lo, int64_t hi, SpyralHostTables t, double* rows, int64_t* labels) {
  for (int64_t r = lo; r < hi; ++r) {
    const unsigned long long b = src[r].bits;
    int pad = (int)((b >> SPYRAL_PACK_CHARGE_BITS) & ((1ull << SPYRAL_PACK_PAD_BITS) - 1));
    pad = pad >= t.n_pads ? t.n_pads - 1 : pad;
    const double q = (double)(b & ((1ull << SPYRAL_PACK_CHARGE_BITS) - 1));
    const double tb = src[r].tb;
    double amp = t.r_max * q;  // detector/response.py:55-57 (maximum of the clipped samples), as spyral.hip amplitude()
    amp = amp > 4095.0 ? 4095.0 : amp;
    double* row = rows + 8 * r;
    nt_store(t.centers[2 * pad], &row[0]);
    nt_store(t.centers[2 * pad + 1], &row[1]);
    nt_store((t.window_edge - tb) / (t.window_edge - t.mm_edge) * t.length * 1000.0, &row[2]);  // writer.py:103-105
    nt_store(amp, &row[3]);
    nt_store(src[r].integral, &row[4]);
    nt_store((double)pad, &row[5]);
    nt_store(tb, &row[6]);
    nt_store(t.sizes[pad], &row[7]);
    nt_store((long long)(b >> (SPYRAL_PACK_CHARGE_BITS + SPYRAL_PACK_PAD_BITS)), reinterpret_cast<long long*>(&labels[r]));
  }
}

void unpack8_slice(const unsigned long long* src, int64_t lo, int64_t hi, const int64_t* offsets, int64_t n_events,
                   uint64_t seed, uint64_t first_event, double* points, int64_t* labels) {
  if (lo >= hi) return;
  // the event of row lo: the last e with offsets[e] - offsets[0] <= lo (events without rows are skipped over)
  const int64_t base = offsets[0];
  int64_t e = std::upper_bound(offsets, offsets + n_events + 1, lo + base) - offsets - 1;
  int64_t e_end = offsets[e + 1] - base;
  for (int64_t r = lo; r < hi; ++r) {
    while (r >= e_end) {
      ++e;
      e_end = offsets[e + 1] - base;
    }
    const unsigned long long b = src[r];
    const uint32_t tb = (uint32_t)((b >> PACK8_CHARGE_BITS) & ((1ull << PACK8_TB_BITS) - 1));
    const uint32_t pad = (uint32_t)((b >> (PACK8_CHARGE_BITS + PACK8_TB_BITS)) & ((1ull << PACK_PAD_BITS) - 1));
    const double u = jitter_uniform_host(seed, first_event + (uint64_t)e, (tb << 14) | pad);
    nt_store((double)pad, &points[3 * r]);
    nt_store((double)tb + u, &points[3 * r + 1]);  // simulator.py:108, as the kernel's flush computes it
    nt_store((double)(b & ((1ull << PACK8_CHARGE_BITS) - 1)), &points[3 * r + 2]);
    nt_store((long long)(b >> (PACK8_CHARGE_BITS + PACK8_TB_BITS + PACK_PAD_BITS)), reinterpret_cast<long long*>(&labels[r]));
  }
}

int pick_threads(int n_threads, int64_t n, int64_t rows_per_thread) {
  // default: up to 32 threads (16 kept up with the 16-byte record; with the jitter regenerated per row 32 deliver
  // 5.5e5 instead of 4.8e5 events/s, tools/deliver_sweep.py), never more than half of the hardware threads
  if (n_threads <= 0) n_threads = (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency() / 2u));
  return (int)std::min<int64_t>(n_threads, std::max<int64_t>(1, n / rows_per_thread));
}

}  // namespace

double jitter_uniform_host(uint64_t seed, uint64_t event, uint32_t key24) {
  const uint32_t seed_lo = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32);
  uint32_t c0 = (uint32_t)event, c1 = ((uint32_t)(event >> 32) << 24) | key24;
  uint32_t k = seed_lo ^ ((seed_hi << 13) | (seed_hi >> 19)) ^ 0x100u;
  for (int r = 0; r < 7; ++r) {  // Philox2x32-7
    const uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;
    const uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
    c1 = (uint32_t)p;
    c0 = n0;
    k += 0x9E3779B9u;
  }
  return ((double)(c0 >> 5) * 67108864.0 + (double)(c1 >> 6)) * (1.0 / 9007199254740992.0);
}

void unpack_rows8(const unsigned long long* src, int64_t n, const int64_t* offsets, int64_t n_events, uint64_t seed,
                  uint64_t first_event, double* points, int64_t* labels, int n_threads) {
  n_threads = pick_threads(n_threads, n, 32768);
  if (n_threads <= 1) {
    unpack8_slice(src, 0, n, offsets, n_events, seed, first_event, points, labels);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t per = (n + n_threads - 1) / n_threads;
  for (int t = 1; t < n_threads; ++t)
    pool.emplace_back(unpack8_slice, src, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per), offsets, n_events, seed,
                      first_event, points, labels);
  unpack8_slice(src, 0, std::min<int64_t>(n, per), offsets, n_events, seed, first_event, points, labels);
  for (std::thread& th : pool) th.join();
}

void unpack_rows(const PackedRow* src, int64_t n, double* points, int64_t* labels, int n_threads) {
  n_threads = pick_threads(n_threads, n, 65536);
  if (n_threads <= 1) {
    unpack_slice(src, 0, n, points, labels);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t per = (n + n_threads - 1) / n_threads;
  for (int t = 1; t < n_threads; ++t)
    pool.emplace_back(unpack_slice, src, std::min<int64_t>(n, t * per), std::min<int64_t>(n, (t + 1) * per), points, labels);
  unpack_slice(src, 0, std::min<int64_t>(n, per), points, labels);
  for (std::thread& th : pool) th.join();
}

void unpack_spyral_rows(const SpyralPacked* src, int64_t n, const SpyralHostTables& t, double* rows, int64_t* labels, int n_threads) {
  n_threads = pick_threads(n_threads, n, 32768);
  if (n_threads <= 1) {
    unpack_spyral_slice(src, 0, n, t, rows, labels);
    return;
  }
  std::vector<std::thread> pool;
  const int64_t per = (n + n_threads - 1) / n_threads;
  for (int k = 1; k < n_threads; ++k)
    pool.emplace_back(unpack_spyral_slice, src, std::min<int64_t>(n, k * per), std::min<int64_t>(n, (k + 1) * per), t, rows, labels);
  unpack_spyral_slice(src, 0, std::min<int64_t>(n, per), t, rows, labels);
  for (std::thread& th : pool) th.join();
}

}  // namespace attpc
