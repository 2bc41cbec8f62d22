// unpack_host.cpp -- see unpack_host.hpp.  Worker threads write disjoint row ranges of the caller's arrays.
#include "unpack_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#if defined(__linux__)
#include <sched.h>
#endif

#if defined(__SANITIZE_ADDRESS__) || defined(__SANITIZE_THREAD__)
#define ATTPC_PLAIN_STORES 1  // sanitizer builds: ordinary stores, which the instrumentation sees
#elif !defined(__clang__) && defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#define ATTPC_HAVE_AVX2_PATH 1  // a second body of unpack8 for CPUs with AVX2, chosen at run time
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
inline void nt_store(double v, double* p) {
  long long bits;
  std::memcpy(&bits, &v, sizeof(bits));
  _mm_stream_si64(reinterpret_cast<long long*>(p), bits);
}
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

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
int64_t unpack_spyral_avx2(const SpyralPacked* src, int64_t r, int64_t hi, const SpyralHostTables& t, double* rows, int64_t* labels);
bool cpu_has_avx2();
#endif

void unpack_spyral_slice(const SpyralPacked* src, int64_t lo, int64_t hi, SpyralHostTables t, double* rows, int64_t* labels) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  // four rows at a time where the CPU has AVX2 and the rows can leave as 16-byte streaming stores
  if (cpu_has_avx2() && (reinterpret_cast<uintptr_t>(rows) & 15u) == 0) lo = unpack_spyral_avx2(src, lo, hi, t, rows, labels);
#endif
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
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(ATTPC_PLAIN_STORES)
  _mm_sfence();  // streaming stores are weakly ordered: made visible before the thread reports back
#endif
}

#ifdef ATTPC_HAVE_AVX2_PATH
__attribute__((target("avx2"))) inline __m256d to_f64(__m256i x) {  // whole numbers below 2^52 in 64-bit lanes -> f64, exactly
  const __m256i magic_i = _mm256_set1_epi64x(0x4330000000000000ll);  // 2^52 as bits
  const __m256d magic_d = _mm256_set1_pd(4503599627370496.0);         // 2^52
  return _mm256_sub_pd(_mm256_castsi256_pd(_mm256_or_si256(x, magic_i)), magic_d);
}

// Spyral rows [r, hi), four per step, `rows` 16-byte aligned (numpy's arrays are): the 24-byte records of four rows
// are taken apart into columns, the columns computed with the operations of unpack_spyral_slice() in its order
// (bit-identical: tests/test_host_api.py), and every row leaves as four 16-byte streaming stores.  Returns the first row
// it did not do.
__attribute__((target("avx2"))) int64_t unpack_spyral_avx2(const SpyralPacked* src, int64_t r, int64_t hi, const SpyralHostTables& t,
                                                           double* rows, int64_t* labels) {
  const __m256d r_max = _mm256_set1_pd(t.r_max), top = _mm256_set1_pd(4095.0), edge = _mm256_set1_pd(t.window_edge);
  const __m256d span = _mm256_set1_pd(t.window_edge - t.mm_edge), length = _mm256_set1_pd(t.length), thousand = _mm256_set1_pd(1000.0);
  const __m256i m14 = _mm256_set1_epi64x(0x3fff), m45 = _mm256_set1_epi64x((1ll << SPYRAL_PACK_CHARGE_BITS) - 1);
  const __m256i last_pad = _mm256_set1_epi64x((long long)(t.n_pads - 1));
  for (; r + 4 <= hi; r += 4) {
    const double* in = reinterpret_cast<const double*>(src + r);
    const __m256d a = _mm256_loadu_pd(in), b = _mm256_loadu_pd(in + 4), c = _mm256_loadu_pd(in + 8);
    // a = tb0 bits0 int0 tb1 | b = bits1 int1 tb2 bits2 | c = int2 tb3 bits3 int3
    const __m256d TB = _mm256_permute4x64_pd(_mm256_blend_pd(_mm256_blend_pd(a, b, 0x4), c, 0x2), 0x6C);     // tb0 tb3 tb2 tb1 -> 0 3 2 1
    const __m256d BT = _mm256_permute4x64_pd(_mm256_blend_pd(_mm256_blend_pd(a, b, 0x9), c, 0x4), 0xB1);     // bits1 bits0 bits3 bits2 -> 1 0 3 2
    const __m256d INT = _mm256_permute4x64_pd(_mm256_blend_pd(_mm256_blend_pd(a, b, 0x2), c, 0x9), 0xC6);    // int2 int1 int0 int3 -> 2 1 0 3
    const __m256i bits = _mm256_castpd_si256(BT);
    const __m256i pad = _mm256_min_epu32(_mm256_and_si256(_mm256_srli_epi64(bits, SPYRAL_PACK_CHARGE_BITS), m14), last_pad);
    const __m256d q = to_f64(_mm256_and_si256(bits, m45));
    const __m256d amp = _mm256_min_pd(_mm256_mul_pd(r_max, q), top);  // detector/response.py:55-57, clipped at 4095
    const __m256d z = _mm256_mul_pd(_mm256_mul_pd(_mm256_div_pd(_mm256_sub_pd(edge, TB), span), length), thousand);  // writer.py:103-105
    const __m256d padf = to_f64(pad);
    alignas(32) long long pads[4], labs[4];
    _mm256_store_si256(reinterpret_cast<__m256i*>(pads), pad);
    _mm256_store_si256(reinterpret_cast<__m256i*>(labs), _mm256_srli_epi64(bits, SPYRAL_PACK_CHARGE_BITS + SPYRAL_PACK_PAD_BITS));
    const __m256d size = _mm256_set_pd(t.sizes[pads[3]], t.sizes[pads[2]], t.sizes[pads[1]], t.sizes[pads[0]]);
    const __m256d za02 = _mm256_unpacklo_pd(z, amp), za13 = _mm256_unpackhi_pd(z, amp);        // z0 a0 | z2 a2,  z1 a1 | z3 a3
    const __m256d ip02 = _mm256_unpacklo_pd(INT, padf), ip13 = _mm256_unpackhi_pd(INT, padf);
    const __m256d ts02 = _mm256_unpacklo_pd(TB, size), ts13 = _mm256_unpackhi_pd(TB, size);
    double* out = rows + 8 * r;
#if defined(ATTPC_PLAIN_STORES)
#define ATTPC_ST128(p, v) _mm_storeu_pd(p, v)
#else
#define ATTPC_ST128(p, v) _mm_stream_pd(p, v)
#endif
    ATTPC_ST128(out + 0, _mm_loadu_pd(t.centers + 2 * pads[0]));
    ATTPC_ST128(out + 2, _mm256_castpd256_pd128(za02));
    ATTPC_ST128(out + 4, _mm256_castpd256_pd128(ip02));
    ATTPC_ST128(out + 6, _mm256_castpd256_pd128(ts02));
    ATTPC_ST128(out + 8, _mm_loadu_pd(t.centers + 2 * pads[1]));
    ATTPC_ST128(out + 10, _mm256_castpd256_pd128(za13));
    ATTPC_ST128(out + 12, _mm256_castpd256_pd128(ip13));
    ATTPC_ST128(out + 14, _mm256_castpd256_pd128(ts13));
    ATTPC_ST128(out + 16, _mm_loadu_pd(t.centers + 2 * pads[2]));
    ATTPC_ST128(out + 18, _mm256_extractf128_pd(za02, 1));
    ATTPC_ST128(out + 20, _mm256_extractf128_pd(ip02, 1));
    ATTPC_ST128(out + 22, _mm256_extractf128_pd(ts02, 1));
    ATTPC_ST128(out + 24, _mm_loadu_pd(t.centers + 2 * pads[3]));
    ATTPC_ST128(out + 26, _mm256_extractf128_pd(za13, 1));
    ATTPC_ST128(out + 28, _mm256_extractf128_pd(ip13, 1));
    ATTPC_ST128(out + 30, _mm256_extractf128_pd(ts13, 1));
#undef ATTPC_ST128
    for (int k = 0; k < 4; ++k) nt_store(labs[k], reinterpret_cast<long long*>(&labels[r + k]));
  }
  return r;
}
#endif

// one row of the 8-byte record, event `ev`
inline void unpack8_row(unsigned long long b, uint64_t seed, uint64_t ev, int64_t r, double* points, int64_t* labels) {
  const uint32_t tb = (uint32_t)((b >> PACK8_CHARGE_BITS) & ((1ull << PACK8_TB_BITS) - 1));
  const uint32_t pad = (uint32_t)((b >> (PACK8_CHARGE_BITS + PACK8_TB_BITS)) & ((1ull << PACK_PAD_BITS) - 1));
  const double u = jitter_uniform_host(seed, ev, (tb << 14) | pad);
  nt_store((double)pad, &points[3 * r]);
  nt_store((double)tb + u, &points[3 * r + 1]);  // simulator.py:108, as the kernel's flush computes it
  nt_store((double)(b & ((1ull << PACK8_CHARGE_BITS) - 1)), &points[3 * r + 2]);
  nt_store((long long)(b >> (PACK8_CHARGE_BITS + PACK8_TB_BITS + PACK_PAD_BITS)), reinterpret_cast<long long*>(&labels[r]));
}

#ifdef ATTPC_HAVE_AVX2_PATH
// Rows [r, r_end) of ONE event, eight rows per step; row r of `points` sits on a 32-byte boundary (so do rows r + 4 k: a
// row is 24 bytes), STREAM_LABELS says whether labels + r does too (numpy's arrays start 16 bytes after a 32-byte
// boundary: then both do from r = 2 on): the
// seven Philox2x32 rounds on eight 32-bit lanes (two widening multiplies per round), whole numbers to f64 through the
// 2^52 trick (exact below 2^52), the three columns transposed into the (n, 3) layout, streaming stores of 32 bytes.
// The same operations on the same values as unpack8_row(): bit-identical (tests/test_host_api.py).  Returns the first
// row it did not do.  (The expansion threads of a GPU box share ~16 cores: the scalar loop's ~60 instructions a row were
// what bounded the delivered rate, not the memory bandwidth -- tools/unpack_rate.py.)
template <bool STREAM_LABELS>
__attribute__((target("avx2"))) int64_t unpack8_event_avx2(const unsigned long long* src, int64_t r, int64_t r_end, uint64_t seed,
                                                           uint64_t ev, double* points, int64_t* labels) {
  const uint32_t seed_lo = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32);
  const uint32_t k0 = seed_lo ^ ((seed_hi << 13) | (seed_hi >> 19)) ^ 0x100u;
  // round 1 starts from the same c0 = event[31:0] in every row: its product is a constant of the event
  const uint64_t p1 = (uint64_t)0xD256D193u * (uint64_t)(uint32_t)ev;
  const __m256i hi1_k0 = _mm256_set1_epi32((int)((uint32_t)(p1 >> 32) ^ k0));
  const __m256i lo1 = _mm256_set1_epi32((int)(uint32_t)p1);
  const __m256i ev_hi24 = _mm256_set1_epi32((int)((uint32_t)(ev >> 32) << 24));
  const __m256i mult = _mm256_set1_epi64x(0xD256D193ll);
  __m256i key[6];
  for (int k = 0; k < 6; ++k) key[k] = _mm256_set1_epi32((int)(k0 + (uint32_t)(k + 1) * 0x9E3779B9u));
  const __m256i m9 = _mm256_set1_epi64x(0x1ff), m14 = _mm256_set1_epi64x(0x3fff), m36 = _mm256_set1_epi64x((1ll << 36) - 1);
  const __m256i m32 = _mm256_set1_epi64x(0xffffffffll);
  const __m256d two26 = _mm256_set1_pd(67108864.0), inv53 = _mm256_set1_pd(1.0 / 9007199254740992.0);
  for (; r + 8 <= r_end; r += 8) {
    const __m256i b0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + r));      // rows r .. r + 3
    const __m256i b1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + r + 4));  // rows r + 4 .. r + 7
    const __m256i tb0 = _mm256_and_si256(_mm256_srli_epi64(b0, PACK8_CHARGE_BITS), m9);
    const __m256i tb1 = _mm256_and_si256(_mm256_srli_epi64(b1, PACK8_CHARGE_BITS), m9);
    const __m256i pad0 = _mm256_and_si256(_mm256_srli_epi64(b0, PACK8_CHARGE_BITS + PACK8_TB_BITS), m14);
    const __m256i pad1 = _mm256_and_si256(_mm256_srli_epi64(b1, PACK8_CHARGE_BITS + PACK8_TB_BITS), m14);
    const __m256i key0 = _mm256_or_si256(_mm256_slli_epi64(tb0, 14), pad0);
    const __m256i key1 = _mm256_or_si256(_mm256_slli_epi64(tb1, 14), pad1);
    // 32-bit lane 2 i = row r + i, lane 2 i + 1 = row r + 4 + i
    __m256i c1 = _mm256_or_si256(_mm256_or_si256(key0, _mm256_slli_epi64(key1, 32)), ev_hi24);
    __m256i c0 = _mm256_xor_si256(hi1_k0, c1);  // round 1
    c1 = lo1;
    for (int k = 0; k < 6; ++k) {  // rounds 2 .. 7
      const __m256i pe = _mm256_mul_epu32(c0, mult);
      const __m256i po = _mm256_mul_epu32(_mm256_srli_epi64(c0, 32), mult);
      const __m256i hi = _mm256_blend_epi32(_mm256_srli_epi64(pe, 32), po, 0xAA);
      const __m256i lo = _mm256_blend_epi32(pe, _mm256_slli_epi64(po, 32), 0xAA);
      c0 = _mm256_xor_si256(_mm256_xor_si256(hi, key[k]), c1);
      c1 = lo;
    }
    for (int half = 0; half < 2; ++half) {
      const __m256i b = half ? b1 : b0;
      const __m256i x0 = half ? _mm256_srli_epi64(c0, 32) : _mm256_and_si256(c0, m32);
      const __m256i x1 = half ? _mm256_srli_epi64(c1, 32) : _mm256_and_si256(c1, m32);
      // ((double)(c0 >> 5) * 2^26 + (double)(c1 >> 6)) * 2^-53: every step exact, as in jitter_uniform_host()
      const __m256d u = _mm256_mul_pd(
          _mm256_add_pd(_mm256_mul_pd(to_f64(_mm256_srli_epi64(x0, 5)), two26), to_f64(_mm256_srli_epi64(x1, 6))), inv53);
      const __m256d P = to_f64(half ? pad1 : pad0);
      const __m256d T = _mm256_add_pd(to_f64(half ? tb1 : tb0), u);
      const __m256d Q = to_f64(_mm256_and_si256(b, m36));
      // (P, T, Q) columns of four rows -> twelve consecutive doubles p0 t0 q0 p1 | t1 q1 p2 t2 | q2 p3 t3 q3
      const __m256d pt = _mm256_unpacklo_pd(P, T);       // p0 t0 | p2 t2
      const __m256d qp = _mm256_shuffle_pd(Q, P, 0xA);   // q0 p1 | q2 p3
      const __m256d tq = _mm256_unpackhi_pd(T, Q);       // t1 q1 | t3 q3
      double* out = points + 3 * (r + 4 * half);
      long long* lab = reinterpret_cast<long long*>(labels + r + 4 * half);
      const __m256d o0 = _mm256_permute2f128_pd(pt, qp, 0x20), o1 = _mm256_blend_pd(tq, pt, 0xC), o2 = _mm256_permute2f128_pd(qp, tq, 0x31);
      const __m256i l = _mm256_srli_epi64(b, PACK8_CHARGE_BITS + PACK8_TB_BITS + PACK_PAD_BITS);
#if defined(ATTPC_PLAIN_STORES)
      _mm256_storeu_pd(out, o0);
      _mm256_storeu_pd(out + 4, o1);
      _mm256_storeu_pd(out + 8, o2);
      _mm256_storeu_si256(reinterpret_cast<__m256i*>(lab), l);
#else
      _mm256_stream_pd(out, o0);
      _mm256_stream_pd(out + 4, o1);
      _mm256_stream_pd(out + 8, o2);
      if (STREAM_LABELS) _mm256_stream_si256(reinterpret_cast<__m256i*>(lab), l);
      else _mm256_storeu_si256(reinterpret_cast<__m256i*>(lab), l);
#endif
    }
  }
  return r;
}

bool cpu_has_avx2() {
  static const bool has = __builtin_cpu_supports("avx2");
  return has;
}
#endif

void unpack8_slice(const unsigned long long* src, int64_t lo, int64_t hi, const int64_t* offsets, int64_t n_events,
                   uint64_t seed, uint64_t first_event, double* points, int64_t* labels) {
  if (lo >= hi) return;
  // the event of row lo: the last e with offsets[e] - offsets[0] <= lo (events without rows are skipped over)
  const int64_t base = offsets[0];
  int64_t e = std::upper_bound(offsets, offsets + n_events + 1, lo + base) - offsets - 1;
#ifdef ATTPC_HAVE_AVX2_PATH
  // rows r = phase (mod 4) of `points` start on a 32-byte boundary (a row is 24 bytes, the array 8-byte aligned)
  const bool wide = cpu_has_avx2() && (reinterpret_cast<uintptr_t>(points) & 7u) == 0 && (reinterpret_cast<uintptr_t>(labels) & 7u) == 0;
  const int64_t phase = (int64_t)((reinterpret_cast<uintptr_t>(points) >> 3) & 3u);
  const bool stream_labels = ((reinterpret_cast<uintptr_t>(labels) + 8u * (uintptr_t)phase) & 31u) == 0;
#endif
  for (int64_t r = lo; r < hi;) {
    while (r >= offsets[e + 1] - base) ++e;
    const int64_t e_end = std::min<int64_t>(hi, offsets[e + 1] - base);  // this event's rows inside the slice
    const uint64_t ev = first_event + (uint64_t)e;
#ifdef ATTPC_HAVE_AVX2_PATH
    if (wide) {
      for (; r < e_end && (r & 3) != phase; ++r) unpack8_row(src[r], seed, ev, r, points, labels);  // up to a 32-byte boundary
      r = stream_labels ? unpack8_event_avx2<true>(src, r, e_end, seed, ev, points, labels)
                        : unpack8_event_avx2<false>(src, r, e_end, seed, ev, points, labels);
    }
#endif
    for (; r < e_end; ++r) unpack8_row(src[r], seed, ev, r, points, labels);
  }
#if defined(ATTPC_HAVE_AVX2_PATH) && !defined(ATTPC_PLAIN_STORES)
  _mm_sfence();  // streaming stores are weakly ordered: made visible before the thread reports back
#endif
}

// CPUs this process may really use: the hardware threads, cut down to the scheduler affinity mask and to the CPU
// quota of its control group (a GPU box hands a job 16 of its 256 hardware threads through cpu.max: twice as many
// expansion threads as that only take turns)
int usable_cpus() {
  static const int cached = [] {
    int n = (int)std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    long long quota = -1, period = -1;
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
      char word[32] = {0};
      if (std::fscanf(f, "%31s %lld", word, &period) == 2 && std::strcmp(word, "max") != 0) quota = std::atoll(word);
      std::fclose(f);
    } else {  // cgroup v1
      if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (std::fscanf(q, "%lld", &quota) != 1) quota = -1;
        std::fclose(q);
      }
      if (FILE* q = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
        if (std::fscanf(q, "%lld", &period) != 1) period = -1;
        std::fclose(q);
      }
    }
    if (quota > 0 && period > 0) n = std::min<long long>(n, std::max<long long>(1, (quota + period - 1) / period));
#endif
    return n;
  }();
  return cached;
}

int pick_threads(int n_threads, int64_t n, int64_t rows_per_thread) {
  // default: up to 32 threads, never more than the CPUs the process may use (half of them on a machine of its own:
  // the other hardware thread of a core adds nothing to a loop of streaming stores); tools/deliver_sweep.py
  if (n_threads <= 0) {
    const int cpus = usable_cpus();
    n_threads = std::min(32, std::max(1, cpus >= 64 ? cpus / 2 : cpus));
  }
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
