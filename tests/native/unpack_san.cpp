// Driver for the sanitizer builds of the host-only expansion code (attpc_engine_amd/csrc/unpack_host.cpp):
// packs synthetic rows the way pack_rows_kernel / spyral_write_kernel do, expands them with 1 ... 16 threads into
// exactly-sized heap arrays (so that AddressSanitizer sees any write past a slice) and checks every value.
// Built and run by tests/test_native_sanitizers.py under -fsanitize=address,undefined and -fsanitize=thread; CPU only.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "unpack_host.hpp"

using namespace attpc;

static unsigned long long lcg(unsigned long long& s) {
  s = s * 6364136223846793005ull + 1442695040888963407ull;
  return s >> 11;
}

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 300000;  // > 4 x 65536: the thread pools really start
  unsigned long long seed = 12345;
  const int n_pads = 10240;
  std::vector<double> centers(2 * n_pads), sizes(n_pads);
  for (int p = 0; p < n_pads; ++p) {
    centers[2 * p] = 0.5 * p - 100.0;
    centers[2 * p + 1] = 300.0 - 0.25 * p;
    sizes[p] = (double)(p % 3);
  }
  std::vector<PackedRow> rows(n);
  std::vector<SpyralPacked> srows(n);
  std::vector<double> want_tb(n), want_q(n);
  std::vector<int> want_pad(n), want_label(n);
  for (long long r = 0; r < n; ++r) {
    const unsigned long long q = lcg(seed) & ((1ull << PACK_CHARGE_BITS) - 1);
    const int pad = (int)(lcg(seed) % n_pads), label = (int)(lcg(seed) % 32);
    const double tb = (double)(lcg(seed) % 512) + (double)(lcg(seed) % 1000) / 1000.0;
    want_tb[r] = tb; want_q[r] = (double)q; want_pad[r] = pad; want_label[r] = label;
    rows[r].tb = tb;
    rows[r].bits = q | ((unsigned long long)pad << PACK_CHARGE_BITS) | ((unsigned long long)label << (PACK_CHARGE_BITS + PACK_PAD_BITS));
    srows[r].tb = tb;
    srows[r].bits = rows[r].bits;
    srows[r].integral = (double)q * 0.5;
  }
  SpyralHostTables t;
  t.centers = centers.data(); t.sizes = sizes.data(); t.n_pads = n_pads;
  t.r_max = 3.2e-5; t.window_edge = 560.0; t.mm_edge = 10.0; t.length = 1.0;
  long long bad = 0;
  for (int threads : {1, 2, 3, 7, 16, 0}) {
    for (long long m : {n, n - 1, (long long)70001, (long long)1, (long long)0}) {  // ragged slices, tiny and empty inputs
      if (m > n || m < 0) continue;
      double* points = (double*)malloc((size_t)(3 * m + 1) * sizeof(double));
      long long* labels = (long long*)malloc((size_t)(m + 1) * sizeof(long long));
      unpack_rows(rows.data(), m, points, (int64_t*)labels, threads);
      for (long long r = 0; r < m; ++r)
        bad += points[3 * r] != (double)want_pad[r] || points[3 * r + 1] != want_tb[r] || points[3 * r + 2] != want_q[r] ||
               labels[r] != want_label[r];
      free(points);
      free(labels);
      double* out = (double*)malloc((size_t)(8 * m + 1) * sizeof(double));
      labels = (long long*)malloc((size_t)(m + 1) * sizeof(long long));
      unpack_spyral_rows(srows.data(), m, t, out, (int64_t*)labels, threads);
      for (long long r = 0; r < m; ++r) {
        const double* row = out + 8 * r;
        double amp = t.r_max * want_q[r];
        amp = amp > 4095.0 ? 4095.0 : amp;
        const double z = (560.0 - want_tb[r]) / 550.0 * 1.0 * 1000.0;
        bad += row[0] != centers[2 * want_pad[r]] || row[1] != centers[2 * want_pad[r] + 1] || row[2] != z || row[3] != amp ||
               row[4] != want_q[r] * 0.5 || row[5] != (double)want_pad[r] || row[6] != want_tb[r] || row[7] != sizes[want_pad[r]] ||
               labels[r] != want_label[r];
      }
      free(out);
      free(labels);
    }
  }
  {  // 8-byte records: the host regenerates the jitter; events of ragged sizes, some empty
    const long long n_events = 997;
    std::vector<int64_t> offsets(n_events + 1, 0);
    for (long long e = 0; e < n_events; ++e) {
      long long c = (long long)(lcg(seed) % (2 * n / n_events));
      if (e % 50 == 3) c = 0;
      offsets[e + 1] = offsets[e] + c;
    }
    const long long m = offsets[n_events] < n ? offsets[n_events] : n;
    for (long long e = 0; e <= n_events; ++e) offsets[e] = offsets[e] > m ? m : offsets[e];
    std::vector<unsigned long long> rows8(m);
    std::vector<long long> ev_of(m);
    for (long long e = 0; e < n_events; ++e)
      for (long long r = offsets[e]; r < offsets[e + 1]; ++r) ev_of[r] = e;
    for (long long r = 0; r < m; ++r) {
      const unsigned long long q = lcg(seed) & ((1ull << PACK8_CHARGE_BITS) - 1), tb = lcg(seed) % 512;
      rows8[r] = q | (tb << PACK8_CHARGE_BITS) | ((unsigned long long)want_pad[r] << (PACK8_CHARGE_BITS + PACK8_TB_BITS)) |
                 ((unsigned long long)want_label[r] << (PACK8_CHARGE_BITS + PACK8_TB_BITS + PACK_PAD_BITS));
    }
    for (int threads : {1, 4, 16}) {
      double* points = (double*)malloc((size_t)(3 * m + 1) * sizeof(double));
      long long* labels = (long long*)malloc((size_t)(m + 1) * sizeof(long long));
      unpack_rows8(rows8.data(), m, offsets.data(), n_events, 77ull, 1000000ull, points, (int64_t*)labels, threads);
      for (long long r = 0; r < m; ++r) {
        const unsigned long long b = rows8[r];
        const unsigned tb = (unsigned)((b >> PACK8_CHARGE_BITS) & 511u);
        const double u = jitter_uniform_host(77ull, 1000000ull + (unsigned long long)ev_of[r], (tb << 14) | (unsigned)want_pad[r]);
        bad += points[3 * r] != (double)want_pad[r] || points[3 * r + 1] != (double)tb + u ||
               points[3 * r + 2] != (double)(b & ((1ull << PACK8_CHARGE_BITS) - 1)) || labels[r] != want_label[r];
      }
      free(points);
      free(labels);
    }
  }
  printf("unpack_san: %lld rows, mismatches %lld\n", n, bad);
  return bad ? 1 : 0;
}
