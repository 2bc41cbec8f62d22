// unpack_host.hpp -- host-only expansion of the compact transfer records (include/attpc_engine.h:
// attpc_unpack_rows, attpc_unpack_spyral_rows).  Plain C++17, no HIP: compiled into libattpc_hip.so with abi.hip and,
// by tests/test_native_sanitizers.py, on its own under -fsanitize=address,undefined and -fsanitize=thread.
#pragma once
#include <stdint.h>

namespace attpc {

// A cloud row in the reference's dtypes is 3 f64 + i64 = 32 bytes, but it holds 14 bits of pad, 5 of label, an integer
// charge and one real number (the jittered time bucket): 16 bytes carry it losslessly --
//   word 0 = the f64 time bucket + jitter as it is, word 1 = charge (45 bits) | pad << 45 (14) | label << 59 (5).
struct PackedRow {
  double tb;
  unsigned long long bits;
};
constexpr int PACK_CHARGE_BITS = 45, PACK_PAD_BITS = 14;

// The 8-byte record of a cloud row: the jitter is a pure function of (seed, global event id, time bucket, pad)
// (csrc/common.hpp jitter_uniform: Philox2x32-7), so it need not cross the link at all --
//   electrons (36 bits) | time bucket << 36 (9) | pad << 45 (14) | label << 59 (5),
// and the host rebuilds column 1 as (double)tb + jitter_uniform(...) with the kernel's own operations: bit-identical.
constexpr int PACK8_CHARGE_BITS = 36, PACK8_TB_BITS = 9;
// the same function as the device's jitter_uniform() (and the oracle's orc_jitter_uniform)
double jitter_uniform_host(uint64_t seed, uint64_t event, uint32_t key24);

// Compact transfer record of a Spyral row (24 instead of 72 bytes; the host rebuilds x, y, z, amplitude and pad
// scale from it): time bucket + jitter, electrons | pad << 45 | label << 59, clipped integral.
struct SpyralPacked {
  double tb;
  unsigned long long bits;
  double integral;
};
constexpr int SPYRAL_PACK_CHARGE_BITS = 45, SPYRAL_PACK_PAD_BITS = 14;

struct SpyralHostTables {  // what convert_to_spyral (writer.py:61-112) needs beside the record
  const double* centers;    // [n_pads, 2]
  const double* sizes;      // [n_pads]
  int32_t n_pads;
  double r_max, window_edge, mm_edge, length;
};

// n_threads <= 0: min(32, the CPUs the process may use -- affinity mask and control-group quota honoured; half of them on
// a machine with 64 or more); short inputs use fewer threads (one per 65 536 / 32 768 rows)
void unpack_rows(const PackedRow* src, int64_t n, double* points, int64_t* labels, int n_threads);
void unpack_spyral_rows(const SpyralPacked* src, int64_t n, const SpyralHostTables& t, double* rows, int64_t* labels,
                        int n_threads);
// rows of events first_event .. first_event + n_events - 1 in event order; offsets [n_events + 1] = CSR offsets of the
// events inside src (offsets[n_events] - offsets[0] == n; only differences are used)
void unpack_rows8(const unsigned long long* src, int64_t n, const int64_t* offsets, int64_t n_events, uint64_t seed,
                  uint64_t first_event, double* points, int64_t* labels, int n_threads);

}  // namespace attpc
