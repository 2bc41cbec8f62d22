// spyral.hip -- GET response, ADC threshold and Spyral row conversion on the device
// ("next" row 1 of SURVEY.md 8f), run on the event-ordered cloud of a chunk before D2H.
//
// Restates (reference src/attpc_engine/detector/): response.py:35-57 (apply_response: every one of
// the 512 response samples scaled by the electrons and clipped at 4095; amplitude = max,
// integral = sum), writer.py:61-112 (convert_to_spyral row layout), writer.py:232-234 (rows with
// amplitude <= adc_threshold are dropped).
//
// The clipped sum has a closed form: with the response samples sorted descending (r_(1) >= r_(2) ...)
// and k = #{i : r_i q > 4095},  integral = 4095 k + q (total - sum of the k largest).  k comes from a
// binary search; sum order differs from the reference's sequential loop, i.e. last-bit differences.
//
// One workgroup per event; the kept rows are written in ascending z (the z-sort of writer.py:236-238 on
// the device: counting sort over the integer time bucket + rank inside the bucket).
// Bound: HBM (32 B read per cloud row, twice; 72 B written per kept row).
#include "tracks_args.hpp"

namespace attpc {

constexpr int SP_THREADS = 256;

__device__ __forceinline__ double amplitude(const SpyralDev& sp, double q) {
  const double a = sp.r_max * q;
  return a > 4095.0 ? 4095.0 : a;
}

__device__ __forceinline__ double clipped_integral(const SpyralDev& sp, double q) {
  // k = number of samples with r*q > 4095  (sorted_desc is descending)
  int lo = 0, hi = ATTPC_NUM_TB;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (sp.sorted_desc[mid] * q > 4095.0) lo = mid + 1; else hi = mid;
  }
  return 4095.0 * (double)lo + q * (sp.total - sp.prefix[lo]);
}

__global__ __launch_bounds__(SP_THREADS) void spyral_count_kernel(SpyralDev sp, const int64_t* __restrict__ event_start,
                                                                   const double* __restrict__ points,
                                                                   uint32_t* __restrict__ kept) {
  __shared__ int total;
  const uint32_t e = blockIdx.x;
  if (threadIdx.x == 0) total = 0;
  block_sync();
  const int64_t lo = event_start[e], hi = event_start[e + 1];
  int mine = 0;
  for (int64_t r = lo + threadIdx.x; r < hi; r += SP_THREADS) mine += amplitude(sp, points[3 * r + 2]) > sp.threshold ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&total, mine);
  block_sync();
  if (threadIdx.x == 0) kept[e] = (uint32_t)total;
}

// Kept rows of one event, converted and written in ascending z (writer.py:232-238: threshold, then
// argsort of column 2).  z falls with the time bucket, and the integer time bucket is a 9-bit key, so the
// sort is a counting sort over the 512 buckets (highest bucket first) followed by a rank inside each
// bucket on the jittered time bucket itself (larger first; equal values keep their cloud order -- the
// reference's argsort is unstable, so ties have no defined order there).  The bucket-grouped
// (row, time bucket) list lives in a global scratch range of the event's own cloud rows.
__global__ __launch_bounds__(SP_THREADS) void spyral_write_kernel(SpyralDev sp, const int64_t* __restrict__ event_start,
                                                                   const int64_t* __restrict__ kept_start,
                                                                   const double* __restrict__ points,
                                                                   const int64_t* __restrict__ labels,
                                                                   double* __restrict__ rows,
                                                                   int64_t* __restrict__ out_labels,
                                                                   uint32_t* __restrict__ sort_idx,
                                                                   double* __restrict__ sort_key) {
  __shared__ uint32_t bin_count[ATTPC_NUM_TB];
  __shared__ uint32_t bin_start[ATTPC_NUM_TB + 1];
  __shared__ uint32_t bin_cursor[ATTPC_NUM_TB];
  __shared__ uint32_t wave_total[SP_THREADS / 64];
  const uint32_t e = blockIdx.x;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t lo = event_start[e], hi = event_start[e + 1];
  const int64_t out0 = kept_start[e];
  const uint32_t n_kept = (uint32_t)(kept_start[e + 1] - out0);
  if (n_kept == 0u) return;  // uniform
  for (int b = t; b < ATTPC_NUM_TB; b += SP_THREADS) bin_count[b] = 0u;
  block_sync();
  auto bin_of = [](double tb) -> int {  // ascending z = descending time bucket
    int b = (int)tb;
    b = b < 0 ? 0 : (b > ATTPC_NUM_TB - 1 ? ATTPC_NUM_TB - 1 : b);
    return ATTPC_NUM_TB - 1 - b;
  };
  for (int64_t r = lo + t; r < hi; r += SP_THREADS)
    if (amplitude(sp, points[3 * r + 2]) > sp.threshold) atomicAdd(&bin_count[bin_of(points[3 * r + 1])], 1u);
  block_sync();
  {  // exclusive prefix over the 512 buckets: two per thread, wave scan, wave offsets
    static_assert(ATTPC_NUM_TB == 2 * SP_THREADS, "two buckets per thread");
    const uint32_t c0 = bin_count[2 * t], c1 = bin_count[2 * t + 1];
    uint32_t incl = c0 + c1;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      incl += lane >= off ? up : 0u;
    }
    if (lane == 63) wave_total[wave] = incl;
    block_sync();
    uint32_t excl = incl - (c0 + c1);
    for (int w = 0; w < wave; ++w) excl += wave_total[w];
    bin_start[2 * t] = excl;
    bin_start[2 * t + 1] = excl + c0;
    bin_cursor[2 * t] = excl;
    bin_cursor[2 * t + 1] = excl + c0;
    if (t == SP_THREADS - 1) bin_start[ATTPC_NUM_TB] = excl + c0 + c1;
  }
  block_sync();
  for (int64_t r = lo + t; r < hi; r += SP_THREADS) {
    const double tb = points[3 * r + 1];
    if (amplitude(sp, points[3 * r + 2]) > sp.threshold) {
      const uint32_t p = atomicAdd(&bin_cursor[bin_of(tb)], 1u);
      sort_idx[lo + p] = (uint32_t)(r - lo);
      sort_key[lo + p] = tb;
    }
  }
  __threadfence();
  block_sync();
  for (uint32_t p = (uint32_t)t; p < n_kept; p += SP_THREADS) {
    const uint32_t ri = sort_idx[lo + p];
    const double tb = sort_key[lo + p];
    const int b = bin_of(tb);
    const uint32_t b_lo = bin_start[b], b_hi = bin_start[b + 1];
    uint32_t rank = 0u;
    for (uint32_t q = b_lo; q < b_hi; ++q) {
      const double other = sort_key[lo + q];
      rank += (other > tb || (other == tb && sort_idx[lo + q] < ri)) ? 1u : 0u;
    }
    const int64_t r = lo + ri;
    const int64_t o = out0 + b_lo + rank;
    const double padf = points[3 * r], q_el = points[3 * r + 2];
    int pad = (int)padf;
    pad = pad < 0 ? 0 : (pad >= sp.n_pads ? sp.n_pads - 1 : pad);
    double* row = rows + 8 * o;
    row[0] = sp.pad_centers[2 * pad];
    row[1] = sp.pad_centers[2 * pad + 1];
    row[2] = (sp.window_edge - tb) / (sp.window_edge - sp.mm_edge) * sp.length * 1000.0;  // writer.py:103-105
    row[3] = amplitude(sp, q_el);
    row[4] = clipped_integral(sp, q_el);
    row[5] = padf;
    row[6] = tb;
    row[7] = sp.pad_sizes[pad];
    out_labels[o] = labels[r];
  }
}

void launch_spyral_count(hipStream_t s, const SpyralDev& sp, uint32_t n_events, const int64_t* event_start,
                         const double* points, uint32_t* kept) {
  hipLaunchKernelGGL(spyral_count_kernel, dim3(n_events), dim3(SP_THREADS), 0, s, sp, event_start, points, kept);
}
void launch_spyral_write(hipStream_t s, const SpyralDev& sp, uint32_t n_events, const int64_t* event_start,
                         const int64_t* kept_start, const double* points, const int64_t* labels, double* rows,
                         int64_t* out_labels, uint32_t* sort_idx, double* sort_key) {
  hipLaunchKernelGGL(spyral_write_kernel, dim3(n_events), dim3(SP_THREADS), 0, s, sp, event_start, kept_start, points,
                     labels, rows, out_labels, sort_idx, sort_key);
}

}  // namespace attpc
