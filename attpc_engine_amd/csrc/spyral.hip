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
// One workgroup per event; the kept rows are compacted in cloud order with a block prefix sum.
// Bound: HBM (32 B read per cloud row, 72 B written per kept row), no reuse.
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
                                                                   int32_t* __restrict__ kept) {
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
  if (threadIdx.x == 0) kept[e] = total;
}

__global__ __launch_bounds__(SP_THREADS) void spyral_write_kernel(SpyralDev sp, const int64_t* __restrict__ event_start,
                                                                   const int64_t* __restrict__ kept_start,
                                                                   const double* __restrict__ points,
                                                                   const int64_t* __restrict__ labels,
                                                                   double* __restrict__ rows,
                                                                   int64_t* __restrict__ out_labels) {
  __shared__ int wave_count[SP_THREADS / 64];
  __shared__ int running;
  const uint32_t e = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) running = 0;
  block_sync();
  const int64_t lo = event_start[e], hi = event_start[e + 1];
  const int64_t out0 = kept_start[e];
  for (int64_t r0 = lo; r0 < hi; r0 += SP_THREADS) {
    const int64_t r = r0 + threadIdx.x;
    double padf = 0.0, tb = 0.0, q = 0.0, amp = 0.0;
    bool keep = false;
    if (r < hi) {
      padf = points[3 * r];
      tb = points[3 * r + 1];
      q = points[3 * r + 2];
      amp = amplitude(sp, q);
      keep = amp > sp.threshold;
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_count[wave] = (int)__popcll(m);
    block_sync();
    int before = running;
    for (int w = 0; w < wave; ++w) before += wave_count[w];
    if (keep) {
      const int64_t o = out0 + before + (int)__popcll(m & ((1ull << lane) - 1ull));
      int pad = (int)padf;
      pad = pad < 0 ? 0 : (pad >= sp.n_pads ? sp.n_pads - 1 : pad);
      double* row = rows + 8 * o;
      row[0] = sp.pad_centers[2 * pad];
      row[1] = sp.pad_centers[2 * pad + 1];
      row[2] = (sp.window_edge - tb) / (sp.window_edge - sp.mm_edge) * sp.length * 1000.0;  // writer.py:103-105
      row[3] = amp;
      row[4] = clipped_integral(sp, q);
      row[5] = padf;
      row[6] = tb;
      row[7] = sp.pad_sizes[pad];
      out_labels[o] = labels[r];
    }
    block_sync();
    if (threadIdx.x == 0) {
      int all = 0;
      for (int w = 0; w < SP_THREADS / 64; ++w) all += wave_count[w];
      running += all;
    }
    block_sync();
  }
}

void launch_spyral_count(hipStream_t s, const SpyralDev& sp, uint32_t n_events, const int64_t* event_start,
                         const double* points, int32_t* kept) {
  hipLaunchKernelGGL(spyral_count_kernel, dim3(n_events), dim3(SP_THREADS), 0, s, sp, event_start, points, kept);
}
void launch_spyral_write(hipStream_t s, const SpyralDev& sp, uint32_t n_events, const int64_t* event_start,
                         const int64_t* kept_start, const double* points, const int64_t* labels, double* rows,
                         int64_t* out_labels) {
  hipLaunchKernelGGL(spyral_write_kernel, dim3(n_events), dim3(SP_THREADS), 0, s, sp, event_start, kept_start, points,
                     labels, rows, out_labels);
}

}  // namespace attpc
