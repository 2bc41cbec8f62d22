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
// the device: counting sort over time bucket x sixteenths of the jitter + rank inside the bin).
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
// argsort of column 2).  z falls with the time bucket, so the sort key is the jittered time bucket,
// descending.  Counting sort over SORT_BINS = 512 time buckets x 16 sixteenths of the jitter (highest
// first; the bin index is a monotone function of the key), then a rank inside each bin on the key itself
// (larger first; equal keys keep their cloud order -- the reference's argsort is unstable, so ties have no
// defined order there).  Bins hold a few rows each, so the rank costs a handful of comparisons per row
// whatever the event looks like (512 bins alone left hundreds of rows per bin for tracks across the
// drift direction, and a quadratic rank).  The bin-grouped (row, key) list lives in a global scratch
// range of the event's own cloud rows.
constexpr int SORT_SUB = 16;
constexpr int SORT_BINS = ATTPC_NUM_TB * SORT_SUB;           // 8192
constexpr int SORT_BINS_PER_THREAD = SORT_BINS / SP_THREADS;  // 32

static_assert(SORT_BINS * sizeof(uint32_t) <= 32 * 1024, "spyral_write_kernel: the bin table is the kernel's LDS footprint");
__global__ __launch_bounds__(SP_THREADS) void spyral_write_kernel(SpyralDev sp, const int64_t* __restrict__ event_start,
                                                                   const int64_t* __restrict__ kept_start,
                                                                   const double* __restrict__ points,
                                                                   const int64_t* __restrict__ labels,
                                                                   double* __restrict__ rows,
                                                                   int64_t* __restrict__ out_labels,
                                                                   uint32_t* __restrict__ sort_idx,
                                                                   double* __restrict__ sort_key,
                                                                   SpyralPacked* __restrict__ packed,
                                                                   int64_t* __restrict__ pack_flag) {
  // counts, then the next free position of every bin; once every row is placed bin_cursor[b] is the END of bin b,
  // i.e. the start of bin b + 1 -- so the rank pass needs no second array (32 KiB of LDS per workgroup instead of
  // 64 KiB: four workgroups per CU instead of two, in a kernel whose cost is memory latency)
  __shared__ uint32_t bin_cursor[SORT_BINS];
  __shared__ uint32_t wave_total[SP_THREADS / 64];
  const uint32_t e = blockIdx.x;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const int64_t lo = event_start[e], hi = event_start[e + 1];
  const int64_t out0 = kept_start[e];
  const uint32_t n_kept = (uint32_t)(kept_start[e + 1] - out0);
  if (n_kept == 0u) return;  // uniform
  for (int b = t; b < SORT_BINS; b += SP_THREADS) bin_cursor[b] = 0u;
  block_sync();
  auto bin_of = [](double tb) -> int {  // ascending z = descending time bucket; monotone in tb
    double c = tb < 0.0 ? 0.0 : tb;
    int whole = (int)c;
    whole = whole > ATTPC_NUM_TB - 1 ? ATTPC_NUM_TB - 1 : whole;
    int sub = (int)((c - (double)whole) * (double)SORT_SUB);  // exact: a power-of-two scale
    sub = sub > SORT_SUB - 1 ? SORT_SUB - 1 : sub;
    return (ATTPC_NUM_TB - 1 - whole) * SORT_SUB + (SORT_SUB - 1 - sub);
  };
  for (int64_t r = lo + t; r < hi; r += SP_THREADS)
    if (amplitude(sp, points[3 * r + 2]) > sp.threshold) atomicAdd(&bin_cursor[bin_of(points[3 * r + 1])], 1u);
  block_sync();
  {  // exclusive prefix over the bins: 32 consecutive bins per thread, wave scan, wave offsets
    uint32_t local = 0u;
    for (int k = 0; k < SORT_BINS_PER_THREAD; ++k) local += bin_cursor[t * SORT_BINS_PER_THREAD + k];
    uint32_t incl = local;
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = __shfl_up(incl, off);
      incl += lane >= off ? up : 0u;
    }
    if (lane == 63) wave_total[wave] = incl;
    block_sync();
    uint32_t run = incl - local;
    for (int w = 0; w < wave; ++w) run += wave_total[w];
    for (int k = 0; k < SORT_BINS_PER_THREAD; ++k) {
      const int b = t * SORT_BINS_PER_THREAD + k;
      const uint32_t c = bin_cursor[b];
      bin_cursor[b] = run;
      run += c;
    }
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
  // workgroup scope is enough (the list is written and read by this workgroup only) -- an agent-scope fence
  // here writes back the whole L2 once per event, with gigabytes of dirty cloud rows in it
  __threadfence_block();
  block_sync();
  for (uint32_t p = (uint32_t)t; p < n_kept; p += SP_THREADS) {
    const uint32_t ri = sort_idx[lo + p];
    const double tb = sort_key[lo + p];
    const int b = bin_of(tb);
    const uint32_t b_lo = b > 0 ? bin_cursor[b - 1] : 0u, b_hi = bin_cursor[b];
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
    if (packed != nullptr) {  // compact transfer (tracks_args.hpp): the host expands this to the 8-column row
      const long long label = labels[r];
      const unsigned long long charge = (unsigned long long)q_el;
      if (!(q_el >= 0.0) || charge >= (1ull << SPYRAL_PACK_CHARGE_BITS) || padf != (double)pad || label < 0 || label >= 32)
        atomicMax(reinterpret_cast<unsigned long long*>(pack_flag), 1ull);
      SpyralPacked rec;
      rec.tb = tb;
      rec.bits = (charge & ((1ull << SPYRAL_PACK_CHARGE_BITS) - 1)) | ((unsigned long long)pad << SPYRAL_PACK_CHARGE_BITS) |
                 ((unsigned long long)label << (SPYRAL_PACK_CHARGE_BITS + SPYRAL_PACK_PAD_BITS));
      rec.integral = clipped_integral(sp, q_el);
      packed[o] = rec;
      continue;
    }
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
                         int64_t* out_labels, uint32_t* sort_idx, double* sort_key, SpyralPacked* packed, int64_t* pack_flag) {
  hipLaunchKernelGGL(spyral_write_kernel, dim3(n_events), dim3(SP_THREADS), 0, s, sp, event_start, kept_start, points,
                     labels, rows, out_labels, sort_idx, sort_key, packed, pack_flag);
}

}  // namespace attpc
