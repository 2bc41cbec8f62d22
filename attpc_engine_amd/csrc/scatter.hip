// scatter.hip -- electron drift with transverse diffusion + pad-plane point cloud.
//
// Restates (reference src/attpc_engine/detector/): transporter.py:252-317 (transport_track:
// sigma_t = sqrt(2 D dv t / E)), :172-249 (transverse_transport: 10x10 mesh over +-3 sigma,
// pad look-up per pixel, int(pdf h^2 n) electrons into a (tb, pad)-keyed dictionary,
// key inserted and label overwritten even when 0 electrons), :123-169 (point_transport),
// :78-120 (position_to_index: floor to whole mm), pairing.py (key), beam_pads.py (folded
// into the LUT), simulator.py:19-49 (dict_to_points), :108-113 (tb jitter, 0 <= tb < 512).
//
// Execution model: one workgroup = one event.  The event's dictionary is an open-addressing
// hash table in LDS (u32 key|label word + u64 charge, 8192 slots = 96 KiB); the 100 mesh
// pixels of every kept track sample are spread over the lanes and accumulated with LDS
// atomics (ds_cmpst_b32 to claim a slot, ds_add_u64 for the charge).  Events with more keys
// than the table holds are cut into time-bucket windows (a key contains its time bucket, so
// windows partition the key space); each window is flushed as one coalesced block of rows
// (one global atomic per window reserves the range) and the table is reused.  The nuclei of
// an event are scattered one after the other (barrier in between) so that "label = last
// nucleus in `indices` order that touched the key" holds without ordering atomics.
//
// pdf(pixel) h^2 depends only on the pixel index: (36/81)/(2 pi) exp(-(2/9)((i-4.5)^2+(j-4.5)^2))
// because the mesh pitch is h = 6 sigma / 9; the 100 weights are a constant table.
//
// Bound: LDS atomics + 2-byte LUT gathers (L2 resident, 625 KB) + f64 VALU; HBM traffic is
// the 32 B per output row (3 f64 + i64, the reference's own dtypes) and 32 B per track sample.
#include "tracks_args.hpp"

namespace attpc {

constexpr int SC_THREADS = 512;
constexpr int HASH_BITS = 13;
constexpr int HASH_CAP = 1 << HASH_BITS;   // slots
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr int WINDOW_BUDGET = 448;          // samples per window to start with (~16 keys/sample)
constexpr int MAX_PROBES = 192;
constexpr int MESH = ATTPC_MESH_STEPS;
constexpr int PIXELS = MESH * MESH;


struct __align__(16) ScatterShared {
  unsigned long long chg[HASH_CAP];
  uint32_t keys[HASH_CAP];
  double st_xlo[SC_THREADS], st_xhi[SC_THREADS], st_ylo[SC_THREADS], st_yhi[SC_THREADS];
  double st_sx[SC_THREADS], st_sy[SC_THREADS], st_n[SC_THREADS];
  int st_tb[SC_THREADS];   // bit 31 set => zero-diffusion point transport
  double wtab[PIXELS];
  int hist[ATTPC_NUM_TB];
  int win_a, win_b, budget, n_stage, overflow, done, failed, retried;
  unsigned int wg_cursor, n_rows;
  unsigned long long base;
  unsigned long long charge_sum, key_sum;
};

__device__ __forceinline__ const double* sample_ptr(const TrackBuffers& trk, uint32_t track, int s) {
  const int blk = trk.block_table[(size_t)track * MAX_BLOCKS_PER_TRACK + (s / ARENA_BLK)];
  return trk.arena + ((size_t)blk * ARENA_BLK + (s & (ARENA_BLK - 1))) * 4;
}

__device__ __forceinline__ void clear_table(ScatterShared& sh) {
  for (int i = threadIdx.x; i < HASH_CAP; i += SC_THREADS) {
    sh.keys[i] = EMPTY;
    sh.chg[i] = 0ull;
  }
}

__global__ __launch_bounds__(SC_THREADS) void scatter_kernel(ScatterArgs a) {
  __shared__ ScatterShared sh;
  const int tid = threadIdx.x;
  const uint32_t e_local = blockIdx.x;
  const uint64_t event = a.first_event + e_local;
  const int n_sim = a.layout.n_sim;
  const uint32_t track0 = e_local * (uint32_t)n_sim;

  // ---- init ----
  for (int p = tid; p < PIXELS; p += SC_THREADS) {
    const double di = (double)(p / MESH) - 4.5, dj = (double)(p % MESH) - 4.5;
    sh.wtab[p] = (36.0 / 81.0) / TWO_PI * exp(-(2.0 / 9.0) * (di * di + dj * dj));
  }
  for (int i = tid; i < ATTPC_NUM_TB; i += SC_THREADS) sh.hist[i] = 0;
  clear_table(sh);
  if (tid == 0) {
    sh.win_a = 0; sh.win_b = 0; sh.budget = WINDOW_BUDGET; sh.overflow = 0; sh.done = 0;
    sh.failed = 0; sh.retried = 0; sh.charge_sum = 0ull; sh.key_sum = 0ull; sh.n_rows = 0u;
  }
  __syncthreads();

  // ---- histogram of kept samples per time bucket (all nuclei) ----
  unsigned int my_samples = 0;
  for (int isim = 0; isim < n_sim; ++isim) {
    const uint32_t track = track0 + (uint32_t)isim;
    const int cnt = a.trk.counts[track];
    for (int s = tid; s < cnt; s += SC_THREADS) {
      const double t = sample_ptr(a.trk, track, s)[2];
      my_samples++;
      if (t >= 0.0 && t < (double)ATTPC_NUM_TB) atomicAdd(&sh.hist[(int)t], 1);
      // t < 0 (sigma_t would be NaN: undefined in the reference) and tb >= 512 (dropped by the
      // 0 <= tb < 512 mask of simulator.py:111-113) never reach the output
    }
  }
  __syncthreads();

  unsigned long long my_charge = 0ull, my_keys = 0ull;

  for (;;) {
    // ---- choose the next window [win_a, win_b) of time buckets ----
    if (tid == 0) {
      int a0 = sh.win_b;
      if (sh.overflow) {  // retry the same start with half the samples
        a0 = sh.win_a;
        sh.overflow = 0;
      }
      while (a0 < ATTPC_NUM_TB && sh.hist[a0] == 0) a0++;
      if (a0 >= ATTPC_NUM_TB) {
        sh.done = 1;
      } else {
        int b0 = a0, cum = 0;
        do { cum += sh.hist[b0]; b0++; } while (b0 < ATTPC_NUM_TB && cum + sh.hist[b0] <= sh.budget);
        sh.win_a = a0;
        sh.win_b = b0;
        sh.n_stage = cum;  // samples in this window (for the budget update on overflow)
      }
    }
    __syncthreads();
    if (sh.done) break;
    const int win_a = sh.win_a, win_b = sh.win_b;
    const int win_samples = sh.n_stage;
    __syncthreads();

    // ---- scatter every nucleus, in `indices` order ----
    for (int isim = 0; isim < n_sim; ++isim) {
      const uint32_t track = track0 + (uint32_t)isim;
      const int cnt = a.trk.counts[track];
      for (int base = 0; base < cnt; base += SC_THREADS) {
        if (tid == 0) sh.n_stage = 0;
        __syncthreads();
        const int s = base + tid;
        if (s < cnt) {
          const double* rec = sample_ptr(a.trk, track, s);
          const double2 xy = reinterpret_cast<const double2*>(rec)[0];
          const double2 tn = reinterpret_cast<const double2*>(rec)[1];
          const double t = tn.x;
          if (t >= 0.0 && t < (double)ATTPC_NUM_TB) {
            const int tb = (int)t;  // transporter.py:238
            if (tb >= win_a && tb < win_b) {
              const int slot = atomicAdd(&sh.n_stage, 1);
              // transporter.py:301
              const double sigma = sqrt(2.0 * a.det.diffusion * a.det.dv * t / a.det.efield);
              const double xlo = xy.x - 3.0 * sigma, xhi = xy.x + 3.0 * sigma;
              const double ylo = xy.y - 3.0 * sigma, yhi = xy.y + 3.0 * sigma;
              sh.st_xlo[slot] = xlo; sh.st_xhi[slot] = xhi;
              sh.st_ylo[slot] = ylo; sh.st_yhi[slot] = yhi;
              sh.st_sx[slot] = (xhi - xlo) / (double)(MESH - 1);  // numpy.linspace step
              sh.st_sy[slot] = (yhi - ylo) / (double)(MESH - 1);
              sh.st_n[slot] = tn.y;
              sh.st_tb[slot] = (sigma == 0.0) ? (tb | (int)0x80000000) : tb;
            }
          }
        }
        __syncthreads();
        const int n_items = sh.n_stage * PIXELS;
        const uint32_t label_bits = (uint32_t)isim << 24;
        for (int item = tid; item < n_items; item += SC_THREADS) {
          if (sh.overflow) break;
          const int st = item / PIXELS;
          const int p = item - st * PIXELS;
          const int i = p / MESH, j = p - i * MESH;
          const int tbw = sh.st_tb[st];
          double x, y, w;
          if (tbw < 0) {  // point_transport: all electrons straight down (transporter.py:123-169)
            if (p != 0) continue;
            x = 0.5 * (sh.st_xlo[st] + sh.st_xhi[st]);
            y = 0.5 * (sh.st_ylo[st] + sh.st_yhi[st]);
            w = 1.0;
          } else {
            x = (i == MESH - 1) ? sh.st_xhi[st] : (double)i * sh.st_sx[st] + sh.st_xlo[st];
            y = (j == MESH - 1) ? sh.st_yhi[st] : (double)j * sh.st_sy[st] + sh.st_ylo[st];
            w = sh.wtab[p];
          }
          // position_to_index, transporter.py:107-118 (whole-mm floor, low inclusive, high exclusive)
          const double fx = floor(x * 1000.0), fy = floor(y * 1000.0);
          const double lo = (double)a.det.lut_lo, hi = (double)(a.det.lut_lo + a.det.lut_n);
          if (!(fx >= lo && fx < hi && fy >= lo && fy < hi)) continue;
          const int ix = (int)fx - a.det.lut_lo, iy = (int)fy - a.det.lut_lo;
          const int pad = a.det.pad_lut[ix * a.det.lut_n + iy];
          if (pad < 0) continue;  // no pad, or a beam pad (folded)
          const unsigned long long q = (unsigned long long)(long long)(w * sh.st_n[st]);  // transporter.py:240-246
          const uint32_t key = ((uint32_t)(tbw & 0x3ff) << 14) | (uint32_t)pad;
          const uint32_t want = key | label_bits;
          uint32_t hslot = (key * 2654435761u) >> (32 - HASH_BITS);
          int probes = 0;
          for (;;) {
            const uint32_t cur = sh.keys[hslot];
            if ((cur & 0x00FFFFFFu) == key) {
              if (cur != want) sh.keys[hslot] = want;  // last-writer label, transporter.py:249
              break;
            }
            if (cur == EMPTY) {
              const uint32_t old = atomicCAS(&sh.keys[hslot], EMPTY, want);
              if (old == EMPTY) break;
              if ((old & 0x00FFFFFFu) == key) {
                sh.keys[hslot] = want;
                break;
              }
            }
            hslot = (hslot + 1) & (HASH_CAP - 1);
            if (++probes > MAX_PROBES) { sh.overflow = 1; break; }
          }
          if (probes <= MAX_PROBES) atomicAdd(&sh.chg[hslot], q);
        }
        __syncthreads();
      }
    }

    if (sh.overflow) {  // uniform: written before the last barrier
      clear_table(sh);
      if (tid == 0) {
        sh.retried++;
        if (win_b - win_a <= 1 && win_samples <= 1) {
          sh.failed = 1;         // a single sample cannot overflow 8192 slots; defensive
          sh.overflow = 0;
          sh.win_b = win_a + 1;  // skip this bucket
        } else if (win_b - win_a <= 1) {
          sh.failed = 1;         // one time bucket alone exceeds the table: event not representable
          sh.overflow = 0;
          sh.win_b = win_a + 1;
        } else {
          sh.budget = win_samples / 2 > 0 ? win_samples / 2 : 1;
        }
      }
      __syncthreads();
      continue;
    }

    // ---- flush: count, reserve one contiguous range, write rows, clear ----
    if (tid == 0) sh.wg_cursor = 0u;
    __syncthreads();
    unsigned int mine = 0;
    for (int i = tid; i < HASH_CAP; i += SC_THREADS) mine += (sh.keys[i] != EMPTY) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((tid & 63) == 0 && mine) atomicAdd(&sh.wg_cursor, mine);
    __syncthreads();
    const unsigned int total = sh.wg_cursor;
    __syncthreads();
    if (tid == 0) {
      sh.wg_cursor = 0u;
      unsigned long long base = 0ull;
      if (total) {
        base = atomicAdd(&a.out.ctrl[0], (unsigned long long)total);
        const unsigned long long si = atomicAdd(&a.out.ctrl[1], 1ull);
        if (base + total > (unsigned long long)a.out.capacity || si >= (unsigned long long)a.out.seg_capacity) {
          a.out.ctrl[6] = 1ull;  // out of capacity: host re-runs the chunk with larger buffers
          base = ~0ull;
        } else {
          Segment sg;
          sg.event = (int32_t)e_local;
          sg.count = (int32_t)total;
          sg.offset = (int64_t)base;
          a.out.segments[si] = sg;
        }
      }
      sh.base = base;
      sh.n_rows += total;
      sh.budget = WINDOW_BUDGET;
    }
    __syncthreads();
    const unsigned long long base = sh.base;
    for (int i0 = 0; i0 < HASH_CAP; i0 += SC_THREADS) {
      const int i = i0 + tid;
      const uint32_t word = sh.keys[i];
      const bool occ = word != EMPTY;
      const unsigned long long m = __ballot(occ);
      unsigned int wbase = 0;
      if ((tid & 63) == 0 && m) wbase = atomicAdd(&sh.wg_cursor, (unsigned int)__popcll(m));
      wbase = __shfl(wbase, 0);
      if (occ) {
        const unsigned long long q = sh.chg[i];
        sh.keys[i] = EMPTY;
        sh.chg[i] = 0ull;
        const uint32_t key = word & 0x00FFFFFFu;
        const int pad = (int)(key & 0x3fffu), tb = (int)(key >> 14);
        my_charge += q;
        my_keys += (event << 24) + (unsigned long long)key;
        if (base != ~0ull) {
          const unsigned long long row = base + wbase + (unsigned int)__popcll(m & ((1ull << (tid & 63)) - 1ull));
          double ua, ub;
          rng_pair(a.seed, event, key, DOMAIN_JITTER, ua, ub);  // simulator.py:108
          double* o = a.out.points + row * 3;
          o[0] = (double)pad;
          o[1] = (double)tb + ua;
          o[2] = (double)(long long)q;
          a.out.labels[row] = (int64_t)a.layout.indices[word >> 24];
        }
      }
    }
    __syncthreads();
  }

  // ---- per-event statistics ----
  for (int off = 32; off > 0; off >>= 1) {
    my_charge += __shfl_down(my_charge, off);
    my_keys += __shfl_down(my_keys, off);
    my_samples += __shfl_down(my_samples, off);
  }
  if ((tid & 63) == 0) {
    atomicAdd(&sh.charge_sum, my_charge);
    atomicAdd(&sh.key_sum, my_keys);
    if (my_samples) atomicAdd(&a.out.ctrl[7], (unsigned long long)my_samples);
  }
  __syncthreads();
  if (tid == 0) {
    if (sh.charge_sum) atomicAdd(&a.out.ctrl[2], sh.charge_sum);
    if (sh.key_sum) atomicAdd(&a.out.ctrl[3], sh.key_sum);
    if (sh.failed) atomicAdd(&a.out.ctrl[4], 1ull);
    if (sh.retried) atomicAdd(&a.out.ctrl[5], (unsigned long long)sh.retried);
  }
}

void launch_scatter_kernel(uint32_t n_events, hipStream_t s, const ScatterArgs& a) {
  hipLaunchKernelGGL(scatter_kernel, dim3(n_events), dim3(SC_THREADS), 0, s, a);
}

}  // namespace attpc
