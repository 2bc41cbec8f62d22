// lone.hip -- a time bucket that alone holds more keys than the scatter kernel's LDS hash table.
//
// The reference's per-event dictionary has no size limit (detector/simulator.py:93-101,
// transporter.py:247-249).  scatter_kernel cuts an event into time-bucket windows that fit its LDS
// table; a single bucket that still does not fit (more than ~4096 / ~8192 lit pads in ONE time bucket of
// ONE event) is left out there, recorded in CloudBuffers::lone_list and handled here, by a kernel the
// host queues right behind every scatter launch (it exits at once when the list is empty, which is
// the normal case).
//
// Inside one time bucket the key is the pad, so the dictionary is a direct-mapped array over the
// 14-bit pad range of the key: u64 electrons[16384] plus an 8-bit mask of the touching nuclei per pad,
// "label = last nucleus in `indices` order that touched the key" (transporter.py:249) being the highest
// set bit.  The arrays live in GLOBAL memory (one pair per workgroup, all-zero between uses): the kernel
// is queued behind every scatter launch and must start at once beside whatever else is resident -- with
// its table in LDS (144 KiB) each empty launch waited for a free compute unit, milliseconds of the
// scatter stream per chunk.  One workgroup per recorded bucket: every thread takes whole entries
// (sample x slice) of the bucket and walks their 100 mesh pixels (transporter.py:172-249) -- or, with the
// Monte-Carlo diffusion extension, their electrons -- with (L2) atomics; then the lit pads are appended
// to the launch's cloud as one more segment of the event, in the same row format and with the same Philox
// time-bucket jitter as scatter_kernel's flush.  Simple rather than fast on purpose.
#include "tracks_args.hpp"

namespace attpc {

constexpr int LONE_THREADS = 256;
constexpr int LN_CTRL_LONE = 29;   // the same out.ctrl[] slots as scatter.hip
constexpr int LN_CTRL_ROWS = 30;

struct LoneShared {
  double wtab[ATTPC_MESH_STEPS * ATTPC_MESH_STEPS];
  int blocks[ATTPC_MAX_SIM][MAX_BLOCKS_PER_TRACK];
  int cnt[ATTPC_MAX_SIM + 1];
  unsigned int n_lit, cursor;
  unsigned long long base, charge_sum, key_sum;
};

template <bool MC>
__global__ __launch_bounds__(LONE_THREADS) void lone_bucket_kernel(ScatterArgs a) {
  __shared__ LoneShared sh;
  const int t = (int)threadIdx.x;
  const unsigned long long n_listed = a.out.ctrl[LN_CTRL_LONE];
  const uint32_t n_lone = (uint32_t)(n_listed < (unsigned long long)a.out.lone_capacity ? n_listed : a.out.lone_capacity);
  if (n_lone == 0u) return;
  constexpr int MESH = ATTPC_MESH_STEPS;
  const int n_sim = a.layout.n_sim;
  const int lut_n = a.det.lut_n, lut_lo = a.det.lut_lo;
  const int16_t* __restrict__ lut = a.det.pad_lut;
  const int n_slices = a.det.longitudinal_diffusion > 0.0 ? ATTPC_LONG_STEPS : 1;
  const double lo_mm = (double)lut_lo, hi_mm = (double)(lut_lo + lut_n);
  for (int p = t; p < MESH * MESH; p += LONE_THREADS) {
    const double di = (double)(p / MESH) - 4.5, dj = (double)(p % MESH) - 4.5;
    sh.wtab[p] = (36.0 / 81.0) / TWO_PI * exp(-(2.0 / 9.0) * (di * di + dj * dj));  // as in scatter.hip
  }
  // electrons per pad / 8 bits per pad (bit k = touched by the nucleus at position k of `indices`)
  unsigned long long* chg = a.out.lone_chg + (size_t)blockIdx.x * LONE_PADS;
  uint32_t* mask = a.out.lone_mask + (size_t)blockIdx.x * (LONE_PADS / 4);
  if (t == 0) { sh.charge_sum = 0ull; sh.key_sum = 0ull; }
  block_sync();

  auto lut_index = [&](double pos_m) -> int {  // transporter.py:107-118
    const double f = floor(pos_m * 1000.0);
    return (f >= lo_mm && f < hi_mm) ? (int)f - lut_lo : lut_n;
  };
  auto add = [&](int pad, unsigned long long q, int isim) {
    if (pad < 0 || pad >= LONE_PADS) return;
    atomicOr(&mask[pad >> 2], 1u << (8 * (pad & 3) + isim));
    atomicAdd(&chg[pad], q);
  };

  for (uint32_t rec = blockIdx.x; rec < n_lone; rec += gridDim.x) {
    const LoneBucket lb = a.out.lone_list[rec];
    const uint32_t e_local = lb.event;
    const int tb = (int)lb.tb;
    const uint64_t event = a.first_event + e_local;
    const uint32_t track0 = (a.event0 + e_local) * (uint32_t)n_sim;
    if (t == 0) {
      int acc = 0;
      for (int k = 0; k < n_sim; ++k) {
        sh.cnt[k] = acc;
        acc += a.trk.counts[track0 + k];
      }
      for (int k = n_sim; k <= ATTPC_MAX_SIM; ++k) sh.cnt[k] = acc;
      sh.n_lit = 0u;
      sh.cursor = 0u;
    }
    for (int i = t; i < n_sim * MAX_BLOCKS_PER_TRACK; i += LONE_THREADS) {
      const int k = i / MAX_BLOCKS_PER_TRACK, b = i - k * MAX_BLOCKS_PER_TRACK;
      sh.blocks[k][b] = a.trk.block_table[(size_t)(track0 + k) * MAX_BLOCKS_PER_TRACK + b];
    }
    block_sync();
    const int total = sh.cnt[ATTPC_MAX_SIM];
    const int total_s = total * n_slices;
    for (int cs = t; cs < total_s; cs += LONE_THREADS) {
      const int c = cs / n_slices, sl = cs - c * n_slices;
      int isim = 0;
      for (int k = 1; k < ATTPC_MAX_SIM; ++k)
        if (k < n_sim && c >= sh.cnt[k]) isim = k;
      const int s = c - sh.cnt[isim];
      const double* r = a.trk.arena + ((size_t)sh.blocks[isim][s / ARENA_BLK] * ARENA_BLK + (s & (ARENA_BLK - 1))) * 4;
      const double x0 = r[0], y0 = r[1], tm = r[2], n_el = r[3];
      if (!(tm >= 0.0)) continue;
      double ts = tm;
      if (n_slices > 1) {  // numpy.linspace(t - 3 sigma_l, t + 3 sigma_l, 5)[sl], as in scatter.hip
        const double sigma_l = sqrt(2.0 * a.det.longitudinal_diffusion * a.det.dv * tm / a.det.efield) / a.det.dv;
        const double lo = tm - 3.0 * sigma_l, hi = tm + 3.0 * sigma_l;
        ts = sl == n_slices - 1 ? hi : (double)sl * ((hi - lo) / (double)(n_slices - 1)) + lo;
      }
      if (!(ts >= 0.0 && ts < (double)ATTPC_NUM_TB) || (int)ts != tb) continue;
      const double sigma = sqrt(2.0 * a.det.diffusion * a.det.dv * tm / a.det.efield);  // transporter.py:301
      const double wl = n_slices == 1 ? 1.0 : a.det.long_weights[sl];
      if constexpr (MC) {
        const uint32_t n_prim = (uint32_t)(n_el / (double)a.det.mpgd_gain32);
        const unsigned long long q = (unsigned long long)(long long)(wl * (double)a.det.mpgd_gain32);
        for (uint32_t k = 0; k < n_prim; ++k) {
          double ua, ub;
          rng_pair(a.seed, event, k, DOMAIN_MC + (uint32_t)cs, ua, ub);
          const double rad = sqrt(-2.0 * log(1.0 - ua));
          double sn, cn;
          sincos(TWO_PI * ub, &sn, &cn);
          const double x = __dadd_rn(x0, __dmul_rn(sigma, __dmul_rn(rad, cn)));
          const double y = __dadd_rn(y0, __dmul_rn(sigma, __dmul_rn(rad, sn)));
          add((int)lut[lut_index(x) * (lut_n + 1) + lut_index(y)], q, isim);
        }
      } else {
        const double n_w = wl * n_el;  // x 1.0 is exact
        if (sigma == 0.0) {  // point_transport, transporter.py:123-169
          add((int)lut[lut_index(x0) * (lut_n + 1) + lut_index(y0)], (unsigned long long)n_w, isim);
          continue;
        }
        const double xlo = x0 - 3.0 * sigma, xhi = x0 + 3.0 * sigma, ylo = y0 - 3.0 * sigma, yhi = y0 + 3.0 * sigma;
        const double sx = (xhi - xlo) / (double)(MESH - 1), sy = (yhi - ylo) / (double)(MESH - 1);
        for (int j = 0; j < MESH; ++j) {    // x mesh lines (numpy.linspace, :221-227)
          const int ixx = lut_index(j == MESH - 1 ? xhi : (double)j * sx + xlo);
          for (int i = 0; i < MESH; ++i) {  // y mesh lines
            const int iyy = lut_index(i == MESH - 1 ? yhi : (double)i * sy + ylo);
            add((int)lut[ixx * (lut_n + 1) + iyy], (unsigned long long)(sh.wtab[i * MESH + j] * n_w), isim);
          }
        }
      }
    }
    block_sync();
    // lit pads -> rows
    unsigned int mine = 0;
    for (int pad = t; pad < LONE_PADS; pad += LONE_THREADS)
      mine += ((__hip_atomic_load(&mask[pad >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (8 * (pad & 3))) & 0xffu) ? 1u : 0u;
    if (mine) atomicAdd(&sh.n_lit, mine);
    block_sync();
    const unsigned int n_rows = sh.n_lit;
    if (t == 0) {
      unsigned long long base = ~0ull;
      if (n_rows) {
        const unsigned long long g_base = atomicAdd(&a.out.ctrl[0], (unsigned long long)n_rows);
        const unsigned long long g_seg = atomicAdd(&a.out.ctrl[1], 1ull);
        const uint32_t before = atomicAdd(&a.out.ev_rows[e_local], n_rows);  // scatter_kernel's total is final
        if (g_base + n_rows > (unsigned long long)a.out.capacity || g_seg >= (unsigned long long)a.out.seg_capacity) {
          a.out.ctrl[6] = 1ull;  // out of capacity: the host re-runs the chunk with larger buffers
        } else {
          base = g_base;
          Segment sg;
          sg.event = (int32_t)e_local;
          sg.count = (int32_t)n_rows;
          sg.offset = (int64_t)base;
          sg.ev_offset = (int64_t)before;
          a.out.segments[g_seg] = sg;
        }
        atomicAdd(&a.out.ctrl[LN_CTRL_ROWS], (unsigned long long)n_rows);
      }
      sh.base = base;
    }
    block_sync();
    const unsigned long long base = sh.base;
    unsigned long long my_charge = 0ull, my_keys = 0ull;
    for (int pad = t; pad < LONE_PADS; pad += LONE_THREADS) {
      const uint32_t m = (__hip_atomic_load(&mask[pad >> 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (8 * (pad & 3))) & 0xffu;
      if (m == 0u) continue;
      const unsigned long long q = __hip_atomic_load(&chg[pad], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&chg[pad], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t key = ((uint32_t)tb << 14) | (uint32_t)pad;
      my_charge += q;
      my_keys += (event << 24) + (unsigned long long)key;
      if (base != ~0ull) {
        const unsigned long long row = base + atomicAdd(&sh.cursor, 1u);
        const double ua = jitter_uniform((uint32_t)a.seed, (uint32_t)(a.seed >> 32), (uint32_t)event, (uint32_t)(event >> 32), key);  // simulator.py:108
        double* o = a.out.points + row * 3;
        o[0] = (double)pad;
        o[1] = (double)tb + ua;
        o[2] = (double)q;
        a.out.labels[row] = (int64_t)a.layout.indices[31 - __clz((int)m)];
      }
    }
    if (my_charge) atomicAdd(&sh.charge_sum, my_charge);
    if (my_keys) atomicAdd(&sh.key_sum, my_keys);
    block_sync();
    for (int i = t; i < LONE_PADS / 4; i += LONE_THREADS) __hip_atomic_store(&mask[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    block_sync();
  }
  if (t == 0) {
    if (sh.charge_sum) atomicAdd(&a.out.ctrl[2], sh.charge_sum);
    if (sh.key_sum) atomicAdd(&a.out.ctrl[3], sh.key_sum);
  }
}

void launch_lone_bucket_kernel(uint32_t n_workgroups, hipStream_t s, const ScatterArgs& a) {
  if (a.det.mc_diffusion)
    hipLaunchKernelGGL(lone_bucket_kernel<true>, dim3(n_workgroups), dim3(LONE_THREADS), 0, s, a);
  else
    hipLaunchKernelGGL(lone_bucket_kernel<false>, dim3(n_workgroups), dim3(LONE_THREADS), 0, s, a);
}

}  // namespace attpc
