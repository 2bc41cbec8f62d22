// tracks.hip -- charged-particle track integration + primary-electron creation.
//
// Restates (reference src/attpc_engine/detector/): solver.py:19-76 (equation_of_motion),
// :80-240 (four terminal events), :243-305 (generate_trajectory, sampled every 1e-10 s for
// at most 10001 samples), :308-347 (generate_electrons), :387-398 (>= 1 electron cut, gain,
// z -> time bucket).  The reference integrates with scipy Radau at rtol 1e-3; here every lane
// runs classical RK4 on the reference's own output grid (DESIGN.md "Track integrator").
//
// Execution model: one lane = one nucleus.  Waves are persistent: a lane whose track has
// ended takes the next track id from its wave's batch (128 ids per returning atomic on the global
// counter), so short tracks (particle leaves the gas after ~40 samples) do not idle next to long
// ones (particle ranges out, up to 10001 samples).  The host launches the kernel per batch of
// several scatter chunks so that every lane sees several tracks.  The stopping-power tables of all
// species live in LDS; the per-lane state (6 phase-space doubles, event functions, cursors) lives
// in registers.  Kept samples (x, y, time bucket, electrons*gain) are appended to per-track chains
// of 4 KiB arena blocks in HBM; waves reserve arena blocks 64 at a time.
//
// Bound: f64 VALU (about 2 rsqrt + 45 fma-class ops per right-hand side, 4 per step) plus one
// Philox + Box-Muller per two ionising samples; HBM traffic is 32 B per kept sample.
#include "tracks_args.hpp"

namespace attpc {

#ifndef ATTPC_TRACK_THREADS
#define ATTPC_TRACK_THREADS 256
#endif
#ifndef ATTPC_TRACK_REFILL
#define ATTPC_TRACK_REFILL 4
#endif
constexpr int TRACK_THREADS = ATTPC_TRACK_THREADS;
constexpr int STEPS_PER_REFILL = ATTPC_TRACK_REFILL;  // steps between two looks for finished lanes
constexpr int TRACK_ID_BATCH = 128;   // track ids a wave takes from the global counter at a time
constexpr int BLOCK_POOL = 64;        // arena blocks a wave reserves at a time (>= 64: one request can need a block per lane)
static_assert(TRACK_ID_BATCH >= 64 && BLOCK_POOL >= 64, "a single request can be one per lane");


struct Decomp {  // |gamma*beta| decomposition of a state, shared by the RHS and the event tests
  double inv_gv, gamma, inv_gamma, ke;
};

__device__ __forceinline__ Decomp decompose(double px, double py, double pz, double mass) {
  const double gv2 = px * px + py * py + pz * pz;
  Decomp d;
  d.inv_gv = rsqrt(gv2);
  const double g2 = 1.0 + gv2;
  d.inv_gamma = rsqrt(g2);
  d.gamma = g2 * d.inv_gamma;
  d.ke = mass * (d.gamma - 1.0);
  return d;
}

struct SpeciesConst {
  double mass;
  double qm_b;   // q/m * (-B) / c   -> d(gv)/dt = qm_b * (vy, -vx, .)
  double qm_e;   // q/m * (-E) / c
  double drag;   // MEV_2_JOULE * density * 100 / mass_kg / c  (times dE/dx = deceleration / c)
};

// solver.py:19-76 with the fields negated as at the call site (solver.py:297-299)
__device__ __forceinline__ void rhs(const double s[6], const Decomp& d, const SpeciesConst& sc,
                                    const double* tab, double r[6]) {
  const double vscale = C_LIGHT * d.inv_gamma;  // velocity = c * (gamma beta) / gamma
  const double vx = s[3] * vscale, vy = s[4] * vscale, vz = s[5] * vscale;
  const double dec = dedx_lookup(tab, d.ke) * sc.drag * d.inv_gv;  // deceleration/c per unit of gv
  r[0] = vx;
  r[1] = vy;
  r[2] = vz;
  r[3] = sc.qm_b * vy - dec * s[3];
  r[4] = -sc.qm_b * vx - dec * s[4];
  r[5] = sc.qm_e - dec * s[5];
}

// PATH: the path-length sampling extension (DetDev::path_step > 0) -- its own instantiation, so that
// the reference time-grid kernel keeps its registers (170 VGPRs, 2 waves per SIMD) and its code.
#ifndef ATTPC_TRACK_MIN_WAVES
#define ATTPC_TRACK_MIN_WAVES 1  // waves per SIMD the register allocation must leave room for (experiments)
#endif
template <bool PATH>
__global__ __launch_bounds__(TRACK_THREADS, ATTPC_TRACK_MIN_WAVES) void track_kernel(TrackArgs a) {
  extern __shared__ double lds_tab[];  // [n_species][ATTPC_DEDX_NODES]
  const int n_tab = a.det.n_species * ATTPC_DEDX_NODES;
  for (int i = threadIdx.x; i < n_tab; i += TRACK_THREADS) lds_tab[i] = a.det.dedx[i];
  block_sync();

  const int lane = threadIdx.x & 63;
  const int nsub = a.det.ode_substeps > 0 ? a.det.ode_substeps : 1;
  const double h_grid = 1.0e-10 / (double)nsub;
  double h = h_grid;      // PATH: per lane and per sample
  double t_now = 0.0;     // PATH: time of the last recorded sample
  const double e_scale = 1.0e6 / a.det.w_value;

  bool active = false, retired = false;
  uint32_t tid = 0;
  uint64_t event = 0;
  uint32_t fano_domain = 0;
  int k = 0;            // index of the last accepted ODE sample
  int count = 0;        // kept samples
  double* blk_ptr = nullptr;
  double s[6] = {0, 0, 0, 0, 0, 0};
  Decomp dc = {0, 0, 0, 0};
  double g_ke = 0, g_zf = 0, g_zb = 0, g_rho = 0, ke_prev = 0;
  SpeciesConst sc = {0, 0, 0, 0};
  const double* tab = lds_tab;
  double z_cache_cos = 0.0, z_cache_sin = 0.0;  // the two Box-Muller normals of the cached Philox pair ...
  int z_cache_idx = -1;                         // ... and its index (sample >> 1)
  uint32_t ids_next = 0, ids_end = 0;    // this wave's batch of track ids (wave uniform)
  uint32_t pool_next = 0, pool_end = 0;  // this wave's reserved arena blocks (wave uniform)

  for (;;) {
    // next track ids: the wave takes them from its own batch and refills the batch from the global
    // counter TRACK_ID_BATCH at a time (a returning atomic on one hot address costs microseconds;
    // once per finished track it dominated this kernel)
    const unsigned long long need = __ballot(!active && !retired);  // wave uniform from here ...
    uint32_t next_id = 0;
    if (need) {
      const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
      const uint32_t want = (uint32_t)__popcll(need);
      if (ids_next + want > ids_end) {
        // ids left in the old batch are handed out first, the rest come from the new one
        const uint32_t left = ids_end - ids_next;
        const int leader = __ffsll((long long)need) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&a.buf.ctrl[0], (uint32_t)TRACK_ID_BATCH);
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
        next_id = rank < left ? ids_next + rank : base + (rank - left);
        ids_next = base + (want - left);
        ids_end = base + (uint32_t)TRACK_ID_BATCH;
      } else {
        next_id = ids_next + rank;
        ids_next += want;
      }
    }  // ... to here
    if (!active && !retired) {
      if (next_id >= a.n_tracks) {
        retired = true;
      } else {
        // id -> (event, nucleus): nucleus-major in the host's order (TrackArgs::sim_order), or event-major
        uint32_t e_local;
        int isim;
        if (a.sim_order[0] != 0xffu) {
          const uint32_t slot = next_id / a.n_events;
          e_local = next_id - slot * a.n_events;
          isim = (int)a.sim_order[slot];
        } else {
          e_local = next_id / (uint32_t)a.layout.n_sim;
          isim = (int)(next_id - e_local * (uint32_t)a.layout.n_sim);
        }
        tid = e_local * (uint32_t)a.layout.n_sim + (uint32_t)isim;  // the track's place in the tables
        const int row = a.layout.indices[isim];
        const int sp = a.layout.species_of_row[row];
        const bool dead = sp < 0 || (a.kin_status != nullptr && a.kin_status[e_local] != 0);
        count = 0;
        k = 0;
        if (dead) {  // simulator.py:97-98 (Z == 0) or an event that hit the sample limit
          a.buf.counts[tid] = 0;
          a.buf.n_steps[tid] = 0;
        } else {
          const double mass = a.det.mass[sp];
          const double* mom = a.p4 + ((size_t)e_local * a.layout.n_rows + row) * 4;
          const double* vtx = a.vertex + (size_t)e_local * 3;
          const double inv_m = 1.0 / mass;
          s[0] = vtx[0]; s[1] = vtx[1]; s[2] = vtx[2];
          s[3] = mom[0] * inv_m; s[4] = mom[1] * inv_m; s[5] = mom[2] * inv_m;  // solver.py:271-273
          const double gv2 = s[3] * s[3] + s[4] * s[4] + s[5] * s[5];
          if (!(gv2 > 0.0) || !(gv2 < 1.0e300)) {  // nothing to integrate: one row, no electrons
            a.buf.counts[tid] = 0;
            a.buf.n_steps[tid] = 1;
          } else {
            const double mass_kg = mass * MEV_2_KG;
            const double q_m = (double)a.det.Z[sp] * E_CHARGE / mass_kg;
            sc.mass = mass;
            sc.qm_b = q_m * (-a.det.bfield) / C_LIGHT;
            sc.qm_e = q_m * (-a.det.efield) / C_LIGHT;
            sc.drag = MEV_2_JOULE * a.det.density * 100.0 / mass_kg / C_LIGHT;
            tab = lds_tab + sp * ATTPC_DEDX_NODES;
            dc = decompose(s[3], s[4], s[5], mass);
            g_ke = dc.ke - KE_LIMIT;
            g_zf = s[2] - 1.0;
            g_zb = s[2];
            g_rho = s[0] * s[0] + s[1] * s[1] - RHO_MAX * RHO_MAX;
            ke_prev = dc.ke;
            event = a.first_event + e_local;
            fano_domain = DOMAIN_FANO0 + (uint32_t)row;
            z_cache_idx = -1;
            t_now = 0.0;
            active = true;
          }
        }
      }
    }
    if (__all(retired)) break;

    for (int it = 0; it < STEPS_PER_REFILL; ++it) {
      bool stop = false, want_z = false;
      long long n_el = 0;
      double mu_s = 0.0, sig_s = 0.0;
      if constexpr (PATH) {
        if (active) {
          // sample every path_step of arc length, never coarser than the reference grid; the speed is
          // that of the last recorded sample (include/attpc_engine.h, attpc_det_desc::path_step)
          const double gv2 = s[3] * s[3] + s[4] * s[4] + s[5] * s[5];
          const double v = C_LIGHT * sqrt(gv2 / (1.0 + gv2));
          double h_sample = a.det.path_step / v;
          h_sample = h_sample < 1.0e-10 ? h_sample : 1.0e-10;
          if (t_now + h_sample > 1.0e-6 * (1.0 + 1.0e-9)) {  // end of the recording window
            a.buf.counts[tid] = count;
            a.buf.n_steps[tid] = k + 1;
            active = false;
          }
          t_now += h_sample;
          h = h_sample / (double)nsub;
        }
      }
      if (active) {
        for (int sub = 0; sub < nsub && !stop; ++sub) {
          double k1[6], k2[6], k3[6], k4[6], y[6];
          rhs(s, dc, sc, tab, k1);
#pragma unroll
          for (int i = 0; i < 6; ++i) y[i] = s[i] + 0.5 * h * k1[i];
          Decomp d2 = decompose(y[3], y[4], y[5], sc.mass);
          rhs(y, d2, sc, tab, k2);
#pragma unroll
          for (int i = 0; i < 6; ++i) y[i] = s[i] + 0.5 * h * k2[i];
          d2 = decompose(y[3], y[4], y[5], sc.mass);
          rhs(y, d2, sc, tab, k3);
#pragma unroll
          for (int i = 0; i < 6; ++i) y[i] = s[i] + h * k3[i];
          d2 = decompose(y[3], y[4], y[5], sc.mass);
          rhs(y, d2, sc, tab, k4);
#pragma unroll
          for (int i = 0; i < 6; ++i) s[i] = s[i] + (h / 6.0) * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
          dc = decompose(s[3], s[4], s[5], sc.mass);
          // terminal events with scipy's sign-change test (solver.py:80-240, :276-283)
          const double n_ke = dc.ke - KE_LIMIT;
          const double n_zf = s[2] - 1.0;
          const double n_zb = s[2];
          const double n_rho = s[0] * s[0] + s[1] * s[1] - RHO_MAX * RHO_MAX;
          stop = (g_ke >= 0.0 && n_ke <= 0.0) || (g_zf <= 0.0 && n_zf >= 0.0) ||
                 (g_zb >= 0.0 && n_zb <= 0.0) || (g_rho <= 0.0 && n_rho >= 0.0) || !(n_ke == n_ke);
          g_ke = n_ke; g_zf = n_zf; g_zb = n_zb; g_rho = n_rho;
        }
        if (!stop) {
          k++;
          // generate_electrons, solver.py:338-346: mu = |dKE| 1e6 / W, n = trunc(N(mu, sqrt(F mu)))
          const double mu = fabs(dc.ke - ke_prev) * e_scale;
          ke_prev = dc.ke;
          const double sig = sqrt(a.det.fano_factor * mu);
          want_z = mu + 9.0 * sig >= 1.0;  // |z| <= 8.6 for a 53-bit uniform: otherwise n = 0 for certain
          mu_s = mu;
          sig_s = sig;
        }
      }
      // The Fano draw: sample k uses the cosine (k even) or sine (k odd) normal of the Philox pair k >> 1, so a lane
      // needs a new pair every other sample -- and the lanes of a wave are at different k.  When some lane needs a
      // pair NOW, every other lane takes part as well: a lane whose cached pair serves its last sample now (k odd)
      // computes its NEXT pair into the cache, a lane at an even k without the pair of k >> 1 computes that one
      // (it may want the sine at k + 1).  The wave then runs the ~200 instructions of Philox + log + sincos every
      // other step with all lanes in it instead of every step with half of them.  Same pairs, same normals.
      {
        const int idx_now = k >> 1;
        const bool have_now = z_cache_idx == idx_now;
        const bool need_now = want_z && !have_now;
        double z = (k & 1) ? z_cache_sin : z_cache_cos;  // the cached normal (meaningful when have_now)
        if (__any(need_now)) {
          const bool sampled = active && !stop;  // k was advanced for this lane
          const bool fill = sampled && (!have_now || (k & 1));
          if (fill) {
            const int idx = have_now ? idx_now + 1 : idx_now;
            double ua, ub;
            rng_pair(a.seed, event, (uint32_t)idx, fano_domain, ua, ub);
            const double rad = sqrt(-2.0 * log(1.0 - ua));
            double sn, cs;
            sincos(TWO_PI * ub, &sn, &cs);
            if (!have_now) z = (k & 1) ? rad * sn : rad * cs;
            z_cache_cos = rad * cs;
            z_cache_sin = rad * sn;
            z_cache_idx = idx;
          }
        }
        if (want_z) n_el = (long long)(mu_s + sig_s * z);
      }
      // arena blocks for the lanes that start a new block, from the wave's reserved pool (wave uniform:
      // one global atomic per BLOCK_POOL blocks instead of one per block)
      const bool new_blk = n_el >= 1 && (count & (ARENA_BLK - 1)) == 0;
      const unsigned long long blk_mask = __ballot(new_blk);
      uint32_t blk = 0;
      if (blk_mask) {
        const uint32_t rank = (uint32_t)__popcll(blk_mask & ((1ull << lane) - 1ull));
        const uint32_t want = (uint32_t)__popcll(blk_mask);
        if (pool_next + want > pool_end) {
          const uint32_t left = pool_end - pool_next;
          const int leader = __ffsll((long long)blk_mask) - 1;
          uint32_t base = 0;
          if (lane == leader) base = atomicAdd(&a.buf.ctrl[1], (uint32_t)BLOCK_POOL);
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
          blk = rank < left ? pool_next + rank : base + (rank - left);
          pool_next = base + (want - left);
          pool_end = base + (uint32_t)BLOCK_POOL;
        } else {
          blk = pool_next + rank;
          pool_next += want;
        }
      }
      if (n_el >= 1) {  // solver.py:387-392
        const int slot = count & (ARENA_BLK - 1);
        if (new_blk) {
          if (blk < a.buf.arena_blocks) {
            blk_ptr = a.buf.arena + (size_t)blk * ARENA_BLK * 4;
            a.buf.block_table[(size_t)tid * MAX_BLOCKS_PER_TRACK + (count / ARENA_BLK)] = (int32_t)blk;
          } else {
            blk_ptr = nullptr;
            a.buf.ctrl[2] = 1u;  // arena exhausted: host re-runs the chunk with a larger arena
          }
        }
        if (blk_ptr != nullptr) {
          double* o = blk_ptr + slot * 4;
          const double tb = (a.det.length - s[2]) * a.det.inv_dv + a.det.mm_edge;  // solver.py:395-398
          reinterpret_cast<double2*>(o)[0] = make_double2(s[0], s[1]);
          reinterpret_cast<double2*>(o)[1] = make_double2(tb, (double)(n_el * a.det.mpgd_gain));
        }
        count++;  // also without a block: the block counter then tells the host exactly how large the arena must be
      }
      if (active) {
        if (!stop && k >= ATTPC_TIME_SAMPLES - 1) {  // t = 1 us: last recorded sample
          stop = true;
          // path-length step: the sample cap came before the end of the 1 us window -- the track is cut short
          // (at 0.1 mm about 1 m of arc length); counted, attpc_run_stats.n_tracks_capped
          if constexpr (PATH) atomicAdd(&a.buf.ctrl[4], 1u);
        }
        if (stop) {
          a.buf.counts[tid] = count;
          a.buf.n_steps[tid] = k + 1;
          active = false;
        }
      }
    }
  }
}

void launch_track_kernel(uint32_t blocks, size_t lds_bytes, hipStream_t s, const TrackArgs& a) {
  if (a.det.path_step > 0.0)
    hipLaunchKernelGGL(track_kernel<true>, dim3(blocks), dim3(TRACK_THREADS), lds_bytes, s, a);
  else
    hipLaunchKernelGGL(track_kernel<false>, dim3(blocks), dim3(TRACK_THREADS), lds_bytes, s, a);
}

}  // namespace attpc
