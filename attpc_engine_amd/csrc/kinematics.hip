// kinematics.hip -- KinematicsPipeline.run() as a HIP kernel: one lane = one event.
//
// Restates (reference src/attpc_engine/): kinematics/pipeline.py:232-283 (sample),
// :285-388 (run: whole-event rejection loop), kinematics/reaction.py:70-101,103-178
// (Reaction), :230-250,252-303 (Decay), kinematics/excitation.py, kinematics/angle.py.
//
// Roofline: HBM-write bound in principle (24 + 32*N bytes per event, no reuse), in practice
// negligible next to the detector kernels (<1 % of the run).  All f64: E ~ 1e4 MeV against
// KE ~ MeV needs it (f32 ulp there is ~1 keV).
#include "tracks_args.hpp"

namespace attpc {

struct P4 { double x, y, z, t; };

// `vector`-package boost_beta3 arithmetic (pinned by the reference's LISE KAT)
__device__ __forceinline__ P4 boost_beta3(const P4& p, double bx, double by, double bz) {
  const double bp2 = bx * bx + by * by + bz * bz;
  const double gam = 1.0 / sqrt(1.0 - bp2);
  const double bgam = gam * gam / (1.0 + gam);
  const double xx = 1.0 + bgam * bx * bx, yy = 1.0 + bgam * by * by, zz = 1.0 + bgam * bz * bz;
  const double xy = bgam * bx * by, xz = bgam * bx * bz, yz = bgam * by * bz;
  const double xt = gam * bx, yt = gam * by, zt = gam * bz;
  P4 o;
  o.x = xx * p.x + xy * p.y + xz * p.z + xt * p.t;
  o.y = xy * p.x + yy * p.y + yz * p.z + yt * p.t;
  o.z = xz * p.x + yz * p.y + zz * p.z + zt * p.t;
  o.t = xt * p.x + yt * p.y + zt * p.z + gam * p.t;
  return o;
}

__device__ __forceinline__ double inv_mass(const P4& p) {
  const double m2 = p.t * p.t - (p.x * p.x + p.y * p.y + p.z * p.z);
  return m2 >= 0.0 ? sqrt(m2) : -sqrt(-m2);
}

// reaction.py:156-176 / :285-302: CM solve, boost to lab, partner by subtraction
__device__ __forceinline__ void two_body(const P4& parent, double m_out, double m_other, double ex,
                                         double polar, double azim, P4& a, P4& b) {
  const double ibt = 1.0 / parent.t;
  const double bx = parent.x * ibt, by = parent.y * ibt, bz = parent.z * ibt;
  const P4 pcm = boost_beta3(parent, -bx, -by, -bz);
  const double ecm = pcm.t;
  const double mo = m_other + ex;
  const double e_a = (m_out * m_out - mo * mo + ecm * ecm) / (2.0 * ecm);
  const double p_a = sqrt(e_a * e_a - m_out * m_out);
  double sp, cp, sa, ca;
  sincos(polar, &sp, &cp);
  sincos(azim, &sa, &ca);
  const P4 cm = {p_a * sp * ca, p_a * sp * sa, p_a * cp, e_a};
  a = boost_beta3(cm, bx, by, bz);
  b = {parent.x - a.x, parent.y - a.y, parent.z - a.z, parent.t - a.t};
}

__device__ __forceinline__ bool reaction_allowed(const double* m, double t, double ex) {
  const double pz = sqrt(t * (t + 2.0 * m[1]));
  const double s = m[0] + t + m[1];
  const double e_cm = sqrt(s * s - pz * pz);
  return (m[2] + m[3] + ex) < e_cm;
}

__device__ __forceinline__ bool below_nr_threshold(const double* m, double t, double ex) {
  const double q = m[0] + m[1] - (m[2] + m[3] + ex);
  const double thr = -q * (m[2] + m[3]) / (m[2] + m[3] - m[1]);
  return t < thr;
}

__device__ __forceinline__ void store_row(double* p4, int row, const P4& v) {
  double* r = p4 + 4 * row;
  r[0] = v.x; r[1] = v.y; r[2] = v.z; r[3] = v.t;
}

__device__ double sample_excitation(const attpc_excitation_desc& d, double ua, double ub) {
  if (d.kind == ATTPC_EX_GAUSSIAN) return d.p0 + d.p1 * normal_from(ua, ub);
  if (d.kind == ATTPC_EX_UNIFORM) return d.p0 + (d.p1 - d.p0) * ua;
  const double* cdf = d.table_cdf;
  const double* x = d.table_x;
  if (ua <= cdf[0]) return x[0] - d.p0;
  int lo = 0, hi = d.table_len - 1;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cdf[mid] <= ua) lo = mid; else hi = mid;
  }
  const double w = cdf[hi] - cdf[lo];
  const double f = w > 0.0 ? (ua - cdf[lo]) / w : 0.0;
  return x[lo] + f * (x[hi] - x[lo]) - d.p0;
}

__device__ double sample_polar(const attpc_polar_desc& d, double ua, double ub) {
  if (d.kind == ATTPC_POLAR_UNIFORM) return acos(d.cos_min + (d.cos_max - d.cos_min) * ua);
  int lo = 0, hi = d.table_len;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (d.cdf[mid] <= ua) lo = mid + 1; else hi = mid;
  }
  if (lo > d.table_len - 1) lo = d.table_len - 1;
  return d.angles[lo] + ub * d.bin_width;
}

__device__ __forceinline__ double eloss_lookup(const attpc_kin_desc& d, double z) {
  if (d.eloss_len < 2 || !(d.z_max > d.z_min)) return d.eloss_len > 0 ? d.eloss[0] : 0.0;
  const double t = (z - d.z_min) / (d.z_max - d.z_min) * (double)(d.eloss_len - 1);
  int i = (int)floor(t);
  i = i < 0 ? 0 : (i > d.eloss_len - 2 ? d.eloss_len - 2 : i);
  const double f = t - (double)i;
  return d.eloss[i] + f * (d.eloss[i + 1] - d.eloss[i]);
}

// One event per lane; the retry loop diverges per lane and is bounded by sample_limit.
// Draws are counter based (index = attempt*64 + slot), so a step's parameters are drawn when the
// step is reached -- identical to drawing them all up front (pipeline.py:268-283) because any
// failure resamples the whole event.
__global__ __launch_bounds__(256) void kin_run_kernel(attpc_kin_desc d, uint64_t seed, uint64_t first_event,
                                                      uint32_t n, double* __restrict__ p4_out,
                                                      double* __restrict__ vertex_out,
                                                      int32_t* __restrict__ status_out,
                                                      uint32_t* __restrict__ attempts_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t event = first_event + i;
  const int n_rows = 4 + 2 * (d.n_steps - 1);
  double* p4 = p4_out + (size_t)i * n_rows * 4;
  double vx = 0.0, vy = 0.0, vz = 0.0;
  uint32_t attempt = 0;
  int32_t status = 0;
  for (;;) {
    if ((int32_t)attempt >= d.sample_limit) {  // pipeline.py:316-319
      status = 1;
      const double nan = __longlong_as_double(0x7ff8000000000000ll);
      for (int k = 0; k < n_rows * 4; ++k) p4[k] = nan;
      break;
    }
    const uint32_t base = attempt * KIN_SLOTS;
    attempt++;
    double ua, ub;
    double e_beam = d.beam_energy;
    vx = vy = vz = 0.0;
    if (d.has_target) {  // pipeline.py:245-264
      rng_pair(seed, event, base + 0, DOMAIN_KIN, ua, ub);
      const double rho = fabs(d.rho_sigma * normal_from(ua, ub));
      rng_pair(seed, event, base + 1, DOMAIN_KIN, ua, ub);
      double st, ct;
      sincos(TWO_PI * ua, &st, &ct);
      vx = rho * ct;
      vy = rho * st;
      rng_pair(seed, event, base + 2, DOMAIN_KIN, ua, ub);
      vz = d.z_min + (d.z_max - d.z_min) * ua;
      e_beam = e_beam - eloss_lookup(d, vz);
    }
    rng_pair(seed, event, base + 3, DOMAIN_KIN, ua, ub);
    double ex = sample_excitation(d.excitation[0], ua, ub);
    if (!reaction_allowed(d.masses, e_beam, ex)) continue;  // pipeline.py:323-326
    // reference raises ValueError here (reaction.py:142-143); restated as "resample"
    if (below_nr_threshold(d.masses, e_beam, ex)) continue;
    rng_pair(seed, event, base + 4, DOMAIN_KIN, ua, ub);
    double th = sample_polar(d.polar[0], ua, ub);
    rng_pair(seed, event, base + 5, DOMAIN_KIN, ua, ub);
    double ph = TWO_PI * ua;
    const P4 target = {0.0, 0.0, 0.0, d.masses[0]};
    const P4 proj = {0.0, 0.0, sqrt(e_beam * (e_beam + 2.0 * d.masses[1])), e_beam + d.masses[1]};
    const P4 parent = {0.0, 0.0, proj.z, target.t + proj.t};
    P4 eject, prev;
    two_body(parent, d.masses[2], d.masses[3], ex, th, ph, eject, prev);
    store_row(p4, 0, target);
    store_row(p4, 1, proj);
    store_row(p4, 2, eject);
    store_row(p4, 3, prev);
    bool allowed = true;
    for (int s = 1; s < d.n_steps; ++s) {  // pipeline.py:350-382
      const double m1 = d.masses[4 + 2 * (s - 1)], m2 = d.masses[5 + 2 * (s - 1)];
      rng_pair(seed, event, base + 3 + 4 * s, DOMAIN_KIN, ua, ub);
      ex = sample_excitation(d.excitation[s], ua, ub);
      if (!((inv_mass(prev) - (m1 + m2 + ex)) > 0.0)) { allowed = false; break; }
      rng_pair(seed, event, base + 4 + 4 * s, DOMAIN_KIN, ua, ub);
      th = sample_polar(d.polar[s], ua, ub);
      rng_pair(seed, event, base + 5 + 4 * s, DOMAIN_KIN, ua, ub);
      ph = TWO_PI * ua;
      P4 r1, r2;
      two_body(prev, m1, m2, ex, th, ph, r1, r2);
      store_row(p4, 4 + 2 * (s - 1), r1);
      store_row(p4, 5 + 2 * (s - 1), r2);
      prev = r2;
    }
    if (allowed) break;
  }
  if (vertex_out) {
    vertex_out[3 * (size_t)i + 0] = vx;
    vertex_out[3 * (size_t)i + 1] = vy;
    vertex_out[3 * (size_t)i + 2] = vz;
  }
  if (status_out) status_out[i] = status;
  if (attempts_out) attempts_out[i] = attempt;
}

// Deterministic map sampled parameters -> 4-vectors (attpc_kin_calculate)
__global__ __launch_bounds__(256) void kin_calculate_kernel(attpc_kin_desc d, uint32_t n,
                                                            const double* __restrict__ beam,
                                                            const double* __restrict__ ex_in,
                                                            const double* __restrict__ th_in,
                                                            const double* __restrict__ ph_in,
                                                            double* __restrict__ p4_out,
                                                            int32_t* __restrict__ status_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int n_rows = 4 + 2 * (d.n_steps - 1);
  double* p4 = p4_out + (size_t)i * n_rows * 4;
  const double nan = __longlong_as_double(0x7ff8000000000000ll);
  for (int k = 0; k < n_rows * 4; ++k) p4[k] = nan;
  const double e_beam = beam[i];
  const double* ex = ex_in + (size_t)i * d.n_steps;
  const double* th = th_in + (size_t)i * d.n_steps;
  const double* ph = ph_in + (size_t)i * d.n_steps;
  const bool allowed0 = reaction_allowed(d.masses, e_beam, ex[0]);          // reaction.py:70-101
  const bool below_nr = below_nr_threshold(d.masses, e_beam, ex[0]);        // reaction.py:136-143
  if (!allowed0 || below_nr) { status_out[i] = allowed0 ? -1 : (below_nr ? -2 : 1); return; }
  const P4 target = {0.0, 0.0, 0.0, d.masses[0]};
  const P4 proj = {0.0, 0.0, sqrt(e_beam * (e_beam + 2.0 * d.masses[1])), e_beam + d.masses[1]};
  const P4 parent = {0.0, 0.0, proj.z, target.t + proj.t};
  P4 eject, prev;
  two_body(parent, d.masses[2], d.masses[3], ex[0], th[0], ph[0], eject, prev);
  store_row(p4, 0, target);
  store_row(p4, 1, proj);
  store_row(p4, 2, eject);
  store_row(p4, 3, prev);
  int32_t status = 0;
  for (int s = 1; s < d.n_steps; ++s) {
    const double m1 = d.masses[4 + 2 * (s - 1)], m2 = d.masses[5 + 2 * (s - 1)];
    if (!((inv_mass(prev) - (m1 + m2 + ex[s])) > 0.0)) { status = s + 1; break; }
    P4 r1, r2;
    two_body(prev, m1, m2, ex[s], th[s], ph[s], r1, r2);
    store_row(p4, 4 + 2 * (s - 1), r1);
    store_row(p4, 5 + 2 * (s - 1), r2);
    prev = r2;
  }
  status_out[i] = status;
}

// Decay.calculate for explicit parent 4-vectors (attpc_decay_calculate)
__global__ __launch_bounds__(256) void decay_calculate_kernel(uint32_t n, const double* __restrict__ parent_in,
                                                              double m1, double m2,
                                                              const double* __restrict__ ex,
                                                              const double* __restrict__ th,
                                                              const double* __restrict__ ph,
                                                              double* __restrict__ out,
                                                              int32_t* __restrict__ status_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const P4 parent = {parent_in[4 * i], parent_in[4 * i + 1], parent_in[4 * i + 2], parent_in[4 * i + 3]};
  double* o = out + (size_t)i * 8;
  if (!((inv_mass(parent) - (m1 + m2 + ex[i])) > 0.0)) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int k = 0; k < 8; ++k) o[k] = nan;
    status_out[i] = 1;
    return;
  }
  P4 r1, r2;
  two_body(parent, m1, m2, ex[i], th[i], ph[i], r1, r2);
  store_row(o, 0, r1);
  store_row(o, 1, r2);
  status_out[i] = 0;
}

void launch_kin_run(hipStream_t s, const attpc_kin_desc& d, uint64_t seed, uint64_t first_event, uint32_t n,
                    double* p4, double* vertex, int32_t* status, uint32_t* attempts) {
  hipLaunchKernelGGL(kin_run_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, seed, first_event, n, p4, vertex,
                     status, attempts);
}
void launch_kin_calculate(hipStream_t s, const attpc_kin_desc& d, uint32_t n, const double* beam, const double* ex,
                          const double* th, const double* ph, double* p4, int32_t* status) {
  hipLaunchKernelGGL(kin_calculate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d, n, beam, ex, th, ph, p4, status);
}
void launch_decay_calculate(hipStream_t s, uint32_t n, const double* parent, double m1, double m2, const double* ex,
                            const double* th, const double* ph, double* out, int32_t* status) {
  hipLaunchKernelGGL(decay_calculate_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, parent, m1, m2, ex, th, ph,
                     out, status);
}

}  // namespace attpc
