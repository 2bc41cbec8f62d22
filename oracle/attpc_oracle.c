/*
 * attpc_oracle.c -- plain-C CPU restatement of the attpc_engine hot path.
 * TEST INFRASTRUCTURE ONLY (see attpc_oracle.h).  Compiled with -ffp-contract=off so
 * every operation is an individually rounded IEEE binary64 operation, like numpy.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/src/attpc_engine/).  Where the reference draws from an unseeded
 * numpy Generator, the oracle draws from a counter-based Philox4x32-10 stream instead
 * (the reference is not bit-reproducible run to run, SURVEY.md section 5 "RNG note").
 */
#include "attpc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* detector/constants.py:23-35 (scipy CODATA values, exact doubles) */
static const double MEV_2_JOULE = 1.6021766339999998e-13;
static const double MEV_2_KG = 1.7826619216278976e-30;
static const double C_LIGHT = 299792458.0;
static const double E_CHARGE = 1.602176634e-19;
static const double KE_LIMIT = 1e-6; /* detector/solver.py:14 */
static const double PI = 3.141592653589793;

/* ------------------------------------------------------------------ RNG ---------- */
/* Philox4x32-R (Salmon et al., SC'11; R = 10 is the generator behind rocRAND's default). */
void orc_philox4x32(const uint32_t ctr_in[4], const uint32_t key_in[2], int32_t rounds, uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < rounds; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  orc_philox4x32(ctr_in, key_in, 10, out);
}

static double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

/* counter = (event_lo, event_hi, index, domain), key = seed; two uniforms in [0,1) */
void orc_rng_pair(uint64_t seed, uint64_t event, uint32_t index, uint32_t domain, double* u_a,
                  double* u_b) {
  uint32_t ctr[4] = {(uint32_t)event, (uint32_t)(event >> 32), index, domain};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  uint32_t r[4];
  orc_philox4x32_10(ctr, key, r);
  *u_a = u53(r[0], r[1]);
  *u_b = u53(r[2], r[3]);
}

/* Philox2x32 (Salmon et al. SC'11), the 64-bit member of the family; pinned by the Random123 known-answer vectors
 * (tests/test_oracle_golden.py). */
void orc_philox2x32(const uint32_t ctr_in[2], uint32_t key, int32_t rounds, uint32_t out[2]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], k = key;
  for (int32_t r = 0; r < rounds; ++r) {
    uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;
    uint32_t n0 = (uint32_t)(p >> 32) ^ k ^ c1;
    c1 = (uint32_t)p;
    c0 = n0;
    k += 0x9E3779B9u;
  }
  out[0] = c0;
  out[1] = c1;
}

/* the time-bucket jitter of a cloud point (simulator.py:108: tb += U[0,1); one draw per point, the most numerous
 * random numbers of the path): Philox2x32-7 -- the fewest rounds that pass BigCrush in the Philox paper --
 * counter = (event[31:0], event[39:32] << 24 | tb << 14 | pad), key = seed[31:0] ^ rotl(seed[63:32], 13) ^ 0x100 */
double orc_jitter_uniform(uint64_t seed, uint64_t event, uint32_t key24) {
  uint32_t seed_lo = (uint32_t)seed, seed_hi = (uint32_t)(seed >> 32);
  uint32_t ctr[2] = {(uint32_t)event, ((uint32_t)(event >> 32) << 24) | key24};
  uint32_t r[2];
  orc_philox2x32(ctr, seed_lo ^ ((seed_hi << 13) | (seed_hi >> 19)) ^ 0x100u, 7, r);
  return u53(r[0], r[1]);
}

/* Box-Muller on (1-u_a) in (0,1] and u_b */
static double normal_from(double u_a, double u_b) {
  return sqrt(-2.0 * log(1.0 - u_a)) * cos(2.0 * PI * u_b);
}

double orc_rng_normal(uint64_t seed, uint64_t event, uint32_t index, uint32_t domain) {
  double a, b;
  orc_rng_pair(seed, event, index, domain, &a, &b);
  return normal_from(a, b);
}

/* ------------------------------------------------------------ kinematics --------- */
/* Lorentz boost of p (px,py,pz,E) by velocity (bx,by,bz): the arithmetic of the `vector`
 * package's boost_beta3 (third-party, absent; pinned by tests/test_kinematics.py:13-36). */
static void boost_beta3(const double p[4], double bx, double by, double bz, double out[4]) {
  double bp2 = bx * bx + by * by + bz * bz;
  double gam = 1.0 / sqrt(1.0 - bp2);
  double bgam = gam * gam / (1.0 + gam);
  double xx = 1.0 + bgam * bx * bx, yy = 1.0 + bgam * by * by, zz = 1.0 + bgam * bz * bz;
  double xy = bgam * bx * by, xz = bgam * bx * bz, yz = bgam * by * bz;
  double xt = gam * bx, yt = gam * by, zt = gam * bz;
  out[0] = xx * p[0] + xy * p[1] + xz * p[2] + xt * p[3];
  out[1] = xy * p[0] + yy * p[1] + yz * p[2] + yt * p[3];
  out[2] = xz * p[0] + yz * p[1] + zz * p[2] + zt * p[3];
  out[3] = xt * p[0] + yt * p[1] + zt * p[2] + gam * p[3];
}
/* vec.boost(by)  == boost_p4: beta = by.p / by.E */
static void boost_p4(const double p[4], const double by[4], double out[4]) {
  boost_beta3(p, by[0] / by[3], by[1] / by[3], by[2] / by[3], out);
}
/* vec.boostCM_of(of) == boost by -beta of `of` */
static void boost_cm_of(const double p[4], const double of[4], double out[4]) {
  boost_beta3(p, -of[0] / of[3], -of[1] / of[3], -of[2] / of[3], out);
}
static double inv_mass(const double p[4]) {
  double m2 = p[3] * p[3] - (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  return m2 >= 0.0 ? sqrt(m2) : -sqrt(-m2);
}

/* kinematics/reaction.py:70-101.  m = target, projectile, ejectile, residual masses */
int32_t orc_reaction_allowed(const double m[4], double t, double ex) {
  double pz = sqrt(t * (t + 2.0 * m[1]));
  double s = m[0] + t + m[1];
  double e_cm = sqrt(s * s - pz * pz);
  return (m[2] + m[3] + ex) < e_cm;
}

/* two-body CM solve + boost shared by Reaction.calculate / Decay.calculate:
 * reaction.py:156-176 and :285-302 */
static void two_body(const double parent[4], double m_out, double m_other, double ex, double polar,
                     double azim, double out_a[4], double out_b[4]) {
  double parent_cm[4];
  boost_cm_of(parent, parent, parent_cm);
  double ecm = parent_cm[3];
  double mo = m_other + ex;
  double e_a = (m_out * m_out - mo * mo + ecm * ecm) / (2.0 * ecm);
  double p_a = sqrt(e_a * e_a - m_out * m_out);
  double cm[4] = {p_a * sin(polar) * cos(azim), p_a * sin(polar) * sin(azim), p_a * cos(polar), e_a};
  boost_p4(cm, parent, out_a);
  for (int i = 0; i < 4; ++i) out_b[i] = parent[i] - out_a[i];
}

/* kinematics/reaction.py:103-178.  returns 0 ok, 1 below the (non-relativistic) threshold */
int32_t orc_reaction_calculate(const double m[4], double t, double polar, double azim, double ex,
                               double out[4][4]) {
  double q = m[0] + m[1] - (m[2] + m[3] + ex);
  double thr = -q * (m[2] + m[3]) / (m[2] + m[3] - m[1]);
  if (t < thr) return 1;
  double target[4] = {0.0, 0.0, 0.0, m[0]};
  double proj[4] = {0.0, 0.0, sqrt(t * (t + 2.0 * m[1])), t + m[1]};
  double parent[4];
  for (int i = 0; i < 4; ++i) parent[i] = target[i] + proj[i];
  memcpy(out[0], target, sizeof target);
  memcpy(out[1], proj, sizeof proj);
  two_body(parent, m[2], m[3], ex, polar, azim, out[2], out[3]);
  return 0;
}

/* kinematics/reaction.py:230-250 */
int32_t orc_decay_allowed(const double parent[4], double m1, double m2, double ex) {
  return (inv_mass(parent) - (m1 + m2 + ex)) > 0.0;
}

/* kinematics/reaction.py:252-303; out[0] = residual_1, out[1] = residual_2 */
int32_t orc_decay_calculate(const double parent[4], double m1, double m2, double polar, double azim,
                            double ex, double out[2][4]) {
  double q = inv_mass(parent) - (m1 + m2 + ex);
  if (q < 0.0) return 1;
  two_body(parent, m1, m2, ex, polar, azim, out[0], out[1]);
  return 0;
}

/* kinematics/excitation.py: Gaussian :58-80, Uniform :107-128, BreitWigner :162-188
 * (BreitWigner as an inverse-CDF table of scipy.stats.rel_breitwigner built by the caller) */
double orc_sample_excitation(const orc_excitation_desc* d, double u_a, double u_b) {
  switch (d->kind) {
    case 0: return d->p0 + d->p1 * normal_from(u_a, u_b);
    case 1: return d->p0 + (d->p1 - d->p0) * u_a;
    default: {
      int32_t n = d->table_len;
      const double* cdf = d->table_cdf;
      const double* x = d->table_x;
      if (u_a <= cdf[0]) return x[0] - d->p0;
      int32_t lo = 0, hi = n - 1; /* cdf[lo] <= u < cdf[hi] */
      while (hi - lo > 1) {
        int32_t mid = (lo + hi) / 2;
        if (cdf[mid] <= u_a) lo = mid; else hi = mid;
      }
      double w = cdf[hi] - cdf[lo];
      double f = w > 0.0 ? (u_a - cdf[lo]) / w : 0.0;
      return x[lo] + f * (x[hi] - x[lo]) - d->p0;
    }
  }
}

/* kinematics/angle.py: PolarUniform :62-80, PolarArbitrary :122-152 (numpy choice =
 * searchsorted(cumsum(p)/sum, u, side='right')) */
double orc_sample_polar(const orc_polar_desc* d, double u_a, double u_b) {
  if (d->kind == 0) return acos(d->cos_min + (d->cos_max - d->cos_min) * u_a);
  int32_t lo = 0, hi = d->table_len; /* first idx with cdf[idx] > u */
  while (lo < hi) {
    int32_t mid = (lo + hi) / 2;
    if (d->cdf[mid] <= u_a) lo = mid + 1; else hi = mid;
  }
  if (lo > d->table_len - 1) lo = d->table_len - 1;
  return d->angles[lo] + u_b * d->bin_width;
}

static double eloss_lookup(const orc_kin_desc* d, double z) {
  if (d->eloss_len < 2 || !(d->z_max > d->z_min)) return d->eloss_len > 0 ? d->eloss[0] : 0.0;
  double t = (z - d->z_min) / (d->z_max - d->z_min) * (double)(d->eloss_len - 1);
  int32_t i = (int32_t)floor(t);
  if (i < 0) i = 0;
  if (i > d->eloss_len - 2) i = d->eloss_len - 2;
  double f = t - (double)i;
  return d->eloss[i] + f * (d->eloss[i + 1] - d->eloss[i]);
}

/* draw slots inside one attempt: 0,1,2 vertex; 3+4*step + {0 Ex, 1 polar, 2 phi} */
#define KIN_SLOTS 64u

/* kinematics/pipeline.py:232-283 (sample) + :285-388 (run).  returns status 0 ok / 1 limit */
int32_t orc_kin_event(const orc_kin_desc* d, uint64_t seed, uint64_t event, double* p4,
                      double* vertex, uint32_t* attempts_out) {
  int32_t n_rows = 4 + 2 * (d->n_steps - 1);
  uint32_t attempt = 0;
  int32_t status = 0;
  for (;;) {
    if ((int32_t)attempt >= d->sample_limit) { /* pipeline.py:316-319 */
      status = 1;
      for (int i = 0; i < n_rows * 4; ++i) p4[i] = NAN;
      break;
    }
    uint32_t base = attempt * KIN_SLOTS;
    attempt++;
    double ua, ub;
    double e_beam = d->beam_energy;
    vertex[0] = vertex[1] = vertex[2] = 0.0;
    if (d->has_target) { /* pipeline.py:245-264 */
      orc_rng_pair(seed, event, base + 0, 0, &ua, &ub);
      double rho = fabs(0.0 + d->rho_sigma * normal_from(ua, ub));
      orc_rng_pair(seed, event, base + 1, 0, &ua, &ub);
      double theta = 0.0 + (2.0 * PI - 0.0) * ua;
      vertex[0] = rho * cos(theta);
      vertex[1] = rho * sin(theta);
      orc_rng_pair(seed, event, base + 2, 0, &ua, &ub);
      vertex[2] = d->z_min + (d->z_max - d->z_min) * ua;
      e_beam = e_beam - eloss_lookup(d, vertex[2]);
    }
    double ex[ORC_MAX_STEPS], th[ORC_MAX_STEPS], ph[ORC_MAX_STEPS];
    for (int s = 0; s < d->n_steps; ++s) { /* pipeline.py:268-283 */
      orc_rng_pair(seed, event, base + 3 + 4 * s + 0, 0, &ua, &ub);
      ex[s] = orc_sample_excitation(&d->excitation[s], ua, ub);
      orc_rng_pair(seed, event, base + 3 + 4 * s + 1, 0, &ua, &ub);
      th[s] = orc_sample_polar(&d->polar[s], ua, ub);
      orc_rng_pair(seed, event, base + 3 + 4 * s + 2, 0, &ua, &ub);
      ph[s] = 0.0 + (2.0 * PI - 0.0) * ua;
    }
    if (!orc_reaction_allowed(d->masses, e_beam, ex[0])) continue; /* pipeline.py:323-326 */
    double rows[4][4];
    /* reference raises ValueError here (reaction.py:142-143); restated as "resample" */
    if (orc_reaction_calculate(d->masses, e_beam, th[0], ph[0], ex[0], rows)) continue;
    memcpy(p4, rows, sizeof rows);
    double prev[4];
    memcpy(prev, rows[3], sizeof prev);
    int allowed = 1;
    for (int s = 1; s < d->n_steps; ++s) { /* pipeline.py:350-382 */
      double m1 = d->masses[4 + 2 * (s - 1)], m2 = d->masses[5 + 2 * (s - 1)];
      if (!orc_decay_allowed(prev, m1, m2, ex[s])) { allowed = 0; break; }
      double r[2][4];
      orc_decay_calculate(prev, m1, m2, th[s], ph[s], ex[s], r);
      memcpy(p4 + (4 + 2 * (s - 1)) * 4, r, sizeof r);
      memcpy(prev, r[1], sizeof prev);
    }
    if (allowed) break;
  }
  if (attempts_out) *attempts_out = attempt;
  return status;
}

void orc_kin_batch(const orc_kin_desc* d, uint64_t seed, uint64_t first, uint64_t n, double* p4,
                   double* vertex, int32_t* status, uint32_t* attempts, int32_t n_threads) {
  int32_t n_rows = 4 + 2 * (d->n_steps - 1);
  (void)n_threads;
#pragma omp parallel for schedule(dynamic, 256) num_threads(n_threads > 0 ? n_threads : 1)
  for (int64_t i = 0; i < (int64_t)n; ++i) {
    uint32_t att;
    int32_t st = orc_kin_event(d, seed, first + (uint64_t)i, p4 + i * n_rows * 4, vertex + i * 3, &att);
    if (status) status[i] = st;
    if (attempts) attempts[i] = att;
  }
}

/* --------------------------------------------------------------- detector -------- */
static int32_t g_beam_pads[512];
static int32_t g_n_beam_pads = 0;
void orc_set_beam_pads(const int32_t* pads, int32_t n) {
  if (n > 512) n = 512;
  memcpy(g_beam_pads, pads, (size_t)n * sizeof(int32_t));
  g_n_beam_pads = n;
}
static int is_beam_pad(int64_t pad) { /* `pad not in BEAM_PADS_ARRAY`, transporter.py:162,237 */
  for (int i = 0; i < g_n_beam_pads; ++i)
    if (g_beam_pads[i] == pad) return 1;
  return 0;
}

/* detector/pairing.py:5-27 */
int64_t orc_pair(int64_t tb, int64_t pad) {
  if (tb < 0 || pad < 0) return -1;
  int64_t mx = tb > pad ? tb : pad;
  return tb == mx ? tb * tb + tb + pad : pad * pad + tb;
}
/* detector/pairing.py:30-55 */
void orc_unpair(int64_t id, int64_t* tb, int64_t* pad) {
  if (id < 0) { *tb = -1; *pad = -1; return; }
  double s = floor(sqrt((double)id));
  double rem = (double)id - s * s;
  if (rem < s) { *tb = (int64_t)rem; *pad = (int64_t)s; }
  else { *tb = (int64_t)s; *pad = (int64_t)(rem - s); }
}

/* dE/dx on the binade grid: E = 2^e (1 + m/32); linear inside a sub-bin.  Restates the
 * configure-time tabulation of target.get_dedx (reference solver.py:64-66 calls it live). */
double orc_dedx_lookup(const double* tab, double ke) {
  const double e_lo = ldexp(1.0, ORC_DEDX_EMIN);
  const double e_hi = ldexp(1.0, ORC_DEDX_EMAX);
  if (!(ke >= e_lo)) return tab[0];
  if (ke >= e_hi) return tab[ORC_DEDX_NODES - 1];
  int ex;
  double f = frexp(ke, &ex); /* ke = f 2^ex, f in [0.5,1) */
  double sub = (2.0 * f - 1.0) * (double)ORC_DEDX_SUB;
  int j = (int)sub;
  double t = sub - (double)j;
  int i = (ex - 1 - ORC_DEDX_EMIN) * ORC_DEDX_SUB + j;
  return tab[i] + t * (tab[i + 1] - tab[i]);
}

/* detector/solver.py:19-76 */
void orc_equation_of_motion(const double s[6], double bfield, double efield, const orc_det_desc* det,
                            const orc_species_desc* sp, double r[6]) {
  double gv = sqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]);
  double beta = sqrt(gv * gv / (1.0 + gv * gv));
  double gamma = gv / beta;
  double ux = s[3] / gv, uy = s[4] / gv, uz = s[5] / gv;
  double vx = ux * beta * C_LIGHT, vy = uy * beta * C_LIGHT, vz = uz * beta * C_LIGHT;
  double ke = sp->mass * (gamma - 1.0);
  double charge_c = (double)sp->Z * E_CHARGE;
  double mass_kg = sp->mass * MEV_2_KG;
  double q_m = charge_c / mass_kg;
  double decel = (orc_dedx_lookup(sp->dedx, ke) * MEV_2_JOULE * det->density * 100.0) / mass_kg;
  r[0] = vx; r[1] = vy; r[2] = vz;
  r[3] = (q_m * vy * bfield - decel * ux) / C_LIGHT;
  r[4] = (q_m * (-1.0 * vx * bfield) - decel * uy) / C_LIGHT;
  r[5] = (q_m * efield - decel * uz) / C_LIGHT;
}

static double kinetic_energy(const double s[6], double mass) { /* solver.py:116-119, :332-335 */
  double gv = sqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]);
  double beta = sqrt(gv * gv / (1.0 + gv * gv));
  double gamma = gv / beta;
  return mass * (gamma - 1.0);
}

/* detector/solver.py:243-305.  The reference integrates with scipy Radau (rtol 1e-3) and
 * samples the dense output on t_k = k 1e-10 s; the restatement uses classical RK4 on that
 * grid (more accurate than the reference's tolerance) and the same four
 * terminal events with scipy's sign-change semantics (solver.py:80-240, :276-283):
 * a sample is kept iff no event fired in (t_{k-1}, t_k]. */
int32_t orc_generate_trajectory(const orc_det_desc* det, const orc_species_desc* sp,
                                const double vertex[3], const double mom[4], double* track) {
  double s[6] = {vertex[0], vertex[1], vertex[2], mom[0] / sp->mass, mom[1] / sp->mass,
                 mom[2] / sp->mass};
  memcpy(track, s, sizeof s);
  double gv0 = sqrt(s[3] * s[3] + s[4] * s[4] + s[5] * s[5]);
  if (!(gv0 > 0.0) || !isfinite(gv0)) return 1; /* nothing to integrate */
  int nsub = det->ode_substeps > 0 ? det->ode_substeps : 1;
  /* EXTENSION (path_step > 0, no reference counterpart: the reference grid is fixed in time,
   * solver.py:16): a sample every `path_step` metres of arc length, i.e. after the time
   * path_step / v(state at the previous sample), but never coarser than the reference's own
   * 1e-10 s grid (so the end of the range is integrated exactly like the default mode);
   * recording stops with the reference's 1 us window and its 10001-sample cap. */
  const int by_path = det->path_step > 0.0;
  double h_sample = 1.0e-10, t = 0.0;
  double bf = det->bfield * -1.0, ef = det->efield * -1.0; /* solver.py:297-299 */
  double g_ke = kinetic_energy(s, sp->mass) - KE_LIMIT;
  double g_zf = s[2] - 1.0;
  double g_zb = s[2];
  double g_rho = sqrt(s[0] * s[0] + s[1] * s[1]) - 0.292;
  int32_t n = 1;
  for (int k = 1; k < ORC_TIME_SAMPLES; ++k) {
    if (by_path) {
      double gv2 = s[3] * s[3] + s[4] * s[4] + s[5] * s[5];
      double v = C_LIGHT * sqrt(gv2 / (1.0 + gv2));
      h_sample = det->path_step / v;
      if (!(h_sample < 1.0e-10)) h_sample = 1.0e-10;
      if (t + h_sample > 1.0e-6 * (1.0 + 1.0e-9)) break; /* end of the recording window */
    }
    double h = h_sample / (double)nsub;
    int stop = 0;
    for (int sub = 0; sub < nsub && !stop; ++sub) {
      double k1[6], k2[6], k3[6], k4[6], y[6];
      orc_equation_of_motion(s, bf, ef, det, sp, k1);
      for (int i = 0; i < 6; ++i) y[i] = s[i] + 0.5 * h * k1[i];
      orc_equation_of_motion(y, bf, ef, det, sp, k2);
      for (int i = 0; i < 6; ++i) y[i] = s[i] + 0.5 * h * k2[i];
      orc_equation_of_motion(y, bf, ef, det, sp, k3);
      for (int i = 0; i < 6; ++i) y[i] = s[i] + h * k3[i];
      orc_equation_of_motion(y, bf, ef, det, sp, k4);
      for (int i = 0; i < 6; ++i) s[i] = s[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
      double n_ke = kinetic_energy(s, sp->mass) - KE_LIMIT;
      double n_zf = s[2] - 1.0;
      double n_zb = s[2];
      double n_rho = sqrt(s[0] * s[0] + s[1] * s[1]) - 0.292;
      /* scipy find_active_events: up = (g<=0)&(g_new>=0), down = (g>=0)&(g_new<=0) */
      if (g_ke >= 0.0 && n_ke <= 0.0) stop = 1;   /* direction -1 */
      if (g_zf <= 0.0 && n_zf >= 0.0) stop = 1;   /* direction +1 */
      if (g_zb >= 0.0 && n_zb <= 0.0) stop = 1;   /* direction -1 */
      if (g_rho <= 0.0 && n_rho >= 0.0) stop = 1; /* direction +1 */
      if (!(n_ke == n_ke)) stop = 1;               /* NaN state: nothing more to record */
      g_ke = n_ke; g_zf = n_zf; g_zb = n_zb; g_rho = n_rho;
    }
    if (stop) break;
    t += h_sample;
    memcpy(track + (size_t)n * 6, s, sizeof s);
    n++;
  }
  return n;
}

/* detector/solver.py:308-347; the Fano draw of sample k uses Philox index k>>1 in `domain` */
void orc_generate_electrons(const orc_det_desc* det, const orc_species_desc* sp, const double* track,
                            int32_t n_rows, uint64_t seed, uint64_t event, uint32_t domain,
                            int64_t* electrons) {
  double prev = 0.0;
  double scale = 1.0e6 / det->w_value;
  for (int32_t k = 0; k < n_rows; ++k) {
    double e = kinetic_energy(track + (size_t)k * 6, sp->mass);
    double mu = 0.0;
    if (k > 0) mu = fabs(e - prev);
    mu *= scale;
    prev = e;
    /* one Philox call serves two consecutive samples (Box-Muller cos / sin branches) */
    double ua, ub;
    orc_rng_pair(seed, event, (uint32_t)k >> 1, domain, &ua, &ub);
    double rad = sqrt(-2.0 * log(1.0 - ua));
    double z = (k & 1) ? rad * sin(2.0 * PI * ub) : rad * cos(2.0 * PI * ub);
    double draw = mu + sqrt(det->fano_factor * mu) * z;
    electrons[k] = (int64_t)draw; /* dtype=np.int64 cast: truncation toward zero */
  }
}

/* ---- insertion-ordered dictionary (stands in for numba.typed.Dict, simulator.py:93-95) ---- */
struct orc_dict {
  int64_t* keys; int64_t* charge; int64_t* label; int64_t len, cap;
  int64_t* slots; int64_t n_slots; /* open addressing: index+1 into keys, 0 = empty */
};
static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
orc_dict* orc_dict_new(void) {
  orc_dict* d = (orc_dict*)calloc(1, sizeof *d);
  d->cap = 4096;
  d->keys = (int64_t*)malloc(sizeof(int64_t) * d->cap);
  d->charge = (int64_t*)malloc(sizeof(int64_t) * d->cap);
  d->label = (int64_t*)malloc(sizeof(int64_t) * d->cap);
  d->n_slots = 16384;
  d->slots = (int64_t*)calloc(d->n_slots, sizeof(int64_t));
  return d;
}
void orc_dict_free(orc_dict* d) {
  if (!d) return;
  free(d->keys); free(d->charge); free(d->label); free(d->slots); free(d);
}
void orc_dict_clear(orc_dict* d) {
  d->len = 0;
  memset(d->slots, 0, sizeof(int64_t) * d->n_slots);
}
int64_t orc_dict_len(const orc_dict* d) { return d->len; }
void orc_dict_item(const orc_dict* d, int64_t i, int64_t* key, int64_t* charge, int64_t* label) {
  *key = d->keys[i]; *charge = d->charge[i]; *label = d->label[i];
}
static void dict_rehash(orc_dict* d) {
  free(d->slots);
  d->n_slots *= 2;
  d->slots = (int64_t*)calloc(d->n_slots, sizeof(int64_t));
  for (int64_t i = 0; i < d->len; ++i) {
    uint64_t h = mix64((uint64_t)d->keys[i]) & (uint64_t)(d->n_slots - 1);
    while (d->slots[h]) h = (h + 1) & (uint64_t)(d->n_slots - 1);
    d->slots[h] = i + 1;
  }
}
/* charge, _ = points.get(id,(0,0)); charge += q; points[id] = (charge, label) */
static void dict_add(orc_dict* d, int64_t key, int64_t q, int64_t label) {
  uint64_t h = mix64((uint64_t)key) & (uint64_t)(d->n_slots - 1);
  while (d->slots[h]) {
    int64_t i = d->slots[h] - 1;
    if (d->keys[i] == key) { d->charge[i] += q; d->label[i] = label; return; }
    h = (h + 1) & (uint64_t)(d->n_slots - 1);
  }
  if (d->len == d->cap) {
    d->cap *= 2;
    d->keys = (int64_t*)realloc(d->keys, sizeof(int64_t) * d->cap);
    d->charge = (int64_t*)realloc(d->charge, sizeof(int64_t) * d->cap);
    d->label = (int64_t*)realloc(d->label, sizeof(int64_t) * d->cap);
  }
  d->keys[d->len] = key; d->charge[d->len] = q; d->label[d->len] = label;
  d->slots[h] = ++d->len;
  if (d->len * 2 > d->n_slots) dict_rehash(d);
}

/* detector/transporter.py:78-120 with grid_edges = (lut_lo, lut_lo + lut_n, 1.0) */
static int position_to_index(const orc_det_desc* det, double px, double py, int64_t* ix, int64_t* iy) {
  double x = px * 1000.0, y = py * 1000.0;
  double low = (double)det->lut_lo, high = (double)(det->lut_lo + det->lut_n), bin = 1.0;
  if (floor(x) >= high || floor(y) >= high) return 0;
  if (floor(x) < low || floor(y) < low) return 0;
  if (!(x == x) || !(y == y)) return 0; /* NaN position: undefined in the reference; dropped */
  *ix = (int64_t)((floor(x) - low) / bin);
  *iy = (int64_t)((floor(y) - low) / bin);
  return 1;
}

/* detector/transporter.py:11-41 */
static double bivariate_normal_pdf(double px, double py, double mx, double my, double sigma) {
  double c1 = 1.0 / 2.0 / PI / (sigma * sigma);
  double c2 = (-1.0 / 2.0 / (sigma * sigma)) * (((px - mx) * (px - mx)) + ((py - my) * (py - my)));
  return c1 * exp(c2);
}

/* numpy.linspace(lo, hi, 10): arange(10)*step + lo, last element forced to hi */
static void linspace10(double lo, double hi, double out[ORC_MESH_STEPS]) {
  double step = (hi - lo) / (double)(ORC_MESH_STEPS - 1);
  for (int i = 0; i < ORC_MESH_STEPS; ++i) out[i] = (double)i * step + lo;
  out[ORC_MESH_STEPS - 1] = hi;
}

/* detector/transporter.py:123-169 */
static void point_transport(const orc_det_desc* det, double time, double cx, double cy, int64_t electrons,
                            orc_dict* points, int64_t label, double wl) {
  int64_t ix, iy;
  if (!position_to_index(det, cx, cy, &ix, &iy)) return;
  int64_t pad = det->pad_lut[ix * det->lut_n + iy];
  if (pad != -1 && !is_beam_pad(pad)) {
    int64_t tb = (int64_t)time;
    dict_add(points, orc_pair(tb, pad), wl == 1.0 ? electrons : (int64_t)(wl * (double)electrons), label);
  }
}

/* detector/transporter.py:172-249 */
static void transverse_transport(const orc_det_desc* det, double time, double cx, double cy,
                                 int64_t electrons, double sigma_t, orc_dict* points, int64_t label, double wl) {
  double xs[ORC_MESH_STEPS], ys[ORC_MESH_STEPS];
  linspace10(cx - 3.0 * sigma_t, cx + 3.0 * sigma_t, xs);
  linspace10(cy - 3.0 * sigma_t, cy + 3.0 * sigma_t, ys);
  double step_x = 2.0 * 3.0 * sigma_t / (double)(ORC_MESH_STEPS - 1);
  double step_y = 2.0 * 3.0 * sigma_t / (double)(ORC_MESH_STEPS - 1);
  for (int i = 0; i < ORC_MESH_STEPS; ++i) {      /* meshgrid: x outer, y inner :69-73 */
    for (int j = 0; j < ORC_MESH_STEPS; ++j) {
      int64_t ix, iy;
      if (!position_to_index(det, xs[i], ys[j], &ix, &iy)) continue;
      int64_t pad = det->pad_lut[ix * det->lut_n + iy];
      if (pad != -1 && !is_beam_pad(pad)) {
        int64_t tb = (int64_t)time;
        int64_t id = orc_pair(tb, pad);
        int64_t pixel =
            (int64_t)(bivariate_normal_pdf(xs[i], ys[j], cx, cy, sigma_t) * (step_x * step_y) * (wl * (double)electrons));
        dict_add(points, id, pixel, label);
      }
    }
  }
}

/* EXTENSION (no reference counterpart): per-electron Monte-Carlo transverse diffusion.  Every
 * primary electron k of entry `cs` lands at (cx + sigma N_x, cy + sigma N_y), Box-Muller on the
 * Philox pair (index k, domain 0x200 + cs), and adds q electrons to the pad under it. */
static void mc_transport(const orc_det_desc* det, double time, double cx, double cy, int64_t n_prim, int64_t q,
                         double sigma_t, orc_dict* points, int64_t label, uint64_t seed, uint64_t event,
                         uint32_t cs) {
  int64_t tb = (int64_t)time;
  for (int64_t k = 0; k < n_prim; ++k) {
    double ua, ub;
    orc_rng_pair(seed, event, (uint32_t)k, 0x200u + cs, &ua, &ub);
    double rad = sqrt(-2.0 * log(1.0 - ua));
    double x = cx + sigma_t * (rad * cos(2.0 * PI * ub));
    double y = cy + sigma_t * (rad * sin(2.0 * PI * ub));
    int64_t ix, iy;
    if (!position_to_index(det, x, y, &ix, &iy)) continue;
    int64_t pad = det->pad_lut[ix * det->lut_n + iy];
    if (pad != -1 && !is_beam_pad(pad)) dict_add(points, orc_pair(tb, pad), q, label);
  }
}

/* detector/transporter.py:252-317; track rows here are (x, y, time bucket).  seed / event /
 * sample_base (kept samples of the event's earlier nuclei) only matter for the Monte-Carlo
 * extension, whose random streams are keyed by the entry number (sample x slice) in the event. */
static void transport_track_ex(const orc_det_desc* det, const double* xyt, const int64_t* electrons, int32_t n,
                               orc_dict* points, int64_t label, uint64_t seed, uint64_t event,
                               int64_t sample_base) {
  double dv = det->length / (double)(det->windows_edge - det->micromegas_edge);
  const int n_slices = det->longitudinal_diffusion > 0.0 ? 5 : 1;
  for (int32_t i = 0; i < n; ++i) {
    double time = xyt[3 * i + 2];
    double sigma_t = sqrt(2.0 * det->diffusion * dv * time / det->efield);
    if (!(sigma_t == sigma_t)) continue; /* NaN (time < 0) is undefined behaviour in the reference: dropped */
    if (det->longitudinal_diffusion > 0.0) {
      /* EXTENSION: 5 time slices over +-3 sigma_l (numpy.linspace semantics), weight long_weights[s] */
      double sigma_l = sqrt(2.0 * det->longitudinal_diffusion * dv * time / det->efield) / dv;
      double lo = time - 3.0 * sigma_l, hi = time + 3.0 * sigma_l, step = (hi - lo) / 4.0;
      for (int sl = 0; sl < 5; ++sl) {
        double ts = sl == 4 ? hi : (double)sl * step + lo;
        if (!(ts >= 0.0)) continue;
        if (det->mc_diffusion)
          mc_transport(det, ts, xyt[3 * i], xyt[3 * i + 1], electrons[i] / det->mpgd_gain,
                       (int64_t)(det->long_weights[sl] * (double)det->mpgd_gain), sigma_t, points, label, seed, event,
                       (uint32_t)((sample_base + i) * n_slices + sl));
        else if (sigma_t == 0.0) point_transport(det, ts, xyt[3 * i], xyt[3 * i + 1], electrons[i], points, label, det->long_weights[sl]);
        else transverse_transport(det, ts, xyt[3 * i], xyt[3 * i + 1], electrons[i], sigma_t, points, label, det->long_weights[sl]);
      }
      continue;
    }
    if (det->mc_diffusion)
      mc_transport(det, time, xyt[3 * i], xyt[3 * i + 1], electrons[i] / det->mpgd_gain, det->mpgd_gain, sigma_t, points,
                   label, seed, event, (uint32_t)(sample_base + i));
    else if (sigma_t == 0.0) point_transport(det, time, xyt[3 * i], xyt[3 * i + 1], electrons[i], points, label, 1.0);
    else transverse_transport(det, time, xyt[3 * i], xyt[3 * i + 1], electrons[i], sigma_t, points, label, 1.0);
  }
}

void orc_transport_track(const orc_det_desc* det, const double* xyt, const int64_t* electrons, int32_t n,
                         orc_dict* points, int64_t label) {
  transport_track_ex(det, xyt, electrons, n, points, label, 0, 0, 0);
}

/* detector/solver.py:350-413 */
static int32_t generate_point_cloud_ex(const orc_det_desc* det, const orc_species_desc* sp, const double mom[4],
                                      const double vertex[3], uint64_t seed, uint64_t event, int64_t label,
                                      orc_dict* points, double* samples_out, int32_t* n_track_rows,
                                      int64_t sample_base) {
  double* track = (double*)malloc(sizeof(double) * 6 * ORC_TIME_SAMPLES);
  int64_t* electrons = (int64_t*)malloc(sizeof(int64_t) * ORC_TIME_SAMPLES);
  int32_t n = orc_generate_trajectory(det, sp, vertex, mom, track);
  if (n_track_rows) *n_track_rows = n;
  orc_generate_electrons(det, sp, track, n, seed, event, 1u + (uint32_t)label, electrons);
  double dv = det->length / (double)(det->windows_edge - det->micromegas_edge); /* parameters.py:172-174 */
  double* xyt = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  int32_t m = 0;
  for (int32_t k = 0; k < n; ++k) {
    if (electrons[k] >= 1) { /* solver.py:387-389 */
      electrons[m] = electrons[k] * det->mpgd_gain; /* :392 */
      xyt[3 * m] = track[6 * k];
      xyt[3 * m + 1] = track[6 * k + 1];
      xyt[3 * m + 2] = (det->length - track[6 * k + 2]) / dv + (double)det->micromegas_edge; /* :395-398 */
      if (samples_out) {
        samples_out[4 * m] = xyt[3 * m]; samples_out[4 * m + 1] = xyt[3 * m + 1];
        samples_out[4 * m + 2] = xyt[3 * m + 2]; samples_out[4 * m + 3] = (double)electrons[m];
      }
      m++;
    }
  }
  transport_track_ex(det, xyt, electrons, m, points, label, seed, event, sample_base);
  free(track); free(electrons); free(xyt);
  return m;
}

int32_t orc_generate_point_cloud(const orc_det_desc* det, const orc_species_desc* sp, const double mom[4],
                                 const double vertex[3], uint64_t seed, uint64_t event, int64_t label,
                                 orc_dict* points, double* samples_out, int32_t* n_track_rows) {
  return generate_point_cloud_ex(det, sp, mom, vertex, seed, event, label, points, samples_out, n_track_rows, 0);
}

/* detector/simulator.py:52-115 (+ dict_to_points :19-49).  The tb jitter of a point is
 * orc_jitter_uniform(seed, event, tb << 14 | pad). */
int64_t orc_simulate(const orc_det_desc* det, const orc_event_layout* lay, uint64_t seed, uint64_t event,
                     const double* p4, const double* vertex, int64_t capacity, double* points,
                     int64_t* labels, uint64_t* n_track_samples) {
  orc_dict* d = orc_dict_new();
  uint64_t samples = 0;
  for (int32_t i = 0; i < lay->n_sim; ++i) {
    int32_t row = lay->indices[i];
    int32_t sp = lay->species_of_row[row];
    if (sp < 0) continue; /* proton_numbers[idx] == 0, simulator.py:97-98 */
    samples += (uint64_t)generate_point_cloud_ex(det, &det->species[sp], p4 + 4 * row, vertex, seed, event,
                                                 (int64_t)row, d, NULL, NULL, (int64_t)samples);
  }
  if (n_track_samples) *n_track_samples = samples;
  int64_t n_out = 0;
  for (int64_t i = 0; i < d->len; ++i) {
    int64_t tb, pad;
    orc_unpair(d->keys[i], &tb, &pad);
    double tbf = (double)tb;
    if (tb >= 0 && pad >= 0) {
      tbf += orc_jitter_uniform(seed, event, (uint32_t)((tb << 14) | pad)); /* simulator.py:108 */
    }
    if (0.0 <= tbf && tbf < (double)ORC_NUM_TB) { /* simulator.py:111-113 */
      if (n_out < capacity) {
        points[3 * n_out] = (double)pad;
        points[3 * n_out + 1] = tbf;
        points[3 * n_out + 2] = (double)d->charge[i];
        labels[n_out] = d->label[i];
      }
      n_out++;
    }
  }
  orc_dict_free(d);
  return n_out <= capacity ? n_out : -n_out;
}

int64_t orc_sim_batch(const orc_kin_desc* kin, const orc_det_desc* det, const orc_event_layout* lay,
                      uint64_t seed, uint64_t first, uint64_t n, double* p4, double* vertex, int32_t* status,
                      int64_t capacity, int64_t* offsets, double* points, int64_t* labels, uint64_t* stats,
                      int32_t n_threads) {
  int32_t n_rows = lay->n_rows;
  int64_t per_event_cap = 1 << 19;
  int64_t* counts = (int64_t*)calloc(n + 1, sizeof(int64_t));
  double** ev_pts = (double**)calloc(n, sizeof(double*));
  int64_t** ev_lab = (int64_t**)calloc(n, sizeof(int64_t*));
  uint64_t tot_samples = 0;
  stats[0] = stats[1] = stats[2] = stats[3] = 0;
  (void)n_threads;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
  {
    double* tp = (double*)malloc(sizeof(double) * 3 * per_event_cap);
    int64_t* tl = (int64_t*)malloc(sizeof(int64_t) * per_event_cap);
    double lp4[ORC_MAX_ROWS * 4], lv[3];
#pragma omp for schedule(dynamic, 4) reduction(+ : tot_samples)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
      uint64_t ev = first + (uint64_t)i;
      double* ep = p4 ? p4 + i * n_rows * 4 : lp4;
      double* evx = vertex ? vertex + i * 3 : lv;
      int32_t st = 0;
      if (kin) st = orc_kin_event(kin, seed, ev, ep, evx, NULL);
      if (status) status[i] = st;
      int64_t c = 0;
      uint64_t ns = 0;
      if (st == 0) {
        c = orc_simulate(det, lay, seed, ev, ep, evx, per_event_cap, tp, tl, &ns);
        if (c < 0) c = 0; /* > 65536 points in one event: not representable here */
      }
      tot_samples += ns;
      counts[i] = c;
      if (points && c > 0) {
        ev_pts[i] = (double*)malloc(sizeof(double) * 3 * c);
        ev_lab[i] = (int64_t*)malloc(sizeof(int64_t) * c);
        memcpy(ev_pts[i], tp, sizeof(double) * 3 * c);
        memcpy(ev_lab[i], tl, sizeof(int64_t) * c);
      } else if (c > 0) { /* stats only: fold checksums here */
        uint64_t cs = 0, ks = 0;
        for (int64_t k = 0; k < c; ++k) {
          cs += (uint64_t)(int64_t)tp[3 * k + 2];
          ks += (ev << 24) + ((uint64_t)(int64_t)floor(tp[3 * k + 1]) << 14) + (uint64_t)(int64_t)tp[3 * k];
        }
#pragma omp critical
        { stats[2] += cs; stats[3] += ks; }
      }
    }
    free(tp); free(tl);
  }
  int64_t total = 0;
  for (uint64_t i = 0; i < n; ++i) { if (offsets) offsets[i] = total; total += counts[i]; }
  if (offsets) offsets[n] = total;
  stats[0] = (uint64_t)total;
  stats[1] = tot_samples;
  int64_t ret = total;
  if (points) {
    if (total > capacity) ret = -total;
    else {
      int64_t off = 0;
      for (uint64_t i = 0; i < n; ++i) {
        if (counts[i] > 0) {
          memcpy(points + 3 * off, ev_pts[i], sizeof(double) * 3 * counts[i]);
          memcpy(labels + off, ev_lab[i], sizeof(int64_t) * counts[i]);
          for (int64_t k = 0; k < counts[i]; ++k) {
            stats[2] += (uint64_t)(int64_t)ev_pts[i][3 * k + 2];
            stats[3] += ((first + i) << 24) + ((uint64_t)(int64_t)floor(ev_pts[i][3 * k + 1]) << 14) +
                        (uint64_t)(int64_t)ev_pts[i][3 * k];
          }
          off += counts[i];
        }
      }
    }
    for (uint64_t i = 0; i < n; ++i) { free(ev_pts[i]); free(ev_lab[i]); }
  }
  free(counts); free(ev_pts); free(ev_lab);
  return ret;
}

/* ------------------------------------------------------- "next": response -------- */
/* detector/response.py:8-32 */
void orc_get_response(double clock_freq, double amp_gain, double shaping_time, double* response) {
  double c1 = 4095.0 * E_CHARGE / amp_gain / 1e-15;
  double step = ((double)ORC_NUM_TB - 0.0) / (double)(ORC_NUM_TB - 1); /* linspace(0,512,512) */
  for (int i = 0; i < ORC_NUM_TB; ++i) {
    double tb = (i == ORC_NUM_TB - 1) ? (double)ORC_NUM_TB : (double)i * step + 0.0;
    double c2 = tb / (shaping_time * clock_freq * 0.001);
    double r = c1 * exp(-3.0 * c2) * (c2 * c2 * c2) * sin(c2);
    response[i] = r < 0.0 ? 0.0 : r;
  }
}
/* detector/response.py:35-57 */
void orc_apply_response(const double* response, double electrons, double* amp, double* integral) {
  double mx = -INFINITY, sum = 0.0;
  for (int i = 0; i < ORC_NUM_TB; ++i) {
    double v = response[i] * electrons;
    if (v > 4095.0) v = 4095.0;
    if (v > mx) mx = v;
    sum += v;
  }
  *amp = mx; *integral = sum;
}
/* detector/writer.py:61-112 */
void orc_convert_to_spyral(const double* points, int64_t n, int32_t window_edge, int32_t mm_edge,
                           double length, const double* response, const double* pad_centers,
                           const double* pad_sizes, double* rows) {
  for (int64_t i = 0; i < n; ++i) {
    const double* pt = points + 3 * i;
    int64_t pad = (int64_t)pt[0];
    double amp, integral;
    orc_apply_response(response, pt[2], &amp, &integral);
    double* r = rows + 8 * i;
    r[0] = pad_centers[2 * pad]; r[1] = pad_centers[2 * pad + 1];
    r[2] = ((double)window_edge - pt[1]) / (double)(window_edge - mm_edge) * length * 1000.0;
    r[3] = amp; r[4] = integral; r[5] = pt[0]; r[6] = pt[1]; r[7] = pad_sizes[pad];
  }
}
