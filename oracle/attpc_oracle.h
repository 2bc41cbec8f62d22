/*
 * attpc_oracle.h -- CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call
 * this library; the product (attpc_engine_amd + libattpc_hip.so) never does.
 *
 * The descriptor structs are declared here independently of include/attpc_engine.h
 * but with the same field layout on purpose, so that a test can hand the *same
 * bytes* to the oracle and to the HIP library.
 *
 * Parity pinning (see DESIGN.md "Oracle"): kinematics pinned by the reference's
 * LISE known-answer test and by golden vectors generated from the reference's own
 * reaction.py; pad-plane transport pinned by golden vectors generated from the
 * reference's own transporter.py/pairing.py/simulator.py; track integration is a
 * fixed-grid RK4 checked against golden scipy-Radau tracks produced through the
 * reference's solver.py within a stated physical tolerance.  Third-party
 * arithmetic absent from the reference tree (vector 1.6.0, spyral-utils 2.0.0 /
 * pycatima 1.96) is "parity unpinned" beyond the LISE value.
 */
#ifndef ATTPC_ORACLE_H
#define ATTPC_ORACLE_H
#include <stdint.h>

#define ORC_MAX_STEPS 8
#define ORC_MAX_ROWS (4 + 2 * (ORC_MAX_STEPS - 1))
#define ORC_MAX_SPECIES 16
#define ORC_MAX_SIM 8
#define ORC_DEDX_EMIN (-30)
#define ORC_DEDX_EMAX 14
#define ORC_DEDX_SUB 32
#define ORC_DEDX_NODES ((ORC_DEDX_EMAX - ORC_DEDX_EMIN) * ORC_DEDX_SUB + 1)
#define ORC_NUM_TB 512
#define ORC_TIME_SAMPLES 10001
#define ORC_MESH_STEPS 10

typedef struct {
  int32_t kind, table_len;
  double p0, p1, p2;
  const double* table_x;
  const double* table_cdf;
} orc_excitation_desc;

typedef struct {
  int32_t kind, table_len;
  double cos_min, cos_max, bin_width;
  const double* angles;
  const double* cdf;
} orc_polar_desc;

typedef struct {
  int32_t n_steps, sample_limit;
  double beam_energy;
  double masses[ORC_MAX_ROWS];
  orc_excitation_desc excitation[ORC_MAX_STEPS];
  orc_polar_desc polar[ORC_MAX_STEPS];
  int32_t has_target, eloss_len;
  double rho_sigma, z_min, z_max;
  const double* eloss;
} orc_kin_desc;

typedef struct {
  int32_t Z, A;
  double mass;
  const double* dedx;
} orc_species_desc;

typedef struct {
  double length, efield, bfield, density, diffusion, fano_factor, w_value;
  int64_t mpgd_gain;
  int32_t micromegas_edge, windows_edge;
  const int16_t* pad_lut; /* the oracle takes the UNFOLDED lut + the beam pad list */
  int32_t lut_n, lut_lo;
  int32_t n_species, ode_substeps;
  orc_species_desc species[ORC_MAX_SPECIES];
  double longitudinal_diffusion; /* extension, 0 = reference behaviour */
  double long_weights[5];
  int32_t mc_diffusion;          /* extension: per-electron Monte-Carlo transverse diffusion */
  int32_t reserved_ext;
  double path_step;              /* extension: track sample every path_step metres (0 = 1e-10 s grid) */
} orc_det_desc;

typedef struct {
  int32_t n_rows, n_sim;
  int32_t indices[ORC_MAX_SIM];
  int32_t species_of_row[ORC_MAX_ROWS];
} orc_event_layout;

/* RNG */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], int32_t rounds, uint32_t out[4]);
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void orc_philox2x32(const uint32_t ctr_in[2], uint32_t key, int32_t rounds, uint32_t out[2]);
double orc_jitter_uniform(uint64_t seed, uint64_t event, uint32_t key24);
void orc_rng_pair(uint64_t seed, uint64_t event, uint32_t index, uint32_t domain, double* u_a,
                  double* u_b);
double orc_rng_normal(uint64_t seed, uint64_t event, uint32_t index, uint32_t domain);

/* kinematics */
int32_t orc_reaction_allowed(const double m[4], double projectile_energy, double excitation);
int32_t orc_reaction_calculate(const double m[4], double projectile_energy, double polar,
                               double azimuthal, double excitation, double out[4][4]);
int32_t orc_decay_allowed(const double parent[4], double m1, double m2, double excitation);
int32_t orc_decay_calculate(const double parent[4], double m1, double m2, double polar,
                            double azimuthal, double excitation, double out[2][4]);
double orc_sample_excitation(const orc_excitation_desc* d, double u_a, double u_b);
double orc_sample_polar(const orc_polar_desc* d, double u_a, double u_b);
int32_t orc_kin_event(const orc_kin_desc* d, uint64_t seed, uint64_t event, double* p4,
                      double* vertex, uint32_t* attempts);
void orc_kin_batch(const orc_kin_desc* d, uint64_t seed, uint64_t first_event, uint64_t n,
                   double* p4, double* vertex, int32_t* status, uint32_t* attempts,
                   int32_t n_threads);

/* detector */
void orc_set_beam_pads(const int32_t* pads, int32_t n);
int64_t orc_pair(int64_t tb, int64_t pad);
void orc_unpair(int64_t id, int64_t* tb, int64_t* pad);
double orc_dedx_lookup(const double* table, double ke);
void orc_equation_of_motion(const double state[6], double bfield, double efield,
                            const orc_det_desc* det, const orc_species_desc* sp, double out[6]);
/* track [<=ORC_TIME_SAMPLES][6]; returns number of rows */
int32_t orc_generate_trajectory(const orc_det_desc* det, const orc_species_desc* sp,
                                const double vertex[3], const double momentum[4], double* track);
void orc_generate_electrons(const orc_det_desc* det, const orc_species_desc* sp,
                            const double* track, int32_t n_rows, uint64_t seed, uint64_t event,
                            uint32_t domain, int64_t* electrons);

/* insertion-ordered (key -> charge,label) dictionary */
typedef struct orc_dict orc_dict;
orc_dict* orc_dict_new(void);
void orc_dict_free(orc_dict* d);
void orc_dict_clear(orc_dict* d);
int64_t orc_dict_len(const orc_dict* d);
void orc_dict_item(const orc_dict* d, int64_t i, int64_t* key, int64_t* charge, int64_t* label);

void orc_transport_track(const orc_det_desc* det, const double* track_xyt /*[n][3] x,y,time*/,
                         const int64_t* electrons, int32_t n, orc_dict* points, int64_t label);
/* samples out (optional): rows (x, y, time bucket, electrons*gain) of kept samples */
int32_t orc_generate_point_cloud(const orc_det_desc* det, const orc_species_desc* sp,
                                 const double momentum[4], const double vertex[3], uint64_t seed,
                                 uint64_t event, int64_t label, orc_dict* points,
                                 double* samples_out, int32_t* n_track_rows);
/* returns number of points written (<= capacity) or -(needed) when capacity is too small */
int64_t orc_simulate(const orc_det_desc* det, const orc_event_layout* lay, uint64_t seed,
                     uint64_t event, const double* p4, const double* vertex, int64_t capacity,
                     double* points, int64_t* labels, uint64_t* n_track_samples);
/* fused batch for timing / bulk parity: offsets[n+1] CSR (events in order) */
int64_t orc_sim_batch(const orc_kin_desc* kin, const orc_det_desc* det, const orc_event_layout* lay,
                      uint64_t seed, uint64_t first_event, uint64_t n, double* p4, double* vertex,
                      int32_t* status, int64_t capacity, int64_t* offsets, double* points,
                      int64_t* labels, uint64_t* stats /*[4]: points, samples, chargesum, keysum*/,
                      int32_t n_threads);

/* "next" rows: response + spyral conversion */
void orc_get_response(double clock_freq, double amp_gain, double shaping_time, double* response);
void orc_apply_response(const double* response, double electrons, double* amp, double* integral);
void orc_convert_to_spyral(const double* points, int64_t n, int32_t window_edge, int32_t mm_edge,
                           double length, const double* response, const double* pad_centers,
                           const double* pad_sizes, double* rows);
#endif
