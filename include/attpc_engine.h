/*
 * attpc_engine.h -- C ABI of the MI355X-native AT-TPC Monte-Carlo hot path.
 *
 * The reference (ATTPC/attpc_engine, pure Python) has no FFI layer; its operator
 * boundary for the hot path is a set of Python call signatures.  Each entry point
 * below names the reference interface it replaces (paths relative to the
 * reference's src/attpc_engine/).  The Python package `attpc_engine_amd` binds
 * these symbols with ctypes (attpc_engine_amd/_abi.py) and re-exposes the
 * reference's own class/function names on top.  INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C, no exceptions cross the boundary; every call returns an int32
 *     status (ATTPC_OK == 0); attpc_last_error(ctx) gives a ctx-owned string.
 *   - the caller owns every host buffer; the library owns device memory, freed in
 *     attpc_ctx_destroy.  A ctx is single-threaded; distinct ctxs (one per GPU,
 *     one per process in the multi-GPU bench) are independent.
 *   - all floating point is IEEE binary64 ("f64"), charges are int64.
 *   - random numbers: Philox4x32-10, key = seed, counter = (global event id, draw index, domain); the
 *     time-bucket jitter of a cloud point is Philox2x32-7 with counter = (event[31:0],
 *     event[39:32] << 24 | tb << 14 | pad) and key word = seed[31:0] ^ rotl(seed[63:32], 13) ^ 0x100
 *     -- every draw a pure function of (seed, global event id, ...): results do not depend on
 *     batch / chunk / GPU count.
 */
#ifndef ATTPC_ENGINE_H
#define ATTPC_ENGINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ATTPC_ABI_VERSION 3
#define ATTPC_API __attribute__((visibility("default")))

/* status codes */
#define ATTPC_OK 0
#define ATTPC_E_INVALID 1       /* bad argument / descriptor                                */
#define ATTPC_E_NODEVICE 2      /* no HIP device                                            */
#define ATTPC_E_HIP 3           /* a HIP runtime call failed (see attpc_last_error)         */
#define ATTPC_E_CAPACITY 4      /* caller buffer too small; required size in stats          */
#define ATTPC_E_NOTCONFIGURED 5 /* run called before the matching configure                 */
#define ATTPC_E_DATALOSS 6      /* the run finished but stats.n_failed or stats.n_inconsistent is
                                   not 0: part of an event's charge is missing from the cloud     */

#define ATTPC_MAX_STEPS 8    /* 1 Reaction + up to 7 Decays                                 */
#define ATTPC_MAX_ROWS (4 + 2 * (ATTPC_MAX_STEPS - 1)) /* nuclei (rows) per event           */
#define ATTPC_MAX_SPECIES 16
#define ATTPC_MAX_SIM 8      /* max simulated nuclei per event (len(indices))               */

/* stopping-power table: nodes on a "binade" grid, E = 2^e (1 + m/32) MeV,
 * e in [ATTPC_DEDX_EMIN, ATTPC_DEDX_EMAX), m in [0,32); one extra closing node. */
#define ATTPC_DEDX_EMIN (-30)
#define ATTPC_DEDX_EMAX 14
#define ATTPC_DEDX_SUB 32
#define ATTPC_DEDX_NODES ((ATTPC_DEDX_EMAX - ATTPC_DEDX_EMIN) * ATTPC_DEDX_SUB + 1)

#define ATTPC_NUM_TB 512        /* detector/constants.py:23                                 */
#define ATTPC_TIME_SAMPLES 10001 /* detector/solver.py:16  TIME_STEPS                       */
#define ATTPC_MESH_STEPS 10     /* detector/transporter.py:8  STEPS                         */
#define ATTPC_LONG_STEPS 5      /* time slices of the longitudinal-diffusion extension       */

/* excitation sampler kinds -- kinematics/excitation.py */
#define ATTPC_EX_GAUSSIAN 0 /* p0 = centroid, p1 = sigma (= FWHM/2.355)         :32-80      */
#define ATTPC_EX_UNIFORM 1  /* p0 = min, p1 = max                               :83-128     */
#define ATTPC_EX_TABLE 2    /* inverse-CDF table: x[table_len], cdf[table_len]; value=x-p0.
                               ExcitationBreitWigner (:131-188) is configured this way
                               (x = total energy, p0 = rest mass).                          */
/* polar sampler kinds -- kinematics/angle.py */
#define ATTPC_POLAR_UNIFORM 0   /* uniform in cos(theta) in [cos_min, cos_max]  :35-80      */
#define ATTPC_POLAR_ARBITRARY 1 /* choice(angles, p) + U*bin_width              :83-152     */

typedef struct attpc_excitation_desc {
  int32_t kind;
  int32_t table_len;
  double p0, p1, p2;
  const double* table_x;   /* [table_len] (ATTPC_EX_TABLE) */
  const double* table_cdf; /* [table_len], nondecreasing, last == 1 */
} attpc_excitation_desc;

typedef struct attpc_polar_desc {
  int32_t kind;
  int32_t table_len;
  double cos_min, cos_max;
  double bin_width;
  const double* angles; /* [table_len] lower bin edges, radians */
  const double* cdf;    /* [table_len] cumulative probabilities (numpy choice semantics) */
} attpc_polar_desc;

/* replaces KinematicsPipeline.__init__ state, kinematics/pipeline.py:125-185 */
typedef struct attpc_kin_desc {
  int32_t n_steps;      /* 1 + number of decays */
  int32_t sample_limit; /* event_sample_limit, pipeline.py:132 */
  double beam_energy;   /* MeV */
  /* nuclear masses (MeV) in result-row order: target, projectile, ejectile, residual,
     then (residual_1, residual_2) per decay -- pipeline.py:398-406 */
  double masses[ATTPC_MAX_ROWS];
  attpc_excitation_desc excitation[ATTPC_MAX_STEPS];
  attpc_polar_desc polar[ATTPC_MAX_STEPS];
  /* KinematicsTargetMaterial, pipeline.py:16-36 / :245-264 */
  int32_t has_target;
  int32_t eloss_len;   /* nodes of the beam energy-loss table */
  double rho_sigma;    /* m */
  double z_min, z_max; /* m */
  const double* eloss; /* [eloss_len] energy loss (MeV) of the projectile at beam_energy after
                          path z_min + i (z_max - z_min)/(eloss_len - 1) */
} attpc_kin_desc;

typedef struct attpc_species_desc {
  int32_t Z;
  int32_t A;
  double mass;        /* ground-state nuclear mass, MeV (solver.py:273) */
  const double* dedx; /* [ATTPC_DEDX_NODES] MeV/(g/cm^2) on the binade grid */
} attpc_species_desc;

/* replaces Config / DetectorParams / ElectronicsParams, detector/parameters.py:10-174 */
typedef struct attpc_det_desc {
  double length;   /* m */
  double efield;   /* V/m */
  double bfield;   /* T */
  double density;  /* g/cm^3, target.density (solver.py:65) */
  double diffusion;   /* V */
  double fano_factor;
  double w_value;  /* eV */
  int64_t mpgd_gain;
  int32_t micromegas_edge; /* time buckets */
  int32_t windows_edge;
  /* pad look-up at whole-millimetre pitch (transporter.py:78-120 floors to mm first).
     pad_lut[ix * lut_n + iy] for floor(x_mm) = lut_lo + ix; -1 = no pad.  Beam pads
     (detector/beam_pads.py) must already be folded to -1 by the caller. */
  const int16_t* pad_lut;
  int32_t lut_n;
  int32_t lut_lo;
  int32_t n_species;
  int32_t ode_substeps; /* RK4 sub-steps per 1e-10 s output sample; 0 -> default (1) */
  attpc_species_desc species[ATTPC_MAX_SPECIES];
  /* EXTENSION (not in the reference, which has no longitudinal diffusion -- docs/user_guide/
     detector/index.md:130-133): > 0 spreads every sample over ATTPC_LONG_STEPS time slices,
     linspace(t - 3 sigma_l, t + 3 sigma_l), sigma_l = sqrt(2 D_l dv t / E) / dv time buckets,
     slice s carrying the fraction long_weights[s] (1-D Gaussian pdf x slice pitch) of each
     pixel: electrons = int(pdf h^2 * (long_weights[s] * n)).  0 = reference behaviour. */
  double longitudinal_diffusion; /* V */
  double long_weights[5];
  /* EXTENSION (north star: "stochastic per-electron diffusion instead of the deterministic 10x10
     mesh"): != 0 moves every primary electron k of an entry (sample x slice) by its own Gaussian
     step, x = x0 + sigma_t N_x, y = y0 + sigma_t N_y (Box-Muller on the Philox pair with index k in
     domain 0x200 + entry number), and adds int(w_slice * gain) electrons to the pad it lands on.
     0 = the reference's mesh. */
  int32_t mc_diffusion;
  int32_t reserved_ext;
  /* EXTENSION (BASELINE configs[4] "0.1 mm dE/dx step"; the reference grid is fixed in time,
     detector/solver.py:16,285-303): > 0 records a track sample -- and creates its electrons from the
     energy lost since the previous sample, solver.py:338-346 -- every `path_step` metres of arc length:
     sample k+1 follows sample k after the time min(path_step / v_k, 1e-10 s) (v_k = speed at sample k;
     never coarser than the reference grid, so the end of the range is integrated as in the default
     mode), each interval integrated with ode_substeps RK4 steps; recording ends with the reference's
     1 us window or after ATTPC_TIME_SAMPLES samples.  0 = the reference's 1e-10 s grid. */
  double path_step; /* m */
} attpc_det_desc;

/* which rows of an event are simulated, detector/simulator.py:96-101,157-158 */
typedef struct attpc_event_layout {
  int32_t n_rows;                          /* nuclei per event (rows of the kinematics result) */
  int32_t n_sim;                           /* len(indices) */
  int32_t indices[ATTPC_MAX_SIM];          /* rows to simulate, in order */
  int32_t species_of_row[ATTPC_MAX_ROWS];  /* index into det_desc.species, -1 => skip (Z == 0) */
} attpc_event_layout;

/* Host output buffers for a run (any pointer may be NULL => that output stays on the device). */
typedef struct attpc_cloud_out {
  int64_t capacity;   /* rows available in points/labels */
  int64_t* offsets;   /* [n_events + 1] CSR offsets into points/labels */
  double* points;     /* [capacity, 3] rows (pad, time bucket, electrons) -- simulator.py:40-46 */
  int64_t* labels;    /* [capacity] row index of the nucleus that last touched the point */
  int64_t* event_points; /* [n_events] or NULL: cloud rows of every event BEFORE any threshold
                            (simulator.py:204-205 decides "empty event" on this count) */
} attpc_cloud_out;

typedef struct attpc_run_stats {
  uint64_t n_events;
  uint64_t n_points;          /* cloud rows produced */
  uint64_t n_track_samples;   /* track samples with >= 1 electron (scatter work items) */
  uint64_t n_sample_limit;    /* events that hit event_sample_limit */
  uint64_t n_lds_overflow;    /* windows redone with a smaller time-bucket range (LDS table too full) */
  uint64_t n_failed;          /* events that lost a time bucket (more than 65 536 lone buckets in one launch);
                                 must be 0 -- the run then returns ATTPC_E_DATALOSS */
  uint64_t charge_checksum;   /* sum of all charges mod 2^64 */
  uint64_t key_checksum;      /* sum over points of (event*2^24 + tb*2^14 + pad) mod 2^64 */
  double ms_kinematics;       /* device time of each kernel family (HIP events on the ctx stream) */
  double ms_tracks;
  double ms_scatter;
  uint32_t launches_kinematics;
  uint32_t launches_tracks;
  uint32_t launches_scatter;
  uint32_t n_inconsistent;    /* self-check: flushed windows whose occupied-slot count differed from the
                                 number of claimed keys; must be 0 */
  uint64_t n_lone_buckets;    /* time buckets that alone exceeded the LDS table and went through the
                                 direct-mapped table of lone_bucket_kernel (complete, just slower) */
  uint64_t n_buffer_growths;  /* device buffers (re)allocated during this run; 0 once the sizes have settled */
  uint64_t n_tracks_capped;   /* path-length dE/dx step only: tracks that reached ATTPC_TIME_SAMPLES samples before the
                                 end of the 1 us recording window and were cut there (about 1 m of arc length at a
                                 0.1 mm step); always 0 on the reference's time grid, whose 10 001st sample IS 1 us */
  uint64_t device_bytes;      /* device memory the context holds at the end of the run (buffers it allocated) */
} attpc_run_stats;

typedef struct attpc_ctx attpc_ctx;

ATTPC_API int32_t attpc_version(void);
ATTPC_API int32_t attpc_device_count(void);
/* device >= 0: that HIP device.  There is no CPU fallback: without a device -> ATTPC_E_NODEVICE. */
ATTPC_API int32_t attpc_ctx_create(int32_t device, attpc_ctx** out);
ATTPC_API int32_t attpc_ctx_destroy(attpc_ctx* ctx);
ATTPC_API const char* attpc_last_error(const attpc_ctx* ctx);
/* events processed per internal chunk (device working set scales with it); 0 -> default */
ATTPC_API int32_t attpc_set_chunk_events(attpc_ctx* ctx, int32_t chunk_events);
ATTPC_API int32_t attpc_sync(attpc_ctx* ctx);
/* Tuning / test switches of a context (no environment variables are read by the library):
 *   "scatter_variant"  0 = automatic, 1 = always the two-workgroups-per-CU build, 2 = always the
 *                      one-workgroup build of the scatter kernel, 3 = always the build with u64 sums per table
 *                      slot (the automatic choice switches to it when u32 sums -- 4.3e9 electrons per pad and
 *                      time bucket -- turn out too small for the detector at hand)
 *   "tiny_buffers"     != 0: the next buffers are allocated far too small (exercises the
 *                      grow-and-rerun path in tests)
 *   "compact_transfer" 2 (default): delivered clouds cross PCIe as 8-byte records (attpc_unpack_rows8: the host
 *                      regenerates the time-bucket jitter), 1: as 16-byte records (attpc_unpack_rows), into
 *                      library-owned pinned staging, and are expanded into the caller's arrays by host threads; a chunk
 *                      with a row that does not fit the record falls back to the next wider form;
 *                      0: rows are copied in the reference's dtypes (32 bytes) straight into the caller's arrays.
 *                      Spyral rows: != 0 = 24-byte records (attpc_unpack_spyral_rows)
 *   "unpack_threads"   host threads of that expansion; 0 (default) = min(32, the CPUs the process may use: affinity
 *                      mask and control-group quota honoured, half of them on a machine with 64 or more)
 *   "deliver_chunk_events"  events per chunk when clouds are delivered (default 8192: a chunk's copy hides the next
 *                      chunk's scatter and assembly; the first chunk's device work and the last chunk's expansion
 *                      stand alone, so smaller chunks shorten a short call)
 *   "serial_tracks"    1: kinematics + track integration of the next batch run on the scatter stream, behind the
 *                      current batch's scatter launches; 0: beside them on a low-priority stream; -1 (default):
 *                      behind on the reference's time grid (beside is 3 % slower there: both kernels are issue
 *                      bound), beside with the path-length dE/dx step (12 % faster there)
 *   "track_species_major"  1 (default): the track kernel takes its tracks nucleus by nucleus, lightest species first
 *                      (its lanes then run dry on the short tracks of the heavy ions); 0: event by event.  Scheduling
 *                      only: results do not depend on it
 *   "first_batch_chunks"  > 0: the first track batch of a call spans at most this many scatter chunks (experiment)
 *   "scatter_merge"    -1 (default) = automatic, 0 = never, 1 = always use the scatter kernel's merge variant, which adds
 *                      up the pixel charges of consecutive track samples that fall on the same pad in the same time
 *                      bucket before the table sees them (same results; automatic = with the path-length dE/dx step,
 *                      attpc_det_desc.path_step > 0, where samples are far closer than a pad)
 *   "chunk_events"     as attpc_set_chunk_events */
ATTPC_API int32_t attpc_set_option(attpc_ctx* ctx, const char* name, int64_t value);
/* Page-locked host memory for output buffers (point clouds are PCIe bound on their way to the host:
 * copies into pinned memory run at the link rate, copies into pageable memory at a fraction of it).
 * Plain memory otherwise: the caller reads/writes it freely and returns it with attpc_host_free. */
ATTPC_API int32_t attpc_host_alloc(attpc_ctx* ctx, uint64_t bytes, void** out);
ATTPC_API int32_t attpc_host_free(attpc_ctx* ctx, void* ptr);
/* The 16-byte transfer record of a cloud row and its expansion (host only, no device, no context):
 *   bytes 0..7   f64  time bucket + jitter (column 1 of the row, as it is)
 *   bytes 8..15  u64  electrons (bits 0..44) | pad << 45 (14 bits) | label << 59 (5 bits)
 * -> points[r] = (pad, time bucket, electrons) as f64, labels[r] as i64 (detector/simulator.py:40-46). */
ATTPC_API int32_t attpc_unpack_rows(const void* packed, int64_t n_rows, double* points, int64_t* labels, int32_t n_threads);
/* The 8-byte transfer record of a cloud row ("compact_transfer" 2, the default) and its expansion, host only.  The
 * jitter of a cloud point is a pure function of (seed, global event id, time bucket, pad) -- Philox2x32-7, see the
 * conventions at the top -- so it does not cross the link: the host regenerates it.
 *   u64  electrons (bits 0..35) | time bucket << 36 (9 bits) | pad << 45 (14 bits) | label << 59 (5 bits)
 * packed [n_rows] holds the rows of events first_event .. first_event + n_events - 1 in event order, offsets
 * [n_events + 1] their CSR offsets (offsets[n_events] - offsets[0] == n_rows)
 * -> points[r] = (pad, time bucket + jitter, electrons) as f64, labels[r] as i64: bit-identical to the rows the device
 * writes (detector/simulator.py:40-46, :108). */
ATTPC_API int32_t attpc_unpack_rows8(const void* packed, int64_t n_rows, const int64_t* offsets, int64_t n_events,
                                     uint64_t seed, uint64_t first_event, double* points, int64_t* labels, int32_t n_threads);
/* The 24-byte transfer record of a Spyral row (attpc_sim_run_spyral with "compact_transfer") and its expansion
 * to the 8 columns of convert_to_spyral (detector/writer.py:61-112), host only:
 *   bytes 0..7 f64 time bucket + jitter, 8..15 u64 electrons | pad << 45 | label << 59, 16..23 f64 clipped integral
 * -> x, y = pad_centers[pad], z = (windows_edge - tb) / (windows_edge - micromegas_edge) * length * 1000,
 *    amplitude = min(r_max * electrons, 4095) with r_max the largest response sample, integral, pad, tb,
 *    pad_sizes[pad]; labels[r] as i64. */
ATTPC_API int32_t attpc_unpack_spyral_rows(const void* packed, int64_t n_rows, const double* pad_centers,
                                           const double* pad_sizes, int32_t n_pads, double r_max, int32_t windows_edge,
                                           int32_t micromegas_edge, double length, double* rows, int64_t* labels,
                                           int32_t n_threads);

/* KinematicsPipeline(...) state -> device.  kinematics/pipeline.py:125-185 */
ATTPC_API int32_t attpc_kin_configure(attpc_ctx* ctx, const attpc_kin_desc* desc);
/* n x KinematicsPipeline.run(), kinematics/pipeline.py:285-388 (sample :232-283,
 * Reaction/Decay.calculate reaction.py:103-178,252-303).
 * p4 [n, n_rows, 4] (px,py,pz,E MeV), vertex [n,3] m, status [n] (0 ok, 1 sample limit),
 * attempts [n]; each may be NULL. */
ATTPC_API int32_t attpc_kin_run(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                      double* p4, double* vertex, int32_t* status, uint32_t* attempts);

/* Deterministic map "sampled parameters -> 4-vectors" for n parameter sets, using the
 * configured masses / n_steps: Reaction.calculate + is_excitation_allowed
 * (kinematics/reaction.py:70-178) followed by Decay.* (:230-303) per decay.
 * beam_energy [n]; ex, polar, azim [n, n_steps]; p4 [n, n_rows, 4];
 * status [n]: 0 ok, k+1 = step k not energetically allowed (rows from step k on are NaN),
 * -1 = reaction allowed but below the non-relativistic threshold formula of
 * Reaction.calculate (reaction.py:136-143, where the reference raises ValueError),
 * -2 = both not allowed and below that threshold. */
ATTPC_API int32_t attpc_kin_calculate(attpc_ctx* ctx, uint64_t n, const double* beam_energy,
                            const double* ex, const double* polar, const double* azim,
                            double* p4, int32_t* status);

/* Decay.is_excitation_allowed + Decay.calculate for explicit parent 4-vectors
 * (kinematics/reaction.py:230-303): parent [n,4]; ex, polar, azim [n];
 * out [n,2,4] = residual_1, residual_2; status [n]: 0 ok, 1 not allowed (rows NaN). */
ATTPC_API int32_t attpc_decay_calculate(attpc_ctx* ctx, uint64_t n, const double* parent, double mass_1,
                              double mass_2, const double* ex, const double* polar,
                              const double* azim, double* out, int32_t* status);

/* Config(...) + nuclei table -> device.  detector/parameters.py:145-174 */
ATTPC_API int32_t attpc_det_configure(attpc_ctx* ctx, const attpc_det_desc* desc);
/* n x simulate(), detector/simulator.py:52-115 (generate_point_cloud solver.py:350-413,
 * transport_track transporter.py:252-317, dict_to_points simulator.py:19-49).
 * p4/vertex are host arrays as produced by attpc_kin_run (or read from a kinematics file). */
ATTPC_API int32_t attpc_det_run(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                      const attpc_event_layout* layout, const double* p4, const double* vertex,
                      attpc_cloud_out* out, attpc_run_stats* stats);

/* Fused kinematics + detector: run_kinematics_pipeline + run_simulation without the
 * file in between (kinematics/pipeline.py:429-495, detector/simulator.py:118-210);
 * kinematics never leaves HBM.  out == NULL keeps the clouds device-resident
 * (chunk buffers are overwritten; stats carry counts and checksums). */
ATTPC_API int32_t attpc_sim_run(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                      const attpc_event_layout* layout, double* p4, double* vertex,
                      int32_t* kin_status, attpc_cloud_out* out, attpc_run_stats* stats);

/* Announces the attpc_sim_run / attpc_sim_run_spyral call AFTER the next one: "when the run I am about to start has
 * queued its last track batch, go on with the kinematics + tracks of events first_event .. of this seed".  The next
 * run then queues that first batch (up to 8 scatter chunks) on its low-priority stream behind its own last scatter
 * launches, and the call that was announced finds it under way instead of starting with track integration that has
 * nothing to run beside (about 5 % of a 1e6-event call of the headline workload).  Purely a scheduling hint: a call
 * for other events, another entry point or a configure call lets the batch finish and drops it; results never depend
 * on it.  n_events == 0 or layout == NULL withdraws the announcement.  The reference has no counterpart (its loop is
 * one event at a time, detector/simulator.py:183-208). */
ATTPC_API int32_t attpc_sim_hint_next(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                                      const attpc_event_layout* layout);

/* Diagnostics used by the parity tests: one track per (event, simulated nucleus) of a
 * det_run-style input.  samples [n_tracks, ATTPC_TIME_SAMPLES, 4] rows (x m, y m,
 * time bucket, electrons*gain) of the samples with >= 1 electron; counts [n_tracks];
 * n_steps [n_tracks] = number of recorded ODE samples (rows of the reference's track). */
ATTPC_API int32_t attpc_det_tracks(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                         const attpc_event_layout* layout, const double* p4,
                         const double* vertex, int64_t max_samples_per_track, double* samples,
                         int32_t* counts, int32_t* n_steps);

/* Diagnostics used by the parity tests: the pad-plane scatter alone -- transport_track
 * (detector/transporter.py:252-317: transverse_transport :172-249, point_transport :123-169,
 * position_to_index :78-120) per simulated nucleus into the event's shared dictionary, then
 * dict_to_points + jitter + 0 <= tb < 512 mask (detector/simulator.py:19-49, :93-113) -- for EXPLICIT
 * track samples, so that reference-generated transport fixtures reach the HIP kernel directly.
 * samples [sum(counts), 4] rows (x m, y m, time bucket, electrons already multiplied by the gain),
 * concatenated track by track, track = event * n_sim + position in layout->indices;
 * counts [n_events * n_sim] (each <= 10112).  layout->species_of_row is ignored. */
ATTPC_API int32_t attpc_det_scatter(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                          const attpc_event_layout* layout, const double* samples, const int32_t* counts,
                          attpc_cloud_out* out, attpc_run_stats* stats);

/* GET response + Spyral row conversion, detector/response.py:8-57, detector/writer.py:61-112:
 * rows [n,8] = x_mm, y_mm, z_mm, amplitude, integral, pad, tb, pad_scale. */
ATTPC_API int32_t attpc_spyral_rows(attpc_ctx* ctx, int64_t n_points, const double* points,
                          const double* response /*[512]*/, const double* pad_centers /*[npads,2]*/,
                          const double* pad_sizes /*[npads]*/, int32_t n_pads, int32_t windows_edge,
                          int32_t micromegas_edge, double length, double* rows);

/* Electronics / geometry the Spyral conversion needs: get_response(config) (detector/response.py:8-32),
 * Config.pad_centers / pad_sizes (detector/parameters.py:207-261), adc_threshold, time-bucket edges. */
typedef struct attpc_spyral_desc {
  const double* response;    /* [ATTPC_NUM_TB] ADC counts per electron and time bucket */
  const double* pad_centers; /* [n_pads, 2] mm */
  const double* pad_sizes;   /* [n_pads] */
  int32_t n_pads;
  int32_t windows_edge;
  int32_t micromegas_edge;
  int32_t reserved;
  double length;             /* m */
  double adc_threshold;      /* rows with amplitude <= threshold are dropped (writer.py:232-234) */
} attpc_spyral_desc;

ATTPC_API int32_t attpc_spyral_configure(attpc_ctx* ctx, const attpc_spyral_desc* desc);

/* attpc_sim_run followed, on the device and before anything crosses PCIe, by what SpyralWriter.write
 * does per event (detector/writer.py:194-238): convert_to_spyral (rows of 8: x_mm, y_mm, z_mm,
 * amplitude, integral, pad, tb, pad_scale), the ADC-threshold cut (:232-234) and the sort by z
 * (:236-238).  out->points receives rows of EIGHT doubles here (capacity counts rows), the rows of
 * every event in ascending z, ready for the writer. */
ATTPC_API int32_t attpc_sim_run_spyral(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                                       const attpc_event_layout* layout, double* p4, double* vertex,
                                       int32_t* kin_status, attpc_cloud_out* out, attpc_run_stats* stats);

/* attpc_det_run followed, on the device, by the same per-event work of SpyralWriter.write as in
 * attpc_sim_run_spyral: the file-driven flow run_simulation(config, kinematics file, SpyralWriter)
 * (detector/simulator.py:183-208 -> detector/writer.py:194-255) with kinematics read from a file (host arrays
 * p4 [n, n_rows, 4], vertex [n, 3]) instead of generated on the device.  out->points receives rows of EIGHT doubles,
 * every event's rows thresholded and in ascending z; out->event_points the cloud rows of every event before the
 * threshold (simulator.py:204-205 decides "empty event" on those). */
ATTPC_API int32_t attpc_det_run_spyral(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                                       const attpc_event_layout* layout, const double* p4, const double* vertex,
                                       attpc_cloud_out* out, attpc_run_stats* stats);

#ifdef __cplusplus
}
#endif
#endif /* ATTPC_ENGINE_H */
