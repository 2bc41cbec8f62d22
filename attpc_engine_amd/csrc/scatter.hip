// scatter.hip -- electron drift with transverse diffusion + pad-plane point cloud.
//
// Restates (reference src/attpc_engine/detector/): transporter.py:252-317 (transport_track:
// sigma_t = sqrt(2 D dv t / E)), :172-249 (transverse_transport: 10x10 mesh over +-3 sigma,
// pad look-up per pixel, int(pdf h^2 n) electrons into a (tb, pad)-keyed dictionary,
// key inserted and label overwritten even when 0 electrons), :123-169 (point_transport),
// :78-120 (position_to_index: floor to whole mm), pairing.py (key), beam_pads.py (folded
// into the LUT), simulator.py:19-49 (dict_to_points), :108-113 (tb jitter, 0 <= tb < 512).
//
// Execution model: persistent workgroups (1024 threads and 8192 table slots, one per compute unit; or,
// built through scatter_small.hip, 512 threads and 4096 slots, two per compute unit) take events from
// a global counter; one event at a time, its dictionary is an open-addressing hash table in LDS
// (u32 key|label word + u64 charge per slot, buckets of 4 keys).
// Per event: the entries (samples x slices) are histogrammed by time bucket, prefix-summed and
// sorted by time bucket once (up to SORT_CAP entries; longer events rank their entry list chunk by chunk
// and remember every chunk's time-bucket range); events with more keys than the table should hold are cut into
// time-bucket windows (a key contains its time bucket, so windows partition the key space) sized by
// an estimated key count and cut back to a whole number of row passes.  Per window:
//   stage  one lane per entry of the window: sigma_t and the whole-mm LUT indices of the entry's
//          10 mesh columns / 10 mesh rows (20 floors instead of 200) -> LDS
//   rows   (per wave, no workgroup barrier) one lane per mesh row: 10 pad look-ups in flight
//          (2-byte gathers from the 625 KB LUT, L2 resident, laid out so that the lines of a sample
//          read neighbouring addresses), the 10 pixel charges truncated to u32, runs of equal pads
//          merged in registers, runs (key|label, charge) written to the wave's LDS queue at positions
//          from a ballot prefix
//   insert (same wave) the queued runs as a stream, one probe step per lane and trip: ds_read_b128 bucket
//          probe, ds_cmpst_b32 to claim a slot, ds_max_u32 for the label, ds_add_u64 for the charge; a lane
//          that is done takes the next run at once (the loop is written in gfx950 assembly, stream_insert())
//   flush  occupied slots compacted per wave, rows written (with the Philox time-bucket jitter) to a
//          range of the output block this workgroup reserved, slots reset on the way
// Merging before inserting cuts hash inserts 2.5x (100 pixels -> ~40 runs on ~15 pads per sample) and the
// queue turns the sparse "which lanes end a run" pattern into dense wave work.  Pixel charges
// are whole numbers far below 2^53, so integer accumulation is exact and order independent.
// "label = last nucleus in `indices` order that touched the key" (transporter.py:249) is the
// MAX position in `indices` over the touching nuclei, kept with ds_max_u32 on the key|label
// word, so all nuclei of an event scatter concurrently.
//
// pdf(pixel) h^2 depends only on the pixel index: (36/81)/(2 pi) exp(-(2/9)((i-4.5)^2+(j-4.5)^2))
// because the mesh pitch is h = 6 sigma / 9; the 100 weights are a constant table.
//
// Bound: VALU issue (DESIGN.md 4.3 has the counters); HBM traffic is the 32 B per output row
// (3 f64 + i64, the reference's own dtypes) and 32 B per track sample read.
#include "tracks_args.hpp"

// This file is compiled three times into the library: as it is ("big": one 1024-thread workgroup with a
// 12 288-slot table per CU), through scatter_small.hip ("small": two 512-thread workgroups with 6 144-slot
// tables per CU, 6 % faster for detectors with the usual diffusion, but a single time bucket with more keys
// than its table holds does not fit) and through scatter_wide.hip ("wide": big's geometry with 8 192 slots of
// u64 sums, for detectors whose sums pass u32).  The host picks the variant per launch (abi.hip).
#ifndef ATTPC_SC_VARIANT
#define ATTPC_SC_VARIANT big
#endif
#define ATTPC_SC_CAT2(a, b) a##b
#define ATTPC_SC_CAT(a, b) ATTPC_SC_CAT2(a, b)

namespace attpc {
namespace ATTPC_SC_CAT(sc_, ATTPC_SC_VARIANT) {

// Diagnostic build only (-DATTPC_PHASE_TIMERS): thread 0 of every workgroup accumulates
// s_memtime deltas per phase into out.ctrl[8 + phase]; never compiled into the shipped library.
#ifdef ATTPC_PHASE_TIMERS
#define PHASE_DECL unsigned long long ph_t0 = __builtin_amdgcn_s_memtime(), ph_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PHASE_MARK(k)                                               \
  do {                                                              \
    const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); \
    ph_acc[k] += ph_now - ph_t0;                                    \
    ph_t0 = ph_now;                                                 \
  } while (0)
#define PHASE_SYNC asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#define PHASE_COUNT(k, v) ph_acc[k] += (unsigned long long)(v)
#define PHASE_FLUSH                                                          \
  do {                                                                       \
    if (tid == 0)                                                            \
      for (int k = 0; k < 20; ++k) atomicAdd(&a.out.ctrl[8 + k], ph_acc[k]); \
    if (lane == 0 && (tid >> 6) < 16) {  /* per wave: wait at the window's last barrier, staging */ \
      atomicAdd(&a.out.ctrl[40 + (tid >> 6)], ph_acc[4]);                    \
      if (((tid >> 6) & 1) == 0) atomicAdd(&a.out.ctrl[56 + (tid >> 7)], ph_acc[3]); \
    }                                                                        \
  } while (0)
#else
#define PHASE_DECL
#define PHASE_MARK(k)
#define PHASE_SYNC
#define PHASE_COUNT(k, v)
#define PHASE_FLUSH
#endif

// Build-time tunables (the A/B variants of tools/ab_scatter.py are builds with other values)
#ifndef ATTPC_SC_THREADS
#define ATTPC_SC_THREADS 1024
#endif
#ifndef ATTPC_SC_HASH_BITS
#define ATTPC_SC_HASH_BITS 13
#endif
#ifndef ATTPC_SC_STAGE
#define ATTPC_SC_STAGE 320
#endif
#ifndef ATTPC_SC_WAVE_QUEUE
#define ATTPC_SC_WAVE_QUEUE 256
#endif
#ifndef ATTPC_SC_WG_PER_CU
#define ATTPC_SC_WG_PER_CU 1
#endif
#ifndef ATTPC_SC_TARGET_PCT
#define ATTPC_SC_TARGET_PCT 50
#endif
constexpr int SC_THREADS = ATTPC_SC_THREADS;
constexpr int N_WAVES = SC_THREADS / 64;
constexpr int STAGE = ATTPC_SC_STAGE;            // entries staged per rows round
constexpr int SORT_CAP = 2048;                   // events with at most this many entries are sorted by time bucket
constexpr int LDS_BLOCKS = SORT_CAP / ARENA_BLK; // arena block ids per track kept in LDS: all a sorted event can use
constexpr int MAX_CHUNKS = SORT_CAP / 4;         // chunks of an unsorted event with a time-bucket range of their own (the last
                                                 // one stands for all further chunks)
constexpr int SORT_PER_THREAD = (SORT_CAP + SC_THREADS - 1) / SC_THREADS;
constexpr int HASH_BITS = ATTPC_SC_HASH_BITS;
// Slots of the table.  Default: 8 bytes each (u32 key | label word + u32 electrons), 1.5 x 2^HASH_BITS of them in the
// 12 bytes x 2^HASH_BITS the first table took (u64 electrons) -- a third fewer windows per event, each with its
// selection, barriers, compaction and partial passes (-6.6 % kernel time on the headline workload).  A key's electrons
// fit u32 in practice (largest sum met in 2.2e7 points of the headline workload: 1.1e9); a window in which one may not
// is done again by lone_bucket_kernel (u64 sums), see ADD_LIMIT -- and a detector where that is the rule (a gain far
// beyond the AT-TPC's) runs the third build of this file, scatter_wide.hip (ATTPC_SC_WIDE_CHARGE: u64 electrons per
// slot, the first table), which the host switches to by itself (abi.hip, prefer_wide).
#ifndef ATTPC_SC_WIDE_CHARGE
#define ATTPC_SC_WIDE_CHARGE 0
#endif
#if ATTPC_SC_WIDE_CHARGE
using charge_t = unsigned long long;
constexpr int HASH_CAP = 1 << HASH_BITS;         // slots (8 192)
#else
using charge_t = uint32_t;
constexpr int HASH_CAP = 3 << (HASH_BITS - 1);   // slots (6 144 / 12 288)
#endif
constexpr int TARGET_KEYS = HASH_CAP * ATTPC_SC_TARGET_PCT / 100;  // aimed-at fill: inserts slow down steeply beyond ~55 %
#ifndef ATTPC_SC_BUDGET_MAX_MULT
#define ATTPC_SC_BUDGET_MAX_MULT 64
#endif
#ifndef ATTPC_SC_BUDGET_GROWTH
#define ATTPC_SC_BUDGET_GROWTH 2
#endif
constexpr int BUDGET_GROWTH = ATTPC_SC_BUDGET_GROWTH;  // ... and the previous window's budget by this factor
constexpr int BUDGET_MAX_MULT = ATTPC_SC_BUDGET_MAX_MULT;  // a window's estimated keys may exceed the aimed-at fill by this factor
constexpr int WAVE_QUEUE = ATTPC_SC_WAVE_QUEUE;  // queued runs per wave and pass (typ. ~190 per 64 mesh rows)
constexpr int MESH = ATTPC_MESH_STEPS;
constexpr int PIXELS = MESH * MESH;
constexpr int BINS_PER_THREAD = (ATTPC_NUM_TB + SC_THREADS - 1) / SC_THREADS;
constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr uint32_t KEY_MASK = 0x00FFFFFFu;
constexpr int SEG_BLOCK = 16;        // segment slots reserved at a time
constexpr int CTRL_NEXT_EVENT = 28;  // out.ctrl[]: next unassigned event of the launch
constexpr int CTRL_LONE = 29;        // out.ctrl[]: time buckets left to lone_bucket_kernel (entries of out.lone_list)
constexpr int CTRL_ROWS = 30;        // out.ctrl[]: rows actually written ([0] is the reservation cursor)
constexpr int CTRL_MISMATCH = 31;    // out.ctrl[]: windows whose occupied-slot count differed from the claimed keys
constexpr int CTRL_DANGER = 32;      // out.ctrl[]: windows given to lone_bucket_kernel because a u32 sum could have wrapped
// merge variant (scatter_kernel<false, true>): a wave works on MERGE_SEQ_PER_WAVE sequences of consecutive entries of
// the window at a time, ten lanes (the mesh lines of constant y) per sequence; MERGE_G entries of every sequence are
// staged per round
#ifndef ATTPC_SC_MERGE_EXTRA  // LDS left beside the default kernel's arrays (one workgroup per CU: 160 KiB, two: 80 KiB each)
#define ATTPC_SC_MERGE_EXTRA (ATTPC_SC_WG_PER_CU == 1 ? 5800 : 976)
#endif
constexpr int MERGE_SEQ_PER_WAVE = 64 / MESH;
constexpr int MERGE_NSEQ = N_WAVES * MERGE_SEQ_PER_WAVE;
constexpr int MERGE_ENTRY_BYTES = 8 + 2 * 2 * MESH + 4;  // electrons f64, 2 x 10 LUT indices i16, time bucket word
constexpr int MERGE_STAGE_BYTES = STAGE * MERGE_ENTRY_BYTES + SORT_CAP * 2 + ATTPC_SC_MERGE_EXTRA;  // st_n .. merge_extra
constexpr int MERGE_G = MERGE_STAGE_BYTES / MERGE_ENTRY_BYTES / MERGE_NSEQ;  // entries of a sequence staged per round
constexpr int MERGE_ROUND = MERGE_NSEQ * MERGE_G;  // entries staged per round (MergeStage capacity)
constexpr uint32_t MERGE_INVALID = 0xFFFFFFFFu;    // meta word of an entry outside 0 <= time bucket < 512
static_assert(MERGE_G >= 1, "at least one staged entry per sequence and round");
static_assert(STAGE <= SC_THREADS, "one lane per staged entry");
static_assert(4 * N_WAVES * (WAVE_QUEUE + 2) >= HASH_CAP && HASH_CAP <= 65536, "the wave queues double as the (16-bit) slot list of a flush");
static_assert(2 * MAX_CHUNKS * sizeof(uint32_t) <= SORT_CAP * sizeof(unsigned short), "chunk ranges overlay perm[]");

// Drift time of slice `sl` of a sample created at time bucket t: the sample itself without the
// longitudinal-diffusion extension, else numpy.linspace(t - 3 sigma_l, t + 3 sigma_l, 5)[sl] with
// sigma_l = sqrt(2 D_l dv t / E) / dv time buckets.
__device__ __forceinline__ double slice_time(const DetDev& det, double t, int sl, int n_slices) {
  if (n_slices == 1) return t;
  const double sigma_l = sqrt(2.0 * det.longitudinal_diffusion * det.dv * t / det.efield) / det.dv;
  const double lo = t - 3.0 * sigma_l, hi = t + 3.0 * sigma_l;
  return sl == n_slices - 1 ? hi : (double)sl * ((hi - lo) / (double)(n_slices - 1)) + lo;
}

// Estimated distinct keys a sample adds: its mesh is 6 sigma_t wide, pads have a ~4.9 mm pitch,
// so it touches about (1 + 6 sigma_t / 4.9 mm)^2 pads; sigma_t^2 = 2 D dv tb / E.  `spread` =
// (6 / 4.9 mm)^2 * 2 D dv / E is a per-configuration constant.  (Default detector: 3 keys at the
// micromegas, 15 at tb 256, 28 at tb 511; the measured mean of the headline workload is 14.6.)
// Windows are cut on the prefix sum of this estimate; the ratio observed/estimated of each
// flushed window rescales the next one.
__device__ __forceinline__ int key_estimate(int tb, float spread) {
  const float r = 1.0f + sqrtf(spread * (float)tb);
  return (int)fminf(r * r + 0.5f, 100.0f);  // a sample has 100 pixels
}
// ... refined (round 3): a key holds its time bucket, so a sample shares pads-as-keys only with samples of its own
// bucket -- and with its predecessor on the track, if that one is in the same bucket, it shares all the pads the two
// meshes have in common: only the strip the mesh moved on by is new, about (S + p) d / p^2 pads for a displacement d
// (S = 6 sigma_t, p = 4.9 mm).  The slow heavy recoils of the headline workload put several samples into every bucket:
// the plain estimate was twice the keys the windows really held (tables a quarter full instead of half); samples
// 0.1 mm apart (configs[4]) share nearly everything.  `rec` = the sample's record (x, y, time bucket, electrons),
// `prev` = its predecessor's or nullptr.
__device__ __forceinline__ int key_estimate_on_track(const double* rec, const double* prev, double t, int tb_of_entry,
                                                     double prev_ts, float spread) {
  const float r = 1.0f + sqrtf(spread * (float)(int)fmin(t, 511.0));
  float est = r * r;
  if (prev != nullptr && prev_ts >= 0.0 && (int)prev_ts == tb_of_entry) {
    const float dx = (float)(rec[0] - prev[0]), dy = (float)(rec[1] - prev[1]);
    const float d_over_p = sqrtf(dx * dx + dy * dy) * (1.0f / 4.9e-3f);
    est = fminf(est, r * d_over_p + 1.0f);
  }
  return (int)fminf(est + 0.5f, 100.0f);
}

// Monte-Carlo diffusion extension: what a staged entry needs instead of the mesh indices (these
// records overlay st_ix / st_iy)
struct McEntry {
  double x, y, sigma;  // sample position [m], sigma_t [m]
  uint32_t n_prim;     // primary electrons
  uint32_t cs;         // entry number in the event (random-stream domain)
};
static_assert(sizeof(McEntry) * STAGE <= 2 * sizeof(short) * STAGE * MESH, "McEntry records overlay st_ix + st_iy");
static_assert(2 * sizeof(short) * STAGE * MESH >= sizeof(uint32_t) * ATTPC_NUM_TB,
              "st_ix + st_iy double as the per-bucket cursors of the entry sort (one u32 per time bucket)");

struct __align__(16) ScatterShared {
  double wtab[PIXELS];        // first member: rows of 10 weights are read as five 16-byte pairs
  uint32_t keys[HASH_CAP];    // bits 0..13 pad, 14..23 time bucket, 24..26 position in `indices`
  charge_t chg[HASH_CAP];     // electrons per key (ds_add_rtn_u32 / ds_add_u64); keys and chg sit below 64 KiB in
                              // every build: stream_insert() addresses them through the 16-bit offset field
  uint2 queue[N_WAVES][WAVE_QUEUE + 2];  // per wave: (key|label, charge) of queued runs (+ dump slot);
                                         // the whole array is the slot list during a flush
  double st_n[STAGE];         // electrons x gain (x the slice weight of the longitudinal extension)
  short st_ix[STAGE][MESH];   // the lane's coordinate: LUT index of the y mesh line i, lut_n = off the pad plane
  short st_iy[STAGE][MESH];   // the stepped coordinate: LUT index of the x mesh line j
  int st_tb[STAGE];           // bits 0..9 time bucket, 24..26 position in `indices`, 30 point transport
  unsigned short perm[SORT_CAP];    // entries (sample x slice) sorted by time bucket, events of <= SORT_CAP entries;
                                    // longer events: lowest / highest time bucket of every chunk of SC_THREADS entries
                                    // (two u32 per chunk: chunk_lo(), chunk_hi())
  // The merge variant keeps its sorted entry list in global memory and has no use for perm[]: its staging arrays
  // (MergeStage below) lie over st_n .. merge_extra as one block, with room for more entries per round than STAGE.
  char merge_extra[ATTPC_SC_MERGE_EXTRA];
  int blocks[ATTPC_MAX_SIM][LDS_BLOCKS];  // the first arena block ids of the event's tracks (a sorted event has no others)
  long long label_of[ATTPC_MAX_SIM];  // row number (label) of each simulated nucleus
  int cnt[ATTPC_MAX_SIM + 1]; // exclusive prefix of kept samples per simulated nucleus
  unsigned long long cum[ATTPC_NUM_TB];  // inclusive prefix sums per time bucket: low word estimated keys,
                                         // high word staged entries (samples x slices)
  unsigned long long wave_sum[SC_THREADS / 64];
  int stage_sum[2][SC_THREADS / 64];  // in-window entries per wave of a staging chunk (double buffered)
  int win_a, win_b, win_samples, win_r0, win_n, budget, overflow, done, ev_failed, failed, retried;
  int danger;  // an add of this window met a sum >= 2^31 (ADD_LIMIT): its time buckets go to lone_bucket_kernel
  unsigned int wg_cursor, n_keys, batch_first;
  unsigned long long base;
  unsigned long long row_cur, row_end, seg_cur, seg_end;  // this workgroup's reserved output rows / segment slots
  unsigned long long wg_samples, wg_rows;                 // workgroup totals (thread 0)
  unsigned long long ev_rows;                             // rows of the current event flushed so far (thread 0)
  unsigned long long charge_sum, key_sum;
  int merge_lead;  // merge variant: steps the window's foremost wave has done (ds_max_u32; wave priorities follow the lag)
  double long_w[ATTPC_LONG_STEPS];  // merge variant: DetDev::long_weights (an indexed kernel argument is a global load)
  double sigma_k[2];  // merge variant: 2 D dv and E of sigma_t^2 = 2 D dv t / E, for the lanes that stage (LDS, not registers:
                      // as loop invariants in vector registers they were spilled, and a reload from scratch memory waits
                      // for every load under way -- the touches of the next round included)
};

static_assert(sizeof(ScatterShared) <= (ATTPC_SC_WG_PER_CU == 1 ? 163840 : 81920), "LDS of a CU / of half a CU");

// The merge variant's staging arrays: the same four arrays as st_n / st_ix / st_iy / st_tb, MERGE_ROUND entries each,
// laid over ScatterShared::st_n .. merge_extra.
struct MergeStage {
  double* n;
  short (*ix)[MESH];
  short (*iy)[MESH];
  int* tb;
};
__device__ __forceinline__ MergeStage merge_stage(ScatterShared& sh) {
  static_assert(offsetof(ScatterShared, merge_extra) + ATTPC_SC_MERGE_EXTRA - offsetof(ScatterShared, st_n) >= MERGE_ROUND * MERGE_ENTRY_BYTES &&
                    offsetof(ScatterShared, st_ix) == offsetof(ScatterShared, st_n) + STAGE * 8 &&
                    offsetof(ScatterShared, perm) == offsetof(ScatterShared, st_tb) + STAGE * 4 &&
                    offsetof(ScatterShared, merge_extra) == offsetof(ScatterShared, perm) + SORT_CAP * 2,
                "the merge staging block is contiguous and large enough");
  MergeStage m;
  m.n = &sh.st_n[0];
  m.ix = reinterpret_cast<short (*)[MESH]>(m.n + MERGE_ROUND);
  m.iy = m.ix + MERGE_ROUND;
  m.tb = reinterpret_cast<int*>(m.iy + MERGE_ROUND);
  return m;
}

// sample c of the event's concatenated tracks -> record pointer and position in `indices`.  `table` = the event's
// rows of the block table in global memory: only events too long to be sorted (more than SORT_CAP entries) have
// tracks with more than LDS_BLOCKS blocks, and only their code path passes it (nullptr: every block id is in LDS).
__device__ __forceinline__ const double* sample_ptr(const ScatterShared& sh, const double* arena, const int32_t* table, int c,
                                                    int& isim) {
  isim = 0;
  // cnt[k] = total for k >= n_sim (per-event init), so c >= cnt[k] is false there: no k < n_sim test,
  // which cost seven loop-invariant lane masks in scalar registers (spilled, reloaded per call)
#pragma unroll
  for (int k = 1; k < ATTPC_MAX_SIM; ++k)
    if (c >= sh.cnt[k]) isim = k;
  const int s = c - sh.cnt[isim];
  const int bi = s / ARENA_BLK;
  int blk;
  if (table != nullptr && bi >= LDS_BLOCKS) blk = table[isim * MAX_BLOCKS_PER_TRACK + bi];
  else blk = sh.blocks[isim][bi];
  return arena + ((size_t)blk * ARENA_BLK + (s & (ARENA_BLK - 1))) * 4;
}

__device__ __forceinline__ void clear_table(ScatterShared& sh) {
  for (int i = threadIdx.x; i < HASH_CAP; i += SC_THREADS) {
    sh.keys[i] = EMPTY;
    sh.chg[i] = (charge_t)0;
  }
}

// first index in [lo, hi) whose prefix sum exceeds value; HIGH selects the entry count (high word)
// instead of the estimated keys (low word).  Executed by one whole wave (ln = lane): a 64-way
// search step over every 8th entry and an 8-way step inside the block found -- two LDS round trips
// instead of the nine of a binary search (the selection sits on the workgroup's critical path).
static_assert(ATTPC_NUM_TB == 64 * 8, "wave_upper_bound covers 8 buckets per lane");
static_assert(HASH_CAP % (SC_THREADS) == 0, "the flush compacts HASH_CAP / N_WAVES slots per wave, 64 at a time");
template <bool HIGH>
__device__ __forceinline__ int wave_upper_bound(const unsigned long long* cum, int lo, int hi, unsigned int value,
                                                int ln) {
  const int ic = ln * 8 + 7;
  const unsigned long long vc = cum[ic];
  const unsigned int fc = HIGH ? (unsigned int)(vc >> 32) : (unsigned int)vc;
  const unsigned long long mc = __ballot(ic >= hi || (ic >= lo && fc > value));
  if (mc == 0ull) return hi;
  const int blk = __ffsll((long long)mc) - 1;
  const int jf = blk * 8 + (ln & 7);
  const unsigned long long vf = cum[jf];
  const unsigned int ff = HIGH ? (unsigned int)(vf >> 32) : (unsigned int)vf;
  const unsigned long long mf = __ballot(jf >= hi || (jf >= lo && ff > value)) & 0xffull;
  return blk * 8 + (__ffsll((long long)mf) - 1);
}

constexpr int BUCKET = 4;                           // keys per bucket = one ds_read_b128
constexpr int N_BUCKETS = HASH_CAP / BUCKET;
constexpr int MAX_BUCKET_PROBES = 48;

// Multiplicative hash (golden ratio, all 32 bits of the product), scaled to the (not 2^n) bucket count.  (Tried, round 3: the
// same in full-rate 24-bit multiplies with a 16-bit intermediate hash -- 70 000 window retries per 400 000 events
// instead of 1 000: time buckets nine apart mapped the same pads to neighbouring buckets.)
__device__ __forceinline__ uint32_t hash_bucket(uint32_t key) {
  return __umulhi(key * 2654435761u, (uint32_t)N_BUCKETS);
}
__device__ __forceinline__ uint32_t next_bucket(uint32_t b) { return b + 1u == (uint32_t)N_BUCKETS ? 0u : b + 1u; }

// Charges are u32 sums.  One add is below 2^31 (a run is at most ten pixels below 2^27 each, table_add() refuses more), so
// a sum can only wrap if it was >= 2^31 before the add: every add RETURNS the old value, and bit 31 of any returned value
// marks the window as dangerous.  A dangerous window is thrown away and its time buckets go to lone_bucket_kernel (u64
// sums in global memory): exact, slow, and not met in the workloads of BASELINE.json.
constexpr uint32_t ADD_LIMIT = 1u << 30;

// points[key] = (charge + q, label) of transporter.py:247-249: find or claim the key's slot,
// raise the label, add the charge.  `want` = key | label bits.  The table is bucketed: one
// 16-byte LDS read shows 4 candidate slots, so a probe step is one round trip and chains stay
// short at 50 % load (linear probing over single slots needed ~15 dependent round trips for
// the slowest of 64 lanes).  False if the table is too full.
__device__ __forceinline__ bool table_add(ScatterShared& sh, uint32_t want, unsigned long long q) {
#if !ATTPC_SC_WIDE_CHARGE
  if (q >= (unsigned long long)ADD_LIMIT) {  // no u32 add for this one: the window goes to lone_bucket_kernel
    sh.danger = 1;
    return true;
  }
#endif
  const uint32_t key = want & KEY_MASK;
  uint32_t b = hash_bucket(key);
  uint32_t h = 0, cur = 0;
  int probes = 0;
  for (;;) {
    const uint4 k4 = *reinterpret_cast<const uint4*>(&sh.keys[b * BUCKET]);
    const bool m0 = (k4.x & KEY_MASK) == key, m1 = (k4.y & KEY_MASK) == key;
    const bool m2 = (k4.z & KEY_MASK) == key, m3 = (k4.w & KEY_MASK) == key;
    if (m0 || m1 || m2 || m3) {
      const uint32_t pos = m0 ? 0u : (m1 ? 1u : (m2 ? 2u : 3u));
      h = b * BUCKET + pos;
      cur = m0 ? k4.x : (m1 ? k4.y : (m2 ? k4.z : k4.w));
      break;
    }
    const bool e0 = k4.x == EMPTY, e1 = k4.y == EMPTY, e2 = k4.z == EMPTY, e3 = k4.w == EMPTY;
    if (e0 || e1 || e2 || e3) {  // claim the first free slot of this bucket
      const uint32_t pos = e0 ? 0u : (e1 ? 1u : (e2 ? 2u : 3u));
      const uint32_t old = atomicCAS(&sh.keys[b * BUCKET + pos], EMPTY, want);
      if (old == EMPTY || (old & KEY_MASK) == key) {
        h = b * BUCKET + pos;
        cur = old == EMPTY ? want : old;
        if (old == EMPTY) atomicAdd(&sh.n_keys, 1u);  // rows of the window's flush
        break;
      }
      continue;  // lost the slot to another key: look at the bucket again
    }
    if (++probes >= MAX_BUCKET_PROBES) return false;
    b = next_bucket(b);
  }
  if (cur < want) atomicMax(&sh.keys[h], want);
#if ATTPC_SC_WIDE_CHARGE
  atomicAdd(&sh.chg[h], q);
#else
  if (atomicAdd(&sh.chg[h], (uint32_t)q) >> 31) sh.danger = 1;
#endif
  return true;
}

// The same operation for a whole wave, one queued run per lane (`pending` lanes only), written as
// one wave-uniform loop with selects instead of per-lane control flow: every trip is a bucket read,
// at most one compare-and-swap, and the two fire-and-forget updates.  `claimed` counts the new
// keys of the wave (scalar).  False if some lane ran out of probes.
__device__ __forceinline__ bool wave_insert(ScatterShared& sh, uint32_t want, uint32_t q, bool pending,
                                            unsigned int& claimed, unsigned int& trips) {
  const uint32_t key = want & KEY_MASK;
  uint32_t b = hash_bucket(key);
  int probes = 0;
  bool fail = false;
  while (__any(pending)) {
    trips++;  // diagnostic builds only (dead code otherwise)
    const uint4 k4 = *reinterpret_cast<const uint4*>(&sh.keys[b * BUCKET]);
    const bool m0 = (k4.x & KEY_MASK) == key, m1 = (k4.y & KEY_MASK) == key;
    const bool m2 = (k4.z & KEY_MASK) == key, m3 = (k4.w & KEY_MASK) == key;
    const bool e0 = k4.x == EMPTY, e1 = k4.y == EMPTY, e2 = k4.z == EMPTY, e3 = k4.w == EMPTY;
    const bool any_m = m0 || m1 || m2 || m3, any_e = e0 || e1 || e2 || e3;
    const uint32_t pos_m = m0 ? 0u : (m1 ? 1u : (m2 ? 2u : 3u));
    const uint32_t pos_e = e0 ? 0u : (e1 ? 1u : (e2 ? 2u : 3u));
    const uint32_t h = b * BUCKET + (any_m ? pos_m : pos_e);
    uint32_t cur = m0 ? k4.x : (m1 ? k4.y : (m2 ? k4.z : k4.w));
    const bool try_claim = pending && !any_m && any_e;
    uint32_t old = 0u;
    if (try_claim) old = atomicCAS(&sh.keys[h], EMPTY, want);
    const bool won = try_claim && old == EMPTY;
    const bool same = try_claim && (old & KEY_MASK) == key;  // another lane claimed it for this key
    cur = won ? want : (same ? old : cur);
    const bool done = pending && (any_m || won || same);
    if (done) {
      if (cur < want) atomicMax(&sh.keys[h], want);
#if ATTPC_SC_WIDE_CHARGE
      atomicAdd(&sh.chg[h], (unsigned long long)q);
#else
      if (atomicAdd(&sh.chg[h], q) >> 31) sh.danger = 1;
#endif
    }
    claimed += (unsigned int)__popcll(__ballot(won));
    const bool advance = pending && !any_m && !any_e;  // full bucket of other keys
    probes += advance ? 1 : 0;
    const bool give_up = advance && probes >= MAX_BUCKET_PROBES;
    fail = fail || give_up;
    b = advance ? next_bucket(b) : b;
    pending = pending && !done && !give_up;  // a lost compare-and-swap looks at the same bucket again
  }
  return !__any(fail);
}

// The table inserts of the mesh path as a STREAM: a lane that has finished its run takes the next one from
// the wave's queue at once, and a lane that has not (lost a compare-and-swap, met a full bucket) keeps its
// run for the next trip -- also across the 64-row blocks of a rows round (`carry`).  wave_insert() above
// loops until the slowest of its 64 lanes is done: 1.75 trips per 64 runs on the headline workload with most
// lanes idle in the later ones, each trip a 1 KiB bucket read of the whole wave; here it is 1.33, every trip
// but the last few of a round on (nearly) 64 runs.  (Tried and slower: linear probing with one returning
// compare-and-swap per probe instead of the bucket read -- 2.3 trips per 64 runs, +12 % kernel time.)
#ifdef ATTPC_SC_CXX_INSERT  // the compiler's version of the loop (A/B builds and diagnostic trip counts)
struct InsertCarry {  // per lane
  uint32_t want, q;
  uint32_t b;         // bits 0..15 bucket, 16.. probes so far
  uint32_t danger;    // OR of the old sums the adds returned (bit 31: see ADD_LIMIT)
  bool have;
  __device__ __forceinline__ void reset() { want = 0u; q = 0u; b = 0u; danger = 0u; have = false; }
};

__device__ __forceinline__ bool stream_insert(ScatterShared& sh, const uint2* __restrict__ queue, int n_q, bool drain,
                                              InsertCarry& c, unsigned int& claimed, unsigned int& trips) {
  int next = 0;  // wave uniform: first queue item nobody has taken yet
  bool fail = false;
  for (;;) {
    const unsigned long long idle_m = __ballot(!c.have);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_m, 0u));
    const bool fresh = next < n_q;  // this trip starts new runs
    if (!c.have && next + rank < n_q) {
      const uint2 item = queue[next + rank];
      c.want = item.x;
      c.q = item.y;
      c.b = hash_bucket(item.x & KEY_MASK);
      c.have = true;
    }
    next = min(n_q, next + (int)__popcll(idle_m));
    if (!__any(c.have)) break;
    if (!fresh && !drain) break;  // queue used up: the unfinished runs ride along with the next block's
    trips++;  // diagnostic builds only (dead code otherwise)
    const uint32_t key = c.want & KEY_MASK;
    const uint32_t b = c.b & 0xffffu;
    const uint4 k4 = *reinterpret_cast<const uint4*>(&sh.keys[b * BUCKET]);
    const bool m0 = (k4.x & KEY_MASK) == key, m1 = (k4.y & KEY_MASK) == key;
    const bool m2 = (k4.z & KEY_MASK) == key, m3 = (k4.w & KEY_MASK) == key;
    const bool e0 = k4.x == EMPTY, e1 = k4.y == EMPTY, e2 = k4.z == EMPTY, e3 = k4.w == EMPTY;
    const bool any_m = m0 || m1 || m2 || m3, any_e = e0 || e1 || e2 || e3;
    const uint32_t pos_m = m0 ? 0u : (m1 ? 1u : (m2 ? 2u : 3u));
    const uint32_t pos_e = e0 ? 0u : (e1 ? 1u : (e2 ? 2u : 3u));
    const uint32_t h = b * BUCKET + (any_m ? pos_m : pos_e);
    uint32_t cur = m0 ? k4.x : (m1 ? k4.y : (m2 ? k4.z : k4.w));
    const bool try_claim = c.have && !any_m && any_e;
    uint32_t old = 0u;
    if (try_claim) old = atomicCAS(&sh.keys[h], EMPTY, c.want);
    const bool won = try_claim && old == EMPTY;
    const bool same = try_claim && (old & KEY_MASK) == key;  // another lane claimed it for this key
    cur = won ? c.want : (same ? old : cur);
    const bool done = c.have && (any_m || won || same);
    if (done) {
      if (cur < c.want) atomicMax(&sh.keys[h], c.want);
#if ATTPC_SC_WIDE_CHARGE
      atomicAdd(&sh.chg[h], (unsigned long long)c.q);
#else
      c.danger |= atomicAdd(&sh.chg[h], c.q);
#endif
    }
    claimed += (unsigned int)__popcll(__ballot(won));
    const bool advance = c.have && !any_m && !any_e;  // full bucket of other keys
    const uint32_t probes = (c.b >> 16) + (advance ? 1u : 0u);
    const bool give_up = advance && probes >= (uint32_t)MAX_BUCKET_PROBES;
    fail = fail || give_up;
    c.b = (advance ? next_bucket(b) : b) | (probes << 16);
    c.have = c.have && !done && !give_up;  // a lost compare-and-swap looks at the same bucket again
  }
  return !__any(fail);
}
#else
// The shipped loop is written in gfx950 assembly.  hipcc's code for the C++ above spends ~75 vector + ~75 scalar
// instructions and 13 branches per trip (per-lane flags that live across the loop are kept as 0/1 in vector
// registers and turned into lane masks and back, every `if` becomes an exec-mask branch), and the trips were 59 % of
// all instructions the kernel issued.  Here a trip is 40 vector + ~35 scalar instructions and 4 branches: `have` and
// every condition are lane masks in scalar registers, the conditional LDS operations run under those masks, and the
// per-lane probe count is replaced by a trip budget of the call (a table too full to take a run ends the window either
// way).  Same table protocol as the C++ version: bucket read, ds_cmpst claim of the first free slot, ds_max label,
// ds_add_u64 charge, a lost claim looks at the same bucket again.  Registers are named explicitly (the 128-bit bucket
// and the {q, 0} pair of the 64-bit add need consecutive, even-aligned registers); the operands bind them.
struct InsertCarry {
  uint32_t want, q, ba;     // per lane: key | label, charge, byte offset of the bucket inside keys[]
  uint32_t danger;          // per lane: OR of the old sums the adds returned (bit 31: see ADD_LIMIT)
  unsigned long long have;  // wave uniform: lanes with a run under way
  __device__ __forceinline__ void reset() { want = 0u; q = 0u; ba = 0u; danger = 0u; have = 0ull; }
};
static_assert(offsetof(ScatterShared, keys) < 65536 && offsetof(ScatterShared, chg) < 65536,
              "keys[] / chg[] are addressed as register + 16-bit offset field");

__device__ __forceinline__ bool stream_insert(ScatterShared& sh, const uint2* __restrict__ queue, int n_q, bool drain,
                                              InsertCarry& c, unsigned int& claimed, unsigned int& trips) {
  (void)sh;
  uint32_t fail, budget_left;
  const uint32_t qbase = (uint32_t)(uintptr_t)queue;  // LDS byte address of the wave's queue
  // wave-uniform values the compiler may hold in vector registers
  c.have = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)c.have) |
           ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(c.have >> 32)) << 32);
  claimed = (uint32_t)__builtin_amdgcn_readfirstlane(claimed);
  asm volatile(
      "s_mov_b32 s92, 0\n"              // next: first queue item nobody has taken yet
#if ATTPC_SC_WIDE_CHARGE
      "v_mov_b32 v115, 0\n"             // v[114:115] = {q, 0}
#else
      "v_mov_b32 v124, 0\n"             // the old sum the lane's last add returned
#endif
      "v_mov_b32 v126, -1\n"            // EMPTY
      "s_mov_b32 s99, 0\n"              // fail
      "v_mov_b32 v127, 0x9e3779b1\n"    // hash_bucket()'s multiplier
      "s_mov_b32 s96, %[nb]\n"          // ... and its scale, the number of buckets
      "s_mov_b64 s[70:71], exec\n"       // every lane of the wave (the callers are wave uniform)
      "s_lshr_b32 s94, s93, 4\n"
      "s_add_i32 s94, s94, 64\n"        // trip budget of this call
      "v_and_b32 v121, 0xffffff, v112\n"  // keys of the runs carried over from the last call
      "1:\n"
      "s_andn2_b64 s[72:73], exec, s[68:69]\n"  // idle lanes
      "s_cmp_lt_i32 s92, s93\n"                 // fresh: this trip starts new runs
      "s_cbranch_scc0 3f\n"
      "v_mbcnt_lo_u32_b32 v120, s72, 0\n"
      "v_mbcnt_hi_u32_b32 v120, s73, v120\n"
      "v_add_u32 v120, s92, v120\n"             // queue item of this idle lane
      "v_cmp_gt_i32 vcc, s93, v120\n"
      "s_and_b64 s[74:75], vcc, s[72:73]\n"     // lanes that take a new run
      "s_bcnt1_i32_b64 s97, s[72:73]\n"
      "s_add_i32 s92, s92, s97\n"
      "s_min_i32 s92, s92, s93\n"
      "s_or_b64 s[68:69], s[68:69], s[74:75]\n"
      "s_mov_b64 exec, s[74:75]\n"
      "v_lshl_add_u32 v120, v120, 3, v111\n"
      "ds_read_b32 v112, v120\n"                // key | label
      "ds_read_b32 v114, v120 offset:4\n"       // charge
      "s_waitcnt lgkmcnt(1)\n"
      "v_and_b32 v121, 0xffffff, v112\n"
      "v_mul_lo_u32 v113, v121, v127\n"
      "v_mul_hi_u32 v113, v113, s96\n"          // bucket = (hash * buckets) >> 32
      "v_lshlrev_b32 v113, 4, v113\n"           // * 16 bytes
      "s_mov_b64 exec, s[70:71]\n"
      "2:\n"
      "s_cmp_eq_u64 s[68:69], 0\n"
      "s_cbranch_scc1 9f\n"
      "s_sub_u32 s94, s94, 1\n"
      "s_cbranch_scc1 8f\n"                     // out of trips: the table is too full for this window
      "s_mov_b64 exec, s[68:69]\n"
      "ds_read_b128 v[116:119], v113 offset:%[keys]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_and_b32 v120, 0xffffff, v116\n"
      "v_cmp_eq_u32 s[76:77], v120, v121\n"
      "v_and_b32 v120, 0xffffff, v117\n"
      "v_cmp_eq_u32 s[78:79], v120, v121\n"
      "v_and_b32 v120, 0xffffff, v118\n"
      "v_cmp_eq_u32 s[80:81], v120, v121\n"
      "v_and_b32 v120, 0xffffff, v119\n"
      "v_cmp_eq_u32 vcc, v120, v121\n"
      "v_cndmask_b32 v122, 12, 8, s[80:81]\n"        // byte offset of the matching slot in the bucket
      "v_cndmask_b32 v122, v122, 4, s[78:79]\n"
      "v_cndmask_b32 v122, v122, 0, s[76:77]\n"
      "s_or_b64 s[82:83], s[76:77], s[78:79]\n"
      "s_or_b64 s[80:81], s[80:81], vcc\n"
      "s_or_b64 s[82:83], s[82:83], s[80:81]\n"      // any slot holds the key
      "v_cmp_eq_u32 s[76:77], -1, v116\n"
      "v_cmp_eq_u32 s[78:79], -1, v117\n"
      "v_cmp_eq_u32 s[80:81], -1, v118\n"
      "v_cmp_eq_u32 vcc, -1, v119\n"
      "v_cndmask_b32 v123, 12, 8, s[80:81]\n"        // first free slot
      "v_cndmask_b32 v123, v123, 4, s[78:79]\n"
      "v_cndmask_b32 v123, v123, 0, s[76:77]\n"
      "s_or_b64 s[84:85], s[76:77], s[78:79]\n"
      "s_or_b64 s[80:81], s[80:81], vcc\n"
      "s_or_b64 s[84:85], s[84:85], s[80:81]\n"      // any slot free
      "v_cndmask_b32 v122, v123, v122, s[82:83]\n"
      "v_add_u32 v122, v113, v122\n"                 // byte offset of the slot inside keys[]
      "s_andn2_b64 s[74:75], s[84:85], s[82:83]\n"   // claim: key not there, a slot free
      "s_mov_b64 exec, s[74:75]\n"
      "ds_cmpst_rtn_b32 v125, v122, v126, v112 offset:%[keys]\n"
      "s_mov_b64 exec, s[68:69]\n"
      "s_waitcnt lgkmcnt(0)\n"
      "v_cmp_eq_u32 vcc, -1, v125\n"
      "s_and_b64 s[86:87], vcc, s[74:75]\n"          // won the slot
      "v_and_b32 v120, 0xffffff, v125\n"
      "v_cmp_eq_u32 vcc, v120, v121\n"
      "s_and_b64 s[88:89], vcc, s[74:75]\n"          // another lane claimed it for the same key
      "s_bcnt1_i32_b64 s97, s[86:87]\n"
      "s_add_i32 s95, s95, s97\n"                    // new keys of the wave
      "s_or_b64 exec, s[82:83], s[88:89]\n"          // the slot was there already: raise its label
      "ds_max_u32 v122, v112 offset:%[keys]\n"       // (a no-op where it is not lower; cheaper than finding out)
      "s_or_b64 s[90:91], exec, s[86:87]\n"          // done: the run's slot is known
      "s_mov_b64 exec, s[90:91]\n"
#if ATTPC_SC_WIDE_CHARGE
      "v_lshlrev_b32 v120, 1, v122\n"
      "ds_add_u64 v120, v[114:115] offset:%[chg]\n"
#else
      "v_or_b32 %[danger], %[danger], v124\n"       // (what these lanes' previous add returned: long since arrived)
      "ds_add_rtn_u32 v124, v122, v114 offset:%[chg]\n"  // the slot's u32 sum; the old value comes back (ADD_LIMIT)
#endif
      "s_or_b64 s[76:77], s[82:83], s[84:85]\n"
      "s_andn2_b64 exec, s[68:69], s[76:77]\n"       // full bucket of other keys: on to the next one
      "s_cbranch_execz 4f\n"                        // (rare at half load: most trips skip it)
      "v_add_u32 v113, 16, v113\n"
      "v_cmp_le_u32 vcc, %[nb16], v113\n"
      "v_cndmask_b32 v113, v113, 0, vcc\n"          // ... around the end of the table
      "4:\n"
      "s_mov_b64 exec, s[70:71]\n"
      "s_andn2_b64 s[68:69], s[68:69], s[90:91]\n"   // a lost claim looks at the same bucket again
      "s_branch 1b\n"
      "3:\n"                                         // the queue is used up
      "s_cmp_eq_u32 s98, 0\n"
      "s_cbranch_scc0 2b\n"                          // draining: go on with the runs under way
      "s_branch 9f\n"                                // else they ride along with the next block's
      "8:\n"
      "s_mov_b32 s99, 1\n"
      "9:\n"
      "s_waitcnt lgkmcnt(0)\n"                       // (the budget exit can come straight after a fresh run's loads: the
                                                     //  compiler does not track LDS reads issued in here)
      "s_mov_b64 exec, s[70:71]\n"
#if !ATTPC_SC_WIDE_CHARGE
      "v_or_b32 %[danger], %[danger], v124\n"       // what the last adds returned
#endif
      : "+{v112}"(c.want), "+{v113}"(c.ba), "+{v114}"(c.q), "+{s[68:69]}"(c.have), "+{s95}"(claimed), "={s99}"(fail), "={s94}"(budget_left),
        [danger] "+v"(c.danger)
      : "{v111}"(qbase), "{s93}"(__builtin_amdgcn_readfirstlane(n_q)), "{s98}"((uint32_t)(drain ? 1u : 0u)),
        [nb] "n"(N_BUCKETS), [nb16] "n"(N_BUCKETS * 16),
        [keys] "n"(offsetof(ScatterShared, keys)), [chg] "n"(offsetof(ScatterShared, chg))
      : "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "s70", "s71",
        "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87",
        "s88", "s89", "s90", "s91", "s92", "s96", "s97", "vcc", "scc", "memory");
  trips += ((uint32_t)n_q >> 4) + 64u - (fail ? 0u : budget_left);  // diagnostic builds only (dead code otherwise)
  return fail == 0u;
}
#endif

// Ten runs per lane at most -> the wave's queue: pixel j's run (key, electrons) goes to LDS address `at` + 8 x (lanes
// below in mask[j]) on the lanes of mask[j]; `at` = scalar address of the queue's next free entry, advanced past the
// pixels' runs in turn.  The masks go to exec as they are (an `if` on a mask that crossed a branch is rebuilt by the
// compiler from a 0/1 vector value: two instructions a pixel), five writes between one save and one restore of exec.
__device__ __forceinline__ void queue_put(uint32_t at, const unsigned long long (&mask)[MESH], uint32_t hi,
                                          const int (&ended)[MESH], const uint32_t (&q)[MESH]) {
  uint32_t addr[MESH];
#pragma unroll
  for (int j = 0; j < MESH; ++j) {
    const unsigned long long mk = mask[j];
    const uint32_t e = __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
    asm("" : "+s"(at));  // (kept a scalar of its own: folded into the lane's count it costs a vector add)
    addr[j] = 8u * e + at;
    at += 8u * (uint32_t)__popcll(mk);
  }
  static_assert(MESH == 10, "two blocks of five writes");
#pragma unroll
  for (int b = 0; b < MESH; b += 5) {
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[saved], exec\n"
        "s_mov_b64 exec, %[m0]\n"
        "ds_write2_b32 %[a0], %[k0], %[q0] offset1:1\n"
        "s_mov_b64 exec, %[m1]\n"
        "ds_write2_b32 %[a1], %[k1], %[q1] offset1:1\n"
        "s_mov_b64 exec, %[m2]\n"
        "ds_write2_b32 %[a2], %[k2], %[q2] offset1:1\n"
        "s_mov_b64 exec, %[m3]\n"
        "ds_write2_b32 %[a3], %[k3], %[q3] offset1:1\n"
        "s_mov_b64 exec, %[m4]\n"
        "ds_write2_b32 %[a4], %[k4], %[q4] offset1:1\n"
        "s_mov_b64 exec, %[saved]\n"
        : [saved] "=&s"(saved)
        : [m0] "s"(mask[b]), [m1] "s"(mask[b + 1]), [m2] "s"(mask[b + 2]), [m3] "s"(mask[b + 3]), [m4] "s"(mask[b + 4]),
          [a0] "v"(addr[b]), [a1] "v"(addr[b + 1]), [a2] "v"(addr[b + 2]), [a3] "v"(addr[b + 3]), [a4] "v"(addr[b + 4]),
          [k0] "v"(hi | (uint32_t)ended[b]), [k1] "v"(hi | (uint32_t)ended[b + 1]), [k2] "v"(hi | (uint32_t)ended[b + 2]),
          [k3] "v"(hi | (uint32_t)ended[b + 3]), [k4] "v"(hi | (uint32_t)ended[b + 4]),
          [q0] "v"(q[b]), [q1] "v"(q[b + 1]), [q2] "v"(q[b + 2]), [q3] "v"(q[b + 3]), [q4] "v"(q[b + 4])
        : "memory");
  }
}

// Merge variant: what a lane carries along its sequence of entries -- ten accumulators (the pad under each of the
// ten pixels of its mesh line and the electrons gathered on it since the pad last changed) for one time bucket
// and nucleus (`hi`, the high bits of the key word), and the fill of the wave's queue.
constexpr uint32_t MERGE_NO_HI = 0xFFFFFFFFu;
struct MergeAcc {
  uint32_t hi;
  int pad[MESH];     // -1: no accumulator for this pixel
  uint32_t q[MESH];
  int fill;          // wave uniform: runs waiting in the wave's queue
  int steps;         // wave uniform: steps of this window done
  uint32_t bound;    // >= every q[j]: the sum of the rows' largest pixels since the accumulators last restarted
  __device__ __forceinline__ void reset() {
    hi = MERGE_NO_HI;
    fill = 0;
    steps = 0;
    bound = 0u;
#pragma unroll
    for (int j = 0; j < MESH; ++j) {
      pad[j] = -1;
      q[j] = 0u;
    }
  }
};

// Time-bucket range of the chunks of an unsorted event (they overlay ScatterShared::perm, which such an event does not
// use): lowest bucket of chunk k in word k, highest in word MAX_CHUNKS + k.
__device__ __forceinline__ uint32_t* chunk_lo(ScatterShared& sh) { return reinterpret_cast<uint32_t*>(&sh.perm[0]); }
__device__ __forceinline__ uint32_t* chunk_hi(ScatterShared& sh) { return reinterpret_cast<uint32_t*>(&sh.perm[0]) + MAX_CHUNKS; }
__device__ __forceinline__ void chunk_range_add(ScatterShared& sh, int chunk, int tb) {
  const int k = min(chunk, MAX_CHUNKS - 1);
  atomicMin(chunk_lo(sh) + k, (uint32_t)tb);
  atomicMax(chunk_hi(sh) + k, (uint32_t)tb);
}
// first chunk >= from (< n_chunks) whose range meets the window [win_a, win_b), n_chunks if there is none.  Every
// wave computes it for itself (64 chunks per step); the ranges do not change while the windows are worked on.
__device__ __forceinline__ int next_chunk_in_window(ScatterShared& sh, int from, int n_chunks, int win_a, int win_b, int ln) {
  for (int k0 = from; k0 < n_chunks; k0 += 64) {
    const int k = min(k0 + ln, MAX_CHUNKS - 1);
    const bool hit = k0 + ln < n_chunks && (int)chunk_lo(sh)[k] < win_b && (int)chunk_hi(sh)[k] >= win_a &&
                     chunk_lo(sh)[k] != 0xffffffffu;
    const unsigned long long m = __ballot(hit);
    if (m) return k0 + __ffsll((long long)m) - 1;
  }
  return n_chunks;
}

// Next window [win_a, win_b) of time buckets: starts at the first non-empty bucket >= `from` and
// extends while the estimated key count stays within the budget (at least one bucket).  A window
// that does not reach the end of the event is then cut back to a whole number of row passes: the
// rows phase works in passes of SC_THREADS mesh rows (64 per wave), a window of 2.2 passes costs 3,
// so the 0.2 is left to the next window.  Wave 0 (every lane with the same arguments).
// `quantum` = entries of one full pass (SC_THREADS / MESH; merge variant: one staging round, MERGE_ROUND).
template <int QUANTUM = SC_THREADS / MESH>
__device__ __forceinline__ void select_window(ScatterShared& sh, int from, int budget, int ln) {
  const unsigned long long before = from > 0 ? sh.cum[from - 1] : 0ull;
  const unsigned int keys0 = (unsigned int)before, entries0 = (unsigned int)(before >> 32);
  const int a0 = wave_upper_bound<true>(sh.cum, from, ATTPC_NUM_TB, entries0, ln);
  if (a0 >= ATTPC_NUM_TB) {
    if (ln == 0) sh.done = 1;
    return;
  }
  int b0 = wave_upper_bound<false>(sh.cum, a0, ATTPC_NUM_TB, keys0 + (unsigned int)budget, ln);
  if (b0 <= a0) b0 = a0 + 1;
  // What is left of the event fits this window with a little stretch: take it all, rather than leave a remainder
  // window that costs its own selection, staging round, barriers and flush for a fraction of a pass (round 3: an event
  // of 3 1/4 windows' worth of keys was cut into 3 whole-pass windows + the quarter).
#ifndef ATTPC_SC_TAIL_PCT
#define ATTPC_SC_TAIL_PCT 25
#endif
  if ((unsigned int)sh.cum[ATTPC_NUM_TB - 1] - keys0 <= (unsigned int)budget + (unsigned int)budget * ATTPC_SC_TAIL_PCT / 100u)
    b0 = ATTPC_NUM_TB;
  const unsigned int entries = (unsigned int)(sh.cum[b0 - 1] >> 32) - entries0;
  const unsigned int passes = QUANTUM == SC_THREADS / MESH ? entries * MESH / SC_THREADS : entries / (unsigned int)QUANTUM;
  if (passes >= 1u && (unsigned int)(sh.cum[ATTPC_NUM_TB - 1] >> 32) > entries0 + entries) {
    const int b1 = wave_upper_bound<true>(sh.cum, a0, b0,
                                          entries0 + (QUANTUM == SC_THREADS / MESH ? passes * SC_THREADS / MESH : passes * (unsigned int)QUANTUM), ln);
    if (b1 > a0) b0 = b1;
  }
  if (ln == 0) {
    const unsigned long long last = sh.cum[b0 - 1];
    sh.budget = budget;
    sh.win_a = a0;
    sh.win_b = b0;
    sh.win_samples = (int)((unsigned int)last - keys0);
    sh.win_r0 = (int)entries0;  // the buckets between `from` and a0 are empty
    sh.win_n = (int)((unsigned int)(last >> 32) - entries0);
  }
}

// a wave-uniform constant the compiler may not hoist out of its loop (hoisted copies of such
// constants ended up in VGPRs that were then spilled to scratch)
__device__ __forceinline__ int local_const(int value) {
  asm volatile("" : "+s"(value));
  return value;
}

__device__ __forceinline__ int fresh_tid() {
  int t = (int)threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// MC: the Monte-Carlo diffusion extension (its own instantiation, so that the default kernel keeps its
// register budget).
// MERGE: the variant for track samples much closer than a pad (path-length dE/dx step, BASELINE configs[4]: samples
// 0.1 mm apart, every one lighting ~100 pads at 10x diffusion -- scattered one by one that is 37 table inserts per key).
// transporter.py:229-249 adds int(pdf h^2 n_k) to points[key] pixel by pixel; the sum over the samples k of a
// track whose pixel (i, j) falls on the same pad in the same time bucket can be formed in a register before the
// table sees it -- every term is still truncated on its own, so the result is the reference's exactly, in any
// grouping.  The entries of an event are sorted by time bucket into a list in global memory with a counting sort
// that keeps runs of consecutive samples together; a window of time buckets is cut into sequences of consecutive
// list entries, ten lanes (mesh lines) per sequence; a lane walks along its sequence with ten accumulators
// (pad, electrons) and emits a run to the wave's queue only when the pad under a pixel, the time bucket or the
// nucleus changes.  Same windows, table, insert loop and flush as the default kernel; its own instantiation.
template <bool MC, bool MERGE>
__global__ __launch_bounds__(SC_THREADS, (SC_THREADS * ATTPC_SC_WG_PER_CU + 255) / 256) void scatter_kernel(ScatterArgs a) {
  __shared__ ScatterShared sh;
  // tid / lane are re-read through an opaque asm at every use (macros below): otherwise the compiler
  // hoists every tid-derived LDS address of every phase to the top of the kernel, runs out of the 128
  // VGPRs a 1024-thread workgroup allows and reloads them from scratch memory (a global round trip
  // each) inside the per-window code
#define tid (fresh_tid())
#define lane (fresh_tid() & 63)
  const int n_sim = a.layout.n_sim;
  const int lut_n = a.det.lut_n, lut_lo = a.det.lut_lo;
  const int16_t* __restrict__ lut = a.det.pad_lut;
  const double* __restrict__ arena = a.trk.arena;
  const float spread = (float)((6.0 / 4.9e-3) * (6.0 / 4.9e-3) * 2.0 * a.det.diffusion * a.det.dv / a.det.efield);
  const int n_slices = a.det.longitudinal_diffusion > 0.0 ? ATTPC_LONG_STEPS : 1;
  constexpr int WQ = MERGE ? MERGE_ROUND : SC_THREADS / MESH;  // entries of one full pass (select_window)
  // aimed-at keys per window.  Merge variant: the runs reach the table in batches of a few hundred after long
  // stretches of arithmetic, so slower probes at a fuller table cost less than the windows they save (every window
  // ends every sequence's accumulators: up to 100 runs per sequence)
#ifndef ATTPC_SC_MERGE_TARGET_PCT
#define ATTPC_SC_MERGE_TARGET_PCT 50
#endif
  constexpr int TK = MERGE ? HASH_CAP * ATTPC_SC_MERGE_TARGET_PCT / 100 : TARGET_KEYS;
  // merge variant: this workgroup's two entry lists in global memory (list order / sorted by time bucket)
  uint2* __restrict__ const mg_list = MERGE ? a.merge_scratch + (size_t)blockIdx.x * 2u * a.merge_cap : nullptr;
  uint2* __restrict__ const mg_perm = MERGE ? mg_list + a.merge_cap : nullptr;

  PHASE_DECL;
  // ---- once per workgroup (persistent: it takes batches of events from a global counter) ----
  for (int p = tid; p < PIXELS; p += SC_THREADS) {
    const double di = (double)(p / MESH) - 4.5, dj = (double)(p % MESH) - 4.5;
    sh.wtab[p] = (36.0 / 81.0) / TWO_PI * exp(-(2.0 / 9.0) * (di * di + dj * dj));
  }
  if (tid < ATTPC_MAX_SIM) sh.label_of[tid] = (long long)a.layout.indices[tid];
  if (tid == 0) {
#pragma unroll
    for (int k = 0; k < ATTPC_LONG_STEPS; ++k) sh.long_w[k] = a.det.long_weights[k];
    sh.sigma_k[0] = 2.0 * a.det.diffusion * a.det.dv;
    sh.sigma_k[1] = a.det.efield;
    sh.failed = 0; sh.retried = 0; sh.charge_sum = 0ull; sh.key_sum = 0ull; sh.n_keys = 0u;
    sh.row_cur = 0ull; sh.row_end = 0ull; sh.seg_cur = 0ull; sh.seg_end = 0ull; sh.wg_samples = 0ull; sh.wg_rows = 0ull;
  }
  clear_table(sh);  // every flush leaves the table empty again
  unsigned long long my_charge = 0ull, my_keys = 0ull;

  // Events come in batches of a.batch from a global counter.  The returning atomic on that hot address
  // takes microseconds, so thread 0 asks for the NEXT batch while it works on the last event of the
  // current one: right after its rows pass of the first window, where wave 0 (which has issue
  // priority and finishes first) would otherwise just wait at the barrier for the other waves.
  auto take_batch = [&]() -> uint32_t {
    int zero = 0;
    asm volatile("" : "+v"(zero));  // opaque address: keeps LLVM's atomic optimizer (readfirstlane) away
    return (uint32_t)atomicAdd(&a.out.ctrl[CTRL_NEXT_EVENT + zero], (unsigned long long)a.batch);
  };
  uint32_t next_first = 0u;  // thread 0: first event of the next batch, once asked for
  bool have_next = false;
  if (tid == 0) sh.batch_first = take_batch();
  for (;;) {
    block_sync();  // the previous event is finished in every wave
    const uint32_t batch_first = sh.batch_first;
    if (batch_first >= a.n_events) break;
    const uint32_t batch_end = min(batch_first + a.batch, a.n_events);
    for (uint32_t e_local = batch_first; e_local < batch_end; ++e_local) {
      if (e_local != batch_first) block_sync();
      const bool last_of_batch = e_local + 1u == batch_end;
      const uint64_t event = a.first_event + e_local;
      const uint32_t track0 = (a.event0 + e_local) * (uint32_t)n_sim;
      // ---- per-event init ----
      for (int i = tid; i < ATTPC_NUM_TB; i += SC_THREADS) {
        sh.cum[i] = 0ull;
        reinterpret_cast<uint32_t*>(&sh.st_ix[0][0])[i] = 0u;  // per-bucket cursors of the entry sort
      }
      if (tid == 0) {
        int acc = 0;
        for (int k = 0; k < n_sim; ++k) {
          sh.cnt[k] = acc;
          acc += a.trk.counts[track0 + k];
        }
        for (int k = n_sim; k <= ATTPC_MAX_SIM; ++k) sh.cnt[k] = acc;
        const int zero = local_const(0);
        sh.win_a = zero; sh.win_b = zero; sh.budget = local_const(TK); sh.overflow = zero; sh.done = zero; sh.danger = zero;
        sh.ev_failed = zero;
        sh.ev_rows = 0ull;
        sh.wg_samples += (unsigned long long)acc;
      }
      // the arena block ids of the event's tracks, loaded together with the counts (entries past a
      // track's last block are never used)
      const int32_t* __restrict__ ev_table = a.trk.block_table + (size_t)track0 * MAX_BLOCKS_PER_TRACK;
      for (int i = tid; i < n_sim * LDS_BLOCKS; i += SC_THREADS) {
        const int k = i / LDS_BLOCKS, b = i - k * LDS_BLOCKS;
        sh.blocks[k][b] = ev_table[k * MAX_BLOCKS_PER_TRACK + b];
      }
      block_sync();
      const int total = sh.cnt[ATTPC_MAX_SIM];
      PHASE_MARK(0);

      // ---- histogram of kept samples per time bucket (all nuclei), then its prefix sum ----
      const int total_s = total * n_slices;  // entries = samples x slices
      // t < 0 (sigma_t would be NaN: undefined in the reference) and tb >= 512 (removed by the
      // 0 <= tb < 512 mask of simulator.py:111-113) never reach the output
      const bool sorted = MERGE || total_s <= SORT_CAP;  // few enough entries: sort them by time bucket once
      int my_tb[SORT_PER_THREAD];  // time bucket of this thread's entries tid, tid + SC_THREADS, ...
#pragma unroll
      for (int k = 0; k < SORT_PER_THREAD; ++k) my_tb[k] = -1;
      if constexpr (MERGE) {
        // Entry e of the list = slice e / total of sample e % total (slice major: the samples of a track stay
        // neighbours).  Every entry is resolved once -- arena record, time bucket, nucleus, slice -- and written to
        // the list; later passes never touch the block table again.
        for (int e0 = 0; e0 < total_s; e0 += 2 * SC_THREADS) {  // two entries per thread in flight
          const double* rec[2];
          const double* prev[2];
          int isim[2], sl[2];
          double t[2], t_prev[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = e0 + u * SC_THREADS + tid;
            sl[u] = 0;
            isim[u] = 0;
            rec[u] = arena;
            prev[u] = nullptr;
            if (e < total_s) {
              sl[u] = n_slices == 1 ? 0 : e / total;
              const int c = e - sl[u] * total;
              rec[u] = sample_ptr(sh, arena, ev_table, c, isim[u]);
              int isim_prev;
              if (c > sh.cnt[isim[u]]) prev[u] = sample_ptr(sh, arena, ev_table, c - 1, isim_prev);  // same track
            }
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            t[u] = rec[u][2];
            t_prev[u] = prev[u] != nullptr ? prev[u][2] : -1.0;
          }
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const int e = e0 + u * SC_THREADS + tid;
            if (e >= total_s) continue;
            uint32_t meta = MERGE_INVALID;
            if (t[u] >= 0.0) {
              const double ts = slice_time(a.det, t[u], sl[u], n_slices);
              if (ts >= 0.0 && ts < (double)ATTPC_NUM_TB) {
                meta = (uint32_t)(int)ts | ((uint32_t)isim[u] << 10) | ((uint32_t)sl[u] << 13);
                const double prev_ts = t_prev[u] >= 0.0 ? slice_time(a.det, t_prev[u], sl[u], n_slices) : -1.0;
                atomicAdd(&sh.cum[(int)ts], (1ull << 32) | (unsigned long long)key_estimate_on_track(rec[u], prev[u], t[u], (int)ts,
                                                                                                    prev_ts, spread));
              }
            }
            mg_list[e] = make_uint2((uint32_t)((rec[u] - arena) >> 2), meta);
          }
        }
        __threadfence();  // the list is read back (by the same threads) in the sort pass below
#ifdef ATTPC_PHASE_TIMERS
        PHASE_SYNC;
        PHASE_MARK(13);  // merge variant: list pass (printed as "staged")
#endif
      } else if (sorted) {
#pragma unroll
        for (int k = 0; k < SORT_PER_THREAD; ++k) {
          const int cs = tid + k * SC_THREADS;
          if (cs < total_s) {
            const int c = cs / n_slices;
            int isim, isim_prev;
            const double* rec = sample_ptr(sh, arena, nullptr, c, isim);
            // (the predecessor on the same track: c - 1 unless c is the track's first sample)
            const double* prev = c > sh.cnt[isim] ? sample_ptr(sh, arena, nullptr, c - 1, isim_prev) : nullptr;
            const double t = rec[2];
            const double t_prev = prev != nullptr ? prev[2] : -1.0;
            if (t >= 0.0) {
              const int sl = cs - c * n_slices;
              const double ts = slice_time(a.det, t, sl, n_slices);
              if (ts >= 0.0 && ts < (double)ATTPC_NUM_TB) {
                my_tb[k] = (int)ts;
                const double prev_ts = t_prev >= 0.0 ? slice_time(a.det, t_prev, sl, n_slices) : -1.0;
                atomicAdd(&sh.cum[my_tb[k]],
                          (1ull << 32) | (unsigned long long)key_estimate_on_track(rec, prev, t, my_tb[k], prev_ts, spread));
              }
            }
          }
        }
      } else {
        // an event too long to be sorted: its windows scan the entry list in chunks of SC_THREADS; the time-bucket
        // range of every chunk is recorded here, so that a window only looks at the chunks that can hold entries of
        // its own (samples follow their track, a window's entries sit in a few stretches of the list: configs[4]
        // has 39 chunks per event and ~27 windows, each of which now reads 3-4 chunks instead of all)
        for (int i = tid; i < MAX_CHUNKS; i += SC_THREADS) {
          chunk_lo(sh)[i] = 0xffffffffu;
          chunk_hi(sh)[i] = 0u;
        }
        block_sync();
        for (int c = tid; c < total; c += SC_THREADS) {
          int isim;
          const double t = sample_ptr(sh, arena, ev_table, c, isim)[2];
          if (!(t >= 0.0)) continue;
          const int est = key_estimate((int)fmin(t, 511.0), spread);
          for (int sl = 0; sl < n_slices; ++sl) {
            const double ts = slice_time(a.det, t, sl, n_slices);
            if (ts >= 0.0 && ts < (double)ATTPC_NUM_TB) {
              atomicAdd(&sh.cum[(int)ts], (1ull << 32) | (unsigned long long)est);
              chunk_range_add(sh, (c * n_slices + sl) / SC_THREADS, (int)ts);
            }
          }
        }
      }
      block_sync();
      {  // inclusive prefix sum over the 512 buckets: per-thread serial part, wave scan, wave offsets
        unsigned long long local[BINS_PER_THREAD];
        unsigned long long v = 0ull;
#pragma unroll
        for (int k = 0; k < BINS_PER_THREAD; ++k) {
          const int bin = tid * BINS_PER_THREAD + k;
          v += bin < ATTPC_NUM_TB ? sh.cum[bin] : 0ull;
          local[k] = v;
        }
        unsigned long long incl = v;
        for (int off = 1; off < 64; off <<= 1) {
          const unsigned long long up = __shfl_up(incl, off);
          incl += lane >= off ? up : 0ull;
        }
        if (lane == 63) sh.wave_sum[tid >> 6] = incl;
        block_sync();
        unsigned long long offset = incl - v;
        for (int w = 0; w < (tid >> 6); ++w) offset += sh.wave_sum[w];
#pragma unroll
        for (int k = 0; k < BINS_PER_THREAD; ++k) {
          const int bin = tid * BINS_PER_THREAD + k;
          if (bin < ATTPC_NUM_TB) sh.cum[bin] = local[k] + offset;
        }
      }
      block_sync();

      if constexpr (MERGE) {
        // Counting sort by time bucket that keeps RUNS together: consecutive list entries of the same nucleus, slice
        // and time bucket (the samples of a track while it stays inside one bucket) are placed as one block, by one
        // returning LDS atomic of the run's first lane -- so the sorted list is, bucket by bucket, a sequence of
        // stretches of track, and neighbours in it are neighbours in space.
        uint32_t* __restrict__ cursors = reinterpret_cast<uint32_t*>(&sh.st_ix[0][0]);
        for (int e0 = 0; e0 < total_s; e0 += SC_THREADS) {  // workgroup-uniform trip count (ballots inside)
          const int e = e0 + tid;
          unsigned long long raw = ~0ull;
          if (e < total_s)
            raw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(mg_list + e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t meta = (uint32_t)(raw >> 32);
          const bool valid = meta != MERGE_INVALID;
          const uint32_t prev = (uint32_t)__shfl_up((int)meta, 1);
          const bool head = valid && (lane == 0 || prev != meta);
          const unsigned long long hm = __ballot(head), vm = __ballot(valid);
          const unsigned long long upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);  // lanes 0 .. lane
          const unsigned long long mine = hm & upto;
          const int h = mine ? 63 - __clzll((long long)mine) : 0;  // first lane of this lane's run
          const unsigned long long breaks = (hm | ~vm) & ~upto;   // the run ends before the next head / invalid entry
          const int next = breaks ? __ffsll((long long)breaks) - 1 : 64;
          uint32_t pos = 0u;
          if (head) {
            const int tb = (int)(meta & 0x3ffu);
            pos = (tb > 0 ? (uint32_t)(sh.cum[tb - 1] >> 32) : 0u) + atomicAdd(&cursors[tb], (uint32_t)(next - lane));
          }
          pos = (uint32_t)__shfl((int)pos, h);
          if (valid) mg_perm[pos + (uint32_t)(lane - h)] = make_uint2((uint32_t)raw, meta);
        }
        __threadfence();  // read by other waves of the workgroup after the barrier below
      }
#pragma unroll
      for (int k = 0; k < SORT_PER_THREAD; ++k) {
        if (my_tb[k] >= 0) {  // counting sort: a window is then a contiguous range of perm[]
          const unsigned int before = my_tb[k] > 0 ? (unsigned int)(sh.cum[my_tb[k] - 1] >> 32) : 0u;
          sh.perm[before + atomicAdd(&reinterpret_cast<uint32_t*>(&sh.st_ix[0][0])[my_tb[k]], 1u)] =
              (unsigned short)(tid + k * SC_THREADS);
        }
      }
      if (tid < 64) select_window<WQ>(sh, 0, local_const(TK), lane);
      block_sync();
      PHASE_MARK(1);

      for (;;) {
        // the window was chosen by thread 0 before the barrier that ended the previous iteration
        if (sh.done) break;
        const int win_a = sh.win_a, win_b = sh.win_b;
        PHASE_MARK(2);

        // ---- scatter the window's samples of all nuclei ----
        // The (sample, slice) list is scanned in chunks of one entry per thread; the entries inside
        // the window are ranked (ballot prefix + wave totals) and staged densely, so a rows round runs
        // on a full staging buffer (or on the window's remainder) whatever the order of the samples.
        int round_lo = 0;  // rank of staging slot 0
        int filled = 0;    // in-window entries of the chunks before this one
        bool stop = false;
        // One rows round over the staged entries.  Returns false when the table is too full.
        // Monte-Carlo diffusion extension: a wave takes one staged entry at a time and moves its primary
        // electrons 64 at a time: Philox pair (index = electron, domain = 0x200 + entry) -> Box-Muller ->
        // position -> pad -> the same wave-level insert as the mesh path, int(w_slice * gain) electrons each.
        auto rows_round_mc = [&](int n_stage) -> bool {
          const int wave = tid >> 6;
          const McEntry* __restrict__ entries = reinterpret_cast<const McEntry*>(&sh.st_ix[0][0]);
          const double lo_mm = (double)lut_lo, hi_mm = (double)(lut_lo + lut_n);
          unsigned int claimed = 0u, diag_trips = 0u;
          bool ok = true;
          for (int st = wave; st < n_stage && ok; st += N_WAVES) {  // wave uniform
            const McEntry m = entries[st];
            const int tbw = sh.st_tb[st];
            const uint32_t word_hi = ((uint32_t)(tbw & 0x3ff) << 14) | ((uint32_t)((tbw >> 24) & 7) << 24);
            const uint32_t q = (uint32_t)sh.st_n[st];
            for (uint32_t k0 = 0; k0 < m.n_prim && ok; k0 += 64u) {
              const uint32_t k = k0 + (uint32_t)lane;
              double ua, ub;
              rng_pair(a.seed, event, k, DOMAIN_MC + m.cs, ua, ub);
              const double rad = sqrt(-2.0 * log(1.0 - ua));
              double sn, cs;
              sincos(TWO_PI * ub, &sn, &cs);
              // no FMA contraction: the oracle rounds the product before the sum
              const double x = __dadd_rn(m.x, __dmul_rn(m.sigma, __dmul_rn(rad, cs)));
              const double y = __dadd_rn(m.y, __dmul_rn(m.sigma, __dmul_rn(rad, sn)));
              const double fx = floor(x * 1000.0), fy = floor(y * 1000.0);
              const int ixx = (fx >= lo_mm && fx < hi_mm) ? (int)fx - lut_lo : lut_n;
              const int iyy = (fy >= lo_mm && fy < hi_mm) ? (int)fy - lut_lo : lut_n;
              const int pad = (int)lut[ixx * (lut_n + 1) + iyy];  // -1 off the plane / beam pad
              ok = wave_insert(sh, word_hi | (uint32_t)max(pad, 0), q, k < m.n_prim && pad >= 0, claimed, diag_trips);
            }
          }
          if (lane == 0 && claimed) atomicAdd(&sh.n_keys, claimed);
          return ok;
        };
        // `carry`: the runs a lane has not finished (lost claim, full bucket) ride along from block to block -- and, in
        // a sorted event, from round to round of a window: only the window's last round drains them (`drain`), so the
        // sparse trips at the end of every round (a quarter of all trips) are not paid 7 times per event.
        auto rows_round = [&](int n_stage, InsertCarry& carry, bool drain) -> bool {
          if constexpr (MC) return rows_round_mc(n_stage);
          const int n_rows = n_stage * MESH;
          const int wave = tid >> 6;
          uint2* __restrict__ queue = sh.queue[wave];
          const uint32_t qbase = (uint32_t)(uintptr_t)queue;  // LDS byte address of the wave's queue
          const char* __restrict__ lut_bytes = reinterpret_cast<const char*>(lut);
          // LUT [x][y] with one extra row and column of -1: index lut_n stands for "off the pad plane",
          // so off-plane pixels, missing rows and rows handled elsewhere need no masks.  A lane holds
          // one y (`ix` below, the fast LUT index) and steps through ten x (`iy[j]`): the lanes of a
          // sample read neighbouring addresses.
          const unsigned int row_pitch = 2u * (unsigned int)(lut_n + 1);  // bytes per iy row
          unsigned int claimed = 0u;  // new keys of this wave (wave uniform)
          unsigned int diag_trips = 0u, diag_calls = 0u;
          (void)diag_calls;  // diagnostic builds only
          bool ok = true;
#ifdef ATTPC_SC_SKEW  // experiment: start half of the waves late, so that their LDS-bound insert phases meet the
                      // VALU-bound row phases of the others
          if (wave >= N_WAVES / 2) __builtin_amdgcn_s_sleep(ATTPC_SC_SKEW);
#endif
          for (int row0 = wave * 64; row0 < n_rows; row0 += SC_THREADS) {  // wave-uniform trip count
            const int row = min(row0 + lane, n_rows - 1);
            const bool have = row0 + lane < n_rows;
            const int st = row / MESH;
            const int i = row - st * MESH;
            const int tbw = sh.st_tb[st];
            const bool point = (tbw & (1 << 30)) != 0;
            const uint32_t word_hi = ((uint32_t)(tbw & 0x3ff) << 14) | ((uint32_t)((tbw >> 24) & 7) << 24);
            const double n_el = sh.st_n[st];
            const int ix = sh.st_ix[st][i];
            // the row's 10 iy indices (5 dwords) and weights (5 x 16 bytes)
            const uint32_t* __restrict__ iy32 = reinterpret_cast<const uint32_t*>(&sh.st_iy[st][0]);
            const double2* __restrict__ w2 = reinterpret_cast<const double2*>(&sh.wtab[i * MESH]);
            unsigned int iy[MESH];
            double w[MESH];
#pragma unroll
            for (int j = 0; j < MESH; j += 2) {
              const uint32_t pair = iy32[j >> 1];
              iy[j] = pair & 0xffffu;
              iy[j + 1] = pair >> 16;
              const double2 ww = w2[j >> 1];
              w[j] = ww.x;
              w[j + 1] = ww.y;
            }
            // per-pixel electrons int(pdf h^2 n) (transporter.py:240-246) as u32; the centre pixel is the
            // largest of the row, so one check bounds every run total of the row below 2^32
            uint32_t el[MESH];
#pragma unroll
            for (int j = 0; j < MESH; ++j) el[j] = (uint32_t)(w[j] * n_el);  // cvt truncates
            const bool big = el[MESH / 2] >= (1u << 28);
            // point_transport (transporter.py:123-169, sigma == 0: all electrons straight down, row 0 /
            // pixel 0 stand for the sample) and rows too large for u32 go pixel by pixel into the table
            const bool slow = have && ix != lut_n && (point || big);
            int pad[MESH];
            {
              // 10 independent gathers in flight (32-bit byte offsets from the uniform base).  The empty
              // asm takes all ten results: without it the compiler sinks each load into a branch on its
              // first use and waits there -- ten serial L2 round trips.
              const unsigned int col = 2u * (unsigned int)((have && !slow) ? ix : lut_n);
#pragma unroll
              for (int j = 0; j < MESH; ++j)
                pad[j] = (int)*reinterpret_cast<const int16_t*>(lut_bytes + (__umul24(iy[j], row_pitch) + col));
              asm volatile("" : "+v"(pad[0]), "+v"(pad[1]), "+v"(pad[2]), "+v"(pad[3]), "+v"(pad[4]), "+v"(pad[5]),
                           "+v"(pad[6]), "+v"(pad[7]), "+v"(pad[8]), "+v"(pad[9]));
            }
#ifdef ATTPC_PHASE_TIMERS
            PHASE_SYNC;
            PHASE_MARK(8);
#endif
            bool slow_ok = true;
            if (__any(slow)) {
              if (slow) {
#pragma unroll 1
                for (int j = 0; j < MESH && slow_ok; ++j) {
                  if (point && (i != 0 || j != 0)) continue;
                  const int p = (int)*reinterpret_cast<const int16_t*>(
                      lut_bytes + (__umul24(iy[j], row_pitch) + 2u * (unsigned int)ix));
                  const double q = (point ? 1.0 : sh.wtab[i * MESH + j]) * n_el;
                  if (p >= 0) slow_ok = table_add(sh, word_hi | (uint32_t)p, (unsigned long long)q);
                }
              }
            }
            // Merge runs of equal pads: a run's total sits with its last pixel, `ended[j]` = the pad of a run that ends at
            // pixel j (-1: none ends there, or the run lies off the pad plane: a stretch of -1 adds up to a total nobody
            // reads).  Conditions stay lane masks in scalar registers (one ballot of a comparison per pixel); the first
            // version's per-lane bit fields cost twice the instructions (round 3, profiles/r03_scatter_phases.md).
            uint32_t run_q[MESH];
            int ended[MESH];
            unsigned long long mask[MESH];
            int wave_total = 0;
            {
              uint32_t acc = 0u;
#pragma unroll
              for (int j = 0; j < MESH; ++j) {
                acc += el[j];
                run_q[j] = acc;
                if (j < MESH - 1) {
                  const bool last = pad[j + 1] != pad[j];
                  ended[j] = last ? pad[j] : -1;
                  acc = last ? 0u : acc;
                } else {
                  ended[j] = pad[j];
                }
                mask[j] = __builtin_amdgcn_ballot_w64(ended[j] >= 0);
                wave_total += (int)__popcll(mask[j]);
              }
            }
#ifdef ATTPC_PHASE_TIMERS
            asm volatile("" ::"v"(ended[0]), "v"(run_q[9]));
            PHASE_SYNC;
            PHASE_MARK(9);
#endif
            if (wave_total <= WAVE_QUEUE) {
              // queue positions pixel by pixel: the runs of the pixels before (scalar popcounts) + the lanes below in
              // the pixel's own mask (mbcnt)
              queue_put((uint32_t)__builtin_amdgcn_readfirstlane((int)qbase), mask, word_hi, ended, run_q);
#ifdef ATTPC_PHASE_TIMERS
              PHASE_SYNC;
              PHASE_MARK(10);
#endif
#ifndef ATTPC_ABL_NOINSERT  // (ablation builds only: how long does the kernel take without the table inserts)
              ok = stream_insert(sh, queue, wave_total, false, carry, claimed, diag_trips);
#endif
              diag_calls += (unsigned int)(wave_total + 63) / 64u;
#ifdef ATTPC_PHASE_TIMERS
              PHASE_SYNC;
              PHASE_MARK(11);
#endif
            } else {  // more runs than the queue holds (nearly every pixel on a pad of its own): in passes
              for (int pass0 = 0; pass0 < wave_total && ok; pass0 += WAVE_QUEUE) {
                int base = -pass0;
#pragma unroll
                for (int j = 0; j < MESH; ++j) {
                  const unsigned long long mk = mask[j];
                  const int e = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                  if (ended[j] >= 0 && e >= 0 && e < WAVE_QUEUE) queue[e] = make_uint2(word_hi | (uint32_t)ended[j], run_q[j]);
                  base += (int)__popcll(mk);
                }
                const int n_q = min(wave_total - pass0, WAVE_QUEUE);
#ifndef ATTPC_ABL_NOINSERT
                ok = stream_insert(sh, queue, n_q, false, carry, claimed, diag_trips);
#endif
                diag_calls += (unsigned int)(n_q + 63) / 64u;
              }
            }
            ok = ok && !__any(!slow_ok);  // the slow path fails in single lanes
            if (!ok) break;    // table too full: the whole wave stops together
          }
          if (ok && drain) ok = stream_insert(sh, queue, 0, true, carry, claimed, diag_trips);  // the runs still under way
          if (carry.danger >> 31) sh.danger = 1;  // (never, for the workloads of BASELINE.json: ADD_LIMIT)
          if (lane == 0 && claimed) atomicAdd(&sh.n_keys, claimed);  // rows of the window's flush
          PHASE_COUNT(5, (unsigned long long)diag_trips + ((unsigned long long)diag_calls << 32));
          return ok;
        };

        // Merge variant of the rows phase.  Lane = mesh line i of sequence `sub` of its wave (lanes 60..63 idle); per
        // round it steps through the MERGE_G staged entries of its sequence.  Per step: the line's ten pads (gathers)
        // and pixel charges as in rows_round(); a pixel whose pad, time bucket and nucleus are those of the lane's
        // accumulator adds to it, any other one first sends the accumulator to the wave's queue as a run.  The queue is
        // handed to stream_insert() when it is nearly full, and drained at the window's last round (`last`).
        auto rows_round_merge = [&](MergeAcc& m, InsertCarry& carry, bool last) -> bool {
          const int wave = tid >> 6;
          const MergeStage ms = merge_stage(sh);
          uint2* __restrict__ queue = sh.queue[wave];
          const uint32_t qbase = (uint32_t)(uintptr_t)queue;  // LDS byte address of the wave's queue
          const char* __restrict__ lut_bytes = reinterpret_cast<const char*>(lut);
          const unsigned int row_pitch = 2u * (unsigned int)(lut_n + 1);
          unsigned int claimed = 0u, diag_trips = 0u;
          bool ok = true;
          const int sub = lane / MESH;
          const int i = lane - sub * MESH;
          const bool lane_ok = sub < MERGE_SEQ_PER_WAVE;
          const int slot0 = (wave * MERGE_SEQ_PER_WAVE + min(sub, MERGE_SEQ_PER_WAVE - 1)) * MERGE_G;
          // The accumulators whose pad is given in `ended[]` (-1: this one goes on) -> runs in the wave's queue, column by
          // column: a run's place is the fill + the runs of the columns before it (scalar popcounts) + the lanes below
          // it in its own column (mbcnt).  Conditions stay lane masks in scalar registers, nothing is turned into 0/1
          // vector values and back (the per-lane bit fields of the first version cost ~15 instructions a pixel; a
          // ballot of anything but a comparison costs two -- hence the -1 convention instead of a flag).
          auto emit = [&](const int (&ended)[MESH]) {
            int wave_total = 0;
            unsigned long long mask[MESH];
#pragma unroll
            for (int j = 0; j < MESH; ++j) {
              mask[j] = __builtin_amdgcn_ballot_w64(ended[j] >= 0);
              wave_total += (int)__popcll(mask[j]);
            }
            if (wave_total == 0) return;
            if (m.fill + wave_total > WAVE_QUEUE) {  // no room: the queued runs go to the table first
#ifndef ATTPC_ABL_NOINSERT  // (ablation builds only)
              ok = stream_insert(sh, queue, m.fill, false, carry, claimed, diag_trips);
#endif
              m.fill = 0;
              if (!ok) return;
            }
            if (wave_total <= WAVE_QUEUE) {
              const uint32_t fill0 = (uint32_t)__builtin_amdgcn_readfirstlane(m.fill);
              queue_put((uint32_t)__builtin_amdgcn_readfirstlane((int)qbase) + 8u * fill0, mask, m.hi, ended, m.q);
              m.fill = (int)fill0 + wave_total;
            } else {  // more runs than the queue holds (every lane changed bucket at once): in passes
              for (int pass0 = 0; pass0 < wave_total && ok; pass0 += WAVE_QUEUE) {
                int base = -pass0;
#pragma unroll  // (not rolled: an index that is not a constant would send the accumulators to scratch memory)
                for (int j = 0; j < MESH; ++j) {
                  const unsigned long long mk = mask[j];
                  const int e = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
                  if (ended[j] >= 0 && e >= 0 && e < WAVE_QUEUE) queue[e] = make_uint2(m.hi | (uint32_t)ended[j], m.q[j]);
                  base += (int)__popcll(mk);
                }
#ifndef ATTPC_ABL_NOINSERT
                ok = stream_insert(sh, queue, min(wave_total - pass0, WAVE_QUEUE), false, carry, claimed, diag_trips);
#endif
              }
            }
          };
          for (int qs = 0; qs < MERGE_G && ok; ++qs) {  // wave-uniform trip count
            const int st = slot0 + qs;
            const int tbw = ms.tb[st];
            const bool have = lane_ok && tbw >= 0;
            const bool point = (tbw & (1 << 30)) != 0;
            const uint32_t word_hi = ((uint32_t)(tbw & 0x3ff) << 14) | ((uint32_t)((tbw >> 24) & 7) << 24);
            const double n_el = have ? ms.n[st] : 0.0;
            const int ix = ms.ix[st][i];
            const uint32_t* __restrict__ iy32 = reinterpret_cast<const uint32_t*>(&ms.iy[st][0]);
            const double2* __restrict__ w2 = reinterpret_cast<const double2*>(&sh.wtab[i * MESH]);
            unsigned int iy[MESH];
            double w[MESH];
#pragma unroll
            for (int j = 0; j < MESH; j += 2) {
              const uint32_t pair = iy32[j >> 1];
              iy[j] = pair & 0xffffu;
              iy[j + 1] = pair >> 16;
              const double2 ww = w2[j >> 1];
              w[j] = ww.x;
              w[j + 1] = ww.y;
            }
            uint32_t el[MESH];
#pragma unroll
            for (int j = 0; j < MESH; ++j) el[j] = (uint32_t)(w[j] * n_el);  // cvt truncates (transporter.py:240-246)
            const bool big = el[MESH / 2] >= (1u << 28);
            const bool slow = have && ix != lut_n && (point || big);  // as in rows_round(): straight into the table
            int pad[MESH];
            {
#ifndef ATTPC_ABL_NOGATHER
              const unsigned int col = 2u * (unsigned int)((have && !slow) ? ix : lut_n);
#pragma unroll
              for (int j = 0; j < MESH; ++j)
                pad[j] = (int)*reinterpret_cast<const int16_t*>(lut_bytes + (__umul24(iy[j], row_pitch) + col));
              asm volatile("" : "+v"(pad[0]), "+v"(pad[1]), "+v"(pad[2]), "+v"(pad[3]), "+v"(pad[4]), "+v"(pad[5]),
                           "+v"(pad[6]), "+v"(pad[7]), "+v"(pad[8]), "+v"(pad[9]));
#else  // (ablation builds only, wrong results: what do the gathers cost)
#pragma unroll
              for (int j = 0; j < MESH; ++j) pad[j] = (have && !slow) ? (int)(((iy[j] >> 2) * 64u + ((unsigned int)ix >> 2)) & 0x1fffu) : -1;
#endif
            }
#ifdef ATTPC_PHASE_TIMERS
            PHASE_SYNC;
            PHASE_MARK(8);
#endif
            bool slow_ok = true;
            if (__any(slow)) {
              if (slow) {
#pragma unroll 1
                for (int j = 0; j < MESH && slow_ok; ++j) {
                  if (point && (i != 0 || j != 0)) continue;
                  const int p = (int)*reinterpret_cast<const int16_t*>(
                      lut_bytes + (__umul24(iy[j], row_pitch) + 2u * (unsigned int)ix));
                  const double q = (point ? 1.0 : sh.wtab[i * MESH + j]) * n_el;
                  if (p >= 0) slow_ok = table_add(sh, word_hi | (uint32_t)p, (unsigned long long)q);
                }
              }
            }
            // which accumulators end here: another time bucket / nucleus ends all ten, another pad ends one; all ten
            // are ended as well before one of them could pass 2^31 (a pixel adds less than 2^28): the u32 sums cannot wrap
            // (the centre pixel bounds the other nine -- the weights fall off from the centre -- and is below 2^28 here)
            // (bitwise on purpose: `&&` / `||` become branches and 0/1 vector values, `&` / `|` stay lane masks)
            const bool restart = have & ((word_hi != m.hi) | (m.bound >= 0x70000000u));
            bool chg[MESH];
            int ended[MESH];
#pragma unroll
            for (int j = 0; j < MESH; ++j) {
              chg[j] = have & (restart | (pad[j] != m.pad[j]));
              ended[j] = chg[j] ? m.pad[j] : -1;
            }
#ifdef ATTPC_PHASE_TIMERS
            PHASE_SYNC;
            PHASE_MARK(9);
#endif
            emit(ended);
#ifdef ATTPC_PHASE_TIMERS
            PHASE_SYNC;
            PHASE_MARK(10);
#endif
#pragma unroll
            for (int j = 0; j < MESH; ++j) {
              m.pad[j] = chg[j] ? pad[j] : m.pad[j];
              m.q[j] = (chg[j] ? 0u : m.q[j]) + el[j];  // (el = 0 without an entry; also where there is no pad: never emitted)
            }
            // (a row that went straight to the table left no real pad behind: every accumulator restarts after it)
            m.bound = slow ? 0u : (restart ? 0u : m.bound) + el[MESH / 2];  // (no entry: no restart, el = 0)
            m.hi = have ? word_hi : m.hi;
            ok = ok && !__any(!slow_ok);
#ifndef ATTPC_ABL_NOPRIO  // (ablation builds only)
            {
              // The waves of a window end at one barrier, and the instruction arbiter serves the oldest wave of a SIMD
              // first: left alone, waves 0-3 reach the barrier 50 k cycles before waves 12-15 and each SIMD ends the
              // window on a single wave (nothing to hide its latencies behind).  A wave therefore takes a priority
              // that grows with the steps it lags behind the foremost wave.
              const int steps = __builtin_amdgcn_readfirstlane(m.steps) + 1;
              m.steps = steps;
              if (lane == 0) atomicMax(&sh.merge_lead, steps);
              const int lag = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int*>(&sh.merge_lead)) - steps;
              if (lag >= 3) __builtin_amdgcn_s_setprio(3);
              else if (lag == 2) __builtin_amdgcn_s_setprio(2);
              else if (lag == 1) __builtin_amdgcn_s_setprio(1);
              else __builtin_amdgcn_s_setprio(0);
            }
#endif
          }
          if (last && ok) {  // the window ends: every accumulator becomes a run, the queue and the runs under way drain
            emit(m.pad);
#ifndef ATTPC_ABL_NOINSERT
            if (ok) ok = stream_insert(sh, queue, m.fill, true, carry, claimed, diag_trips);
#endif
            m.fill = 0;
#ifdef ATTPC_PHASE_TIMERS
            PHASE_SYNC;
            PHASE_MARK(11);
#endif
          }
          if (carry.danger >> 31) sh.danger = 1;
          if (lane == 0 && claimed) atomicAdd(&sh.n_keys, claimed);  // rows of the window's flush
          PHASE_COUNT(5, (unsigned long long)diag_trips);
          return ok;
        };

        // one (sample, slice) entry -> staging slot: sigma_t and the LUT indices of its 20 mesh lines
        auto stage_entry = [&](int slot, double2 xy, double2 tn, int isim, int sl, int cs) {
          const int tb = (int)slice_time(a.det, tn.x, sl, n_slices);  // transporter.py:238
          const double sigma = sqrt(2.0 * a.det.diffusion * a.det.dv * tn.x / a.det.efield);  // :301
          if constexpr (MC) {  // extension: the electrons are moved one by one in rows_round_mc
            McEntry m;
            m.x = xy.x; m.y = xy.y; m.sigma = sigma;
            m.n_prim = (uint32_t)(tn.y / (double)a.det.mpgd_gain32);  // tn.y = electrons x gain, an exact multiple
            m.cs = (uint32_t)cs;
            reinterpret_cast<McEntry*>(&sh.st_ix[0][0])[slot] = m;
            sh.st_n[slot] = (double)(long long)((n_slices == 1 ? 1.0 : a.det.long_weights[sl]) * (double)a.det.mpgd_gain32);
            sh.st_tb[slot] = tb | (isim << 24);
            return;
          }
          // numpy.linspace(c - 3 sigma, c + 3 sigma, 10) (:221-227) and position_to_index
          // (:107-118: whole-mm floor, low edge inclusive, high edge exclusive) per mesh line
          const bool sane = sigma >= 0.0;  // (a NaN would convert to 0, the middle of the table: sent far outside instead)
          const double xlo = sane ? xy.x - 3.0 * sigma : 1.0e300, xhi = sane ? xy.x + 3.0 * sigma : 1.0e300;
          const double ylo = sane ? xy.y - 3.0 * sigma : 1.0e300, yhi = sane ? xy.y + 3.0 * sigma : 1.0e300;
          const double sx = (xhi - xlo) / (double)(MESH - 1), sy = (yhi - ylo) / (double)(MESH - 1);
          auto mesh_line = [&](int i) {
            const double x = (i == MESH - 1) ? xhi : (double)i * sx + xlo;
            const double y = (i == MESH - 1) ? yhi : (double)i * sy + ylo;
            // lanes are mesh lines of constant y that step through x: 8 % fewer runs than the other way
            // round on the AT-TPC pad plane (the weights are symmetric, so the pixels are the same).
            // The range test is made on the whole number (v_cvt saturates: a floor outside int32 stays outside the
            // table; unsigned, so below the low edge is above the high one): two instructions instead of five.
            const short vy = (short)min((unsigned int)((int)floor(x * 1000.0) - lut_lo), (unsigned int)lut_n);
            const short vx = (short)min((unsigned int)((int)floor(y * 1000.0) - lut_lo), (unsigned int)lut_n);
            if constexpr (MERGE) {
              const MergeStage ms = merge_stage(sh);
              ms.iy[slot][i] = vy;
              ms.ix[slot][i] = vx;
            } else {
              sh.st_iy[slot][i] = vy;
              sh.st_ix[slot][i] = vx;
            }
          };
          if constexpr (MERGE) {
            // staged inside the rows loop with the accumulators live: rolled, so that the ten lines' temporaries do
            // not all need registers at once
#pragma unroll 1
            for (int i = 0; i < MESH; ++i) mesh_line(i);
          } else {
#pragma unroll
            for (int i = 0; i < MESH; ++i) mesh_line(i);
          }
          const double n_staged = (n_slices == 1 ? 1.0 : a.det.long_weights[sl]) * tn.y;  // x 1.0 is exact
          const int tb_staged = tb | (isim << 24) | ((sigma == 0.0) ? (1 << 30) : 0);
          if constexpr (MERGE) {
            const MergeStage ms = merge_stage(sh);
            ms.n[slot] = n_staged;
            ms.tb[slot] = tb_staged;
          } else {
            sh.st_n[slot] = n_staged;
            sh.st_tb[slot] = tb_staged;
          }
        };
        // Merge variant: the same entry -> staging slot, by TWO lanes (the rounds of this variant stage a few entries
        // per wave with most lanes idle, and staging was a third of the variant's time): both compute sigma_t, lane
        // `axis` 0 then takes the ten mesh lines of constant x, lane 1 those of constant y.  Same arithmetic per line
        // as stage_entry() (the range test is made on the whole number instead of the double: v_cvt saturates, so a
        // floor outside int32 stays outside the table).
        auto stage_entry_merge = [&](int slot, int axis, double2 xy, double2 tn, int isim, int sl) {
          const MergeStage ms = merge_stage(sh);
          const int ns = local_const(n_slices);  // (as a hoisted 0/1 vector value it was spilled: see sigma_k)
          const int tb = (int)slice_time(a.det, tn.x, sl, ns);  // transporter.py:238
          const double sigma = sqrt(sh.sigma_k[0] * tn.x / sh.sigma_k[1]);  // :301 (2 D dv t / E, left to right)
          const double c = axis ? xy.y : xy.x;
          const bool sane = sigma >= 0.0;  // (a NaN would convert to 0, the middle of the table: sent far outside instead)
          const double lo = sane ? c - 3.0 * sigma : 1.0e300, hi = sane ? c + 3.0 * sigma : 1.0e300;
          const double step = (hi - lo) / (double)(MESH - 1);  // numpy.linspace(c - 3 sigma, c + 3 sigma, 10) (:221-227)
          // x lines are the ones a lane of the rows phase steps through (iy), its own line is a y line (ix): stage_entry()
          short* __restrict__ out = axis ? &ms.ix[slot][0] : &ms.iy[slot][0];
          auto line = [&](int i, double v) {
            const int k = (int)floor(v * 1000.0) - lut_lo;  // position_to_index (:107-118): whole-mm floor
            out[i] = (short)min((unsigned int)k, (unsigned int)lut_n);  // low edge inclusive, high edge exclusive
          };
#pragma unroll 1
          for (int i = 0; i < MESH - 1; ++i) line(i, (double)i * step + lo);
          line(MESH - 1, hi);
          if (axis == 0) {
            ms.n[slot] = (ns == 1 ? 1.0 : sh.long_w[sl]) * tn.y;  // x 1.0 is exact
            ms.tb[slot] = tb | (isim << 24) | ((sigma == 0.0) ? (1 << 30) : 0);
          }
        };
        if constexpr (MERGE) {
          // The window is mg_perm[win_r0 .. win_r0 + win_n), cut into MERGE_NSEQ sequences of seq_len consecutive
          // entries.  Every WAVE works on its own MERGE_SEQ_PER_WAVE sequences without a workgroup barrier inside the
          // window: per round a few of its lanes stage the next MERGE_G entries of each of its sequences into the wave's
          // own staging slots, then all of its lanes step through them (rows_round_merge), accumulators in registers
          // from round to round.  The waves drift apart, so one wave's staging loads and table inserts overlap the
          // others' arithmetic.
          const int r0 = sh.win_r0, n_win = sh.win_n;
          const int n_rounds = (n_win + MERGE_ROUND - 1) / MERGE_ROUND;
          const int seq_len = n_rounds * MERGE_G;
          constexpr int WAVE_SLOTS = MERGE_SEQ_PER_WAVE * MERGE_G;  // staging slots of one wave
          static_assert(WAVE_SLOTS <= 32, "two lanes per staging slot of the wave");
          MergeAcc acc;
          acc.reset();
          InsertCarry carry;
          carry.reset();
          bool ok = true;
          // the previous window's flush resets its table slots and reads its slot list (the queues) without a barrier
          // behind it (it counts on the first barrier of the next staging): here the waves start on their own
          if (tid == 0) sh.merge_lead = 0;
          block_sync();
          for (int r = 0; r < n_rounds && ok; ++r) {  // wave-uniform
            // (Tried and dropped, profiles/r03_scatter_phases.md: loads that "touch" what the next round will read -- the
            //  next words of the list, the next records of the arena -- so that the round's two dependent loads hit the
            //  L2.  Loads return in order: the first wait for a gather is a wait for the touches as well, so they only
            //  move the miss from the staging to the first step; +3 % kernel time.)
            if ((lane & 31) < WAVE_SLOTS) {  // lanes 0.. take the x lines of a slot, lanes 32.. its y lines
              const int axis = lane >> 5;
              const int slot = (tid >> 6) * WAVE_SLOTS + (lane & 31);
              const int sq = slot / MERGE_G;
              const int rho = sq * seq_len + r * MERGE_G + (slot - sq * MERGE_G);
              if (rho < n_win) {
                const unsigned long long raw = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(mg_perm + r0 + rho),
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double* rec = arena + (size_t)(uint32_t)raw * 4u;
                const uint32_t meta = (uint32_t)(raw >> 32);
                const double2 rec_xy = reinterpret_cast<const double2*>(rec)[0], rec_tn = reinterpret_cast<const double2*>(rec)[1];
                stage_entry_merge(slot, axis, rec_xy, rec_tn, (int)((meta >> 10) & 7u), (int)((meta >> 13) & 7u));
              } else {
                // no entry: the lanes of this slot skip the step -- their gathers still run, so the slot's indices
                // must point into the table ("off the pad plane"), not at whatever the LDS held
                const MergeStage ms = merge_stage(sh);
                if (axis == 0) {
                  ms.tb[slot] = -1;
                  ms.n[slot] = 0.0;
                }
                short* __restrict__ out = axis ? &ms.ix[slot][0] : &ms.iy[slot][0];
#pragma unroll
                for (int k = 0; k < MESH; ++k) out[k] = (short)lut_n;
              }
            }
            // (LDS operations of one wave complete in order: a fence for the compiler is all that is needed between
            //  the lanes that stage and the lanes that read, and between this round's reads and the next round's staging)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            PHASE_MARK(3);
            PHASE_COUNT(12, 1);
            ok = rows_round_merge(acc, carry, r + 1 == n_rounds);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (*reinterpret_cast<volatile int*>(&sh.overflow)) break;  // another wave gave up on this window already
          }
          __builtin_amdgcn_s_setprio(0);
          if (!ok) sh.overflow = 1;
          if (last_of_batch && !have_next && tid == 0) {
            next_first = take_batch();
            have_next = true;
          }
          block_sync();
          PHASE_MARK(4);
        } else if (sorted) {  // the window is perm[win_r0 .. win_r0 + win_n): stage it densely, STAGE entries per round
          const int r0 = sh.win_r0, n_win = sh.win_n;
          InsertCarry carry;
          carry.reset();
          for (int base = 0; base < n_win; base += STAGE) {
            const int n_stage = min(STAGE, n_win - base);
            if (tid < n_stage) {
              const int cs = (int)sh.perm[r0 + base + tid];
              const int c = cs / n_slices;
              int isim;
              const double* rec = sample_ptr(sh, arena, nullptr, c, isim);
              stage_entry(tid, reinterpret_cast<const double2*>(rec)[0], reinterpret_cast<const double2*>(rec)[1], isim,
                          cs - c * n_slices, cs);
            }
            block_sync();
            PHASE_MARK(3);
            PHASE_COUNT(12, 1);
            PHASE_COUNT(13, n_stage);
            PHASE_COUNT(14, (n_stage * MESH + SC_THREADS - 1) / SC_THREADS);
            const bool ok = rows_round(n_stage, carry, base + STAGE >= n_win);
            if (!ok) sh.overflow = 1;
            if (last_of_batch && !have_next && tid == 0) {
              next_first = take_batch();
              have_next = true;
            }
            block_sync();
            PHASE_MARK(4);
            if (sh.overflow) break;  // uniform: every thread sees the flag after the barrier
          }
        }
        // (an unsorted event: only the chunks whose time-bucket range meets the window are read)
        const int n_chunks = sorted ? 0 : (total_s + SC_THREADS - 1) / SC_THREADS;
        int chunk_next = sorted ? 0 : next_chunk_in_window(sh, 0, n_chunks, win_a, win_b, lane);
        for (int chunk = 0; chunk_next < n_chunks && !stop; ++chunk) {  // `chunk` counts the chunks read (buffer parity)
          const int c0 = chunk_next * SC_THREADS;
          chunk_next = next_chunk_in_window(sh, chunk_next + 1, n_chunks, win_a, win_b, lane);
          bool in_win = false;
          double2 xy = {0.0, 0.0}, tn = {0.0, 0.0};
          int isim = 0, sl = 0;
          if (c0 + tid < total_s) {  // the whole record in one round trip, in the window or not
            const int c = (c0 + tid) / n_slices;
            sl = (c0 + tid) - c * n_slices;
            const double* rec = sample_ptr(sh, arena, ev_table, c, isim);
            xy = reinterpret_cast<const double2*>(rec)[0];
            tn = reinterpret_cast<const double2*>(rec)[1];
            const double ts = tn.x >= 0.0 ? slice_time(a.det, tn.x, sl, n_slices) : -1.0;
            in_win = ts >= 0.0 && ts < (double)ATTPC_NUM_TB && (int)ts >= win_a && (int)ts < win_b;
          }
          const unsigned long long bal = __ballot(in_win);
          if (lane == 0) sh.stage_sum[chunk & 1][tid >> 6] = __popcll(bal);
          block_sync();
          // rank among the window's entries; entries outside the window never match a staging slot
          int rank = in_win ? filled + (int)__popcll(bal & ((1ull << lane) - 1ull)) : -(1 << 30);
          int filled_new = filled;
#pragma unroll
          for (int w = 0; w < N_WAVES; ++w) {
            const int n_w = sh.stage_sum[chunk & 1][w];
            rank += w < (tid >> 6) ? n_w : 0;
            filled_new += n_w;
          }
          const bool last_chunk = chunk_next >= n_chunks;  // no later chunk holds entries of this window
          if (rank - round_lo >= 0 && rank - round_lo < STAGE) stage_entry(rank - round_lo, xy, tn, isim, sl, c0 + tid);
          for (;;) {
            const int pending = filled_new - round_lo;  // staged entries, workgroup uniform
            if (pending < STAGE && !(last_chunk && pending > 0)) break;  // keep filling / nothing left
            const int n_stage = min(pending, STAGE);
            block_sync();
            PHASE_MARK(3);
            PHASE_COUNT(12, 1);                                          // rows rounds
            PHASE_COUNT(13, n_stage);                                    // staged entries
            PHASE_COUNT(14, (n_stage * MESH + SC_THREADS - 1) / SC_THREADS);  // 64-row passes of the busiest wave

            // rows -> runs -> this wave's queue -> table, 64 mesh rows per wave at a time
            InsertCarry round_carry;
            round_carry.reset();
            const bool ok = rows_round(n_stage, round_carry, true);
            if (!ok) sh.overflow = 1;
            if (last_of_batch && !have_next && tid == 0) {
              next_first = take_batch();
              have_next = true;
            }
            block_sync();
            PHASE_MARK(4);
            if (sh.overflow) {  // uniform: every thread sees the flag after the barrier
              stop = true;
              break;
            }
            round_lo += n_stage;
            if (filled_new <= round_lo) break;
            if (rank - round_lo >= 0 && rank - round_lo < STAGE) {  // rest of this chunk: the records were not
              const int c = (c0 + tid) / n_slices;                   // kept in registers across the rows phase
              int isim2;
              const double* rec = sample_ptr(sh, arena, ev_table, c, isim2);
              stage_entry(rank - round_lo, reinterpret_cast<const double2*>(rec)[0], reinterpret_cast<const double2*>(rec)[1],
                          isim2, (c0 + tid) - c * n_slices, c0 + tid);
            }
          }
          filled = filled_new;
        }

        if (sh.danger) {  // (uniform: every thread sees the flag after the barrier that ended the rows phase)
          // A sum of this window reached 2^31: a later add could have wrapped its u32.  Nothing of the window is kept:
          // each of its time buckets that holds entries goes to lone_bucket_kernel, which scatters it again into u64
          // sums (exact; as slow as that kernel is, and not met short of gains / ions far beyond BASELINE.json's)
          block_sync();
          clear_table(sh);
          for (int tb = win_a + tid; tb < win_b; tb += SC_THREADS) {
            const unsigned int upto = (unsigned int)(sh.cum[tb] >> 32), before = tb > 0 ? (unsigned int)(sh.cum[tb - 1] >> 32) : 0u;
            if (upto == before) continue;
            const unsigned long long slot = atomicAdd(&a.out.ctrl[CTRL_LONE], 1ull);
            if (slot < (unsigned long long)a.out.lone_capacity) {
              LoneBucket lb;
              lb.event = e_local;
              lb.tb = (uint32_t)tb;
              a.out.lone_list[slot] = lb;
            } else {
              atomicMax(reinterpret_cast<unsigned int*>(&sh.ev_failed), 2u);  // list full: the bucket is lost, the host raises
            }
          }
          block_sync();
          if (tid < 64) {
            if (tid == 0) {
              if (sh.ev_failed == 2) {
                sh.failed++;
                sh.ev_failed = 1;
              }
              sh.retried++;
              sh.overflow = 0;
              sh.danger = 0;
              sh.n_keys = 0u;
              atomicAdd(&a.out.ctrl[CTRL_DANGER], 1ull);  // (rare: the host watches it and switches to the wide build)
            }
            select_window<WQ>(sh, win_b, sh.budget, lane);
          }
          block_sync();
          continue;
        }
        if (sh.overflow) {
          block_sync();
          clear_table(sh);
          if (tid < 64) {  // wave 0: same window start, smaller budget (a lone bucket is left to lone_bucket_kernel)
            const bool lone = win_b - win_a <= 1;  // one time bucket alone exceeds the table
            const int budget = lone ? local_const(TK) : (sh.win_samples / 2 > 0 ? sh.win_samples / 2 : 1);
            if (tid == 0) {
              sh.retried++;
              if (lone) {
                const unsigned long long slot = atomicAdd(&a.out.ctrl[CTRL_LONE], 1ull);
                if (slot < (unsigned long long)a.out.lone_capacity) {
                  LoneBucket lb;
                  lb.event = e_local;
                  lb.tb = (uint32_t)win_a;
                  a.out.lone_list[slot] = lb;
                } else {  // list full: the bucket is lost and the host raises (n_failed)
                  if (!sh.ev_failed) sh.failed++;
                  sh.ev_failed = 1;
                }
              }
              sh.overflow = 0;
              sh.n_keys = 0u;
            }
            select_window<WQ>(sh, lone ? win_a + 1 : win_a, budget, lane);
          }
          block_sync();
          continue;
        }

        // ---- flush: reserve one contiguous range of rows, compact the occupied slots, write rows ----
        // (the last barrier of the rows phase made every claim visible: n_keys is final)
        const unsigned int n_rows = sh.n_keys;
        unsigned long long g_base = 0ull, g_seg = 0ull;
        if (tid == 0) {
          sh.wg_cursor = 0u;
          if (n_rows) {
            // Output rows and segment slots come from blocks this workgroup reserved earlier: one
            // global atomic per ~row_block rows instead of two per window (a hot-address atomic costs
            // microseconds here, and every wave waits for it at the next barrier).  The unused tail of
            // a block stays a hole; consumers find rows through the segment list only.
            int zero = 0;
            asm volatile("" : "+v"(zero));  // opaque: keeps LLVM's atomic optimizer (readfirstlane) away
            if (sh.row_cur + n_rows > sh.row_end) {
              const unsigned long long need = max((unsigned long long)a.row_block, (unsigned long long)n_rows);
              sh.row_cur = atomicAdd(&a.out.ctrl[zero], need);
              sh.row_end = sh.row_cur + need;
            }
            g_base = sh.row_cur;
            sh.row_cur += n_rows;
            if (sh.seg_cur >= sh.seg_end) {
              sh.seg_cur = atomicAdd(&a.out.ctrl[zero + 1], (unsigned long long)SEG_BLOCK);
              sh.seg_end = sh.seg_cur + SEG_BLOCK;
            }
            g_seg = sh.seg_cur++;
            sh.wg_rows += n_rows;
          }
        }
        block_sync();
        PHASE_MARK(19);
        {  // every wave compacts its own contiguous slice of the table with a single LDS atomic.  A lane looks at four
           // consecutive slots per step (one 16-byte read), counts its occupied ones, gets its place in the wave's
           // part of the slot list from a ballot prefix of those counts (0..12: four bits) and writes them there:
           // 3 reads + 4 ballots per wave and window where the slot-per-lane version took 12 + 12 (compaction was 8 %
           // of the kernel's instructions with the 6 144-slot table).
          constexpr int PER_WAVE = HASH_CAP / N_WAVES;
          constexpr int STEPS = PER_WAVE / 256;
          static_assert(PER_WAVE % 256 == 0 && STEPS * 4 < 16, "four slots per lane and step; a lane's count fits four bits");
          const int wave = tid >> 6;
          uint32_t occ = 0u;  // bit 4 * step + k: slot k of this lane's four of that step is occupied
#pragma unroll
          for (int st = 0; st < STEPS; ++st) {
            const uint4 k4 = *reinterpret_cast<const uint4*>(&sh.keys[wave * PER_WAVE + st * 256 + lane * 4]);
            occ |= ((k4.x != EMPTY ? 1u : 0u) | (k4.y != EMPTY ? 2u : 0u) | (k4.z != EMPTY ? 4u : 0u) | (k4.w != EMPTY ? 8u : 0u)) << (4 * st);
          }
          const uint32_t mine = (uint32_t)__popc(occ);
          unsigned int before = 0u, cnt = 0u;
#pragma unroll
          for (int bit = 0; bit < 4; ++bit) {
            const unsigned long long mk = __ballot((mine >> bit) & 1u);
            before += (unsigned int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u)) << bit;
            cnt += (unsigned int)__popcll(mk) << bit;
          }
          unsigned int wbase = 0;
          if (lane == 0 && cnt) wbase = atomicAdd(&sh.wg_cursor, cnt);
          wbase = __shfl(wbase, 0);
          unsigned short* __restrict__ list = reinterpret_cast<unsigned short*>(&sh.queue[0][0]);
          // (branch-free: a free slot's "entry" goes to a dump position behind the list -- twelve short store sequences
          //  instead of twelve exec-mask branches)
          constexpr unsigned int DUMP = (unsigned int)(sizeof(sh.queue) / sizeof(unsigned short)) - 1u;
          static_assert(sizeof(sh.queue) / sizeof(unsigned short) > HASH_CAP, "room for the dump position behind the slot list");
          unsigned int pos = wbase + before;
          const unsigned int first = (unsigned int)(wave * PER_WAVE + lane * 4);
#pragma unroll
          for (int st = 0; st < STEPS; ++st) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const unsigned int bit = (occ >> (4 * st + k)) & 1u;
              list[bit ? pos : DUMP] = (unsigned short)(first + (unsigned int)(st * 256 + k));
              pos += bit;
            }
          }
        }
        PHASE_MARK(15);  // compaction done (wave 0)
        if (tid == 0) {
          unsigned long long base = 0ull;
          if (n_rows) {
            base = g_base;
#ifdef ATTPC_PHASE_TIMERS
            asm volatile("" ::"v"(g_base), "v"(g_seg));
            PHASE_SYNC;
            PHASE_MARK(16);  // global atomics returned
#endif
            if (base + n_rows > (unsigned long long)a.out.capacity || g_seg >= (unsigned long long)a.out.seg_capacity) {
              a.out.ctrl[6] = 1ull;  // out of capacity: host re-runs the chunk with larger buffers
              base = ~0ull;
            } else {
              Segment sg;
              sg.event = (int32_t)e_local;
              sg.count = (int32_t)n_rows;
              sg.offset = (int64_t)base;
              sg.ev_offset = (int64_t)sh.ev_rows;
              a.out.segments[g_seg] = sg;
            }
          }
          sh.base = base;
          sh.n_keys = 0u;
          sh.ev_rows += n_rows;
          PHASE_MARK(17);  // segment written
        }
        if (tid < 64) {  // wave 0 chooses the next window; the barrier after the row stores publishes it
          // adapt the estimate to this event: observed keys per estimated key of the last window
          // (the estimate counts every sample's pads anew; samples 0.1 mm apart share nearly all of theirs, so the
          // factor reaches 1/40 with the path-length step at 10x diffusion: hence the wide upper bound)
          const unsigned long long scaled =
              n_rows ? (unsigned long long)TK * (unsigned int)max(sh.win_samples, 1) / n_rows
                     : (unsigned long long)TK * BUDGET_MAX_MULT;
          const unsigned long long top = min((unsigned long long)TK * BUDGET_MAX_MULT,
                                             (unsigned long long)max(sh.budget, TK) * BUDGET_GROWTH);
          const int budget = (int)min(max(scaled, (unsigned long long)(TK / 8)), top);
          select_window<WQ>(sh, win_b, budget, lane);
          PHASE_MARK(18);
        }
        block_sync();
        PHASE_MARK(6);
        if (tid == 0 && sh.wg_cursor != n_rows) atomicAdd(&a.out.ctrl[CTRL_MISMATCH], 1ull);  // self-check, never seen
        const unsigned long long base = sh.base;
        // per-thread row pointers and the event id in VECTOR registers: as wave-uniform values they sat in
        // scalar registers spilled to vector lanes, read back with one VALU instruction each per row
        double* prow = a.out.points + (base + (unsigned long long)tid) * 3ull;
        int64_t* plab = a.out.labels + (base + (unsigned long long)tid);
        uint32_t ev_lo = (uint32_t)event, ev_hi = (uint32_t)(event >> 32);
        asm volatile("" : "+v"(ev_lo), "+v"(ev_hi));
        const uint32_t jitter_word =
            (uint32_t)__builtin_amdgcn_readfirstlane((int)jitter_key_word((uint32_t)a.seed, (uint32_t)(a.seed >> 32)));
        for (unsigned int r = tid; r < n_rows; r += SC_THREADS, prow += SC_THREADS * 3, plab += SC_THREADS) {
          const uint32_t slot = reinterpret_cast<const unsigned short*>(&sh.queue[0][0])[r];
          const uint32_t word = sh.keys[slot];
          const unsigned long long q = sh.chg[slot];
          sh.keys[slot] = EMPTY;
          sh.chg[slot] = (charge_t)0;
          const uint32_t key = word & KEY_MASK;
          const int pad = (int)(key & 0x3fffu), tb = (int)(key >> 14);
          my_charge += q;
          my_keys += (((unsigned long long)ev_hi << 32 | ev_lo) << 24) + (unsigned long long)key;
          if (base != ~0ull) {
            // The key word goes through an opaque asm per row: the Philox round keys are then scalar adds inside the
            // loop instead of scalar registers held (and, at the limit of 102, spilled to vector lanes and read back
            // with a VALU instruction each) across the whole kernel
            uint32_t jkey = jitter_word;
            asm volatile("" : "+s"(jkey));
            const double ua = jitter_uniform_k(jkey, ev_lo, ev_hi, key);  // simulator.py:108
            prow[0] = (double)pad;
            prow[1] = (double)tb + ua;
            prow[2] = (double)q;
            *plab = (int64_t)sh.label_of[word >> 24];  // from LDS: a global load here would
                                                                    // make every store wait (one vmcnt)
          }
        }
        // no barrier here: the next window was published before the row stores, and the first barrier of
        // its staging orders these LDS resets before any new insert while the sample loads of that
        // staging overlap the store acknowledgements
        PHASE_MARK(7);
      }
      if (tid == 0) a.out.ev_rows[e_local] = (uint32_t)sh.ev_rows;  // thread 0 is the only writer of sh.ev_rows
    }
    if (tid == 0) {  // events without any window never reached the request above
      sh.batch_first = have_next ? next_first : take_batch();
      have_next = false;
    }
  }
  PHASE_FLUSH;

  // ---- workgroup totals ----
  for (int off = 32; off > 0; off >>= 1) {
    my_charge += __shfl_down(my_charge, off);
    my_keys += __shfl_down(my_keys, off);
  }
  if (lane == 0) {
    if (my_charge) atomicAdd(&sh.charge_sum, my_charge);
    if (my_keys) atomicAdd(&sh.key_sum, my_keys);
  }
  block_sync();
  if (tid == 0) {
    if (sh.wg_samples) atomicAdd(&a.out.ctrl[7], sh.wg_samples);
    if (sh.wg_rows) atomicAdd(&a.out.ctrl[CTRL_ROWS], sh.wg_rows);
    if (sh.charge_sum) atomicAdd(&a.out.ctrl[2], sh.charge_sum);
    if (sh.key_sum) atomicAdd(&a.out.ctrl[3], sh.key_sum);
    if (sh.failed) atomicAdd(&a.out.ctrl[4], (unsigned long long)sh.failed);
    if (sh.retried) atomicAdd(&a.out.ctrl[5], (unsigned long long)sh.retried);
    // reserved but unused segment slots read as empty segments
    Segment none;
    none.event = 0; none.count = 0; none.offset = 0; none.ev_offset = 0;
    for (unsigned long long g = sh.seg_cur; g < sh.seg_end && g < (unsigned long long)a.out.seg_capacity; ++g)
      a.out.segments[g] = none;
  }
}

#undef tid
#undef lane

}  // namespace sc_<variant>

void ATTPC_SC_CAT(launch_scatter_kernel_, ATTPC_SC_VARIANT)(uint32_t n_workgroups, hipStream_t s, const ScatterArgs& a) {
  using namespace ATTPC_SC_CAT(sc_, ATTPC_SC_VARIANT);
  if (a.det.mc_diffusion)
    hipLaunchKernelGGL((scatter_kernel<true, false>), dim3(n_workgroups), dim3(SC_THREADS), 0, s, a);
  else if (a.merge_scratch != nullptr)
    hipLaunchKernelGGL((scatter_kernel<false, true>), dim3(n_workgroups), dim3(SC_THREADS), 0, s, a);
  else
    hipLaunchKernelGGL((scatter_kernel<false, false>), dim3(n_workgroups), dim3(SC_THREADS), 0, s, a);
}

}  // namespace attpc
