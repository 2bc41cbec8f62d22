// abi.hip -- host side of the C ABI (include/attpc_engine.h): context, configuration upload, the
// launch pipeline kinematics -> tracks -> scatter (-> Spyral rows) and output assembly.
//
// HBM layout (N rows/event, S simulated nuclei/event):
//   two TRACK SETS, each for one track batch of B events (up to 8 scatter chunks, T = B*S tracks):
//     p4 f64[B][N][4], vertex f64[B][3], status i32[B], attempts u32[B]          kinematics
//     arena f64[blocks][128][4]  (x, y, time bucket, electrons)                   track samples
//     block_table i32[T][79], counts i32[T], n_steps i32[T]                       track index
//   one CLOUD of one scatter chunk of C events:
//     points f64[cap][3], labels i64[cap], segments {event,count,offset,ev_offset}[..], ev_rows u32[C]
//   two ASSEMBLY SETS (only when clouds are delivered to the host): the chunk's cloud in event order
//     (CSR) or its Spyral rows, filled on the device while the previous chunk's set crosses PCIe.
// Buffers grow on demand and are reused (device-resident mode overwrites the cloud chunk by chunk).
//
// Streams: S (scatter, lone buckets, assembly), T (kinematics + tracks of the NEXT batch, low
// priority: it fills the compute units the persistent scatter workgroups leave at the end of each
// launch), C (device-to-host copies).  The host does not wait per launch: control words of every
// launch are copied to pinned host memory behind it and read once per batch (device-resident) or per
// chunk (delivered clouds, where the copy needs the row total anyway).  A launch whose buffers
// turned out too small is repeated with larger ones (results are deterministic, so a re-run is exact).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tracks_args.hpp"
#include "unpack_host.hpp"

namespace {

using namespace attpc;

constexpr int MAX_SLOTS = 8;           // scatter chunks per track batch
#ifdef ATTPC_PHASE_TIMERS
constexpr int CTRL_WORDS = 64;         // (diagnostic build: + per-wave timers in 40..63)
#else
constexpr int CTRL_WORDS = 40;         // u64 control words per scatter launch (scatter.hip: 0..32 used)
#endif
constexpr uint32_t LONE_CAPACITY = 65536;
constexpr int64_t CLOUD_BUDGET_BYTES = 24ll << 30;  // points + labels of one chunk
constexpr int64_t DELIVER_CHUNK_ROWS = 96ll << 20;  // cloud rows of a chunk whose cloud is delivered (3 GB: ~60 ms of PCIe)
constexpr uint64_t ARENA_BUDGET_BYTES = 24ull << 30;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct TrackSet {  // kinematics + tracks of one track batch
  DevBuf p4, vertex, status, attempts, arena, block_table, counts, n_steps, ctrl;
  size_t arena_blocks = 0;
  uint32_t* h_ctrl = nullptr;  // pinned [16]
  hipEvent_t done = nullptr, k0 = nullptr, k1 = nullptr, t0 = nullptr, t1 = nullptr;
  bool timed_kin = false;
};

struct AsmSet {  // one chunk's cloud in event order, or its Spyral rows
  DevBuf ev_start, points, labels, kept, kept_start, sp_rows, sp_labels;
  size_t row_cap = 0;  // rows the row-sized buffers of the set are kept at (grown with headroom: a launch's row
                       // capacity follows the observed rows per event and moves by fractions of a percent)
  int64_t* h_start = nullptr;  // pinned [h_start_len]: CSR offsets of the chunk (n + 1 entries)
  size_t h_start_len = 0;
  uint32_t* h_ev_rows = nullptr;  // pinned [h_start_len]: cloud rows of every event before any threshold
  int64_t* h_total = nullptr;  // pinned [2]: [0] != 0: a row of the chunk does not fit the 16-byte transfer record
  hipEvent_t ready = nullptr, copied = nullptr;
  // compact transfer: the chunk's rows as 16-byte records, device and (pinned, library-owned) host side, and the
  // expansion into the caller's arrays that is still to be done once the copy has arrived
  DevBuf packed;
  void* h_packed = nullptr;
  size_t h_packed_bytes = 0;
  uint64_t unpack_ticket = 0;  // number of the expansion job that last used h_packed (0: none)
};

struct UnpackJob {  // one chunk's records in pinned staging -> the caller's arrays, once `copied` has fired
  hipEvent_t copied = nullptr;
  const void* src = nullptr;
  int64_t rows = 0;
  double* points = nullptr;   // [rows, 3] cloud rows, or [rows, 8] Spyral rows when `spyral`
  int64_t* labels = nullptr;
  bool spyral = false;
  // 8-byte cloud records: what the host needs to regenerate the jitter (the job's own copy of the chunk's CSR
  // offsets: the set's pinned copy is overwritten by the chunk after next while this job may still run)
  bool tight = false;
  uint64_t seed = 0, first_event = 0;
  std::vector<int64_t> offsets;
};

}  // namespace

struct attpc_ctx {
  int device = 0;
  int n_cus = 256;                 // compute units
  hipStream_t stream = nullptr;    // S
  hipStream_t stream_t = nullptr;  // T (== stream when the option "serial_tracks" is on)
  hipStream_t stream_t_own = nullptr;  // the low-priority stream the context created
  hipStream_t stream_c = nullptr;  // C
  std::string error;
  int32_t chunk_events = 65536;
  int opt_variant = 0;             // 0 auto, 1 small, 2 big, 3 wide (u64 sums)
  bool opt_tiny = false;
  int opt_compact = 2;             // delivered clouds cross PCIe as 8-byte (2) / 16-byte (1) records and are expanded
                                   // by host threads, or in the reference's dtypes (0)
  int opt_unpack_threads = 0;      // 0: min(32, half of the hardware threads)
  int opt_deliver_chunk = 8192;    // events per chunk when clouds are delivered (the pipeline's fill and drain time)
  int opt_merge = -1;              // scatter kernel's merge variant: -1 automatic (path-length dE/dx step), 0 never, 1 always
  int opt_first_batch_chunks = 0;  // > 0: the first track batch of a call spans at most this many scatter chunks
  int opt_track_species_major = 1; // tracks handed out nucleus by nucleus, lightest species first (0: event by event)
  int opt_track_blocks_per_cu = 8; // track_kernel workgroups (256 threads) launched per CU at most
  int opt_serial_tracks = -1;      // -1 automatic (see pick_track_stream), 0 beside the scatter launches, 1 behind them

  bool kin_ready = false;
  attpc_kin_desc kin{};            // device pointers inside
  std::vector<void*> kin_allocs;

  bool det_ready = false;
  DetDev det{};
  std::vector<void*> det_allocs;

  TrackSet tset[2];
  // cloud of one chunk
  DevBuf points, labels, segments, ev_rows, lone_list, lone_chg, lone_mask, out_ctrl, merge_scratch;
  unsigned long long* h_out_ctrl = nullptr;  // pinned [MAX_SLOTS][CTRL_WORDS]
  hipEvent_t s0[MAX_SLOTS] = {}, s1[MAX_SLOTS] = {};
  int64_t cloud_capacity = 0, seg_capacity = 0;
  AsmSet aset[2];
  DevBuf sort_idx, sort_key;
  // running estimates that size the next launches (reset by configure)
  double rows_per_event = 0.0;     // observed cloud rows per event, 0 = unknown
  double segs_per_event = 0.0;
  double blocks_per_track = 0.0;   // observed arena blocks per track
  bool prefer_big = false;         // sticky: the small scatter variant met too many lone buckets
  bool prefer_wide = false;        // sticky: u32 sums per table slot are not enough for this detector (scatter_wide.hip)
  bool slot_wide[8] = {};          // per control-word slot: the launch queued last used the wide build
  bool lone_ready = false;         // lone_bucket_kernel's tables are allocated AND their zeroing has been queued
  uint64_t n_growths = 0;          // device buffers (re)allocated so far (a steady workload stops growing)
  uint64_t device_bytes = 0;       // bytes of the grow-only device buffers (ensure()) held right now
  int64_t launch_row_cap = 0;      // row capacity given to the scatter launch queued last (<= cloud_capacity)
  uint32_t max_batch_events = 0;   // largest track batch so far: both track sets are sized for it (the set that
                                   // first meets the shorter last batch of a call would otherwise grow in the next call)

  // attpc_sim_hint_next: the call after the one that comes next.  `hint_*` is what the caller announced; once the
  // run it was given to has queued that call's first track batch (behind its own last scatter launches) `pre_valid`
  // says which track set holds it.
  bool hint_valid = false, pre_valid = false;
  uint64_t hint_seed = 0, hint_first = 0, hint_n = 0;
  attpc_event_layout hint_lay{};
  int pre_set = 0;
  uint64_t pre_seed = 0, pre_first = 0;
  uint32_t pre_nb = 0;
  attpc_event_layout pre_lay{};

  bool spyral_ready = false;
  SpyralDev spyral{};
  std::vector<void*> spyral_allocs;
  std::vector<double> h_pad_centers, h_pad_sizes;  // host copies: the expansion of compact Spyral records needs them
  DevBuf scratch[8];
  std::vector<void*> host_allocs;  // attpc_host_alloc

  // expansion of compact transfer records: one helper thread takes the jobs in order (it waits for the copy,
  // then fans the rows out over worker threads), so that the thread driving the GPU never stands in a memcpy
  // while the copy engine waits for its next order
  std::thread unpacker;
  std::mutex unpack_mutex;
  std::condition_variable unpack_cv;
  std::deque<UnpackJob> unpack_jobs;
  uint64_t unpack_submitted = 0, unpack_done = 0;
  bool unpack_stop = false, unpack_failed = false;
};

namespace {

int32_t fail(attpc_ctx* ctx, int32_t code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->error = buf;
  return code;
}

#define HIP_TRY(ctx, call)                                                                     \
  do {                                                                                         \
    hipError_t err__ = (call);                                                                 \
    if (err__ != hipSuccess)                                                                   \
      return fail(ctx, ATTPC_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(err__), \
                  __FILE__, __LINE__);                                                         \
  } while (0)

// grow-only device buffer; the caller makes sure nothing in flight uses it when it has to grow
int32_t ensure(attpc_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.bytes) return ATTPC_OK;
  ctx->n_growths++;
#ifdef ATTPC_DEBUG_GROWTH  // (diagnostic builds only, tools/build_variant.sh: which buffer is re-allocated, and when)
  fprintf(stderr, "[attpc grow] buffer at +%zu of the context: %zu -> %zu bytes\n",
          (size_t)(reinterpret_cast<const char*>(&b) - reinterpret_cast<const char*>(ctx)), b.bytes, bytes);
#endif
  if (b.p) HIP_TRY(ctx, hipFree(b.p));
  ctx->device_bytes -= b.bytes;
  b.p = nullptr;
  b.bytes = 0;
  HIP_TRY(ctx, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
  ctx->device_bytes += bytes;
  return ATTPC_OK;
}

template <typename T>
int32_t upload(attpc_ctx* ctx, std::vector<void*>& owner, const T* host, size_t n, const T** dev) {
  void* p = nullptr;
  HIP_TRY(ctx, hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
  owner.push_back(p);
  if (n) HIP_TRY(ctx, hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
  *dev = static_cast<const T*>(p);
  return ATTPC_OK;
}

void free_all(std::vector<void*>& v) {
  for (void* p : v) (void)hipFree(p);
  v.clear();
}

int32_t sync_all(attpc_ctx* ctx) {
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_t));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_c));
  return ATTPC_OK;
}

// A first track batch queued ahead for a call that does not come (another entry point, other events, a new
// configuration): let it finish and forget it.
int32_t drop_prefetch(attpc_ctx* ctx) {
  ctx->hint_valid = false;
  if (!ctx->pre_valid) return ATTPC_OK;
  ctx->pre_valid = false;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_t));
  return ATTPC_OK;
}

bool same_layout(const attpc_event_layout& a, const attpc_event_layout& b) {
  if (a.n_rows != b.n_rows || a.n_sim != b.n_sim) return false;
  for (int i = 0; i < a.n_sim; ++i)
    if (a.indices[i] != b.indices[i]) return false;
  for (int i = 0; i < a.n_rows; ++i)
    if (a.species_of_row[i] != b.species_of_row[i]) return false;
  return true;
}

// Where the kinematics + tracks of the NEXT batch run while the current one is scattered: on the scatter stream itself,
// behind its launches (serial), or beside them on the low-priority stream.  Both kernels are bound by instruction issue;
// beside the default scatter kernel the track kernel (170 VGPRs, 2 waves per SIMD) displaces co-resident scatter
// workgroups and costs more than it hides (headline: 171.6 ms per step beside, 166.5 behind) -- beside the merge variant
// of the path-length step, where track integration is a quarter of the device time and the scatter kernel leaves more
// idle issue slots, it pays (configs[4]: 1.80e5 events/s beside, 1.61e5 behind).  Nothing may be in flight when it changes.
void pick_track_stream(attpc_ctx* ctx) {
  const bool serial = ctx->opt_serial_tracks >= 0 ? ctx->opt_serial_tracks != 0 : !(ctx->det_ready && ctx->det.path_step > 0.0);
  ctx->stream_t = serial ? ctx->stream : ctx->stream_t_own;
}

int32_t validate_layout(attpc_ctx* ctx, const attpc_event_layout* lay, bool with_species) {
  if (!lay || lay->n_rows < 1 || lay->n_rows > ATTPC_MAX_ROWS || lay->n_sim < 0 || lay->n_sim > ATTPC_MAX_SIM)
    return fail(ctx, ATTPC_E_INVALID, "bad event layout");
  for (int i = 0; i < lay->n_sim; ++i) {
    const int row = lay->indices[i];
    if (row < 0 || row >= lay->n_rows) return fail(ctx, ATTPC_E_INVALID, "indices[%d]=%d out of range", i, row);
    const int sp = lay->species_of_row[row];
    if (with_species && sp >= ctx->det.n_species) return fail(ctx, ATTPC_E_INVALID, "species_of_row[%d]=%d out of range", row, sp);
  }
  return ATTPC_OK;
}

struct ChunkResult {
  unsigned long long rows = 0, reserved = 0, segs = 0, charge = 0, keys = 0, failed = 0, retried = 0, samples = 0,
                     mismatch = 0, lone = 0, danger = 0;
  bool overflow = false;
  float ms_scatter = 0;
};

// ------------------------------------------------------------------ small device helpers ----
// out[i] = sum of in[0..i), i = 0..n (one workgroup; n is a chunk's event count), *total = out[n].
// `ctrl` (may be null): control words of the scatter launch that produced the counts -- if that launch ran
// out of cloud or segment capacity (ctrl[6]) its rows were not all written and its segment list has
// unwritten slots, so every offset becomes 0: the kernels behind this one then see empty events and touch
// nothing, and the host repeats the launch with larger buffers.
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(const uint32_t* __restrict__ in, uint32_t n,
                                                              int64_t* __restrict__ out, int64_t* __restrict__ total,
                                                              const unsigned long long* __restrict__ ctrl) {
  __shared__ long long wave_sum[16];
  __shared__ long long carry;
  const int t = (int)threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool dead = ctrl != nullptr && ctrl[6] != 0ull;
  if (t == 0) carry = 0;
  block_sync();
  for (uint32_t base = 0; base < n; base += 1024u) {
    const uint32_t i = base + (uint32_t)t;
    const long long v = (i < n && !dead) ? (long long)in[i] : 0ll;
    long long incl = v;
    for (int off = 1; off < 64; off <<= 1) {
      const long long up = __shfl_up(incl, off);
      incl += lane >= off ? up : 0ll;
    }
    if (lane == 63) wave_sum[wave] = incl;
    block_sync();
    long long before = carry;
    for (int w = 0; w < wave; ++w) before += wave_sum[w];
    if (i < n) out[i] = before + incl - v;
    block_sync();
    if (t == 1023) carry = before + incl;
    block_sync();
  }
  if (t == 0) {
    out[n] = carry;
    if (total) *total = carry;
  }
}

// Device-side CSR assembly: segment s (one flushed window of one event) is copied to rows
// ev_start[event] + ev_offset of the event-ordered arrays.  n_segs is read from the launch's control
// words (the host never sees the segment list).
__global__ __launch_bounds__(256) void gather_segments_kernel(const Segment* __restrict__ segs,
                                                              const unsigned long long* __restrict__ ctrl,
                                                              int64_t seg_capacity,
                                                              const int64_t* __restrict__ ev_start,
                                                              const double* __restrict__ points,
                                                              const int64_t* __restrict__ labels,
                                                              double* __restrict__ out_points,
                                                              int64_t* __restrict__ out_labels) {
  if (ctrl[6] != 0ull) return;  // the launch ran out of capacity: unwritten segment slots, see exclusive_scan_kernel
  const unsigned long long n_all = ctrl[1];
  const uint32_t n_segs = (uint32_t)(n_all < (unsigned long long)seg_capacity ? n_all : (unsigned long long)seg_capacity);
  for (uint32_t s = blockIdx.x; s < n_segs; s += gridDim.x) {
    const Segment sg = segs[s];
    if (sg.count <= 0) continue;
    const int64_t dst = ev_start[sg.event] + sg.ev_offset;
    const double* src_p = points + sg.offset * 3;
    double* dst_p = out_points + dst * 3;
    for (int i = threadIdx.x; i < sg.count * 3; i += 256) dst_p[i] = src_p[i];
    const int64_t* src_l = labels + sg.offset;
    int64_t* dst_l = out_labels + dst;
    for (int i = threadIdx.x; i < sg.count; i += 256) dst_l[i] = src_l[i];
  }
}

// ---- compact transfer of delivered clouds ----
// The delivered path is PCIe bound (234 KB per event in the reference's dtypes), so a chunk crosses the link as 16-byte
// records (PackedRow, unpack_host.hpp) into library-owned pinned staging and host threads expand it into the caller's
// arrays -- which then need not be page-locked either.  A chunk with a row that does not fit (charge >= 2^45,
// label >= 32) goes the plain way.
// tight != 0: the 8-byte record (PackedRow8, unpack_host.hpp) -- the jitter is not sent at all: it is a pure function of
// (seed, event, time bucket, pad), and the host regenerates it with the same Philox2x32-7.  flag[0] != 0: a row does not fit
// the 16-byte record; flag[1] != 0: a row does not fit the 8-byte one (charge >= 2^36, or a jittered time bucket that
// is a whole number -- tb + U rounded up to tb + 1, about one row in 1e13 -- from which the bucket cannot be read back).
__global__ __launch_bounds__(256) void pack_rows_kernel(const int64_t* __restrict__ ev_start, uint32_t n_events,
                                                        const double* __restrict__ points, const int64_t* __restrict__ labels,
                                                        PackedRow* __restrict__ packed, int64_t* __restrict__ flag, int tight) {
  const int64_t total = ev_start[n_events];
  bool bad = false, bad8 = false;
  unsigned long long* __restrict__ packed8 = reinterpret_cast<unsigned long long*>(packed);
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < total; r += (int64_t)gridDim.x * 256) {
    const double padf = points[3 * r], tbj = points[3 * r + 1], q = points[3 * r + 2];
    const long long label = labels[r];
    const unsigned long long charge = (unsigned long long)q, pad = (unsigned long long)padf;
    bad = bad || !(q >= 0.0) || charge >= (1ull << PACK_CHARGE_BITS) || pad >= (1ull << PACK_PAD_BITS) || label < 0 || label >= 32;
    if (tight) {
      const double tbf = floor(tbj);
      bad8 = bad8 || charge >= (1ull << PACK8_CHARGE_BITS) || !(tbf >= 0.0) || tbf >= (double)(1 << PACK8_TB_BITS) || tbf == tbj;
      packed8[r] = (charge & ((1ull << PACK8_CHARGE_BITS) - 1)) | ((unsigned long long)tbf << PACK8_CHARGE_BITS) |
                   (pad << (PACK8_CHARGE_BITS + PACK8_TB_BITS)) |
                   ((unsigned long long)label << (PACK8_CHARGE_BITS + PACK8_TB_BITS + PACK_PAD_BITS));
    } else {
      PackedRow row;
      row.tb = tbj;
      row.bits = (charge & ((1ull << PACK_CHARGE_BITS) - 1)) | (pad << PACK_CHARGE_BITS) |
                 ((unsigned long long)label << (PACK_CHARGE_BITS + PACK_PAD_BITS));
      packed[r] = row;
    }
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(flag), 1ull);
  if (__any(bad8 || bad) && (threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(flag) + 1, 1ull);
}

__global__ __launch_bounds__(256) void count_status_kernel(const int32_t* __restrict__ status, uint32_t n,
                                                           uint32_t* __restrict__ counter) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool bad = i < n && status[i] != 0;
  const unsigned long long m = __ballot(bad);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(counter, (uint32_t)__popcll(m));
}

// ------------------------------------------------------------------ tracks ----
int32_t ensure_kin_buffers(attpc_ctx* ctx, TrackSet& ts, uint32_t n, int n_rows) {
  int32_t rc;
  ctx->max_batch_events = std::max(ctx->max_batch_events, n);
  n = ctx->max_batch_events;
  if ((rc = ensure(ctx, ts.p4, (size_t)n * n_rows * 4 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ts.vertex, (size_t)n * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ts.status, (size_t)n * sizeof(int32_t)))) return rc;
  if ((rc = ensure(ctx, ts.attempts, (size_t)n * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(ctx, ts.ctrl, 16 * sizeof(uint32_t)))) return rc;
  return ATTPC_OK;
}

struct TrackLaunch {  // what launch_tracks queued, for finish_tracks
  attpc_event_layout lay{};
  uint64_t seed = 0, first_event = 0;
  uint32_t n = 0;
  bool use_status = false;
};

// Queue the track integration of a BATCH of `n` events whose kinematics are (being) written to the
// set's p4 / vertex (/ status) on stream T.  A batch spans several scatter chunks: the track kernel
// hands tracks to lanes dynamically, and with fewer tracks than a few times the 200 k lanes of the
// chip the launch is one generation of tracks whose length is set by its longest member.
int32_t launch_tracks(attpc_ctx* ctx, TrackSet& ts, const TrackLaunch& tl) {
  const uint32_t n_tracks = tl.n * (uint32_t)tl.lay.n_sim;
  int32_t rc;
  if ((rc = ensure(ctx, ts.ctrl, 16 * sizeof(uint32_t)))) return rc;
  HIP_TRY(ctx, hipMemsetAsync(ts.ctrl.p, 0, 16 * sizeof(uint32_t), ctx->stream_t));
  if (n_tracks) {
    ctx->max_batch_events = std::max(ctx->max_batch_events, tl.n);
    const size_t alloc_tracks = (size_t)ctx->max_batch_events * (size_t)tl.lay.n_sim;
    if ((rc = ensure(ctx, ts.block_table, alloc_tracks * MAX_BLOCKS_PER_TRACK * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ts.counts, alloc_tracks * sizeof(int32_t)))) return rc;
    if ((rc = ensure(ctx, ts.n_steps, alloc_tracks * sizeof(int32_t)))) return rc;
    const size_t lds = (size_t)ctx->det.n_species * ATTPC_DEDX_NODES * sizeof(double);
    const uint32_t waves_needed = (n_tracks + 63) / 64;
    const uint32_t blocks = std::min<uint32_t>((waves_needed + 3) / 4, (uint32_t)ctx->n_cus * (uint32_t)ctx->opt_track_blocks_per_cu);
    // every wave reserves arena blocks 64 at a time: that slack comes on top of what the samples need
    // keep the arena while it covers the observed need with 3 % to spare, grow it by 25 % when it does not:
    // the need per track moves by fractions of a percent from batch to batch, and re-allocating tens of GB
    // for each such step costs more than the kernels (a too small arena is caught and the batch repeated)
    const size_t pools = (size_t)blocks * 4 * 64 + 1024;
    const double bpt = ctx->blocks_per_track;
    const size_t need_blocks = bpt > 0.0 ? (size_t)((double)n_tracks * (bpt * 1.03 + 0.05)) + pools : (size_t)n_tracks * 3 + pools;
    size_t want_blocks = ts.arena_blocks;
    if (need_blocks > ts.arena_blocks)
      want_blocks = bpt > 0.0 ? (size_t)((double)n_tracks * (bpt * 1.25 + 0.25)) + pools : need_blocks;
    if (ctx->opt_tiny && ts.arena_blocks == 0) want_blocks = 4;  // test hook: grow-and-rerun path
    if ((rc = ensure(ctx, ts.arena, want_blocks * ARENA_BLK * 4 * sizeof(double)))) return rc;
    ts.arena_blocks = want_blocks;
    TrackArgs ta;
    ta.det = ctx->det;
    ta.layout = tl.lay;
    ta.buf.arena = static_cast<double*>(ts.arena.p);
    ta.buf.block_table = static_cast<int32_t*>(ts.block_table.p);
    ta.buf.counts = static_cast<int32_t*>(ts.counts.p);
    ta.buf.n_steps = static_cast<int32_t*>(ts.n_steps.p);
    ta.buf.ctrl = static_cast<uint32_t*>(ts.ctrl.p);
    ta.buf.arena_blocks = (uint32_t)std::min<size_t>(want_blocks, 0xFFFFFFFFu);
    ta.p4 = static_cast<const double*>(ts.p4.p);
    ta.vertex = static_cast<const double*>(ts.vertex.p);
    ta.kin_status = tl.use_status ? static_cast<const int32_t*>(ts.status.p) : nullptr;
    ta.seed = tl.seed;
    ta.first_event = tl.first_event;
    ta.n_events = tl.n;
    ta.n_tracks = n_tracks;
    {  // lightest species first (they travel farthest): the kernel's last tracks are then the short ones
      int order[ATTPC_MAX_SIM];
      const int n_sim = tl.lay.n_sim;
      for (int i = 0; i < n_sim; ++i) order[i] = i;
      auto weight = [&](int isim) -> double {
        const int sp = tl.lay.species_of_row[tl.lay.indices[isim]];
        return sp < 0 ? 1.0e30 : (double)ctx->det.Z[sp] * 1.0e6 + ctx->det.mass[sp];  // by charge, then by mass
      };
      std::stable_sort(order, order + n_sim, [&](int x, int y) { return weight(x) < weight(y); });
      for (int i = 0; i < ATTPC_MAX_SIM; ++i) ta.sim_order[i] = (uint8_t)(i < n_sim ? order[i] : 0);
      if (!ctx->opt_track_species_major || n_sim <= 1) ta.sim_order[0] = 0xffu;
    }
    HIP_TRY(ctx, hipEventRecord(ts.t0, ctx->stream_t));
    launch_track_kernel(blocks, lds, ctx->stream_t, ta);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ts.t1, ctx->stream_t));
  }
  if (tl.use_status && tl.n) {  // events that hit event_sample_limit -> ctrl[3]
    hipLaunchKernelGGL(count_status_kernel, dim3((tl.n + 255) / 256), dim3(256), 0, ctx->stream_t,
                       static_cast<const int32_t*>(ts.status.p), tl.n, static_cast<uint32_t*>(ts.ctrl.p) + 3);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipMemcpyAsync(ts.h_ctrl, ts.ctrl.p, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream_t));
  HIP_TRY(ctx, hipEventRecord(ts.done, ctx->stream_t));
  return ATTPC_OK;
}

// Wait for the batch queued by launch_tracks; a batch whose arena was too small is run again with a
// larger one.  ms / n_limit accumulate.
int32_t finish_tracks(attpc_ctx* ctx, TrackSet& ts, const TrackLaunch& tl, TrackBuffers* out, double* ms_tracks,
                      uint64_t* n_limit, uint64_t* n_capped = nullptr) {
  const uint32_t n_tracks = tl.n * (uint32_t)tl.lay.n_sim;
  for (int attempt = 0; attempt < 8; ++attempt) {
    HIP_TRY(ctx, hipEventSynchronize(ts.done));
    if (n_tracks) {
      float ms_t = 0;
      HIP_TRY(ctx, hipEventElapsedTime(&ms_t, ts.t0, ts.t1));
      *ms_tracks += ms_t;  // timings of discarded attempts stay counted: they were spent
    }
    if (ts.h_ctrl[2] == 0) {  // no sample was refused
      if (n_tracks) ctx->blocks_per_track = (double)ts.h_ctrl[1] / (double)n_tracks;
      if (n_limit) *n_limit += ts.h_ctrl[3];
      if (n_capped) *n_capped += ts.h_ctrl[4];
      *out = TrackBuffers{};
      out->arena = static_cast<double*>(ts.arena.p);
      out->block_table = static_cast<int32_t*>(ts.block_table.p);
      out->counts = static_cast<int32_t*>(ts.counts.p);
      out->n_steps = static_cast<int32_t*>(ts.n_steps.p);
      out->ctrl = static_cast<uint32_t*>(ts.ctrl.p);
      out->arena_blocks = (uint32_t)std::min<size_t>(ts.arena_blocks, 0xFFFFFFFFu);
      return ATTPC_OK;
    }
    // arena exhausted (the block counter kept counting): grow and run the batch again
    const size_t want = std::max<size_t>((size_t)ts.h_ctrl[1] + (size_t)ts.h_ctrl[1] / 8 + 1024, ts.arena_blocks * 2);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_t));
    ts.arena_blocks = std::max(ts.arena_blocks, want);
    int32_t rc = launch_tracks(ctx, ts, tl);
    if (rc) return rc;
  }
  return fail(ctx, ATTPC_E_HIP, "track arena did not fit after repeated growth");
}

// events per track batch: up to MAX_SLOTS scatter chunks, bounded so that the sample arena stays below
// ARENA_BUDGET_BYTES (with the observed blocks per track once a batch has run)
uint64_t track_batch_events(const attpc_ctx* ctx, const attpc_event_layout& lay, uint64_t chunk) {
  const double per_track = ctx->blocks_per_track > 0.0 ? ctx->blocks_per_track * 1.15 + 0.25 : 3.0;
  const double bytes_per_event = (double)std::max(1, lay.n_sim) * per_track * ARENA_BLK * 4 * sizeof(double);
  const uint64_t by_memory = (uint64_t)((double)ARENA_BUDGET_BYTES / bytes_per_event);
  const uint64_t chunks = std::max<uint64_t>(1, std::min<uint64_t>(MAX_SLOTS, by_memory / std::max<uint64_t>(1, chunk)));
  return std::max<uint64_t>(1, std::min<uint64_t>(chunk * chunks, std::max<uint64_t>(by_memory, 1)));
}

// events of the next scatter chunk: a small pilot while the cloud size per event is unknown, then the
// configured chunk bounded by the cloud budget
uint32_t next_chunk_events(const attpc_ctx* ctx, uint64_t remaining) {
  uint64_t n = (uint64_t)std::max(1, ctx->chunk_events);
  if (ctx->rows_per_event <= 0.0) n = std::min<uint64_t>(n, 4096);
  else n = std::min<uint64_t>(n, std::max<uint64_t>(256, (uint64_t)((double)CLOUD_BUDGET_BYTES / (ctx->rows_per_event * 1.15 * 32.0 + 1.0))));
  return (uint32_t)std::min<uint64_t>(n, remaining);
}

// ------------------------------------------------------------------ scatter ----
struct ScatterPlan {  // launch geometry of one chunk
  bool use_small = false;
  bool use_wide = false;  // u64 sums per table slot (scatter_wide.hip), one workgroup per CU like the big build
  uint32_t wgs = 0, batch = 1, row_block = 1;
  int64_t need_rows = 0, need_segs = 0;  // the launch should fit in this much
  int64_t grow_rows = 0, grow_segs = 0;  // what to allocate when the buffers are smaller than that
};

ScatterPlan plan_scatter(const attpc_ctx* ctx, uint32_t n) {
  ScatterPlan p;
  // Kernel variant: "small" (two 512-thread workgroups with 6144-slot tables per CU) is ~6 % faster
  // for detectors with the usual diffusion; "big" (one 1024-thread workgroup, 12288 slots) holds twice
  // as many keys per time bucket.  Small is used when a sample is expected to touch at most 40 pads at
  // the far end of the drift (default detector: 28; the same estimate as key_estimate() in scatter.hip)
  // and no extension is on.  A time bucket that fits neither table goes through lone_bucket_kernel; a
  // context whose small launches meet many of those switches to big for good (prefer_big).
  const double spread = (6.0 / 4.9e-3) * (6.0 / 4.9e-3) * 2.0 * ctx->det.diffusion * ctx->det.dv / ctx->det.efield;
  const double far_keys = (1.0 + std::sqrt(spread * (ATTPC_NUM_TB - 1))) * (1.0 + std::sqrt(spread * (ATTPC_NUM_TB - 1)));
  p.use_small = far_keys <= 40.0 && !ctx->det.mc_diffusion && !(ctx->det.longitudinal_diffusion > 0.0) && !ctx->prefer_big;
  if (ctx->opt_variant == 1) p.use_small = true;
  if (ctx->opt_variant == 2) p.use_small = false;
  // u32 sums per slot hold what the AT-TPC makes (largest key of the headline workload: 1.1e9 electrons); a launch in
  // which many windows had to be given to lone_bucket_kernel (read_scatter) switches the context to the u64 build
  p.use_wide = ctx->opt_variant == 3 || (ctx->prefer_wide && ctx->opt_variant == 0);
  if (p.use_wide) p.use_small = false;
#ifndef ATTPC_SC_SMALL_WGS
#define ATTPC_SC_SMALL_WGS 2  // workgroups per CU of the small variant (scatter_small.hip)
#endif
  // persistent workgroups that take `batch` events per visit to the event counter and reserve output
  // rows `row_block` at a time (small launches: exact reservations, so that short runs waste no rows)
  p.wgs = std::min<uint32_t>((uint32_t)ctx->n_cus * (p.use_small ? (uint32_t)ATTPC_SC_SMALL_WGS : 1u), n);
  p.batch = n / p.wgs >= 64u ? 2u : 1u;  // the request for the next batch is hidden (scatter.hip)
  const double per_event = ctx->rows_per_event > 0.0 ? ctx->rows_per_event * 1.10 : 16384.0;
  const int64_t est_rows = (int64_t)((double)n * per_event) + 4096;
  p.row_block = est_rows / ((int64_t)p.wgs * 16) >= 16384 ? (uint32_t)std::min<int64_t>(est_rows / ((int64_t)p.wgs * 16), 1 << 18) : 1u;
  const int64_t hole_rows = p.row_block > 1u ? (int64_t)p.wgs * p.row_block + est_rows / 16 : 0;
  // keep the buffers while they cover the estimate with 3 % to spare, grow them by 25 % when they do not:
  // rows per event move by fractions of a percent from chunk to chunk, and re-allocating tens of GB for each
  // such step costs more than the kernels (a too small buffer is caught and the launch repeated)
  const double known = ctx->rows_per_event > 0.0 ? ctx->rows_per_event : 16384.0;
  p.need_rows = (int64_t)((double)n * known * 1.03) + hole_rows + 65536;
  p.grow_rows = (int64_t)((double)n * known * 1.25) + hole_rows + 65536;
  const double segs = ctx->segs_per_event > 0.0 ? ctx->segs_per_event : 5.0;
  p.need_segs = (int64_t)((double)n * (segs * 1.05 + 0.5)) + 4096 + (int64_t)p.wgs * 16;
  // (24 bytes a segment: when the list has to grow it is sized for a full chunk at once -- a call's chunks are not all
  //  of one length, and the next call's would re-allocate it inside a caller's timed region)
  p.grow_segs = (int64_t)((double)std::max<uint32_t>(n, (uint32_t)std::max(1, ctx->chunk_events)) * (segs * 1.5 + 1.0)) + 4096 + (int64_t)p.wgs * 16;
  if (ctx->opt_tiny && ctx->cloud_capacity == 0) {
    p.need_rows = p.grow_rows = 64;  // test hook: start with buffers that are certainly too small
    p.need_segs = p.grow_segs = 2;
  }
  return p;
}

// Queue the scatter of the `n` events starting at event `e0` of a track batch (global id
// `first_event` = batch first + e0) on stream S, control words in slot `slot`.  `grow` may enlarge the
// cloud (the caller guarantees S is idle then).
int32_t enqueue_scatter(attpc_ctx* ctx, int slot, const attpc_event_layout& lay, const TrackBuffers& trk, uint64_t seed,
                        uint64_t first_event, uint32_t e0, uint32_t n, int64_t min_rows, int64_t min_segs) {
  int32_t rc;
  const ScatterPlan p = plan_scatter(ctx, n);
  int64_t want_rows = ctx->cloud_capacity, want_segs = ctx->seg_capacity;
  if (std::max(p.need_rows, min_rows) > ctx->cloud_capacity) want_rows = std::max(p.grow_rows, min_rows);
  if (std::max(p.need_segs, min_segs) > ctx->seg_capacity) want_segs = std::max(p.grow_segs, min_segs);
  // merge variant (scatter.hip): track samples far closer than a pad, i.e. the path-length dE/dx step; every
  // workgroup sorts its event's entries into two lists of its own in global memory
  const bool merge = !ctx->det.mc_diffusion && (ctx->opt_merge == 1 || (ctx->opt_merge < 0 && ctx->det.path_step > 0.0));
  const size_t merge_cap = (size_t)std::max(1, lay.n_sim) * MAX_BLOCKS_PER_TRACK * ARENA_BLK *
                           (ctx->det.longitudinal_diffusion > 0.0 ? ATTPC_LONG_STEPS : 1);
  const size_t merge_bytes = merge ? (size_t)p.wgs * 2 * merge_cap * sizeof(uint2) : 0;
  if (want_rows > ctx->cloud_capacity || want_segs > ctx->seg_capacity || (size_t)n * sizeof(uint32_t) > ctx->ev_rows.bytes ||
      merge_bytes > ctx->merge_scratch.bytes) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // nothing in flight may use the old buffers
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_c));
    if ((rc = ensure(ctx, ctx->merge_scratch, merge_bytes))) return rc;
    if ((rc = ensure(ctx, ctx->points, (size_t)want_rows * 3 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->labels, (size_t)want_rows * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->segments, (size_t)want_segs * sizeof(Segment)))) return rc;
    if ((rc = ensure(ctx, ctx->ev_rows, (size_t)std::max<uint32_t>(n, (uint32_t)std::max(1, ctx->chunk_events)) * sizeof(uint32_t)))) return rc;
    ctx->cloud_capacity = want_rows;
    ctx->seg_capacity = want_segs;
    // test hook: poison the segment list, so that a consumer of slots a launch did not write (one that ran
    // out of capacity) cannot go unnoticed on freshly allocated, zero-filled memory
    if (ctx->opt_tiny) HIP_TRY(ctx, hipMemsetAsync(ctx->segments.p, 0x7f, ctx->segments.bytes, ctx->stream));
  }
  unsigned long long* d_ctrl = static_cast<unsigned long long*>(ctx->out_ctrl.p) + (size_t)slot * CTRL_WORDS;
  HIP_TRY(ctx, hipMemsetAsync(d_ctrl, 0, CTRL_WORDS * sizeof(unsigned long long), ctx->stream));
  ScatterArgs sa;
  sa.det = ctx->det;
  sa.layout = lay;
  sa.trk = trk;
  sa.out.points = static_cast<double*>(ctx->points.p);
  sa.out.labels = static_cast<int64_t*>(ctx->labels.p);
  sa.out.segments = static_cast<Segment*>(ctx->segments.p);
  sa.out.ctrl = d_ctrl;
  sa.out.ev_rows = static_cast<uint32_t*>(ctx->ev_rows.p);
  sa.out.lone_list = static_cast<LoneBucket*>(ctx->lone_list.p);
  sa.out.lone_capacity = LONE_CAPACITY;
  sa.out.lone_chg = static_cast<unsigned long long*>(ctx->lone_chg.p);
  sa.out.lone_mask = static_cast<uint32_t*>(ctx->lone_mask.p);
  // The launch may use what its plan asked for (or what a repeated launch was found to need), not the whole
  // buffer: whoever assembles its cloud sizes the event-ordered copy, the Spyral rows and the transfer records by
  // this number, and a device-resident run of large chunks may have left a buffer of many times that size.
  ctx->launch_row_cap = std::min<int64_t>(ctx->cloud_capacity, std::max<int64_t>(p.grow_rows, min_rows));
  sa.out.capacity = ctx->launch_row_cap;
  sa.out.seg_capacity = ctx->seg_capacity;
  sa.seed = seed;
  sa.first_event = first_event;
  sa.n_events = n;
  sa.event0 = e0;
  sa.batch = p.batch;
  sa.row_block = p.row_block;
  sa.merge_scratch = merge ? static_cast<uint2*>(ctx->merge_scratch.p) : nullptr;
  sa.merge_cap = (uint32_t)merge_cap;
  HIP_TRY(ctx, hipEventRecord(ctx->s0[slot], ctx->stream));
  ctx->slot_wide[slot] = p.use_wide;
  if (p.use_wide) launch_scatter_kernel_wide(p.wgs, ctx->stream, sa);
  else if (p.use_small) launch_scatter_kernel_small(p.wgs, ctx->stream, sa);
  else launch_scatter_kernel_big(p.wgs, ctx->stream, sa);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipEventRecord(ctx->s1[slot], ctx->stream));
  launch_lone_bucket_kernel((uint32_t)LONE_WORKGROUPS, ctx->stream, sa);  // exits at once without lone buckets
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_out_ctrl + (size_t)slot * CTRL_WORDS, d_ctrl, CTRL_WORDS * sizeof(unsigned long long),
                              hipMemcpyDeviceToHost, ctx->stream));
#ifdef ATTPC_PHASE_TIMERS
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const unsigned long long* octrl = ctx->h_out_ctrl + (size_t)slot * CTRL_WORDS;
  fprintf(stderr, "[attpc phase cycles] init %llu hist %llu select %llu stage %llu items %llu insert-calls %llu insert-trips %llu flushcount %llu flushwrite %llu (events %u)\n",
          octrl[8], octrl[9], octrl[10], octrl[11], octrl[12], octrl[13] >> 32, octrl[13] & 0xffffffffull, octrl[14], octrl[15], n);
  fprintf(stderr, "[attpc rows-phase cycles] gathers %llu runs %llu scan+queue %llu drain %llu\n", octrl[16], octrl[17], octrl[18], octrl[19]);
  fprintf(stderr, "[attpc rounds] rows-rounds %llu staged %llu busiest-wave passes %llu\n", octrl[20], octrl[21], octrl[22]);
  fprintf(stderr, "[attpc flush cycles] to-barrier %llu to-compacted %llu atomics-wait %llu segment %llu select %llu barrier %llu\n", octrl[27], octrl[23], octrl[24], octrl[25], octrl[26], octrl[14]);
  fprintf(stderr, "[attpc ctrl] rows %llu segments %llu failed %llu retried %llu samples %llu\n", octrl[0], octrl[1],
          octrl[4], octrl[5], octrl[7]);
  fprintf(stderr, "[attpc per-wave wait at the window's last barrier]");
  for (int w = 0; w < 16; ++w) fprintf(stderr, " %llu", octrl[40 + w]);
  fprintf(stderr, "\n[attpc staging, even waves]");
  for (int w = 0; w < 8; ++w) fprintf(stderr, " %llu", octrl[56 + w]);
  fprintf(stderr, "\n");
#endif
  return ATTPC_OK;
}

// Read the control words of slot `slot` (its copy has completed).  Updates the context's size
// estimates; r->overflow says the launch has to be repeated with at least min_rows / min_segs.
void read_scatter(attpc_ctx* ctx, int slot, uint32_t n, ChunkResult* r, int64_t* min_rows, int64_t* min_segs) {
  const unsigned long long* o = ctx->h_out_ctrl + (size_t)slot * CTRL_WORDS;
  float ms = 0;
  if (hipEventElapsedTime(&ms, ctx->s0[slot], ctx->s1[slot]) == hipSuccess) r->ms_scatter += ms;
  r->overflow = o[6] != 0;
  r->reserved = o[0];
  r->segs = o[1];
  r->danger = o[32];
  // Windows whose u32 sums could have wrapped were done again by lone_bucket_kernel, one time bucket at a time: exact,
  // but meant for the odd window.  Where they are many (more than one per 64 events), or the list of lone buckets ran
  // over because of them, this detector needs u64 sums: the context switches to the wide build for good and this
  // launch is repeated with it (results do not depend on the build).
  if (!ctx->slot_wide[slot] && ctx->opt_variant == 0 && r->danger && (r->danger * 64ull > (unsigned long long)n || o[4] != 0)) {
    ctx->prefer_wide = true;
    r->overflow = true;
    *min_rows = std::max<int64_t>(*min_rows, ctx->launch_row_cap);
    *min_segs = std::max<int64_t>(*min_segs, ctx->seg_capacity);
    return;
  }
  if (r->overflow) {  // cloud / segment capacity exceeded (the cursors kept counting)
    *min_rows = (int64_t)(o[0] + o[0] / 8) + 65536;
    *min_segs = (int64_t)(o[1] + o[1] / 8) + 4096;
    return;
  }
  r->rows = o[30];  // rows written; o[0] is the reservation cursor (holes included)
  r->charge = o[2];
  r->keys = o[3];
  r->failed = o[4];
  r->retried = o[5];
  r->samples = o[7];
  r->mismatch = o[31];
  r->lone = std::min<unsigned long long>(o[29], LONE_CAPACITY);
  if (n) {
    ctx->rows_per_event = std::max((double)r->rows / (double)n, 1.0e-3);  // > 0 = known
    ctx->segs_per_event = (double)r->segs / (double)n;
    if (r->lone * 100ull > (unsigned long long)n) ctx->prefer_big = true;  // > 1 % of the events: the small table is too small here
  }
}

void accumulate(attpc_run_stats* st, const ChunkResult& r) {
  st->n_points += r.rows;
  st->n_track_samples += r.samples;
  st->n_failed += r.failed;
  st->n_lds_overflow += r.retried;
  st->charge_checksum += r.charge;
  st->key_checksum += r.keys;
  st->ms_scatter += r.ms_scatter;
  st->launches_scatter += 1;
  st->n_inconsistent += (uint32_t)r.mismatch;
  st->n_lone_buckets += r.lone;
}

// ------------------------------------------------------------------ assembly (delivered clouds) ----
int32_t ensure_pinned_start(attpc_ctx* ctx, AsmSet& as, size_t len) {
  if (len <= as.h_start_len) return ATTPC_OK;
  if (as.h_start) HIP_TRY(ctx, hipHostFree(as.h_start));
  if (as.h_ev_rows) HIP_TRY(ctx, hipHostFree(as.h_ev_rows));
  as.h_start = nullptr;
  as.h_ev_rows = nullptr;
  as.h_start_len = 0;
  HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&as.h_start), len * sizeof(int64_t), hipHostMallocDefault));
  HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&as.h_ev_rows), len * sizeof(uint32_t), hipHostMallocDefault));
  as.h_start_len = len;
  return ATTPC_OK;
}

// Queue, behind the scatter of slot `slot` on S, the assembly of its cloud into `as`: CSR offsets by a
// device scan of the per-event row counts, rows gathered into event order; for Spyral output also the
// kept-row counts, their scan and the converted, thresholded, z-sorted rows.  The row totals and the
// offsets are copied to pinned memory; as.ready is recorded at the end.  `rows_bound` >= the rows the
// launch can have produced (the reservation capacity).
int32_t enqueue_assembly(attpc_ctx* ctx, int slot, AsmSet& as, uint32_t n, bool spyral) {
  int32_t rc;
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, as.copied, 0));  // the set's previous contents have left
  // rows of the scatter launch queued just before (same stream, same slot), kept with 12 % headroom
  if ((size_t)ctx->launch_row_cap > as.row_cap) as.row_cap = (size_t)ctx->launch_row_cap + (size_t)ctx->launch_row_cap / 8;
  const size_t cap = as.row_cap;
  if ((rc = ensure(ctx, as.ev_start, ((size_t)n + 1) * sizeof(int64_t)))) return rc;
  if ((rc = ensure(ctx, as.points, cap * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, as.labels, cap * sizeof(int64_t)))) return rc;
  if ((rc = ensure_pinned_start(ctx, as, (size_t)n + 1))) return rc;
  const unsigned long long* d_ctrl = static_cast<const unsigned long long*>(ctx->out_ctrl.p) + (size_t)slot * CTRL_WORDS;
  hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, static_cast<const uint32_t*>(ctx->ev_rows.p), n,
                     static_cast<int64_t*>(as.ev_start.p), static_cast<int64_t*>(nullptr), d_ctrl);
  HIP_TRY(ctx, hipGetLastError());
  hipLaunchKernelGGL(gather_segments_kernel, dim3(4096), dim3(256), 0, ctx->stream, static_cast<const Segment*>(ctx->segments.p),
                     d_ctrl, ctx->seg_capacity, static_cast<const int64_t*>(as.ev_start.p),
                     static_cast<const double*>(ctx->points.p), static_cast<const int64_t*>(ctx->labels.p),
                     static_cast<double*>(as.points.p), static_cast<int64_t*>(as.labels.p));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(as.h_ev_rows, ctx->ev_rows.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (!spyral) {
    HIP_TRY(ctx, hipMemcpyAsync(as.h_start, as.ev_start.p, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->opt_compact) {
      if ((rc = ensure(ctx, as.packed, cap * sizeof(PackedRow) + 2 * sizeof(int64_t)))) return rc;
      int64_t* d_flag = reinterpret_cast<int64_t*>(static_cast<char*>(as.packed.p) + cap * sizeof(PackedRow));
      HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 2 * sizeof(int64_t), ctx->stream));
      hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)ctx->n_cus * 8u), dim3(256), 0, ctx->stream,
                         static_cast<const int64_t*>(as.ev_start.p), n, static_cast<const double*>(as.points.p),
                         static_cast<const int64_t*>(as.labels.p), static_cast<PackedRow*>(as.packed.p), d_flag,
                         ctx->opt_compact == 2 ? 1 : 0);
      HIP_TRY(ctx, hipGetLastError());
      HIP_TRY(ctx, hipMemcpyAsync(as.h_total, d_flag, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    }
  } else {
    if ((rc = ensure(ctx, as.kept, (size_t)n * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, as.kept_start, ((size_t)n + 1) * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, as.sp_rows, cap * 8 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, as.sp_labels, cap * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sort_idx, cap * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sort_key, cap * sizeof(double)))) return rc;
    launch_spyral_count(ctx->stream, ctx->spyral, n, static_cast<const int64_t*>(as.ev_start.p),
                        static_cast<const double*>(as.points.p), static_cast<uint32_t*>(as.kept.p));
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, static_cast<const uint32_t*>(as.kept.p), n,
                       static_cast<int64_t*>(as.kept_start.p), static_cast<int64_t*>(nullptr),
                       static_cast<const unsigned long long*>(nullptr));
    HIP_TRY(ctx, hipGetLastError());
    SpyralPacked* d_packed = nullptr;
    int64_t* d_flag = nullptr;
    if (ctx->opt_compact) {  // 24-byte records instead of rows of 8 doubles + label; the flag sits behind them
      if ((rc = ensure(ctx, as.packed, cap * sizeof(SpyralPacked) + 2 * sizeof(int64_t)))) return rc;
      d_packed = static_cast<SpyralPacked*>(as.packed.p);
      d_flag = reinterpret_cast<int64_t*>(static_cast<char*>(as.packed.p) + cap * sizeof(SpyralPacked));
      HIP_TRY(ctx, hipMemsetAsync(d_flag, 0, 2 * sizeof(int64_t), ctx->stream));
    }
    launch_spyral_write(ctx->stream, ctx->spyral, n, static_cast<const int64_t*>(as.ev_start.p),
                        static_cast<const int64_t*>(as.kept_start.p), static_cast<const double*>(as.points.p),
                        static_cast<const int64_t*>(as.labels.p), static_cast<double*>(as.sp_rows.p),
                        static_cast<int64_t*>(as.sp_labels.p), static_cast<uint32_t*>(ctx->sort_idx.p),
                        static_cast<double*>(ctx->sort_key.p), d_packed, d_flag);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->opt_compact) HIP_TRY(ctx, hipMemcpyAsync(as.h_total, d_flag, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(as.h_start, as.kept_start.p, ((size_t)n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  HIP_TRY(ctx, hipEventRecord(as.ready, ctx->stream));
  return ATTPC_OK;
}

void unpacker_main(attpc_ctx* ctx) {
  (void)hipSetDevice(ctx->device);
  for (;;) {
    UnpackJob job;
    {
      std::unique_lock<std::mutex> lock(ctx->unpack_mutex);
      ctx->unpack_cv.wait(lock, [&] { return ctx->unpack_stop || !ctx->unpack_jobs.empty(); });
      if (ctx->unpack_jobs.empty()) return;  // stop requested and nothing left
      job = ctx->unpack_jobs.front();
      ctx->unpack_jobs.pop_front();
    }
    const bool ok = hipEventSynchronize(job.copied) == hipSuccess;
    if (ok && job.spyral) {
      SpyralHostTables t;
      t.centers = ctx->h_pad_centers.data();
      t.sizes = ctx->h_pad_sizes.data();
      t.n_pads = ctx->spyral.n_pads;
      t.r_max = ctx->spyral.r_max;
      t.window_edge = ctx->spyral.window_edge;
      t.mm_edge = ctx->spyral.mm_edge;
      t.length = ctx->spyral.length;
      unpack_spyral_rows(static_cast<const SpyralPacked*>(job.src), job.rows, t, job.points, job.labels, ctx->opt_unpack_threads);
    } else if (ok && job.tight) {
      unpack_rows8(static_cast<const unsigned long long*>(job.src), job.rows, job.offsets.data(), (int64_t)job.offsets.size() - 1,
                   job.seed, job.first_event, job.points, job.labels, ctx->opt_unpack_threads);
    } else if (ok) {
      unpack_rows(static_cast<const PackedRow*>(job.src), job.rows, job.points, job.labels, ctx->opt_unpack_threads);
    }
    {
      std::lock_guard<std::mutex> lock(ctx->unpack_mutex);
      ctx->unpack_done++;
      if (!ok) ctx->unpack_failed = true;
    }
    ctx->unpack_cv.notify_all();
  }
}

uint64_t submit_unpack(attpc_ctx* ctx, const UnpackJob& job) {
  uint64_t ticket;
  {
    std::lock_guard<std::mutex> lock(ctx->unpack_mutex);
    if (!ctx->unpacker.joinable()) ctx->unpacker = std::thread(unpacker_main, ctx);
    ctx->unpack_jobs.push_back(job);
    ticket = ++ctx->unpack_submitted;
  }
  ctx->unpack_cv.notify_all();
  return ticket;
}

int32_t wait_unpacked(attpc_ctx* ctx, uint64_t ticket);

// Every run entry point holds one of these: whatever way the call ends, no expansion job may outlive it (the
// jobs write into the caller's arrays).
struct UnpackDrain {
  attpc_ctx* ctx;
  explicit UnpackDrain(attpc_ctx* c) : ctx(c) {}
  ~UnpackDrain() {
    std::unique_lock<std::mutex> lock(ctx->unpack_mutex);
    ctx->unpack_cv.wait(lock, [&] { return ctx->unpack_done >= ctx->unpack_submitted; });
  }
};

// The chunk in `as` is ready on the device: write its offsets, queue its copy to the caller's arrays on C.
int32_t deliver_chunk(attpc_ctx* ctx, AsmSet& as, uint32_t n, uint64_t chunk_first_local, bool spyral, attpc_cloud_out* out,
                      int64_t* row_cursor, bool* over_capacity, uint64_t seed, uint64_t chunk_first_global) {
  const int64_t base = *row_cursor;
  const int64_t total = as.h_start[n];
  if (out->offsets)
    for (uint32_t i = 0; i <= n; ++i) out->offsets[chunk_first_local + i] = base + as.h_start[i];
  if (out->event_points)
    for (uint32_t i = 0; i < n; ++i) out->event_points[chunk_first_local + i] = (int64_t)as.h_ev_rows[i];
  *row_cursor = base + total;
  if (!out->points || !out->labels || *row_cursor > out->capacity) {
    if (*row_cursor > out->capacity) *over_capacity = true;
    HIP_TRY(ctx, hipEventRecord(as.copied, ctx->stream_c));
    return ATTPC_OK;
  }
  const bool compact = ctx->opt_compact && as.h_total[0] == 0;  // [0]: a row of the chunk does not fit the record
  const bool tight = compact && !spyral && ctx->opt_compact == 2 && as.h_total[1] == 0;  // [1]: ... the 8-byte record
  if (total > 0 && compact && !spyral && ctx->opt_compact == 2 && !tight) {
    // the pack kernel wrote 8-byte records and one of them does not hold its row: pack again, 16 bytes per row (rare)
    hipLaunchKernelGGL(pack_rows_kernel, dim3((unsigned)ctx->n_cus * 8u), dim3(256), 0, ctx->stream,
                       static_cast<const int64_t*>(as.ev_start.p), n, static_cast<const double*>(as.points.p),
                       static_cast<const int64_t*>(as.labels.p), static_cast<PackedRow*>(as.packed.p),
                       reinterpret_cast<int64_t*>(static_cast<char*>(as.packed.p) + as.row_cap * sizeof(PackedRow)), 0);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (total > 0 && spyral && ctx->opt_compact && !compact) {
    // the write kernel produced records only: produce the rows themselves for the plain copy below (rare)
    launch_spyral_write(ctx->stream, ctx->spyral, n, static_cast<const int64_t*>(as.ev_start.p),
                        static_cast<const int64_t*>(as.kept_start.p), static_cast<const double*>(as.points.p),
                        static_cast<const int64_t*>(as.labels.p), static_cast<double*>(as.sp_rows.p),
                        static_cast<int64_t*>(as.sp_labels.p), static_cast<uint32_t*>(ctx->sort_idx.p),
                        static_cast<double*>(ctx->sort_key.p), nullptr, nullptr);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (total > 0 && compact) {
    const size_t bytes = (size_t)total * (spyral ? sizeof(SpyralPacked) : (tight ? sizeof(unsigned long long) : sizeof(PackedRow)));
    {  // the staging's previous occupant (two chunks back) must have been expanded
      int32_t rcw = wait_unpacked(ctx, as.unpack_ticket);
      if (rcw) return rcw;
    }
    if (bytes > as.h_packed_bytes) {  // library-owned pinned staging, grow-only with headroom
      if (as.h_packed) HIP_TRY(ctx, hipHostFree(as.h_packed));
      as.h_packed = nullptr;
      as.h_packed_bytes = 0;
      HIP_TRY(ctx, hipHostMalloc(&as.h_packed, bytes + bytes / 4, hipHostMallocDefault));
      as.h_packed_bytes = bytes + bytes / 4;
    }
    HIP_TRY(ctx, hipMemcpyAsync(as.h_packed, as.packed.p, bytes, hipMemcpyDeviceToHost, ctx->stream_c));
    HIP_TRY(ctx, hipEventRecord(as.copied, ctx->stream_c));
    UnpackJob job;
    job.copied = as.copied;
    job.src = as.h_packed;
    job.rows = total;
    job.points = out->points + base * (spyral ? 8 : 3);
    job.labels = out->labels + base;
    job.spyral = spyral;
    if (tight) {
      job.tight = true;
      job.seed = seed;
      job.first_event = chunk_first_global;
      job.offsets.assign(as.h_start, as.h_start + n + 1);
    }
    as.unpack_ticket = submit_unpack(ctx, job);
    return ATTPC_OK;
  } else if (total > 0) {
    const size_t width = spyral ? 8 : 3;
    HIP_TRY(ctx, hipMemcpyAsync(out->points + base * width, spyral ? as.sp_rows.p : as.points.p, (size_t)total * width * sizeof(double),
                                hipMemcpyDeviceToHost, ctx->stream_c));
    HIP_TRY(ctx, hipMemcpyAsync(out->labels + base, spyral ? as.sp_labels.p : as.labels.p, (size_t)total * sizeof(int64_t),
                                hipMemcpyDeviceToHost, ctx->stream_c));
  }
  HIP_TRY(ctx, hipEventRecord(as.copied, ctx->stream_c));
  return ATTPC_OK;
}

// Wait until expansion job `ticket` (and every earlier one) is done: the staging it read is free again and its
// rows are in the caller's arrays.
int32_t wait_unpacked(attpc_ctx* ctx, uint64_t ticket) {
  std::unique_lock<std::mutex> lock(ctx->unpack_mutex);
  ctx->unpack_cv.wait(lock, [&] { return ctx->unpack_done >= ticket || ctx->unpack_failed; });
  if (ctx->unpack_failed) return fail(ctx, ATTPC_E_HIP, "waiting for a device-to-host copy failed in the expansion thread");
  return ATTPC_OK;
}

// ------------------------------------------------------------------ the run loop ----
struct RunSource {   // where a batch's kinematics come from
  bool from_kernel = false;        // attpc_sim_run: kin_run_kernel on T
  const double* h_p4 = nullptr;    // attpc_det_run: host arrays
  const double* h_vertex = nullptr;
};

struct RunSink {     // optional host copies of the kinematics (attpc_sim_run)
  double* p4 = nullptr;
  double* vertex = nullptr;
  int32_t* status = nullptr;
};

int32_t queue_batch(attpc_ctx* ctx, TrackSet& ts, TrackLaunch& tl, const attpc_event_layout& lay, const RunSource& src,
                    uint64_t seed, uint64_t first_event, uint64_t b0, uint32_t nb, int n_rows) {
  int32_t rc;
  if ((rc = ensure_kin_buffers(ctx, ts, nb, n_rows))) return rc;
  ts.timed_kin = false;
  if (src.from_kernel) {
    HIP_TRY(ctx, hipEventRecord(ts.k0, ctx->stream_t));
    launch_kin_run(ctx->stream_t, ctx->kin, seed, first_event + b0, nb, static_cast<double*>(ts.p4.p),
                   static_cast<double*>(ts.vertex.p), static_cast<int32_t*>(ts.status.p), static_cast<uint32_t*>(ts.attempts.p));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ts.k1, ctx->stream_t));
    ts.timed_kin = true;
  } else {
    HIP_TRY(ctx, hipMemcpyAsync(ts.p4.p, src.h_p4 + b0 * n_rows * 4, (size_t)nb * n_rows * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream_t));
    HIP_TRY(ctx, hipMemcpyAsync(ts.vertex.p, src.h_vertex + b0 * 3, (size_t)nb * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream_t));
  }
  tl.lay = lay;
  tl.seed = seed;
  tl.first_event = first_event + b0;
  tl.n = nb;
  tl.use_status = src.from_kernel;
  return launch_tracks(ctx, ts, tl);
}

// Scatter (and deliver) the chunks of one finished track batch.
// `queue_next` is called once, right after the first launches of this batch are queued on S: it queues the
// next batch's kinematics + tracks on the low-priority stream T, BEHIND scatter work, so that the track
// workgroups only take the compute units the persistent scatter workgroups leave.
template <typename QueueNext>
int32_t run_batch_chunks(attpc_ctx* ctx, const attpc_event_layout& lay, const TrackBuffers& trk, uint64_t seed,
                         uint64_t batch_first_global, uint64_t batch_first_local, uint32_t nb, attpc_cloud_out* out,
                         bool spyral, attpc_run_stats* st, int64_t* row_cursor, bool* over, QueueNext queue_next) {
  int32_t rc;
  bool next_queued = false;
  auto queue_next_once = [&]() -> int32_t {
    if (next_queued) return ATTPC_OK;
    next_queued = true;
    return queue_next();
  };
  if (lay.n_sim == 0 || nb == 0) {  // nothing to scatter: empty clouds
    if ((rc = queue_next_once())) return rc;
    if (out && out->offsets)
      for (uint32_t i = 0; i <= nb; ++i) out->offsets[batch_first_local + i] = *row_cursor;
    if (out && out->event_points)
      for (uint32_t i = 0; i < nb; ++i) out->event_points[batch_first_local + i] = 0;
    return ATTPC_OK;
  }
  if ((rc = ensure(ctx, ctx->out_ctrl, (size_t)MAX_SLOTS * CTRL_WORDS * sizeof(unsigned long long)))) return rc;
  if ((rc = ensure(ctx, ctx->lone_list, (size_t)LONE_CAPACITY * sizeof(LoneBucket)))) return rc;
  if (!ctx->lone_ready) {  // lone_bucket_kernel's tables: zero once, the kernel leaves them zero.  The flag is set
                           // only when both allocations and both memsets went through (a call that failed half-way
                           // is repeated from the start by the next run)
    if ((rc = ensure(ctx, ctx->lone_chg, (size_t)LONE_WORKGROUPS * LONE_PADS * sizeof(unsigned long long)))) return rc;
    if ((rc = ensure(ctx, ctx->lone_mask, (size_t)LONE_WORKGROUPS * (LONE_PADS / 4) * sizeof(uint32_t)))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->lone_chg.p, 0, ctx->lone_chg.bytes, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->lone_mask.p, 0, ctx->lone_mask.bytes, ctx->stream));
    ctx->lone_ready = true;
  }
  struct Chunk { uint32_t e0, n; int slot; };
  if (!out) {
    // device resident: queue up to MAX_SLOTS chunks back to back, read their control words once
    uint32_t e0 = 0;
    while (e0 < nb) {
      std::vector<Chunk> group;
      while (e0 < nb && (int)group.size() < MAX_SLOTS) {
        const uint32_t n = next_chunk_events(ctx, nb - e0);
        const Chunk c{e0, n, (int)group.size()};
        if ((rc = enqueue_scatter(ctx, c.slot, lay, trk, seed, batch_first_global + e0, e0, n, 0, 0))) return rc;
        group.push_back(c);
        e0 += n;
        if (ctx->rows_per_event <= 0.0) break;  // pilot chunk: size the rest from what it produced
      }
      if ((rc = queue_next_once())) return rc;
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      for (const Chunk& c : group) {
        ChunkResult r;
        int64_t min_rows = 0, min_segs = 0;
        read_scatter(ctx, c.slot, c.n, &r, &min_rows, &min_segs);
        for (int attempt = 0; r.overflow && attempt < 8; ++attempt) {  // too small: run this chunk again
          if ((rc = enqueue_scatter(ctx, c.slot, lay, trk, seed, batch_first_global + c.e0, c.e0, c.n, min_rows, min_segs))) return rc;
          HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
          read_scatter(ctx, c.slot, c.n, &r, &min_rows, &min_segs);
        }
        if (r.overflow) return fail(ctx, ATTPC_E_HIP, "point cloud did not fit after repeated buffer growth");
        accumulate(st, r);
      }
    }
    return ATTPC_OK;
  }
  // delivered clouds: chunk c+1 is scattered and assembled while chunk c crosses PCIe
  Chunk prev{0, 0, -1};
  int prev_set = 0;
  uint32_t e0 = 0;
  int seq = 0;
  auto complete = [&](const Chunk& c, int set) -> int32_t {
    AsmSet& as = ctx->aset[set];
    HIP_TRY(ctx, hipEventSynchronize(as.ready));
    ChunkResult r;
    int64_t min_rows = 0, min_segs = 0;
    read_scatter(ctx, c.slot, c.n, &r, &min_rows, &min_segs);
    for (int attempt = 0; r.overflow && attempt < 8; ++attempt) {
      int32_t rc2;
      if ((rc2 = enqueue_scatter(ctx, c.slot, lay, trk, seed, batch_first_global + c.e0, c.e0, c.n, min_rows, min_segs))) return rc2;
      if ((rc2 = enqueue_assembly(ctx, c.slot, as, c.n, spyral))) return rc2;
      HIP_TRY(ctx, hipEventSynchronize(as.ready));
      read_scatter(ctx, c.slot, c.n, &r, &min_rows, &min_segs);
    }
    if (r.overflow) return fail(ctx, ATTPC_E_HIP, "point cloud did not fit after repeated buffer growth");
    accumulate(st, r);
    return deliver_chunk(ctx, as, c.n, batch_first_local + c.e0, spyral, out, row_cursor, over, seed, batch_first_global + c.e0);
  };
  while (e0 < nb) {
    const bool pilot = ctx->rows_per_event <= 0.0;  // only ever true with nothing in flight
    // delivered clouds are PCIe bound: small chunks, so that the copy of one hides the scatter and
    // assembly of the next from the first chunk on
    // ... and bounded in rows as well: the assembly sets, the Spyral sort scratch and the pinned staging all scale
    // with the chunk's cloud (configs[4] has 54 k points per event: 8192 such events would be 100 GB of them)
    uint32_t n = std::min<uint32_t>(next_chunk_events(ctx, nb - e0), (uint32_t)ctx->opt_deliver_chunk);
    if (ctx->rows_per_event > 0.0)
      n = std::min<uint32_t>(n, (uint32_t)std::max(256.0, (double)DELIVER_CHUNK_ROWS / ctx->rows_per_event));
    const Chunk c{e0, n, seq % MAX_SLOTS};
    const int set = seq & 1;
    // an overflow of the chunk in flight is repaired inside complete(); queue this one behind it
    if ((rc = enqueue_scatter(ctx, c.slot, lay, trk, seed, batch_first_global + e0, e0, n, 0, 0))) return rc;
    if ((rc = enqueue_assembly(ctx, c.slot, ctx->aset[set], n, spyral))) return rc;
    if ((rc = queue_next_once())) return rc;
    if (prev.slot >= 0 && (rc = complete(prev, prev_set))) return rc;
    prev = c;
    prev_set = set;
    e0 += n;
    seq++;
    if (pilot) {  // size the following chunks from the pilot
      if ((rc = complete(prev, prev_set))) return rc;
      prev.slot = -1;
    }
  }
  if (prev.slot >= 0 && (rc = complete(prev, prev_set))) return rc;
  if ((rc = wait_unpacked(ctx, ctx->unpack_submitted))) return rc;  // every row is in the caller's arrays
  return queue_next_once();
}

int32_t run_events(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events, const attpc_event_layout& lay,
                   const RunSource& src, const RunSink& sink, attpc_cloud_out* out, bool spyral, attpc_run_stats* stats) {
  int32_t rc;
  UnpackDrain drain(ctx);
  attpc_run_stats st{};
  st.n_events = n_events;
  const uint64_t growths_before = ctx->n_growths;
  const int n_rows = lay.n_rows;
  int64_t row_cursor = 0;
  bool over = false;
  if (out && out->offsets) out->offsets[0] = 0;
  // batches of up to MAX_SLOTS chunks (a small pilot batch while the arena need per track is unknown),
  // each integrated on T while the previous batch is scattered on S
  uint64_t b0 = 0;
  // the call starts on the track set with the larger buffers: a context's first call leaves set 0 sized for its pilot
  // batch and set 1 for a full one, and a later call of one batch per call would grow set 0 (arena and all) to the
  // same size for nothing
  int cur = ctx->tset[1].p4.bytes > ctx->tset[0].p4.bytes ? 1 : 0;
  TrackLaunch tl[2];
  auto batch_size = [&](uint64_t at) -> uint32_t {
    const uint64_t chunk = (uint64_t)std::max(1, ctx->chunk_events);
    uint64_t want = track_batch_events(ctx, lay, chunk);
    if (ctx->blocks_per_track <= 0.0) {
      want = std::min<uint64_t>(want, std::min<uint64_t>(chunk, 16384));  // pilot: sizes the arena
    } else {
      // the per-batch buffers -- the sample arena: tens of GB -- are sized for the largest batch a call of this length
      // has, not for what is left of THIS call behind its pilot batch: the next call of the same length would grow
      // them all again (8 re-allocations inside a caller's timed region)
      ctx->max_batch_events = std::max<uint32_t>(ctx->max_batch_events, (uint32_t)std::min<uint64_t>(want, n_events));
      // the first batch of a call is integrated with nothing to run beside: a shorter one (opt_first_batch_chunks
      // scatter chunks) leaves less of the call's track time exposed
      if (at == 0 && ctx->opt_first_batch_chunks > 0) want = std::min<uint64_t>(want, chunk * (uint64_t)ctx->opt_first_batch_chunks);
    }
    return (uint32_t)std::min<uint64_t>(want, n_events - at);
  };
  // the caller's announcement of the call AFTER this one (attpc_sim_hint_next) belongs to this run alone
  const bool hinted = ctx->hint_valid && src.from_kernel;
  const uint64_t hint_seed = ctx->hint_seed, hint_first = ctx->hint_first, hint_n = ctx->hint_n;
  const attpc_event_layout hint_lay = ctx->hint_lay;
  ctx->hint_valid = false;
  uint32_t nb = n_events ? batch_size(0) : 0;
  if (ctx->pre_valid && src.from_kernel && nb && ctx->pre_seed == seed && ctx->pre_first == first_event &&
      ctx->pre_nb <= n_events && same_layout(ctx->pre_lay, lay)) {
    // the previous run queued this call's first track batch behind its own last scatter launches: take it over
    ctx->pre_valid = false;
    cur = ctx->pre_set;
    nb = ctx->pre_nb;
    tl[cur].lay = lay;
    tl[cur].seed = seed;
    tl[cur].first_event = first_event;
    tl[cur].n = nb;
    tl[cur].use_status = true;
  } else {
    if ((rc = drop_prefetch(ctx))) return rc;
    if (nb && (rc = queue_batch(ctx, ctx->tset[cur], tl[cur], lay, src, seed, first_event, 0, nb, n_rows))) return rc;
  }
  while (nb) {
    TrackSet& ts = ctx->tset[cur];
    TrackBuffers trk;
    if ((rc = finish_tracks(ctx, ts, tl[cur], &trk, &st.ms_tracks, &st.n_sample_limit, &st.n_tracks_capped))) return rc;
    st.launches_tracks += 1;
    if (ts.timed_kin) {
      float ms_k = 0;
      HIP_TRY(ctx, hipEventElapsedTime(&ms_k, ts.k0, ts.k1));
      st.ms_kinematics += ms_k;
      st.launches_kinematics += 1;
    }
    // the next batch goes to the other set: the scatter chunks of the batch before this one have all
    // been read, so that set is free
    const uint64_t next_b0 = b0 + nb;
    uint32_t next_nb = 0;
    auto queue_next = [&]() -> int32_t {
      if (next_b0 >= n_events) {
        // the last batch of this call: the other track set is free, and the caller said what comes next -- that call's
        // first kinematics + track batch goes onto the low-priority stream now, behind this call's last scatter
        // launches, instead of standing alone at the head of the next call
        if (!hinted || hint_n == 0) return ATTPC_OK;
        const uint64_t chunk = (uint64_t)std::max(1, ctx->chunk_events);
        const uint32_t pre_nb = (uint32_t)std::min<uint64_t>(track_batch_events(ctx, hint_lay, chunk), hint_n);
        TrackLaunch pre_tl;
        int32_t rcq = queue_batch(ctx, ctx->tset[cur ^ 1], pre_tl, hint_lay, src, hint_seed, hint_first, 0, pre_nb, hint_lay.n_rows);
        if (rcq) return rcq;
        ctx->pre_valid = true;
        ctx->pre_set = cur ^ 1;
        ctx->pre_seed = hint_seed;
        ctx->pre_first = hint_first;
        ctx->pre_nb = pre_nb;
        ctx->pre_lay = hint_lay;
        return ATTPC_OK;
      }
      next_nb = batch_size(next_b0);
      return queue_batch(ctx, ctx->tset[cur ^ 1], tl[cur ^ 1], lay, src, seed, first_event, next_b0, next_nb, n_rows);
    };
    if (sink.status) HIP_TRY(ctx, hipMemcpyAsync(sink.status + b0, ts.status.p, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (sink.p4) HIP_TRY(ctx, hipMemcpyAsync(sink.p4 + b0 * n_rows * 4, ts.p4.p, (size_t)nb * n_rows * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (sink.vertex) HIP_TRY(ctx, hipMemcpyAsync(sink.vertex + b0 * 3, ts.vertex.p, (size_t)nb * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = run_batch_chunks(ctx, lay, trk, seed, first_event + b0, b0, nb, out, spyral, &st, &row_cursor, &over, queue_next))) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // this set's readers are done before it is refilled
    b0 = next_b0;
    nb = next_nb;
    cur ^= 1;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_c));
  if (spyral) st.n_points = (uint64_t)row_cursor;  // rows that survive the threshold
  st.n_buffer_growths = ctx->n_growths - growths_before;
  st.device_bytes = ctx->device_bytes;
  if (stats) *stats = st;
  if (over) return fail(ctx, ATTPC_E_CAPACITY, "cloud needs %lld rows, capacity %lld", (long long)row_cursor, (long long)out->capacity);
  if (st.n_failed || st.n_inconsistent)
    return fail(ctx, ATTPC_E_DATALOSS, "%llu events lost a time bucket (n_failed), %u table self-check failures (n_inconsistent) in %llu events",
                (unsigned long long)st.n_failed, st.n_inconsistent, (unsigned long long)n_events);
  return ATTPC_OK;
}

}  // namespace

extern "C" {

int32_t attpc_version(void) { return ATTPC_ABI_VERSION; }

int32_t attpc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t attpc_ctx_create(int32_t device, attpc_ctx** out) {
  if (!out) return ATTPC_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return ATTPC_E_NODEVICE;
  if (hipSetDevice(device) != hipSuccess) return ATTPC_E_HIP;
  attpc_ctx* ctx = new attpc_ctx();
  ctx->device = device;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
  }
  bool ok = true;
  int prio_low = 0, prio_high = 0;
  if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) prio_low = prio_high = 0;
  ok = ok && hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_high) == hipSuccess;
  ok = ok && hipStreamCreateWithPriority(&ctx->stream_t_own, hipStreamNonBlocking, prio_low) == hipSuccess;
  pick_track_stream(ctx);
  ok = ok && hipStreamCreateWithPriority(&ctx->stream_c, hipStreamNonBlocking, prio_high) == hipSuccess;
  auto make_event = [&](hipEvent_t* e) { ok = ok && hipEventCreate(e) == hipSuccess; };
  for (TrackSet& ts : ctx->tset) {
    make_event(&ts.done); make_event(&ts.k0); make_event(&ts.k1); make_event(&ts.t0); make_event(&ts.t1);
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&ts.h_ctrl), 16 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
  }
  for (int i = 0; i < MAX_SLOTS; ++i) { make_event(&ctx->s0[i]); make_event(&ctx->s1[i]); }
  for (AsmSet& as : ctx->aset) {
    make_event(&as.ready); make_event(&as.copied);
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&as.h_total), 2 * sizeof(int64_t), hipHostMallocDefault) == hipSuccess;
  }
  ok = ok && hipHostMalloc(reinterpret_cast<void**>(&ctx->h_out_ctrl), (size_t)MAX_SLOTS * CTRL_WORDS * sizeof(unsigned long long),
                           hipHostMallocDefault) == hipSuccess;
  if (!ok) {
    attpc_ctx_destroy(ctx);
    return ATTPC_E_HIP;
  }
  *out = ctx;
  return ATTPC_OK;
}

int32_t attpc_ctx_destroy(attpc_ctx* ctx) {
  if (!ctx) return ATTPC_OK;
  (void)hipSetDevice(ctx->device);
  {
    std::lock_guard<std::mutex> lock(ctx->unpack_mutex);
    ctx->unpack_stop = true;
  }
  ctx->unpack_cv.notify_all();
  if (ctx->unpacker.joinable()) ctx->unpacker.join();
  if (ctx->stream_t_own) (void)hipStreamSynchronize(ctx->stream_t_own);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->stream_c) (void)hipStreamSynchronize(ctx->stream_c);
  free_all(ctx->kin_allocs);
  free_all(ctx->det_allocs);
  free_all(ctx->spyral_allocs);
  std::vector<DevBuf*> bufs = {&ctx->points, &ctx->labels, &ctx->segments, &ctx->ev_rows, &ctx->lone_list, &ctx->lone_chg,
                               &ctx->lone_mask, &ctx->out_ctrl, &ctx->merge_scratch,
                               &ctx->sort_idx, &ctx->sort_key};
  for (TrackSet& ts : ctx->tset) {
    for (DevBuf* b : {&ts.p4, &ts.vertex, &ts.status, &ts.attempts, &ts.arena, &ts.block_table, &ts.counts, &ts.n_steps, &ts.ctrl})
      bufs.push_back(b);
    for (hipEvent_t e : {ts.done, ts.k0, ts.k1, ts.t0, ts.t1})
      if (e) (void)hipEventDestroy(e);
    if (ts.h_ctrl) (void)hipHostFree(ts.h_ctrl);
  }
  for (AsmSet& as : ctx->aset) {
    for (DevBuf* b : {&as.ev_start, &as.points, &as.labels, &as.kept, &as.kept_start, &as.sp_rows, &as.sp_labels, &as.packed}) bufs.push_back(b);
    if (as.h_packed) (void)hipHostFree(as.h_packed);
    for (hipEvent_t e : {as.ready, as.copied})
      if (e) (void)hipEventDestroy(e);
    if (as.h_start) (void)hipHostFree(as.h_start);
    if (as.h_ev_rows) (void)hipHostFree(as.h_ev_rows);
    if (as.h_total) (void)hipHostFree(as.h_total);
  }
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto& b : ctx->scratch)
    if (b.p) (void)hipFree(b.p);
  for (int i = 0; i < MAX_SLOTS; ++i) {
    if (ctx->s0[i]) (void)hipEventDestroy(ctx->s0[i]);
    if (ctx->s1[i]) (void)hipEventDestroy(ctx->s1[i]);
  }
  if (ctx->h_out_ctrl) (void)hipHostFree(ctx->h_out_ctrl);
  for (void* p : ctx->host_allocs) (void)hipHostFree(p);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream_t_own) (void)hipStreamDestroy(ctx->stream_t_own);
  if (ctx->stream_c) (void)hipStreamDestroy(ctx->stream_c);
  delete ctx;
  return ATTPC_OK;
}

const char* attpc_last_error(const attpc_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int32_t attpc_set_chunk_events(attpc_ctx* ctx, int32_t chunk_events) {
  if (!ctx) return ATTPC_E_INVALID;
  ctx->chunk_events = chunk_events > 0 ? chunk_events : 65536;
  return ATTPC_OK;
}

int32_t attpc_set_option(attpc_ctx* ctx, const char* name, int64_t value) {
  if (!ctx || !name) return ATTPC_E_INVALID;
  const std::string key(name);
  if (key == "scatter_variant") {
    if (value < 0 || value > 3) return fail(ctx, ATTPC_E_INVALID, "scatter_variant must be 0, 1, 2 or 3");
    ctx->opt_variant = (int)value;
  } else if (key == "tiny_buffers") {
    ctx->opt_tiny = value != 0;
  } else if (key == "compact_transfer") {
    if (value < 0 || value > 2) return fail(ctx, ATTPC_E_INVALID, "compact_transfer must be 0, 1 or 2");
    ctx->opt_compact = (int)value;
  } else if (key == "deliver_chunk_events") {
    if (value < 64 || value > (1 << 20)) return fail(ctx, ATTPC_E_INVALID, "deliver_chunk_events must be 64..1048576");
    ctx->opt_deliver_chunk = (int)value;
  } else if (key == "unpack_threads") {
    if (value < 0 || value > 1024) return fail(ctx, ATTPC_E_INVALID, "unpack_threads must be 0..1024");
    ctx->opt_unpack_threads = (int)value;
  } else if (key == "serial_tracks") {
    // != 0: kinematics + track integration of the next batch are queued on the scatter stream, behind the current
    // batch's scatter launches, instead of beside them on the low-priority stream
    int32_t rcs = drop_prefetch(ctx);
    if (rcs) return rcs;
    if ((rcs = sync_all(ctx))) return rcs;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_t_own));
    if (value < -1 || value > 1) return fail(ctx, ATTPC_E_INVALID, "serial_tracks must be -1, 0 or 1");
    ctx->opt_serial_tracks = (int)value;
    pick_track_stream(ctx);
  } else if (key == "track_blocks_per_cu") {
    if (value < 1 || value > 64) return fail(ctx, ATTPC_E_INVALID, "track_blocks_per_cu must be 1..64");
    ctx->opt_track_blocks_per_cu = (int)value;
  } else if (key == "track_species_major") {
    ctx->opt_track_species_major = value != 0;
  } else if (key == "first_batch_chunks") {
    if (value < 0 || value > MAX_SLOTS) return fail(ctx, ATTPC_E_INVALID, "first_batch_chunks must be 0..8");
    ctx->opt_first_batch_chunks = (int)value;
  } else if (key == "scatter_merge") {
    if (value < -1 || value > 1) return fail(ctx, ATTPC_E_INVALID, "scatter_merge must be -1, 0 or 1");
    ctx->opt_merge = (int)value;
  } else if (key == "chunk_events") {
    return attpc_set_chunk_events(ctx, (int32_t)value);
  } else {
    return fail(ctx, ATTPC_E_INVALID, "unknown option '%s'", name);
  }
  return ATTPC_OK;
}

int32_t attpc_host_alloc(attpc_ctx* ctx, uint64_t bytes, void** out) {
  if (!ctx || !out) return ATTPC_E_INVALID;
  *out = nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  void* p = nullptr;
  HIP_TRY(ctx, hipHostMalloc(&p, std::max<uint64_t>(bytes, 1), hipHostMallocDefault));
  ctx->host_allocs.push_back(p);
  *out = p;
  return ATTPC_OK;
}

int32_t attpc_host_free(attpc_ctx* ctx, void* ptr) {
  if (!ctx) return ATTPC_E_INVALID;
  if (!ptr) return ATTPC_OK;
  auto it = std::find(ctx->host_allocs.begin(), ctx->host_allocs.end(), ptr);
  if (it == ctx->host_allocs.end()) return fail(ctx, ATTPC_E_INVALID, "attpc_host_free: not an attpc_host_alloc pointer of this context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_c));
  HIP_TRY(ctx, hipHostFree(ptr));
  ctx->host_allocs.erase(it);
  return ATTPC_OK;
}

int32_t attpc_unpack_rows(const void* packed, int64_t n_rows, double* points, int64_t* labels, int32_t n_threads) {
  if (n_rows < 0 || (n_rows > 0 && (!packed || !points || !labels))) return ATTPC_E_INVALID;
  unpack_rows(static_cast<const PackedRow*>(packed), n_rows, points, labels, n_threads);
  return ATTPC_OK;
}

int32_t attpc_unpack_rows8(const void* packed, int64_t n_rows, const int64_t* offsets, int64_t n_events, uint64_t seed,
                           uint64_t first_event, double* points, int64_t* labels, int32_t n_threads) {
  if (n_rows < 0 || n_events < 0 || !offsets || (n_rows > 0 && (!packed || !points || !labels))) return ATTPC_E_INVALID;
  if (offsets[n_events] != offsets[0] + n_rows) return ATTPC_E_INVALID;
  unpack_rows8(static_cast<const unsigned long long*>(packed), n_rows, offsets, n_events, seed, first_event, points, labels, n_threads);
  return ATTPC_OK;
}

int32_t attpc_unpack_spyral_rows(const void* packed, int64_t n_rows, const double* pad_centers, const double* pad_sizes,
                                 int32_t n_pads, double r_max, int32_t windows_edge, int32_t micromegas_edge, double length,
                                 double* rows, int64_t* labels, int32_t n_threads) {
  if (n_rows < 0 || n_pads < 1 || (n_rows > 0 && (!packed || !pad_centers || !pad_sizes || !rows || !labels))) return ATTPC_E_INVALID;
  SpyralHostTables t;
  t.centers = pad_centers;
  t.sizes = pad_sizes;
  t.n_pads = n_pads;
  t.r_max = r_max;
  t.window_edge = (double)windows_edge;
  t.mm_edge = (double)micromegas_edge;
  t.length = length;
  unpack_spyral_rows(static_cast<const SpyralPacked*>(packed), n_rows, t, rows, labels, n_threads);
  return ATTPC_OK;
}

int32_t attpc_sync(attpc_ctx* ctx) {
  if (!ctx) return ATTPC_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return sync_all(ctx);
}

int32_t attpc_kin_configure(attpc_ctx* ctx, const attpc_kin_desc* d) {
  if (!ctx || !d) return ATTPC_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (d->n_steps < 1 || d->n_steps > ATTPC_MAX_STEPS) return fail(ctx, ATTPC_E_INVALID, "n_steps=%d", d->n_steps);
  if (d->sample_limit < 1) return fail(ctx, ATTPC_E_INVALID, "sample_limit=%d", d->sample_limit);
  { int32_t rc0 = drop_prefetch(ctx); if (rc0) return rc0; }
  { int32_t rc0 = sync_all(ctx); if (rc0) return rc0; }
  free_all(ctx->kin_allocs);
  ctx->kin_ready = false;
  ctx->rows_per_event = ctx->segs_per_event = ctx->blocks_per_track = 0.0;  // another workload: size estimates start over
  attpc_kin_desc k = *d;
  int32_t rc;
  for (int s = 0; s < d->n_steps; ++s) {
    attpc_excitation_desc& e = k.excitation[s];
    if (e.kind == ATTPC_EX_TABLE) {
      if (e.table_len < 2 || !e.table_x || !e.table_cdf) return fail(ctx, ATTPC_E_INVALID, "excitation %d: bad table", s);
      if ((rc = upload(ctx, ctx->kin_allocs, d->excitation[s].table_x, (size_t)e.table_len, &e.table_x))) return rc;
      if ((rc = upload(ctx, ctx->kin_allocs, d->excitation[s].table_cdf, (size_t)e.table_len, &e.table_cdf))) return rc;
    } else if (e.kind != ATTPC_EX_GAUSSIAN && e.kind != ATTPC_EX_UNIFORM) {
      return fail(ctx, ATTPC_E_INVALID, "excitation %d: unknown kind %d", s, e.kind);
    } else {
      e.table_x = e.table_cdf = nullptr;
    }
    attpc_polar_desc& p = k.polar[s];
    if (p.kind == ATTPC_POLAR_ARBITRARY) {
      if (p.table_len < 1 || !p.angles || !p.cdf) return fail(ctx, ATTPC_E_INVALID, "polar %d: bad table", s);
      if ((rc = upload(ctx, ctx->kin_allocs, d->polar[s].angles, (size_t)p.table_len, &p.angles))) return rc;
      if ((rc = upload(ctx, ctx->kin_allocs, d->polar[s].cdf, (size_t)p.table_len, &p.cdf))) return rc;
    } else if (p.kind != ATTPC_POLAR_UNIFORM) {
      return fail(ctx, ATTPC_E_INVALID, "polar %d: unknown kind %d", s, p.kind);
    } else {
      p.angles = p.cdf = nullptr;
    }
  }
  if (k.has_target) {
    if (k.eloss_len < 1 || !d->eloss) return fail(ctx, ATTPC_E_INVALID, "target without energy-loss table");
    if ((rc = upload(ctx, ctx->kin_allocs, d->eloss, (size_t)k.eloss_len, &k.eloss))) return rc;
  } else {
    k.eloss = nullptr;
    k.eloss_len = 0;
  }
  ctx->kin = k;
  ctx->kin_ready = true;
  return ATTPC_OK;
}

int32_t attpc_kin_run(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events, double* p4,
                      double* vertex, int32_t* status, uint32_t* attempts) {
  if (!ctx) return ATTPC_E_INVALID;
  if (!ctx->kin_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_kin_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int n_rows = 4 + 2 * (ctx->kin.n_steps - 1);
  // buffers of its own (ctx->scratch): the track sets are sized by the largest track batch met so far
  // (max_batch_events), and a kinematics-only call of 4 chunks per launch must not set that mark for every later
  // detector run; and nothing a failed earlier run may have left queued on any stream is overtaken
  int32_t rc = sync_all(ctx);
  if (rc) return rc;
  const uint64_t chunk = (uint64_t)std::max(1, ctx->chunk_events) * 4;
  const size_t cap = (size_t)std::min<uint64_t>(chunk, std::max<uint64_t>(n_events, 1));
  if ((rc = ensure(ctx, ctx->scratch[4], cap * n_rows * 4 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[5], cap * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[6], cap * sizeof(int32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[7], cap * sizeof(uint32_t)))) return rc;
  double* d_p4 = static_cast<double*>(ctx->scratch[4].p);
  double* d_vertex = static_cast<double*>(ctx->scratch[5].p);
  int32_t* d_status = static_cast<int32_t*>(ctx->scratch[6].p);
  uint32_t* d_attempts = static_cast<uint32_t*>(ctx->scratch[7].p);
  for (uint64_t done = 0; done < n_events; done += chunk) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, n_events - done);
    launch_kin_run(ctx->stream, ctx->kin, seed, first_event + done, n, d_p4, d_vertex, d_status, d_attempts);
    HIP_TRY(ctx, hipGetLastError());
    if (p4) HIP_TRY(ctx, hipMemcpyAsync(p4 + done * n_rows * 4, d_p4, (size_t)n * n_rows * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (vertex) HIP_TRY(ctx, hipMemcpyAsync(vertex + done * 3, d_vertex, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (status) HIP_TRY(ctx, hipMemcpyAsync(status + done, d_status, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (attempts) HIP_TRY(ctx, hipMemcpyAsync(attempts + done, d_attempts, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return ATTPC_OK;
}

int32_t attpc_kin_calculate(attpc_ctx* ctx, uint64_t n, const double* beam_energy, const double* ex,
                            const double* polar, const double* azim, double* p4, int32_t* status) {
  if (!ctx || !beam_energy || !ex || !polar || !azim || !p4 || !status) return ATTPC_E_INVALID;
  if (!ctx->kin_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_kin_configure has not been called");
  if (n == 0) return ATTPC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const int ns = ctx->kin.n_steps, n_rows = 4 + 2 * (ns - 1);
  int32_t rc;
  if ((rc = ensure(ctx, ctx->scratch[0], n * sizeof(double)))) return rc;
  for (int i = 1; i <= 3; ++i)
    if ((rc = ensure(ctx, ctx->scratch[i], n * ns * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[4], n * n_rows * 4 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[5], n * sizeof(int32_t)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0].p, beam_energy, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1].p, ex, n * ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[2].p, polar, n * ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[3].p, azim, n * ns * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  launch_kin_calculate(ctx->stream, ctx->kin, (uint32_t)n, static_cast<const double*>(ctx->scratch[0].p),
                     static_cast<const double*>(ctx->scratch[1].p), static_cast<const double*>(ctx->scratch[2].p),
                     static_cast<const double*>(ctx->scratch[3].p), static_cast<double*>(ctx->scratch[4].p),
                     static_cast<int32_t*>(ctx->scratch[5].p));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(p4, ctx->scratch[4].p, n * n_rows * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(status, ctx->scratch[5].p, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ATTPC_OK;
}

int32_t attpc_decay_calculate(attpc_ctx* ctx, uint64_t n, const double* parent, double mass_1, double mass_2,
                              const double* ex, const double* polar, const double* azim, double* out,
                              int32_t* status) {
  if (!ctx || !parent || !ex || !polar || !azim || !out || !status) return ATTPC_E_INVALID;
  if (n == 0) return ATTPC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc;
  if ((rc = ensure(ctx, ctx->scratch[0], n * 4 * sizeof(double)))) return rc;
  for (int i = 1; i <= 3; ++i)
    if ((rc = ensure(ctx, ctx->scratch[i], n * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[4], n * 8 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[5], n * sizeof(int32_t)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0].p, parent, n * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1].p, ex, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[2].p, polar, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[3].p, azim, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  launch_decay_calculate(ctx->stream, (uint32_t)n, static_cast<const double*>(ctx->scratch[0].p), mass_1, mass_2,
                     static_cast<const double*>(ctx->scratch[1].p), static_cast<const double*>(ctx->scratch[2].p),
                     static_cast<const double*>(ctx->scratch[3].p), static_cast<double*>(ctx->scratch[4].p),
                     static_cast<int32_t*>(ctx->scratch[5].p));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, ctx->scratch[4].p, n * 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(status, ctx->scratch[5].p, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ATTPC_OK;
}

int32_t attpc_det_configure(attpc_ctx* ctx, const attpc_det_desc* d) {
  if (!ctx || !d) return ATTPC_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (d->n_species < 1 || d->n_species > ATTPC_MAX_SPECIES) return fail(ctx, ATTPC_E_INVALID, "n_species=%d", d->n_species);
  if ((size_t)d->n_species * ATTPC_DEDX_NODES * sizeof(double) > 150 * 1024)
    return fail(ctx, ATTPC_E_INVALID, "stopping-power tables of %d species do not fit LDS (max 13)", d->n_species);
  if (!d->pad_lut || d->lut_n < 1) return fail(ctx, ATTPC_E_INVALID, "missing pad look-up table");
  if (d->lut_n > 32000) return fail(ctx, ATTPC_E_INVALID, "pad look-up table larger than 32000 x 32000 (indices are staged as 16 bit)");
  if (d->windows_edge <= d->micromegas_edge) return fail(ctx, ATTPC_E_INVALID, "windows_edge <= micromegas_edge");
  if (!(d->length > 0.0) || !(d->w_value > 0.0)) return fail(ctx, ATTPC_E_INVALID, "length and w_value must be > 0");
  {  // the scatter key packs the pad into 14 bits (tb << 14 | pad): ids outside [-1, 16383] would corrupt the
     // time bucket bits; the reference only treats -1 as "no pad" (transporter.py:162,237)
    const size_t cells = (size_t)d->lut_n * (size_t)d->lut_n;
    for (size_t i = 0; i < cells; ++i)
      if (d->pad_lut[i] < -1 || d->pad_lut[i] >= LONE_PADS)
        return fail(ctx, ATTPC_E_INVALID, "pad look-up table holds pad id %d at cell %zu: ids must be in [-1, %d]", (int)d->pad_lut[i], i, LONE_PADS - 1);
  }
  { int32_t rc0 = drop_prefetch(ctx); if (rc0) return rc0; }
  { int32_t rc0 = sync_all(ctx); if (rc0) return rc0; }
  free_all(ctx->det_allocs);
  ctx->det_ready = false;
  ctx->rows_per_event = ctx->segs_per_event = ctx->blocks_per_track = 0.0;  // size estimates start over
  ctx->prefer_big = false;
  ctx->prefer_wide = false;
  DetDev dv{};
  dv.length = d->length; dv.efield = d->efield; dv.bfield = d->bfield; dv.density = d->density;
  dv.diffusion = d->diffusion; dv.fano_factor = d->fano_factor; dv.w_value = d->w_value;
  dv.dv = d->length / (double)(d->windows_edge - d->micromegas_edge);  // parameters.py:172-174
  dv.inv_dv = 1.0 / dv.dv;
  dv.mm_edge = (double)d->micromegas_edge;
  dv.mpgd_gain = d->mpgd_gain;
  dv.lut_n = d->lut_n; dv.lut_lo = d->lut_lo;
  dv.n_species = d->n_species; dv.ode_substeps = d->ode_substeps > 0 ? d->ode_substeps : 1;
  dv.longitudinal_diffusion = d->longitudinal_diffusion > 0.0 ? d->longitudinal_diffusion : 0.0;
  for (int s = 0; s < ATTPC_LONG_STEPS; ++s) dv.long_weights[s] = d->long_weights[s];
  dv.mc_diffusion = d->mc_diffusion != 0 ? 1 : 0;
  dv.mpgd_gain32 = (int32_t)d->mpgd_gain;
  if (!(d->path_step >= 0.0) || !(d->path_step < 1.0e300)) return fail(ctx, ATTPC_E_INVALID, "path_step must be >= 0 and finite");
  dv.path_step = d->path_step;
  if (dv.mc_diffusion && (d->mpgd_gain < 1 || d->mpgd_gain > 0x7fffffff)) return fail(ctx, ATTPC_E_INVALID, "mc_diffusion needs 1 <= mpgd_gain < 2^31");
  int32_t rc;
  {  // device copy: [x][y] as given, padded with one extra row and column of -1 (index lut_n = "off
     // the pad plane").  The scatter kernel's lanes are mesh lines of constant y that step through x
     // together, so one gather instruction reads neighbouring y of the same x row -- 1-2 cache lines
     // per sample instead of one per lane.
    const size_t n = (size_t)d->lut_n, pitch = n + 1;
    std::vector<int16_t> lut_t(pitch * pitch, (int16_t)-1);
    for (size_t ix = 0; ix < n; ++ix)
      for (size_t iy = 0; iy < n; ++iy) lut_t[ix * pitch + iy] = d->pad_lut[ix * n + iy];
    if ((rc = upload(ctx, ctx->det_allocs, lut_t.data(), pitch * pitch, &dv.pad_lut))) return rc;
  }
  std::vector<double> tabs((size_t)d->n_species * ATTPC_DEDX_NODES);
  for (int s = 0; s < d->n_species; ++s) {
    if (!d->species[s].dedx) return fail(ctx, ATTPC_E_INVALID, "species %d: missing dE/dx table", s);
    if (!(d->species[s].mass > 0.0)) return fail(ctx, ATTPC_E_INVALID, "species %d: mass must be > 0", s);
    std::memcpy(tabs.data() + (size_t)s * ATTPC_DEDX_NODES, d->species[s].dedx, ATTPC_DEDX_NODES * sizeof(double));
    dv.mass[s] = d->species[s].mass;
    dv.Z[s] = d->species[s].Z;
  }
  if ((rc = upload(ctx, ctx->det_allocs, tabs.data(), tabs.size(), &dv.dedx))) return rc;
  ctx->det = dv;
  ctx->det_ready = true;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream_t_own));  // (sync_all above covered stream_t, whichever it was)
  pick_track_stream(ctx);
  return ATTPC_OK;
}


int32_t attpc_det_run(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                      const attpc_event_layout* layout, const double* p4, const double* vertex,
                      attpc_cloud_out* out, attpc_run_stats* stats) {
  if (!ctx || !p4 || !vertex) return ATTPC_E_INVALID;
  if (!ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_det_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc = validate_layout(ctx, layout, true);
  if (rc) return rc;
  RunSource src;
  src.h_p4 = p4;
  src.h_vertex = vertex;
  return run_events(ctx, seed, first_event, n_events, *layout, src, RunSink{}, out, false, stats);
}

int32_t attpc_det_run_spyral(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                             const attpc_event_layout* layout, const double* p4, const double* vertex,
                             attpc_cloud_out* out, attpc_run_stats* stats) {
  if (!ctx || !p4 || !vertex) return ATTPC_E_INVALID;
  if (!out) return fail(ctx, ATTPC_E_INVALID, "attpc_det_run_spyral needs output buffers");
  if (!ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_det_configure has not been called");
  if (!ctx->spyral_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_spyral_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc = validate_layout(ctx, layout, true);
  if (rc) return rc;
  RunSource src;
  src.h_p4 = p4;
  src.h_vertex = vertex;
  return run_events(ctx, seed, first_event, n_events, *layout, src, RunSink{}, out, true, stats);
}

static int32_t sim_run_impl(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                            const attpc_event_layout* layout, double* p4, double* vertex, int32_t* kin_status,
                            attpc_cloud_out* out, attpc_run_stats* stats, bool spyral) {
  if (!ctx) return ATTPC_E_INVALID;
  if (spyral && !ctx->spyral_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_spyral_configure has not been called");
  if (!ctx->kin_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_kin_configure has not been called");
  if (!ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_det_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc = validate_layout(ctx, layout, true);
  if (rc) return rc;
  const int n_rows = 4 + 2 * (ctx->kin.n_steps - 1);
  if (layout->n_rows != n_rows) return fail(ctx, ATTPC_E_INVALID, "layout.n_rows=%d but the pipeline has %d rows", layout->n_rows, n_rows);
  RunSource src;
  src.from_kernel = true;
  RunSink sink;
  sink.p4 = p4;
  sink.vertex = vertex;
  sink.status = kin_status;
  return run_events(ctx, seed, first_event, n_events, *layout, src, sink, out, spyral, stats);
}

int32_t attpc_sim_run(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                      const attpc_event_layout* layout, double* p4, double* vertex, int32_t* kin_status,
                      attpc_cloud_out* out, attpc_run_stats* stats) {
  return sim_run_impl(ctx, seed, first_event, n_events, layout, p4, vertex, kin_status, out, stats, false);
}

int32_t attpc_sim_run_spyral(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                             const attpc_event_layout* layout, double* p4, double* vertex, int32_t* kin_status,
                             attpc_cloud_out* out, attpc_run_stats* stats) {
  if (!out) return fail(ctx, ATTPC_E_INVALID, "attpc_sim_run_spyral needs output buffers");
  return sim_run_impl(ctx, seed, first_event, n_events, layout, p4, vertex, kin_status, out, stats, true);
}

int32_t attpc_sim_hint_next(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                            const attpc_event_layout* layout) {
  if (!ctx) return ATTPC_E_INVALID;
  ctx->hint_valid = false;
  if (!layout || n_events == 0) return ATTPC_OK;  // "nothing known about the next call"
  if (!ctx->kin_ready || !ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_sim_hint_next before the configure calls");
  int32_t rc = validate_layout(ctx, layout, true);
  if (rc) return rc;
  if (layout->n_rows != 4 + 2 * (ctx->kin.n_steps - 1)) return fail(ctx, ATTPC_E_INVALID, "layout.n_rows does not match the pipeline");
  ctx->hint_valid = true;
  ctx->hint_seed = seed;
  ctx->hint_first = first_event;
  ctx->hint_n = n_events;
  ctx->hint_lay = *layout;
  return ATTPC_OK;
}

int32_t attpc_spyral_configure(attpc_ctx* ctx, const attpc_spyral_desc* d) {
  if (!ctx || !d || !d->response || !d->pad_centers || !d->pad_sizes || d->n_pads < 1) return ATTPC_E_INVALID;
  if (d->windows_edge <= d->micromegas_edge) return fail(ctx, ATTPC_E_INVALID, "windows_edge <= micromegas_edge");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  { int32_t rc0 = sync_all(ctx); if (rc0) return rc0; }
  free_all(ctx->spyral_allocs);
  ctx->spyral_ready = false;
  SpyralDev sp{};
  std::vector<double> sorted(d->response, d->response + ATTPC_NUM_TB);
  std::sort(sorted.begin(), sorted.end(), [](double a, double b) { return a > b; });
  std::vector<double> prefix(ATTPC_NUM_TB + 1, 0.0);
  for (int i = 0; i < ATTPC_NUM_TB; ++i) prefix[i + 1] = prefix[i] + sorted[i];
  int32_t rc;
  if ((rc = upload(ctx, ctx->spyral_allocs, d->response, (size_t)ATTPC_NUM_TB, &sp.response))) return rc;
  if ((rc = upload(ctx, ctx->spyral_allocs, sorted.data(), sorted.size(), &sp.sorted_desc))) return rc;
  if ((rc = upload(ctx, ctx->spyral_allocs, prefix.data(), prefix.size(), &sp.prefix))) return rc;
  if ((rc = upload(ctx, ctx->spyral_allocs, d->pad_centers, (size_t)d->n_pads * 2, &sp.pad_centers))) return rc;
  if ((rc = upload(ctx, ctx->spyral_allocs, d->pad_sizes, (size_t)d->n_pads, &sp.pad_sizes))) return rc;
  ctx->h_pad_centers.assign(d->pad_centers, d->pad_centers + (size_t)d->n_pads * 2);
  ctx->h_pad_sizes.assign(d->pad_sizes, d->pad_sizes + (size_t)d->n_pads);
  sp.n_pads = d->n_pads;
  sp.r_max = sorted[0];
  sp.total = prefix[ATTPC_NUM_TB];
  sp.window_edge = (double)d->windows_edge;
  sp.mm_edge = (double)d->micromegas_edge;
  sp.length = d->length;
  sp.threshold = d->adc_threshold;
  ctx->spyral = sp;
  ctx->spyral_ready = true;
  return ATTPC_OK;
}

int32_t attpc_det_tracks(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                         const attpc_event_layout* layout, const double* p4, const double* vertex,
                         int64_t max_samples_per_track, double* samples, int32_t* counts, int32_t* n_steps) {
  if (!ctx || !p4 || !vertex || !counts || !n_steps) return ATTPC_E_INVALID;
  if (!ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_det_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc = validate_layout(ctx, layout, true);
  if (rc) return rc;
  if (n_events > (uint64_t)ctx->chunk_events) return fail(ctx, ATTPC_E_INVALID, "attpc_det_tracks handles at most one chunk");
  if ((rc = drop_prefetch(ctx))) return rc;
  const uint32_t n = (uint32_t)n_events;
  TrackSet& ts = ctx->tset[0];
  TrackLaunch tl;
  RunSource src;
  src.h_p4 = p4;
  src.h_vertex = vertex;
  if ((rc = queue_batch(ctx, ts, tl, *layout, src, seed, first_event, 0, n, layout->n_rows))) return rc;
  TrackBuffers trk;
  double ms = 0;
  if ((rc = finish_tracks(ctx, ts, tl, &trk, &ms, nullptr))) return rc;
  const uint32_t n_tracks = n * (uint32_t)layout->n_sim;
  if (n_tracks == 0) return ATTPC_OK;
  std::vector<int32_t> table((size_t)n_tracks * MAX_BLOCKS_PER_TRACK);
  HIP_TRY(ctx, hipMemcpy(table.data(), ts.block_table.p, table.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(counts, ts.counts.p, (size_t)n_tracks * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(n_steps, ts.n_steps.p, (size_t)n_tracks * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (samples) {
    // h_ctrl[1] counts reserved blocks (waves reserve pools), all below arena_blocks after a good run
    std::vector<double> arena(std::min<size_t>(ts.h_ctrl[1], ts.arena_blocks) * ARENA_BLK * 4);
    if (!arena.empty()) HIP_TRY(ctx, hipMemcpy(arena.data(), ts.arena.p, arena.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (uint32_t t = 0; t < n_tracks; ++t) {
      const int64_t c = std::min<int64_t>(counts[t], max_samples_per_track);
      for (int64_t s = 0; s < c; ++s) {
        const int32_t blk = table[(size_t)t * MAX_BLOCKS_PER_TRACK + s / ARENA_BLK];
        std::memcpy(samples + ((size_t)t * max_samples_per_track + s) * 4,
                    arena.data() + ((size_t)blk * ARENA_BLK + (s % ARENA_BLK)) * 4, 4 * sizeof(double));
      }
    }
  }
  return ATTPC_OK;
}

int32_t attpc_det_scatter(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                          const attpc_event_layout* layout, const double* samples, const int32_t* counts,
                          attpc_cloud_out* out, attpc_run_stats* stats) {
  if (!ctx || !counts || !layout) return ATTPC_E_INVALID;
  if (!ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_det_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc = validate_layout(ctx, layout, false);
  if (rc) return rc;
  if (n_events > (uint64_t)ctx->chunk_events) return fail(ctx, ATTPC_E_INVALID, "attpc_det_scatter handles at most one chunk");
  if ((rc = drop_prefetch(ctx))) return rc;
  const uint32_t n = (uint32_t)n_events;
  const uint32_t n_tracks = n * (uint32_t)layout->n_sim;
  // pack the samples into arena blocks, one chain of consecutive blocks per track
  std::vector<int32_t> table((size_t)n_tracks * MAX_BLOCKS_PER_TRACK, 0);
  size_t total_samples = 0, n_blocks = 0;
  for (uint32_t t = 0; t < n_tracks; ++t) {
    if (counts[t] < 0 || counts[t] > MAX_BLOCKS_PER_TRACK * ARENA_BLK)
      return fail(ctx, ATTPC_E_INVALID, "counts[%u]=%d: a track holds 0..%d samples", t, counts[t], MAX_BLOCKS_PER_TRACK * ARENA_BLK);
    total_samples += (size_t)counts[t];
    n_blocks += ((size_t)counts[t] + ARENA_BLK - 1) / ARENA_BLK;
  }
  if (total_samples && !samples) return ATTPC_E_INVALID;
  std::vector<double> arena(std::max<size_t>(n_blocks, 1) * ARENA_BLK * 4, 0.0);
  size_t blk = 0, src_row = 0;
  for (uint32_t t = 0; t < n_tracks; ++t) {
    const size_t nb = ((size_t)counts[t] + ARENA_BLK - 1) / ARENA_BLK;
    for (size_t b = 0; b < nb; ++b) table[(size_t)t * MAX_BLOCKS_PER_TRACK + b] = (int32_t)(blk + b);
    if (counts[t]) std::memcpy(arena.data() + blk * ARENA_BLK * 4, samples + src_row * 4, (size_t)counts[t] * 4 * sizeof(double));
    blk += nb;
    src_row += (size_t)counts[t];
  }
  if ((rc = sync_all(ctx))) return rc;
  TrackSet& ts = ctx->tset[0];
  if ((rc = ensure(ctx, ts.arena, arena.size() * sizeof(double)))) return rc;
  ts.arena_blocks = std::max(ts.arena_blocks, arena.size() / ((size_t)ARENA_BLK * 4));
  if ((rc = ensure(ctx, ts.block_table, std::max<size_t>(table.size(), 1) * sizeof(int32_t)))) return rc;
  if ((rc = ensure(ctx, ts.counts, std::max<size_t>(n_tracks, 1) * sizeof(int32_t)))) return rc;
  HIP_TRY(ctx, hipMemcpy(ts.arena.p, arena.data(), arena.size() * sizeof(double), hipMemcpyHostToDevice));
  if (n_tracks) {
    HIP_TRY(ctx, hipMemcpy(ts.block_table.p, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(ts.counts.p, counts, (size_t)n_tracks * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  TrackBuffers trk{};
  trk.arena = static_cast<double*>(ts.arena.p);
  trk.block_table = static_cast<int32_t*>(ts.block_table.p);
  trk.counts = static_cast<int32_t*>(ts.counts.p);
  trk.arena_blocks = (uint32_t)ts.arena_blocks;
  UnpackDrain drain(ctx);
  attpc_run_stats st{};
  st.n_events = n_events;
  int64_t row_cursor = 0;
  bool over = false;
  if (out && out->offsets) out->offsets[0] = 0;
  const double keep_rows = ctx->rows_per_event, keep_segs = ctx->segs_per_event;
  ctx->rows_per_event = ctx->segs_per_event = 0.0;  // explicit samples say nothing about the configured workload
  rc = run_batch_chunks(ctx, *layout, trk, seed, first_event, 0, n, out, false, &st, &row_cursor, &over, []() -> int32_t { return ATTPC_OK; });
  ctx->rows_per_event = keep_rows;
  ctx->segs_per_event = keep_segs;
  if (rc) return rc;
  if ((rc = sync_all(ctx))) return rc;
  if (stats) *stats = st;
  if (over) return fail(ctx, ATTPC_E_CAPACITY, "cloud needs %lld rows, capacity %lld", (long long)row_cursor, (long long)out->capacity);
  if (st.n_failed || st.n_inconsistent)
    return fail(ctx, ATTPC_E_DATALOSS, "%llu events lost a time bucket (n_failed), %u table self-check failures (n_inconsistent)",
                (unsigned long long)st.n_failed, st.n_inconsistent);
  return ATTPC_OK;
}

}  // extern "C"

// ---- response + Spyral rows ("next" row 1) ----
namespace attpc {
// detector/response.py:35-57 (clip each of the 512 samples at 4095, max and sum) and
// detector/writer.py:61-112 (row layout).  One lane = one point.
__global__ __launch_bounds__(256) void spyral_rows_kernel(int64_t n, const double* __restrict__ points,
                                                          const double* __restrict__ response,
                                                          const double* __restrict__ centers,
                                                          const double* __restrict__ sizes, int32_t n_pads,
                                                          double window_edge, double mm_edge, double length,
                                                          double* __restrict__ rows) {
  __shared__ double resp[ATTPC_NUM_TB];
  for (int i = threadIdx.x; i < ATTPC_NUM_TB; i += 256) resp[i] = response[i];
  block_sync();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double padf = points[3 * i], tb = points[3 * i + 1], q = points[3 * i + 2];
  int pad = (int)padf;
  pad = pad < 0 ? 0 : (pad >= n_pads ? n_pads - 1 : pad);
  double amp = -1.0 / 0.0, integral = 0.0;
  for (int k = 0; k < ATTPC_NUM_TB; ++k) {
    double v = resp[k] * q;
    v = v > 4095.0 ? 4095.0 : v;
    amp = v > amp ? v : amp;
    integral += v;
  }
  double* r = rows + 8 * i;
  r[0] = centers[2 * pad];
  r[1] = centers[2 * pad + 1];
  r[2] = (window_edge - tb) / (window_edge - mm_edge) * length * 1000.0;
  r[3] = amp;
  r[4] = integral;
  r[5] = padf;
  r[6] = tb;
  r[7] = sizes[pad];
}
}  // namespace attpc

extern "C" int32_t attpc_spyral_rows(attpc_ctx* ctx, int64_t n_points, const double* points, const double* response,
                                     const double* pad_centers, const double* pad_sizes, int32_t n_pads,
                                     int32_t windows_edge, int32_t micromegas_edge, double length, double* rows) {
  if (!ctx || !points || !response || !pad_centers || !pad_sizes || !rows || n_pads < 1) return ATTPC_E_INVALID;
  if (n_points <= 0) return ATTPC_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc;
  const size_t n = (size_t)n_points;
  if ((rc = ensure(ctx, ctx->scratch[0], n * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[1], ATTPC_NUM_TB * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[2], (size_t)n_pads * 2 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[3], (size_t)n_pads * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[4], n * 8 * sizeof(double)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[0].p, points, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[1].p, response, ATTPC_NUM_TB * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[2].p, pad_centers, (size_t)n_pads * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[3].p, pad_sizes, (size_t)n_pads * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(attpc::spyral_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     n_points, static_cast<const double*>(ctx->scratch[0].p),
                     static_cast<const double*>(ctx->scratch[1].p), static_cast<const double*>(ctx->scratch[2].p),
                     static_cast<const double*>(ctx->scratch[3].p), n_pads, (double)windows_edge,
                     (double)micromegas_edge, length, static_cast<double*>(ctx->scratch[4].p));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(rows, ctx->scratch[4].p, n * 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ATTPC_OK;
}
