// abi.hip -- host side of the C ABI (include/attpc_engine.h): context, configuration upload,
// chunked launch sequence kinematics -> tracks -> scatter on one HIP stream, output assembly.
//
// HBM layout per chunk of C events (N rows/event, S simulated nuclei/event, T = C*S tracks):
//   p4 f64[C][N][4], vertex f64[C][3], status i32[C], attempts u32[C]        kinematics
//   arena f64[blocks][128][4]  (x, y, time bucket, electrons)                 track samples
//   block_table i32[T][79], counts i32[T], n_steps i32[T]                     track index
//   points f64[cap][3], labels i64[cap], segments {event,count,offset}[..]    point cloud
// Buffers grow on demand and are reused by every chunk (device-resident mode overwrites them).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tracks_args.hpp"

namespace {

using namespace attpc;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct attpc_ctx {
  int device = 0;
  int n_cus = 256;                 // compute units (one scatter workgroup each)
  hipStream_t stream = nullptr;
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  std::string error;
  int32_t chunk_events = 65536;

  bool kin_ready = false;
  attpc_kin_desc kin{};            // device pointers inside
  std::vector<void*> kin_allocs;

  bool det_ready = false;
  DetDev det{};
  std::vector<void*> det_allocs;

  // chunk buffers
  DevBuf p4, vertex, status, attempts;
  DevBuf arena, block_table, counts, n_steps, trk_ctrl;
  DevBuf points, labels, segments, out_ctrl, asm_labels;

  bool spyral_ready = false;
  SpyralDev spyral{};
  std::vector<void*> spyral_allocs;
  DevBuf sp_rows, sp_labels, sp_event_start, sp_kept, sp_kept_start;
  DevBuf scratch[8];
  size_t arena_blocks = 0;
  int64_t cloud_capacity = 0, seg_capacity = 0;
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

int32_t ensure(attpc_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.bytes) return ATTPC_OK;
  if (b.p) HIP_TRY(ctx, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  HIP_TRY(ctx, hipMalloc(&b.p, bytes));
  b.bytes = bytes;
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

int32_t validate_layout(attpc_ctx* ctx, const attpc_event_layout* lay) {
  if (!lay || lay->n_rows < 1 || lay->n_rows > ATTPC_MAX_ROWS || lay->n_sim < 0 || lay->n_sim > ATTPC_MAX_SIM)
    return fail(ctx, ATTPC_E_INVALID, "bad event layout");
  for (int i = 0; i < lay->n_sim; ++i) {
    const int row = lay->indices[i];
    if (row < 0 || row >= lay->n_rows) return fail(ctx, ATTPC_E_INVALID, "indices[%d]=%d out of range", i, row);
    const int sp = lay->species_of_row[row];
    if (sp >= ctx->det.n_species) return fail(ctx, ATTPC_E_INVALID, "species_of_row[%d]=%d out of range", row, sp);
  }
  return ATTPC_OK;
}

struct ChunkResult {
  unsigned long long rows = 0, segs = 0, charge = 0, keys = 0, failed = 0, retried = 0, samples = 0, mismatch = 0;
  float ms_tracks = 0, ms_scatter = 0;
};

// Track integration for a BATCH of `n` events whose kinematics already sit in ctx->p4 / vertex
// (/status).  A batch spans several scatter chunks: the track kernel hands tracks to lanes
// dynamically, and with fewer tracks than a few times the 200 k lanes of the chip the launch is one
// generation of tracks whose length is set by its longest member.
int32_t run_tracks(attpc_ctx* ctx, const attpc_event_layout& lay, uint64_t seed, uint64_t first_event, uint32_t n,
                   bool use_status, TrackBuffers* out_buf, double* ms) {
  const uint32_t n_tracks = n * (uint32_t)lay.n_sim;
  *out_buf = TrackBuffers{};
  if (n_tracks == 0) return ATTPC_OK;
  int32_t rc;
  if ((rc = ensure(ctx, ctx->block_table, (size_t)n_tracks * MAX_BLOCKS_PER_TRACK * sizeof(int32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->counts, (size_t)n_tracks * sizeof(int32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->n_steps, (size_t)n_tracks * sizeof(int32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->trk_ctrl, 16 * sizeof(uint32_t)))) return rc;
  const size_t lds = (size_t)ctx->det.n_species * ATTPC_DEDX_NODES * sizeof(double);
  const uint32_t waves_needed = (n_tracks + 63) / 64;
  const uint32_t blocks = std::min<uint32_t>((waves_needed + 3) / 4, (uint32_t)ctx->n_cus * 8u);
  // every wave reserves arena blocks 64 at a time: that slack comes on top of what the samples need
  size_t want_blocks = std::max<size_t>(ctx->arena_blocks, (size_t)n_tracks * 3 + (size_t)blocks * 4 * 64 + 1024);
  if (std::getenv("ATTPC_TEST_TINY_BUFFERS") && ctx->arena_blocks == 0) want_blocks = 4;  // test hook: grow-and-rerun path

  for (int attempt = 0; attempt < 8; ++attempt) {
    if ((rc = ensure(ctx, ctx->arena, want_blocks * ARENA_BLK * 4 * sizeof(double)))) return rc;
    ctx->arena_blocks = want_blocks;
    HIP_TRY(ctx, hipMemsetAsync(ctx->trk_ctrl.p, 0, 16 * sizeof(uint32_t), ctx->stream));
    TrackArgs ta;
    ta.det = ctx->det;
    ta.layout = lay;
    ta.buf.arena = static_cast<double*>(ctx->arena.p);
    ta.buf.block_table = static_cast<int32_t*>(ctx->block_table.p);
    ta.buf.counts = static_cast<int32_t*>(ctx->counts.p);
    ta.buf.n_steps = static_cast<int32_t*>(ctx->n_steps.p);
    ta.buf.ctrl = static_cast<uint32_t*>(ctx->trk_ctrl.p);
    ta.buf.arena_blocks = (uint32_t)std::min<size_t>(want_blocks, 0xFFFFFFFFu);
    ta.p4 = static_cast<const double*>(ctx->p4.p);
    ta.vertex = static_cast<const double*>(ctx->vertex.p);
    ta.kin_status = use_status ? static_cast<const int32_t*>(ctx->status.p) : nullptr;
    ta.seed = seed;
    ta.first_event = first_event;
    ta.n_events = n;
    ta.n_tracks = n_tracks;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
    launch_track_kernel(blocks, lds, ctx->stream, ta);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    uint32_t tctrl[4];
    HIP_TRY(ctx, hipMemcpyAsync(tctrl, ctx->trk_ctrl.p, sizeof tctrl, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms_t = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms_t, ctx->ev[0], ctx->ev[1]));
    *ms += ms_t;  // timings of discarded attempts stay counted: they were spent
    if (tctrl[2] == 0) {  // no sample was refused
      *out_buf = ta.buf;
      return ATTPC_OK;
    }
    want_blocks = std::max<size_t>((size_t)tctrl[1] + 1024, want_blocks * 2);  // arena exhausted
  }
  return fail(ctx, ATTPC_E_HIP, "track arena did not fit after repeated growth");
}

// Scatter for the `n` events starting at event `e0` of the current track batch (global id
// `first_event` = batch first + e0).
int32_t run_scatter(attpc_ctx* ctx, const attpc_event_layout& lay, const TrackBuffers& trk, uint64_t seed,
                    uint64_t first_event, uint32_t e0, uint32_t n, ChunkResult* res) {
  if (n == 0 || lay.n_sim == 0) {
    *res = ChunkResult{};
    return ATTPC_OK;
  }
  int32_t rc;
  if ((rc = ensure(ctx, ctx->out_ctrl, 32 * sizeof(unsigned long long)))) return rc;
  // Kernel variant: "small" (two 512-thread workgroups with 4096-slot tables per CU) is ~6 % faster
  // for detectors with the usual diffusion; "big" (one 1024-thread workgroup, 8192 slots) holds twice
  // as many keys per time bucket.  Small is used when a sample is expected to touch at most 40 pads at
  // the far end of the drift (default detector: 28; the same estimate as key_estimate() in scatter.hip)
  // and no extension is on; if a launch of the small variant meets a time bucket that does not fit
  // (n_failed), the chunk is simply run again with the big one (results are deterministic).
  const double spread = (6.0 / 4.9e-3) * (6.0 / 4.9e-3) * 2.0 * ctx->det.diffusion * ctx->det.dv / ctx->det.efield;
  const double far_keys = (1.0 + std::sqrt(spread * (ATTPC_NUM_TB - 1))) * (1.0 + std::sqrt(spread * (ATTPC_NUM_TB - 1)));
  bool use_small = far_keys <= 40.0 && !ctx->det.mc_diffusion && !(ctx->det.longitudinal_diffusion > 0.0);
  if (const char* force = std::getenv("ATTPC_SC_VARIANT")) use_small = std::string(force) == "small";  // tests
  // launch geometry: persistent workgroups that take `batch` events per visit to the event counter
  // and reserve output rows `row_block` at a time (small launches: exact reservations, so that short
  // runs waste no rows)
#ifndef ATTPC_SC_SMALL_WGS
#define ATTPC_SC_SMALL_WGS 2  // workgroups per CU of the small variant (scatter_small.hip)
#endif
  uint32_t sc_wgs = std::min<uint32_t>((uint32_t)ctx->n_cus * (use_small ? (uint32_t)ATTPC_SC_SMALL_WGS : 1u), n);
  uint32_t sc_batch = n / sc_wgs >= 64u ? 2u : 1u;  // the request for the next batch is hidden (scatter.hip)
  const int64_t est_rows = (int64_t)n * 9216;
  uint32_t sc_row_block = est_rows / ((int64_t)sc_wgs * 16) >= 16384
                              ? (uint32_t)std::min<int64_t>(est_rows / ((int64_t)sc_wgs * 16), 1 << 18) : 1u;
  const int64_t hole_rows = sc_row_block > 1u ? (int64_t)sc_wgs * sc_row_block + est_rows / 16 : 0;
  int64_t want_rows = std::max<int64_t>(ctx->cloud_capacity, est_rows + hole_rows + 65536);
  int64_t want_segs = std::max<int64_t>(ctx->seg_capacity, (int64_t)n * 6 + 4096 + (int64_t)sc_wgs * 16);
  if (std::getenv("ATTPC_TEST_TINY_BUFFERS") && ctx->cloud_capacity == 0) {
    want_rows = 64;  // test hook: start with buffers that are certainly too small
    want_segs = 2;
  }

  for (int attempt = 0; attempt < 9; ++attempt) {
    if ((rc = ensure(ctx, ctx->points, (size_t)want_rows * 3 * sizeof(double)))) return rc;
    if ((rc = ensure(ctx, ctx->labels, (size_t)want_rows * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->segments, (size_t)want_segs * sizeof(Segment)))) return rc;
    ctx->cloud_capacity = want_rows;
    ctx->seg_capacity = want_segs;
    HIP_TRY(ctx, hipMemsetAsync(ctx->out_ctrl.p, 0, 32 * sizeof(unsigned long long), ctx->stream));
    ScatterArgs sa;
    sa.det = ctx->det;
    sa.layout = lay;
    sa.trk = trk;
    sa.out.points = static_cast<double*>(ctx->points.p);
    sa.out.labels = static_cast<int64_t*>(ctx->labels.p);
    sa.out.segments = static_cast<Segment*>(ctx->segments.p);
    sa.out.ctrl = static_cast<unsigned long long*>(ctx->out_ctrl.p);
    sa.out.capacity = want_rows;
    sa.out.seg_capacity = want_segs;
    sa.seed = seed;
    sa.first_event = first_event;
    sa.n_events = n;
    sa.event0 = e0;
    sa.batch = sc_batch;
    sa.row_block = sc_row_block;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    if (use_small) launch_scatter_kernel_small(sc_wgs, ctx->stream, sa);
    else launch_scatter_kernel_big(sc_wgs, ctx->stream, sa);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    unsigned long long octrl[32];
    HIP_TRY(ctx, hipMemcpyAsync(octrl, ctx->out_ctrl.p, sizeof octrl, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    float ms_s = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms_s, ctx->ev[2], ctx->ev[3]));
    res->ms_scatter += ms_s;
#ifdef ATTPC_PHASE_TIMERS
    fprintf(stderr, "[attpc phase cycles] init %llu hist %llu select %llu stage %llu items %llu overflow %llu flushcount %llu flushwrite %llu (events %u)\n",
            octrl[8], octrl[9], octrl[10], octrl[11], octrl[12], octrl[13], octrl[14], octrl[15], n);
    fprintf(stderr, "[attpc rows-phase cycles] gathers %llu runs %llu scan+queue %llu drain %llu\n", octrl[16], octrl[17], octrl[18], octrl[19]);
    fprintf(stderr, "[attpc rounds] rows-rounds %llu staged %llu busiest-wave passes %llu\n", octrl[20], octrl[21], octrl[22]);
    fprintf(stderr, "[attpc flush cycles] to-barrier %llu to-compacted %llu atomics-wait %llu segment %llu select %llu barrier %llu\n", octrl[27], octrl[23], octrl[24], octrl[25], octrl[26], octrl[14]);
    fprintf(stderr, "[attpc ctrl] rows %llu segments %llu failed %llu retried %llu samples %llu\n", octrl[0], octrl[1],
            octrl[4], octrl[5], octrl[7]);
#endif
    if (use_small && octrl[4] != 0) {  // a time bucket with more keys than the small table: run the chunk with the big one
      use_small = false;
      sc_wgs = std::min<uint32_t>((uint32_t)ctx->n_cus, n);
      sc_batch = n / sc_wgs >= 64u ? 2u : 1u;
      sc_row_block = est_rows / ((int64_t)sc_wgs * 16) >= 16384 ? (uint32_t)std::min<int64_t>(est_rows / ((int64_t)sc_wgs * 16), 1 << 18) : 1u;
      continue;
    }
    if (octrl[6] == 0) {
      res->rows = octrl[30];  // rows written; octrl[0] is the reservation cursor (holes included)
      res->segs = octrl[1];
      res->charge = octrl[2];
      res->keys = octrl[3];
      res->failed = octrl[4];
      res->retried = octrl[5];
      res->samples = octrl[7];
      res->mismatch = octrl[31];
      return ATTPC_OK;
    }
    // cloud / segment capacity exceeded (the cursors kept counting)
    want_rows = std::max<int64_t>(want_rows, (int64_t)(octrl[0] + octrl[0] / 8) + 65536);
    want_segs = std::max<int64_t>(want_segs, (int64_t)(octrl[1] + octrl[1] / 8) + 4096);
  }
  return fail(ctx, ATTPC_E_HIP, "point cloud did not fit after repeated buffer growth");
}

// events per track batch: several scatter chunks, bounded so that the sample arena stays below ~24 GB
uint64_t track_batch_events(const attpc_ctx* ctx, const attpc_event_layout& lay) {
  const uint64_t chunk = (uint64_t)std::max(1, ctx->chunk_events);
  const uint64_t by_memory = (24ull << 30) / ((uint64_t)std::max(1, lay.n_sim) * 3ull * ARENA_BLK * 4 * sizeof(double));
  const uint64_t chunks = std::max<uint64_t>(1, std::min<uint64_t>(8, by_memory / chunk));
  return chunk * chunks;
}

// Device-side CSR assembly: segment s (one flushed window of one event) is copied to row
// dst[s] of the event-ordered arrays, so the host receives one contiguous block per chunk.
__global__ __launch_bounds__(256) void gather_segments_kernel(const Segment* __restrict__ segs,
                                                              const int64_t* __restrict__ dst, uint32_t n_segs,
                                                              const double* __restrict__ points,
                                                              const int64_t* __restrict__ labels,
                                                              double* __restrict__ out_points,
                                                              int64_t* __restrict__ out_labels) {
  for (uint32_t s = blockIdx.x; s < n_segs; s += gridDim.x) {
    const Segment sg = segs[s];
    const double* src_p = points + sg.offset * 3;
    double* dst_p = out_points + dst[s] * 3;
    for (int i = threadIdx.x; i < sg.count * 3; i += 256) dst_p[i] = src_p[i];
    const int64_t* src_l = labels + sg.offset;
    int64_t* dst_l = out_labels + dst[s];
    for (int i = threadIdx.x; i < sg.count; i += 256) dst_l[i] = src_l[i];
  }
}

// Event-ordered (CSR) copy of one chunk's cloud on the device: scratch[7] = points, asm_labels =
// labels; `start` receives the chunk-local row offset of every event (n + 1 entries).
int32_t gather_chunk_csr(attpc_ctx* ctx, const ChunkResult& r, uint32_t n, std::vector<int64_t>* start, bool run_gather) {
  std::vector<Segment> segs(r.segs);
  if (r.segs) HIP_TRY(ctx, hipMemcpy(segs.data(), ctx->segments.p, r.segs * sizeof(Segment), hipMemcpyDeviceToHost));
  std::vector<int64_t> counts(n, 0);
  for (const Segment& s : segs) counts[s.event] += s.count;
  start->assign(n + 1, 0);
  for (uint32_t i = 0; i < n; ++i) (*start)[i + 1] = (*start)[i] + counts[i];
  if (!run_gather || r.rows == 0) return ATTPC_OK;
  std::vector<int64_t> dst(r.segs);
  std::vector<int64_t> fill(start->begin(), start->end() - 1);
  for (size_t s = 0; s < segs.size(); ++s) {  // segments of one event appear in window order
    dst[s] = fill[segs[s].event];
    fill[segs[s].event] += segs[s].count;
  }
  int32_t rc;
  if ((rc = ensure(ctx, ctx->scratch[6], r.segs * sizeof(int64_t)))) return rc;
  if ((rc = ensure(ctx, ctx->scratch[7], (size_t)r.rows * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->asm_labels, (size_t)r.rows * sizeof(int64_t)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->scratch[6].p, dst.data(), r.segs * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)std::min<uint64_t>(r.segs, 65535)), dim3(256), 0,
                     ctx->stream, static_cast<const Segment*>(ctx->segments.p),
                     static_cast<const int64_t*>(ctx->scratch[6].p), (uint32_t)r.segs,
                     static_cast<const double*>(ctx->points.p), static_cast<const int64_t*>(ctx->labels.p),
                     static_cast<double*>(ctx->scratch[7].p), static_cast<int64_t*>(ctx->asm_labels.p));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // dst is a local vector
  return ATTPC_OK;
}

// copy one chunk's cloud to the caller's CSR arrays (events in order)
int32_t assemble_chunk(attpc_ctx* ctx, const ChunkResult& r, uint32_t n, uint64_t chunk_first_local,
                       attpc_cloud_out* out, int64_t* row_cursor, bool* over_capacity) {
  const int64_t base = *row_cursor;
  const bool fits = out->points && out->labels && base + (int64_t)r.rows <= out->capacity;
  std::vector<int64_t> start;
  int32_t rc = gather_chunk_csr(ctx, r, n, &start, fits);
  if (rc) return rc;
  if (out->offsets)
    for (uint32_t i = 0; i <= n; ++i) out->offsets[chunk_first_local + i] = base + start[i];
  *row_cursor = base + start[n];
  if (!fits) {
    if (*row_cursor > out->capacity) *over_capacity = true;
    return ATTPC_OK;
  }
  if (r.rows == 0) return ATTPC_OK;
  HIP_TRY(ctx, hipMemcpyAsync(out->points + base * 3, ctx->scratch[7].p, (size_t)r.rows * 3 * sizeof(double),
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out->labels + base, ctx->asm_labels.p, (size_t)r.rows * sizeof(int64_t),
                              hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ATTPC_OK;
}

// response + threshold + Spyral rows of one chunk on the device, then D2H (rows of 8 doubles)
int32_t assemble_chunk_spyral(attpc_ctx* ctx, const ChunkResult& r, uint32_t n, uint64_t chunk_first_local,
                              attpc_cloud_out* out, int64_t* row_cursor, bool* over_capacity) {
  std::vector<int64_t> start;
  int32_t rc = gather_chunk_csr(ctx, r, n, &start, true);
  if (rc) return rc;
  const int64_t base = *row_cursor;
  std::vector<int64_t> kept_start(n + 1, 0);
  if (r.rows) {
    if ((rc = ensure(ctx, ctx->sp_event_start, (n + 1) * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sp_kept_start, (n + 1) * sizeof(int64_t)))) return rc;
    if ((rc = ensure(ctx, ctx->sp_kept, n * sizeof(int32_t)))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_event_start.p, start.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    launch_spyral_count(ctx->stream, ctx->spyral, n, static_cast<const int64_t*>(ctx->sp_event_start.p),
                        static_cast<const double*>(ctx->scratch[7].p), static_cast<int32_t*>(ctx->sp_kept.p));
    HIP_TRY(ctx, hipGetLastError());
    std::vector<int32_t> kept(n);
    HIP_TRY(ctx, hipMemcpyAsync(kept.data(), ctx->sp_kept.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < n; ++i) kept_start[i + 1] = kept_start[i] + kept[i];
  }
  const int64_t n_kept = kept_start[n];
  if (out->offsets)
    for (uint32_t i = 0; i <= n; ++i) out->offsets[chunk_first_local + i] = base + kept_start[i];
  *row_cursor = base + n_kept;
  if (*row_cursor > out->capacity || !out->points || !out->labels) {
    if (*row_cursor > out->capacity) *over_capacity = true;
    return ATTPC_OK;
  }
  if (n_kept == 0) return ATTPC_OK;
  if ((rc = ensure(ctx, ctx->sp_rows, (size_t)n_kept * 8 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->sp_labels, (size_t)n_kept * sizeof(int64_t)))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->sp_kept_start.p, kept_start.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
  launch_spyral_write(ctx->stream, ctx->spyral, n, static_cast<const int64_t*>(ctx->sp_event_start.p),
                      static_cast<const int64_t*>(ctx->sp_kept_start.p), static_cast<const double*>(ctx->scratch[7].p),
                      static_cast<const int64_t*>(ctx->asm_labels.p), static_cast<double*>(ctx->sp_rows.p),
                      static_cast<int64_t*>(ctx->sp_labels.p));
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out->points + base * 8, ctx->sp_rows.p, (size_t)n_kept * 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(out->labels + base, ctx->sp_labels.p, (size_t)n_kept * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ATTPC_OK;
}

int32_t ensure_kin_buffers(attpc_ctx* ctx, uint32_t n, int n_rows) {
  int32_t rc;
  if ((rc = ensure(ctx, ctx->p4, (size_t)n * n_rows * 4 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->vertex, (size_t)n * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(ctx, ctx->status, (size_t)n * sizeof(int32_t)))) return rc;
  if ((rc = ensure(ctx, ctx->attempts, (size_t)n * sizeof(uint32_t)))) return rc;
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
  if (hipStreamCreate(&ctx->stream) != hipSuccess) {
    delete ctx;
    return ATTPC_E_HIP;
  }
  for (auto& e : ctx->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      delete ctx;
      return ATTPC_E_HIP;
    }
  *out = ctx;
  return ATTPC_OK;
}

int32_t attpc_ctx_destroy(attpc_ctx* ctx) {
  if (!ctx) return ATTPC_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  free_all(ctx->kin_allocs);
  free_all(ctx->det_allocs);
  free_all(ctx->spyral_allocs);
  DevBuf* bufs[] = {&ctx->p4, &ctx->vertex, &ctx->status, &ctx->attempts, &ctx->arena, &ctx->block_table,
                    &ctx->counts, &ctx->n_steps, &ctx->trk_ctrl, &ctx->points, &ctx->labels, &ctx->segments,
                    &ctx->out_ctrl, &ctx->asm_labels, &ctx->sp_rows, &ctx->sp_labels,
                    &ctx->sp_event_start, &ctx->sp_kept, &ctx->sp_kept_start};
  for (DevBuf* b : bufs)
    if (b->p) (void)hipFree(b->p);
  for (auto& b : ctx->scratch)
    if (b.p) (void)hipFree(b.p);
  for (auto& e : ctx->ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return ATTPC_OK;
}

const char* attpc_last_error(const attpc_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int32_t attpc_set_chunk_events(attpc_ctx* ctx, int32_t chunk_events) {
  if (!ctx) return ATTPC_E_INVALID;
  ctx->chunk_events = chunk_events > 0 ? chunk_events : 65536;
  return ATTPC_OK;
}

int32_t attpc_sync(attpc_ctx* ctx) {
  if (!ctx) return ATTPC_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ATTPC_OK;
}

int32_t attpc_kin_configure(attpc_ctx* ctx, const attpc_kin_desc* d) {
  if (!ctx || !d) return ATTPC_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (d->n_steps < 1 || d->n_steps > ATTPC_MAX_STEPS) return fail(ctx, ATTPC_E_INVALID, "n_steps=%d", d->n_steps);
  if (d->sample_limit < 1) return fail(ctx, ATTPC_E_INVALID, "sample_limit=%d", d->sample_limit);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_all(ctx->kin_allocs);
  ctx->kin_ready = false;
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
  const uint64_t chunk = (uint64_t)std::max(1, ctx->chunk_events) * 4;
  for (uint64_t done = 0; done < n_events; done += chunk) {
    const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, n_events - done);
    int32_t rc = ensure_kin_buffers(ctx, n, n_rows);
    if (rc) return rc;
    launch_kin_run(ctx->stream, ctx->kin, seed, first_event + done, n, static_cast<double*>(ctx->p4.p),
                   static_cast<double*>(ctx->vertex.p), static_cast<int32_t*>(ctx->status.p),
                   static_cast<uint32_t*>(ctx->attempts.p));
    HIP_TRY(ctx, hipGetLastError());
    if (p4) HIP_TRY(ctx, hipMemcpyAsync(p4 + done * n_rows * 4, ctx->p4.p, (size_t)n * n_rows * 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (vertex) HIP_TRY(ctx, hipMemcpyAsync(vertex + done * 3, ctx->vertex.p, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (status) HIP_TRY(ctx, hipMemcpyAsync(status + done, ctx->status.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    if (attempts) HIP_TRY(ctx, hipMemcpyAsync(attempts + done, ctx->attempts.p, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
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
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  free_all(ctx->det_allocs);
  ctx->det_ready = false;
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
  return ATTPC_OK;
}

static void accumulate(attpc_run_stats* st, const ChunkResult& r) {
  st->n_points += r.rows;
  st->n_track_samples += r.samples;
  st->n_failed += r.failed;
  st->n_lds_overflow += r.retried;
  st->charge_checksum += r.charge;
  st->key_checksum += r.keys;
  st->ms_scatter += r.ms_scatter;
  st->launches_scatter += 1;
  st->n_inconsistent += (uint32_t)r.mismatch;
}

int32_t attpc_det_run(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                      const attpc_event_layout* layout, const double* p4, const double* vertex,
                      attpc_cloud_out* out, attpc_run_stats* stats) {
  if (!ctx || !p4 || !vertex) return ATTPC_E_INVALID;
  if (!ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_det_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc = validate_layout(ctx, layout);
  if (rc) return rc;
  attpc_run_stats st{};
  st.n_events = n_events;
  const int n_rows = layout->n_rows;
  const uint64_t chunk = (uint64_t)std::max(1, ctx->chunk_events);
  const uint64_t batch = track_batch_events(ctx, *layout);
  int64_t row_cursor = 0;
  bool over = false;
  if (out && out->offsets) out->offsets[0] = 0;
  for (uint64_t b0 = 0; b0 < n_events; b0 += batch) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(batch, n_events - b0);
    if ((rc = ensure_kin_buffers(ctx, nb, n_rows))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->p4.p, p4 + b0 * n_rows * 4, (size_t)nb * n_rows * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->vertex.p, vertex + b0 * 3, (size_t)nb * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    TrackBuffers trk;
    if ((rc = run_tracks(ctx, *layout, seed, first_event + b0, nb, false, &trk, &st.ms_tracks))) return rc;
    st.launches_tracks += 1;
    for (uint32_t e0 = 0; e0 < nb; e0 += (uint32_t)chunk) {
      const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, nb - e0);
      ChunkResult r;
      if ((rc = run_scatter(ctx, *layout, trk, seed, first_event + b0 + e0, e0, n, &r))) return rc;
      accumulate(&st, r);
      if (out && (rc = assemble_chunk(ctx, r, n, b0 + e0, out, &row_cursor, &over))) return rc;
    }
  }
  if (stats) *stats = st;
  if (over) return fail(ctx, ATTPC_E_CAPACITY, "cloud needs %lld rows, capacity %lld", (long long)row_cursor, (long long)out->capacity);
  return ATTPC_OK;
}

static int32_t sim_run_impl(attpc_ctx* ctx, uint64_t seed, uint64_t first_event, uint64_t n_events,
                            const attpc_event_layout* layout, double* p4, double* vertex, int32_t* kin_status,
                            attpc_cloud_out* out, attpc_run_stats* stats, bool spyral) {
  if (!ctx) return ATTPC_E_INVALID;
  if (spyral && !ctx->spyral_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_spyral_configure has not been called");
  if (!ctx->kin_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_kin_configure has not been called");
  if (!ctx->det_ready) return fail(ctx, ATTPC_E_NOTCONFIGURED, "attpc_det_configure has not been called");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int32_t rc = validate_layout(ctx, layout);
  if (rc) return rc;
  const int n_rows = 4 + 2 * (ctx->kin.n_steps - 1);
  if (layout->n_rows != n_rows) return fail(ctx, ATTPC_E_INVALID, "layout.n_rows=%d but the pipeline has %d rows", layout->n_rows, n_rows);
  attpc_run_stats st{};
  st.n_events = n_events;
  const uint64_t chunk = (uint64_t)std::max(1, ctx->chunk_events);
  const uint64_t batch = track_batch_events(ctx, *layout);
  int64_t row_cursor = 0;
  bool over = false;
  if (out && out->offsets) out->offsets[0] = 0;
  std::vector<int32_t> hstatus;
  for (uint64_t b0 = 0; b0 < n_events; b0 += batch) {
    const uint32_t nb = (uint32_t)std::min<uint64_t>(batch, n_events - b0);
    if ((rc = ensure_kin_buffers(ctx, nb, n_rows))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev[4], ctx->stream));
    launch_kin_run(ctx->stream, ctx->kin, seed, first_event + b0, nb, static_cast<double*>(ctx->p4.p),
                   static_cast<double*>(ctx->vertex.p), static_cast<int32_t*>(ctx->status.p),
                   static_cast<uint32_t*>(ctx->attempts.p));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev[5], ctx->stream));
    TrackBuffers trk;
    if ((rc = run_tracks(ctx, *layout, seed, first_event + b0, nb, true, &trk, &st.ms_tracks))) return rc;
    float ms_k = 0;
    HIP_TRY(ctx, hipEventElapsedTime(&ms_k, ctx->ev[4], ctx->ev[5]));
    st.ms_kinematics += ms_k;
    st.launches_kinematics += 1;
    st.launches_tracks += 1;
    hstatus.resize(nb);
    HIP_TRY(ctx, hipMemcpy(hstatus.data(), ctx->status.p, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int32_t s : hstatus) st.n_sample_limit += (s != 0);
    if (kin_status) std::memcpy(kin_status + b0, hstatus.data(), (size_t)nb * sizeof(int32_t));
    if (p4) HIP_TRY(ctx, hipMemcpy(p4 + b0 * n_rows * 4, ctx->p4.p, (size_t)nb * n_rows * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (vertex) HIP_TRY(ctx, hipMemcpy(vertex + b0 * 3, ctx->vertex.p, (size_t)nb * 3 * sizeof(double), hipMemcpyDeviceToHost));
    for (uint32_t e0 = 0; e0 < nb; e0 += (uint32_t)chunk) {
      const uint32_t n = (uint32_t)std::min<uint64_t>(chunk, nb - e0);
      ChunkResult r;
      if ((rc = run_scatter(ctx, *layout, trk, seed, first_event + b0 + e0, e0, n, &r))) return rc;
      accumulate(&st, r);
      if (out) {
        rc = spyral ? assemble_chunk_spyral(ctx, r, n, b0 + e0, out, &row_cursor, &over)
                    : assemble_chunk(ctx, r, n, b0 + e0, out, &row_cursor, &over);
        if (rc) return rc;
      }
    }
  }
  if (spyral) st.n_points = (uint64_t)row_cursor;  // rows that survive the threshold
  if (stats) *stats = st;
  if (over) return fail(ctx, ATTPC_E_CAPACITY, "cloud needs %lld rows, capacity %lld", (long long)row_cursor, (long long)out->capacity);
  return ATTPC_OK;
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

int32_t attpc_spyral_configure(attpc_ctx* ctx, const attpc_spyral_desc* d) {
  if (!ctx || !d || !d->response || !d->pad_centers || !d->pad_sizes || d->n_pads < 1) return ATTPC_E_INVALID;
  if (d->windows_edge <= d->micromegas_edge) return fail(ctx, ATTPC_E_INVALID, "windows_edge <= micromegas_edge");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
  int32_t rc = validate_layout(ctx, layout);
  if (rc) return rc;
  if (n_events > (uint64_t)ctx->chunk_events) return fail(ctx, ATTPC_E_INVALID, "attpc_det_tracks handles at most one chunk");
  const uint32_t n = (uint32_t)n_events;
  const int n_rows = layout->n_rows;
  if ((rc = ensure_kin_buffers(ctx, n, n_rows))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->p4.p, p4, (size_t)n * n_rows * 4 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->vertex.p, vertex, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  TrackBuffers trk;
  double ms = 0;
  if ((rc = run_tracks(ctx, *layout, seed, first_event, n, false, &trk, &ms))) return rc;
  const uint32_t n_tracks = n * (uint32_t)layout->n_sim;
  uint32_t tctrl[4];
  HIP_TRY(ctx, hipMemcpy(tctrl, ctx->trk_ctrl.p, sizeof tctrl, hipMemcpyDeviceToHost));
  std::vector<int32_t> table((size_t)n_tracks * MAX_BLOCKS_PER_TRACK);
  HIP_TRY(ctx, hipMemcpy(table.data(), ctx->block_table.p, table.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(counts, ctx->counts.p, (size_t)n_tracks * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(n_steps, ctx->n_steps.p, (size_t)n_tracks * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (samples) {
    // tctrl[1] counts reserved blocks (waves reserve pools), all below arena_blocks after a good run
    std::vector<double> arena(std::min<size_t>(tctrl[1], ctx->arena_blocks) * ARENA_BLK * 4);
    if (!arena.empty()) HIP_TRY(ctx, hipMemcpy(arena.data(), ctx->arena.p, arena.size() * sizeof(double), hipMemcpyDeviceToHost));
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
