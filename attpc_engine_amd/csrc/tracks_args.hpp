// tracks_args.hpp -- kernel argument blocks and host launch wrappers (one per kernel file).
#pragma once
#include "common.hpp"
#include "unpack_host.hpp"

namespace attpc {

struct TrackArgs {
  DetDev det;
  attpc_event_layout layout;
  TrackBuffers buf;
  const double* p4;           // [n_events][n_rows][4]
  const double* vertex;       // [n_events][3]
  const int32_t* kin_status;  // [n_events] or nullptr; != 0 -> event has no tracks
  uint64_t seed;
  uint64_t first_event;       // global id of chunk-local event 0
  uint32_t n_events;
  uint32_t n_tracks;          // n_events * n_sim
  // Order in which the tracks are handed to the lanes: all events' nucleus sim_order[0] first, then sim_order[1], ...
  // (the host puts the species with the longest tracks first: the lanes run dry on short tracks at the kernel's end).
  // sim_order[0] == 0xff: event by event, as the tables are laid out.
  uint8_t sim_order[ATTPC_MAX_SIM];
};

struct ScatterArgs {
  DetDev det;
  attpc_event_layout layout;
  TrackBuffers trk;
  CloudBuffers out;
  uint64_t seed;
  uint64_t first_event;
  uint32_t n_events;
  uint32_t event0;     // first event of this launch within the track batch (track ids start at event0 * n_sim)
  uint32_t batch;      // events a workgroup takes per visit to the event counter
  uint32_t row_block;  // output rows a workgroup reserves at a time (1: exactly what each window needs)
  // merge variant of the kernel (scatter.hip, scatter_kernel<false, true>; path-length dE/dx step): per workgroup two
  // lists of merge_cap 8-byte entries {arena record, time bucket | nucleus << 10 | slice << 13}, the event's entries
  // in list order and sorted by time bucket; merge_cap >= the entries an event can have.  nullptr: the default kernel
  uint2* merge_scratch;
  uint32_t merge_cap;
};

void launch_kin_run(hipStream_t s, const attpc_kin_desc& d, uint64_t seed, uint64_t first_event, uint32_t n,
                    double* p4, double* vertex, int32_t* status, uint32_t* attempts);
void launch_kin_calculate(hipStream_t s, const attpc_kin_desc& d, uint32_t n, const double* beam, const double* ex,
                          const double* th, const double* ph, double* p4, int32_t* status);
void launch_decay_calculate(hipStream_t s, uint32_t n, const double* parent, double m1, double m2, const double* ex,
                            const double* th, const double* ph, double* out, int32_t* status);
void launch_track_kernel(uint32_t blocks, size_t lds_bytes, hipStream_t s, const TrackArgs& a);
// scatter.hip compiled as it is / through scatter_small.hip (workgroups per CU: 1 / 2)
void launch_scatter_kernel_big(uint32_t n_workgroups, hipStream_t s, const ScatterArgs& a);
void launch_scatter_kernel_small(uint32_t n_workgroups, hipStream_t s, const ScatterArgs& a);
void launch_scatter_kernel_wide(uint32_t n_workgroups, hipStream_t s, const ScatterArgs& a);  // scatter_wide.hip: u64 sums
// lone.hip: the time buckets scatter_kernel recorded in out.lone_list (normally none: exits at once)
void launch_lone_bucket_kernel(uint32_t n_workgroups, hipStream_t s, const ScatterArgs& a);

// response + threshold + Spyral rows on device (spyral.hip)
struct SpyralDev {
  const double* response;      // [512]
  const double* sorted_desc;   // [512] response sorted descending
  const double* prefix;        // [513] prefix sums of sorted_desc (prefix[k] = sum of the k largest)
  const double* pad_centers;   // [n_pads][2]
  const double* pad_sizes;     // [n_pads]
  int32_t n_pads;
  double r_max, total;
  double window_edge, mm_edge, length, threshold;
};
void launch_spyral_count(hipStream_t s, const SpyralDev& sp, uint32_t n_events, const int64_t* event_start,
                         const double* points, uint32_t* kept);
// rows of every event sorted by z (writer.py:236-238); sort_scratch: one u32 + one f64 per cloud row of the chunk
// (SpyralPacked: the 24-byte transfer record of a Spyral row, unpack_host.hpp)
// packed != nullptr: write SpyralPacked records (and raise *pack_flag for a row that does not fit) instead of rows / labels
void launch_spyral_write(hipStream_t s, const SpyralDev& sp, uint32_t n_events, const int64_t* event_start,
                         const int64_t* kept_start, const double* points, const int64_t* labels, double* rows,
                         int64_t* out_labels, uint32_t* sort_idx, double* sort_key, SpyralPacked* packed, int64_t* pack_flag);

}  // namespace attpc
