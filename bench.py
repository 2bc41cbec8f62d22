#!/usr/bin/env python3
"""Benchmark of the hot path: events/s for a fused kinematics + full-detector batch.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--events E] [--workload o16aa]

One "step" = one pass of the hot path over one batch of E synthetic events per GPU
(default: the BASELINE.json headline, 1e6-event two-step 16O(a,a')16O* -> a + 12C with the
full pad-plane point cloud).  Inputs are generated on the device from Philox streams keyed
by the global event id; the clouds stay resident in HBM (chunk buffers are overwritten),
so `value` is device-resident whole-job throughput.  N > 1 = one process per GPU: either the
driver starts this script once per GPU through torch.distributed.run (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment), or -- `python bench.py --gpus N` with no such
environment -- this process starts the N ranks itself, as fresh children and before it has
touched a GPU, relays rank 0's JSON line and exits non-zero if any rank does.  A `--gpus`
that disagrees with the world found in the environment is an error, never a silent 1-GPU run.
Rank r simulates the event-id range [r*E, (r+1)*E) (weak scaling) or, with --global-events G
(BASELINE configs[3]: 1e7 events over 8 GPUs), its contiguous share of G (strong scaling); no
data-path collective; the timed region is bracketed by a barrier + device sync on both sides
and the MAX over ranks is taken.

The JSON line also carries
  roofline     -- algorithmic HBM bytes of the dominant kernel's launches / its measured
                  average launch duration (HIP events on the engine's stream) vs 8 TB/s
  cpu_baseline -- the CPU oracle (plain-C restatement of the reference algorithm, OpenMP
                  over events) timed on this host's cores on a bounded sample, rank 0, N = 1
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
CLOCK_HZ = 2.4e9       # peak engine clock (same guide); 256 CUs x 4 SIMDs, a wave64 VALU instruction issues over 4 cycles
N_CUS, SIMDS_PER_CU, N_XCDS = 256, 4, 8
# LDS atomics: an LDS instruction of a wave64 is serviced 32 lanes per LDS cycle at best (two lane groups,
# MI355X_MICROARCH.md "LDS"), one LDS pipeline per CU -> at most CLOCK / 2 wave-level atomics per CU and second
LDS_ATOMIC_PEAK_PER_S = N_CUS * CLOCK_HZ / 2.0
PMC_PROFILE = "r03_pmc.json"


def parse_args(argv=None) -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--events", type=int, default=1_000_000, help="events per GPU per step (weak scaling)")
    ap.add_argument("--global-events", type=int, default=0,
                    help="events per step over ALL GPUs, split into contiguous id ranges (strong scaling; "
                         "BASELINE configs[3] = 10000000 with --gpus 8); overrides --events")
    ap.add_argument("--workload", default="o16aa")
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--chunk-events", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-delivered", action="store_true", help="skip the delivered-to-host leg (clouds over PCIe)")
    ap.add_argument("--delivered-events", type=int, default=60_000)
    ap.add_argument("--hint", action="store_true",
                    help="announce the next step to the engine (attpc_sim_hint_next): its first track batch is then integrated "
                         "behind this step's last scatter launches instead of standing alone at its head.  Off by default: "
                         "measured slower (197.4 against 192.7 ms per 1e6-event step, profiles/r03_hint_ab.md) -- both "
                         "kernels are issue bound, and the scatter launches lose more beside the track kernel than the "
                         "8 ms the track batch takes alone")
    ap.add_argument("--first-batch-chunks", type=int, default=-1, help="engine option first_batch_chunks (experiment)")
    ap.add_argument("--track-blocks-per-cu", type=int, default=None, help="engine option track_blocks_per_cu (experiment)")
    ap.add_argument("--serial-tracks", type=int, default=None, help="engine option serial_tracks (A/B: 0 / 1; default automatic)")
    ap.add_argument("--stub-engine", action="store_true",
                    help="TEST ONLY: no GPU, no library -- a stand-in engine with made-up statistics, so that the "
                         "launcher / sharding / reduction path runs on a CPU box; the line says data = 'stub'")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def launch_ranks(args: argparse.Namespace, argv: list[str]) -> int:
    """`python bench.py --gpus N` (N > 1) without a torchrun environment: start the N ranks as fresh child processes
    of this script -- this parent has made no GPU call and makes none -- with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, pass rank 0's stdout (the JSON line) through, and return non-zero if any rank
    failed.  A rank that dies takes the others down after a grace period instead of leaving them in a barrier."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    deadline = None
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                deadline = deadline or time.monotonic() + 30.0  # the others may be waiting for it in a barrier
        if deadline is not None and live and time.monotonic() > deadline:
            for p in live:
                p.kill()  # exactly the children started above
        time.sleep(0.05)
    if rc:
        print(f"bench.py: a rank exited with status {rc}", file=sys.stderr)
    return rc if 0 <= rc < 256 else 1


class StubEngine:
    """TEST ONLY (--stub-engine): stands where Engine stands, touches no GPU and no library; statistics are a pure
    function of the event-id range, so that sums over ranks can be checked against one rank over the union."""

    def __init__(self, n_rows: int):
        self.n_rows = n_rows

    def run(self, n_events: int, seed: int = 0, first_event: int = 0, **_kw) -> dict:
        ids = range(first_event, first_event + n_events)
        points = sum(100 + (i % 7) for i in ids)
        time.sleep(0.01)
        return {"stats": {"ms_kinematics": 0.0, "ms_tracks": 0.1, "ms_scatter": 1.0, "launches_kinematics": 0,
                          "launches_tracks": 1, "launches_scatter": 1, "n_points": points, "n_track_samples": 10 * n_events,
                          "n_failed": 0, "n_sample_limit": 0, "n_lone_buckets": 0, "n_inconsistent": 0,
                          "n_buffer_growths": 0, "n_tracks_capped": 0, "charge_checksum": (sum(ids) * 0x9E3779B97F4A7C15) % (1 << 64),
                          "key_checksum": (sum(ids) * (2 * seed + 1)) % (1 << 64)}}


def main(argv=None) -> int:
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)  # before anything that could touch a GPU (the imports below do not either)

    from attpc_engine_amd import sharding, workloads

    rank, local_rank, world_size = sharding.world()
    if args.gpus != world_size:
        # never a silent n_gpus: 1 line for a --gpus N request, nor an N-rank job labelled as something else
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_size}", file=sys.stderr)
        return 2
    # control plane only: barrier + scalar reductions.  (gloo announces its connections on the C-level stdout:
    # keep stdout for the one JSON line)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        dist = sharding.init_process_group("gloo")
        sharding.barrier(dist)
    finally:
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)

    pipeline, config, indices = workloads.WORKLOADS[args.workload](seed=args.seed)
    if args.stub_engine:
        ctx = None
        engine = StubEngine(len(pipeline.get_proton_numbers()))
    else:
        from attpc_engine_amd import _abi
        from attpc_engine_amd.engine import Engine

        n_dev = max(1, _abi.load_library().attpc_device_count())
        ctx = _abi.Context(local_rank % n_dev)  # one rank per GPU; wraps only when rehearsing on fewer GPUs
        engine = Engine(pipeline, config, indices, context=ctx, chunk_events=args.chunk_events or None)
        if args.first_batch_chunks >= 0:
            ctx.set_option("first_batch_chunks", args.first_batch_chunks)
        if args.track_blocks_per_cu is not None:
            ctx.set_option("track_blocks_per_cu", args.track_blocks_per_cu)
        if args.serial_tracks is not None:
            ctx.set_option("serial_tracks", args.serial_tracks)
    strong = args.global_events > 0
    step_events = args.global_events if strong else args.events * world_size  # events of one step over all ranks

    def sync() -> None:
        if ctx is not None:
            ctx.check(ctx.lib.attpc_sync(ctx.handle), "attpc_sync")

    # every step simulates its own range of global event ids (no replays): step s covers [s * G, (s + 1) * G) with
    # G = events of a step over all ranks, of which rank r takes its contiguous share; warm-up steps use the
    # ranges after the timed ones
    def step_range(step: int) -> tuple[int, int]:
        if strong:
            return sharding.strong_shard(step_events, rank, world_size, first_event=step * step_events)
        return sharding.weak_shard(args.events, rank, first_event=step * step_events)

    for w in range(args.warmup):
        first, n = step_range(args.steps + w)
        engine.run(n, seed=args.seed, first_event=first)
    sync()
    sharding.barrier(dist)
    t0 = time.perf_counter()
    stats = None
    agg = {"ms_kinematics": 0.0, "ms_tracks": 0.0, "ms_scatter": 0.0, "launches_kinematics": 0,
           "launches_tracks": 0, "launches_scatter": 0}
    totals = {"n_points": 0, "n_track_samples": 0, "n_failed": 0, "n_sample_limit": 0, "n_lone_buckets": 0,
              "n_inconsistent": 0, "n_buffer_growths": 0, "n_tracks_capped": 0}
    charge_acc = key_acc = 0
    my_events = 0
    for step in range(args.steps):
        first, n = step_range(step)
        if step + 1 < args.steps and hasattr(engine, "hint_next") and args.hint:
            # a stream of calls: say what the next one will be, so that its first track batch is integrated behind
            # this call's last scatter launches.  Never across the edges of the timed region: the last warm-up step
            # announces nothing (step 0 does all of its own work in here), and the last timed step neither.
            nxt_first, nxt_n = step_range(step + 1)
            engine.hint_next(nxt_n, seed=args.seed, first_event=nxt_first)
        stats = engine.run(n, seed=args.seed, first_event=first)["stats"]
        my_events += n
        for k in agg:
            agg[k] += stats[k]
        for k in totals:
            totals[k] += stats[k]
        charge_acc = (charge_acc + stats["charge_checksum"]) % (1 << 64)
        key_acc = (key_acc + stats["key_checksum"]) % (1 << 64)
    sync()
    sharding.barrier(dist)
    elapsed = time.perf_counter() - t0
    (elapsed_max,) = sharding.reduce_scalars(dist, [elapsed], "max")
    points, samples, failed, limit, lone, events_all, ranks_seen = sharding.reduce_scalars(
        dist, [float(totals["n_points"]), float(totals["n_track_samples"]), float(totals["n_failed"]),
               float(totals["n_sample_limit"]), float(totals["n_lone_buckets"]), float(my_events), 1.0], "sum")
    charge_sum, key_sum = sharding.reduce_checksums(dist, [charge_acc, key_acc])
    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return 0
    if int(ranks_seen) != world_size or int(events_all) != step_events * args.steps:
        print(f"bench.py: {int(ranks_seen)} ranks / {int(events_all)} events seen, expected {world_size} / "
              f"{step_events * args.steps}", file=sys.stderr)
        return 3

    value = step_events * args.steps / elapsed_max
    n_rows = len(pipeline.get_proton_numbers())
    p_event = totals["n_points"] / max(1, my_events)  # rank 0's events (every rank draws from the same distributions)
    # vertex f64[3] + p4 f64[N,4] + one i64 CSR offset + P points of 3 f64 + i64
    bytes_per_event = 24 + 32 * n_rows + 8 + 32 * p_event
    kernels = {"track_kernel": ("ms_tracks", "launches_tracks"), "scatter_kernel": ("ms_scatter", "launches_scatter")}
    dominant = max(kernels, key=lambda k: agg[kernels[k][0]])
    ms_key, launch_key = kernels[dominant]
    launches = max(1, agg[launch_key])
    events_per_launch = my_events / launches
    avg_ms = agg[ms_key] / launches
    achieved = events_per_launch * bytes_per_event / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_note = measured_traffic(args.workload, dominant, events_per_launch)
    issue = issue_roofs(args.workload, dominant, events_per_launch, avg_ms)
    line = {
        "metric": "events/sec (whole node); kinematics + full detector batch, point clouds device-resident (no D2H)",
        "value": value,
        "unit": "events/s",
        "n_gpus": world_size,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "stub (TEST ONLY: no device work behind these numbers)" if args.stub_engine else "synthetic",
        "config": {
            "workload": f"{args.workload}: {workloads.describe(args.workload)}",
            "events_per_gpu_per_step": step_events / world_size,
            "global_events_per_step": step_events,
            "parallelism": f"contiguous event-id shards x{world_size} ({'even split of a fixed total' if strong else 'fixed work per GPU'}), "
                           "one process per GPU, no collective in the data path",
            "ranks_reporting": int(ranks_seen),
            "output": "device-resident point clouds (3 f64 + i64 per point), chunk buffers overwritten",
            "points_per_event": points / max(1.0, events_all),
            "track_samples_per_event": samples / max(1.0, events_all),
            "event_ids": "step s: [s*G, (s+1)*G), G = global events per step; rank r its contiguous share -- no step replays another",
            "algorithmic_bytes_per_event": bytes_per_event,
            "failed_events": failed,
            "lone_time_buckets": lone,
            "buffer_growths_in_timed_steps": totals["n_buffer_growths"],
            "table_self_check_failures": totals["n_inconsistent"],
            "sample_limit_events": limit,
            # path-length dE/dx step only: tracks cut at the 10 001-sample cap before the end of the 1 us window
            # (rank 0's share; always 0 on the reference's time grid)
            "tracks_cut_at_the_sample_cap": totals["n_tracks_capped"],
            "charge_checksum": str(charge_sum),  # u64 sums over all steps and ranks, as strings (beyond int64 / f64)
            "key_checksum": str(key_sum),
        },
        "roofline": {
            "bound": "hbm",
            "kernel": dominant,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_unit": "bytes per launch",
            "traffic_source": traffic_note,
            "algorithmic_bytes_per_launch": events_per_launch * bytes_per_event,
            "avg_launch_ms": avg_ms,
            "events_per_launch": events_per_launch,
            "per": "GPU (rank 0's launches)",
            # kin_run_kernel is left out: its HIP events sit on the low-priority stream and measure the wait for
            # compute units behind the scatter workgroups, not the ~50 us the kernel runs (profiles/ kernel stats)
            "kernel_ms_total": {"scatter_kernel": agg["ms_scatter"],
                                "track_kernel_incl_wait_on_the_low_priority_stream": agg["ms_tracks"]},
            "note": "the kernel is VALU-issue bound, not HBM bound (DESIGN.md 4.3): frac is vs the HBM roof as the "
                    "contract asks, valu_issue_frac is the roof that binds",
            **issue,
        },
    }
    if world_size == 1 and not args.no_delivered and config is not None and not args.stub_engine:
        try:
            # (bounded by the rows as well: 60 000 events of configs[4] would be 100 GB of cloud)
            n_deliver = min(args.delivered_events, step_events, max(1000, int(4.5e8 / max(p_event, 1.0))))
            line["delivered"] = delivered(engine, n_deliver, args.seed, bytes_per_event, p_event)
        except Exception as exc:  # the headline line must not depend on this leg
            line["delivered"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world_size == 1 and not args.no_cpu_baseline and not args.stub_engine:
        line["cpu_baseline"] = cpu_baseline(args.workload, args.seed)
    print(json.dumps(line), flush=True)
    return 0


def _pmc(workload: str, kernel: str):
    """Counter totals of `kernel` and the events of the profiled run from the committed rocprofv3 PMC passes
    (profiles/r02_pmc.json, written by tools/pmc_profile.sh), or None."""
    try:
        prof = json.loads((ROOT / "profiles" / PMC_PROFILE).read_text())[workload]
        return prof[kernel], float(prof["events"])
    except (OSError, KeyError, ValueError):
        return None


def measured_traffic(workload: str, kernel: str, events_per_launch: float):
    """HBM bytes per launch of `kernel`: FETCH_SIZE + WRITE_SIZE (KiB -> bytes) of the committed PMC passes,
    per event, times this run's events per launch.  WRITE_SIZE equals the algorithmic output bytes to 0.1 %;
    FETCH_SIZE is quoted uncorrected: the guide's x2 correction is for wide (16 B / lane) streaming reads,
    this kernel reads 32-byte sample records and 2-byte LUT gathers.  None when no profile exists."""
    got = _pmc(workload, kernel)
    if got is None or "FETCH_SIZE" not in got[0] or "WRITE_SIZE" not in got[0]:
        return None, "no PMC profile for this workload"
    counters, events = got
    per_event = (counters["FETCH_SIZE"] + counters["WRITE_SIZE"]) * 1024.0 / events
    return per_event * events_per_launch, f"profiles/{PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in passes of their own, FETCH uncorrected)"


def issue_roofs(workload: str, kernel: str, events_per_launch: float, avg_launch_ms: float) -> dict:
    """The roofs that do bind: VALU issue and LDS atomics.  Instruction counts per event from the committed PMC
    pass, time from THIS run's launches (HIP events)."""
    got = _pmc(workload, kernel)
    if got is None or avg_launch_ms <= 0:
        return {"valu_issue_frac": None, "lds_atomics_per_s": None}
    counters, events = got
    seconds = avg_launch_ms * 1e-3
    out = {}
    if "SQ_INSTS_VALU" in counters:
        valu = counters["SQ_INSTS_VALU"] / events * events_per_launch
        out["valu_insts_per_event"] = counters["SQ_INSTS_VALU"] / events
        # wave64 VALU instruction = 4 issue cycles of one SIMD (16 lanes per cycle; f64 ones take longer, so
        # this is a lower bound of the busy share)
        out["valu_issue_frac"] = valu * 4.0 / (N_CUS * SIMDS_PER_CU * CLOCK_HZ * seconds)
    if "SQ_ACTIVE_INST_VALU" in counters and counters.get("GRBM_GUI_ACTIVE", 0) > 0:
        # share of the profiled run's SIMD cycles with a VALU instruction under way: SQ_ACTIVE_INST_VALU counts
        # quad-cycles summed over the chip's SIMDs, GRBM_GUI_ACTIVE the busy cycles of the launches summed over
        # the 8 XCDs
        simd_cycles = counters["GRBM_GUI_ACTIVE"] / N_XCDS * N_CUS * SIMDS_PER_CU
        out["valu_busy_frac_profiled"] = counters["SQ_ACTIVE_INST_VALU"] * 4.0 / simd_cycles
    if "SQ_INSTS_LDS_ATOMIC" in counters:
        atomics = counters["SQ_INSTS_LDS_ATOMIC"] / events * events_per_launch
        out["lds_atomics_per_event"] = counters["SQ_INSTS_LDS_ATOMIC"] / events
        out["lds_atomics_per_s"] = atomics / seconds
        out["lds_atomics_peak_per_s"] = LDS_ATOMIC_PEAK_PER_S
        out["lds_atomics_frac"] = atomics / seconds / LDS_ATOMIC_PEAK_PER_S
        out["lds_atomics_unit"] = "wave-level LDS atomic instructions (ds_cmpst / ds_max / ds_add_u64) per second, whole chip"
    out["issue_source"] = f"profiles/{PMC_PROFILE} (instruction counts per event) x this run's launch time"
    return out


def delivered(engine, n: int, seed: int, bytes_per_event: float, p_event: float) -> dict:
    """Delivered-to-host throughput: the same events with their clouds in host arrays (reference
    dtypes, CSR) and, second, with the GET response / ADC threshold / Spyral rows / z-sort done on the device and
    only those rows delivered.  One untimed pass sizes the buffers and touches their pages."""
    out = {"events": n,
           "host_buffers": "ordinary numpy arrays, reused across calls (rows cross PCIe as 8-byte / 24-byte transfer records "
                           "into library-owned pinned staging; host threads expand them to the reference's dtypes)"}
    cap = int(p_event * 1.25) + 64
    for name, run in (("cloud", lambda first: engine.run(n, seed=seed, first_event=first, fetch=True, pinned=False,
                                                         reuse_buffers=True, capacity_per_event=cap)),
                      # (about 43 % of the rows survive the ADC threshold; 0.6 leaves room, and a too small buffer only
                      #  costs a retry in the untimed pass)
                      ("spyral_rows", lambda first: engine.run_spyral(n, seed=seed, first_event=first, pinned=False,
                                                                      reuse_buffers=True, capacity_per_event=int(0.6 * cap)))):
        try:  # the legs are independent: one that does not fit this box must not hide the other
            run(10_000_000)  # untimed: allocates and pins the host arrays (seconds for tens of GB)
            times = []
            for rep in range(3):  # three calls on fresh event ranges; the line reports their median and all three
                t0 = time.perf_counter()
                res = run(20_000_000 + rep * 1_000_000)
                times.append(time.perf_counter() - t0)
            dt = sorted(times)[1]
            rows = int(res["offsets"][-1])
            del res                    # un-pinning the arrays takes a second as well: outside every timed region
        except Exception as exc:
            out[name] = {"error": f"{type(exc).__name__}: {exc}"}
            continue
        finally:
            engine._out_cache = None
        width = 3 if name == "cloud" else 8
        delivered_bytes = rows * (width + 1) * 8 + (n + 1) * 8
        link_bytes = rows * (8 if name == "cloud" else 24) + (n + 1) * 8  # 8-byte / 24-byte transfer records
        out[name] = {"events_per_s": n / dt, "rows_per_event": rows / n, "bytes_per_event": delivered_bytes / n,
                     "pcie_bytes_per_event": link_bytes / n, "pcie_GBps": link_bytes / dt / 1e9,
                     "events_per_s_of_each_call": [round(n / t) for t in times]}
    return out


def cpu_baseline(workload: str, seed: int) -> dict:
    """The CPU oracle on this host's cores, bounded sample (~20 s), same workload/seeds."""
    from oracle import pyoracle as orc
    from tests.helpers import Inputs

    cores, cpu_note = usable_cpus()
    inp = Inputs(workload, seed=seed)
    probe = 4 * cores
    t0 = time.perf_counter()
    orc.sim_batch(inp.kin, inp.det_raw, inp.layout, seed=seed, first=0, n=probe, threads=cores)
    rate = probe / (time.perf_counter() - t0)
    n = int(min(max(probe, rate * 20.0), 200_000))
    t0 = time.perf_counter()
    orc.sim_batch(inp.kin, inp.det_raw, inp.layout, seed=seed, first=0, n=n, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "events/s", "cores": cores, "kind": "port",
            "sample": f"{n} events of the same workload (ids 0..{n - 1}, same seed), OpenMP over events, {dt:.1f} s",
            "host_cpus": cpu_note}


def usable_cpus() -> tuple[int, str]:
    """CPUs this process may really use: the affinity mask cut down to the CPU quota of its control group (a GPU box
    shows 256 hardware threads and grants a job 16 of them through cpu.max) -- the thread count of the CPU baseline."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    note = f"{n} in the affinity mask"
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            granted = max(1, -(-int(quota) // int(period)))
            note += f", control-group quota {quota}/{period} = {granted}"
            n = min(n, granted)
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                granted = max(1, -(-quota // period))
                note += f", control-group quota {quota}/{period} = {granted}"
                n = min(n, granted)
        except (OSError, ValueError):
            pass
    return n, note


if __name__ == "__main__":
    sys.exit(main())
