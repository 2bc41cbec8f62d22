#!/bin/bash
# Counter passes over the bench workload (run on the GPU box from the repo root, through gpurun):
#   tools/pmc_profile.sh <tag> [workload] [events]  ->  gpurun_out/<tag>/pmc.json (+ the rocprofv3 outputs)
# One rocprofv3 run per counter group, --pmc with --kernel-trace only (never sys/hip/hsa tracing: gpurun
# refuses that mix), the program itself behind "--".  FETCH_SIZE and WRITE_SIZE in passes of their own, as
# MI355X_MICROARCH.md prescribes.  Then the kernel-trace --stats run of the default bench command.
# pmc.json holds, per kernel, the counter TOTALS over the run and the events of the run; bench.py turns
# them into per-event figures (roofline.traffic, valu_issue_frac, lds_atomics_per_s).
set -e
TAG=${1:-pmc}
WL=${2:-o16aa}
N=${3:-65536}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
groups=(
  "FETCH_SIZE"
  "WRITE_SIZE"
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS"
  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"
  "GRBM_GUI_ACTIVE SQ_WAVES SQ_CYCLES"
)
i=0
for g in "${groups[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$OUT/g$i" -o run -- python3 bench.py --workload $WL --events $N --steps 1 --warmup 0 --no-cpu-baseline --no-delivered > "$OUT/g$i.log" 2>&1 || echo "group $i failed (see $OUT/g$i.log)"
  echo "pass $i done: $g"
  i=$((i+1))
done
python3 - "$OUT" $N $WL <<'PY'
import csv, glob, json, sys, collections
out, n, wl = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = collections.OrderedDict()
launches = collections.Counter()
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for row in csv.DictReader(open(f)):
        for kern in ("scatter_kernel", "track_kernel", "kin_run_kernel", "lone_bucket_kernel"):
            if kern in row["Kernel_Name"]:
                acc.setdefault(kern, collections.OrderedDict())
                acc[kern][row["Counter_Name"]] = acc[kern].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                seen.add((kern, row.get("Dispatch_Id")))
    for kern, _ in seen:
        launches[(f.split("/g")[1].split("/")[0], kern)] += 1
doc = {wl: {"events": n, "command": f"bench.py --workload {wl} --events {n} --steps 1 --warmup 0", "counters_are": "totals over all launches of the kernel in that run",
            "units": "FETCH_SIZE / WRITE_SIZE in KiB (rocprofv3), SQ_* as reported (wave-instructions; *_CYCLES and ACTIVE/WAIT in quad-cycles summed over waves or SIMDs)",
            **acc}}
json.dump(doc, open(out + "/pmc.json", "w"), indent=1)
for kern, c in acc.items():
    for k, v in c.items():
        print(f"{kern:22s} {k:28s} {v / n:14.2f} per event")
PY
echo "---- track_kernel in batches of 8 scatter chunks (lane occupancy of the persistent-lane scheme) ----"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d "$OUT/tracks8" -o run -- python3 bench.py --workload $WL --events 600000 --steps 1 --warmup 0 --no-cpu-baseline --no-delivered > "$OUT/tracks8.log" 2>&1 || echo "tracks8 pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
per_dispatch = collections.OrderedDict()
for f in sorted(glob.glob(out + "/tracks8/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "track_kernel" in row["Kernel_Name"]:
            d = per_dispatch.setdefault(row["Dispatch_Id"], {"grid": row.get("Grid_Size")})
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
rows = []
for disp, c in per_dispatch.items():
    if c.get("SQ_INSTS_VALU"):
        c["active_lanes_per_valu_instruction"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_INSTS_VALU"] if c.get("SQ_THREAD_CYCLES_VALU") else None
    rows.append(c)
    print("track_kernel dispatch", disp, c)
doc = json.load(open(out + "/pmc.json"))
doc["track_kernel_batches"] = {"command": "bench.py --events 600000 --steps 1 --warmup 0 (pilot batch of 16 384 events, then 524 288 = 8 scatter chunks, then the rest)",
                               "note": "active lanes of 64 = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU (scatter_kernel: 54)",
                               "dispatches": rows}
json.dump(doc, open(out + "/pmc.json", "w"), indent=1)
PY
echo "---- kernel-trace --stats of the default bench command ----"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py --workload $WL --no-cpu-baseline --no-delivered > "$OUT/stats_bench.json" 2> "$OUT/stats.err" || echo "stats run failed"
for f in $(find "$OUT/stats" -name "*kernel_stats.csv"); do cp "$f" "$OUT/kernel_stats.csv"; cat "$f"; done
cut -c1-400 "$OUT/stats_bench.json"
