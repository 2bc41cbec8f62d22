#!/bin/bash
# Vector-memory path counters (TA / TCP = vL1D / TD) of the scatter kernel (run on the GPU box through gpurun):
#   tools/pmc_mem.sh <tag> [events]  ->  gpurun_out/<tag>/mem_pmc.txt
set -e
TAG=${1:-pmc_mem}
N=${2:-65536}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
groups=(
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum"
  "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TD_TD_BUSY_sum TD_TC_STALL_sum"
  "TA_BUSY_avr GRBM_GUI_ACTIVE TA_FLAT_WRITE_WAVEFRONTS_sum TCP_TCC_WRITE_REQ_sum"
)
i=0
for g in "${groups[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$OUT/m$i" -o run -- python3 bench.py --workload ${WL:-o16aa} --events $N --steps 1 --warmup 0 --no-cpu-baseline --no-delivered > "$OUT/m$i.log" 2>&1 || echo "group $i failed (see $OUT/m$i.log)"
  echo "pass $i done: $g"
  i=$((i+1))
done
python3 - "$OUT" $N <<'PY' | tee "$OUT/mem_pmc.txt"
import csv, glob, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
acc = collections.OrderedDict()
for f in sorted(glob.glob(out + "/m*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        if "scatter_kernel" in row["Kernel_Name"]:
            acc[row["Counter_Name"]] = acc.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
for k, v in acc.items():
    print(f"scatter_kernel {k:40s} {v:18.0f} total {v / n:14.2f} per event")
PY
