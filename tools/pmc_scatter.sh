#!/bin/bash
# PMC passes over one 20000-event launch of the headline workload (run on the GPU box, from the repo root):
#   tools/pmc_scatter.sh [tag]   ->  gpurun_out/pmc_<tag>/summary.txt  (per-event counts of scatter_kernel)
# One rocprofv3 run per counter group, --pmc with --kernel-trace only (no sys/hip/hsa tracing).
set -e
TAG=${1:-run}
OUT=gpurun_out/pmc_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
N=20000
groups=(
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS"
  "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_CVT SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN"
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"
)
i=0
for g in "${groups[@]}"; do
  rocprofv3 --pmc $g --kernel-trace --output-format csv -d "$OUT/g$i" -o run -- python3 bench.py --events $N --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/g$i.log" 2>&1
  i=$((i+1))
done
python3 - "$OUT" $N <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
acc = collections.OrderedDict()
for f in sorted(glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        for kern in ("scatter_kernel", "track_kernel"):
            if kern in row["Kernel_Name"]:
                key = kern + " " + row["Counter_Name"]
                acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
with open(out + "/summary.txt", "w") as fh:
    for k, v in acc.items():
        line = f"{k:48s} {v / n:14.1f} per event"
        print(line); fh.write(line + "\n")
PY
