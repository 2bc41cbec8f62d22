#!/bin/bash
# End-of-round measurements on the GPU box (run from the repo root through gpurun):
#   tools/round_end.sh <tag>   ->  gpurun_out/<tag>_{tests.log,bench.json,stats/,pmc_fetch/,pmc_write/}
# 1. the GPU parity suite, 2. the default bench line, 3. rocprofv3 --kernel-trace --stats of the same
# bench command, 4. HBM traffic counters (separate --pmc passes with --kernel-trace only).
set -e
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1 || { tail -20 "$OUT/tests.log"; exit 1; }
tail -2 "$OUT/tests.log"
timeout -k 10 400 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cut -c1-300 "$OUT/bench.json"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o run -- python3 bench.py > "$OUT/stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/pmc_$c" -o run -- python3 bench.py --events 65536 --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_$c.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
acc = collections.OrderedDict()
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        key = (name, row["Counter_Name"])
        acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
for k, v in acc.items():
    print(k, v)
PY
