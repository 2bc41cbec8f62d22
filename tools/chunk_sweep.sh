#!/bin/bash
# Sensitivity of the default bench to the scatter chunk size (run on the GPU box): tools/chunk_sweep.sh
for c in 65536 131072 262144; do
  python3 bench.py --no-cpu-baseline --no-delivered --steps 4 --chunk-events $c > /tmp/chunk_$c.json
  python3 - /tmp/chunk_$c.json $c <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("chunk", sys.argv[2], "events/s", round(d["value"]), "ms_per_step", round(d["ms_per_step"], 2),
      "scatter ms/step", round(d["roofline"]["kernel_ms_total"]["scatter_kernel"] / d["steps"], 2),
      "tracks ms/step", round(d["roofline"]["kernel_ms_total"]["track_kernel"] / d["steps"], 2))
PY
done
