#!/bin/bash
# usage (GPU box, repo root): ROUNDS=3 [BENCH_ARGS="--config 5"] bash tools/ab_bench.sh libA.so libB.so [libC.so ...]
# Alternates fresh `bench.py` processes (the driver's flags, no CPU baseline) between two builds of the library and
# prints each run's mean / median kernel time: the comparison that includes the clock ramp of the first launches,
# which an interleaved A/B inside one warmed-up process (tools/ab_libs.py) does not see.
set -u
n=${ROUNDS:-3}
for i in $(seq $n); do
  for lib in "$@"; do
    VMF_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 bench.py --warmup 5 --steps 20 --cpu-rows 0 ${BENCH_ARGS:-} > /tmp/ab_line.json || exit 1
    python3 - "$lib" <<'PY'
import json, sys
d = json.load(open("/tmp/ab_line.json"))
r = d["roofline"]
print(f"{sys.argv[1]:48s} step {d['ms_per_step']:.4f}  kernel mean {r['kernel_ms']:.4f}  median {r['kernel_ms_median']:.4f}  min {r['kernel_ms_min']:.4f}")
PY
  done
done
