#!/bin/bash
# usage: [BENCH_ARGS="--config 5"] tools/pmc_pass.sh <tag> <counters...>   -- one rocprofv3 --pmc pass over a short bench run (GPU box only)
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-rows 0 ${BENCH_ARGS:-} > $out.log 2>&1
echo "pmc $tag exit $?"
