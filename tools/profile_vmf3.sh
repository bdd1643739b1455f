#!/bin/bash
# usage (GPU box, from the repo root): bash tools/profile_vmf3.sh <tag>
# The m = 3 kernels (tools/bench_vmf3.py): one --kernel-trace --stats run, then separate --pmc passes (HBM fetch, HBM
# write, SQ activity); outputs under gpurun_out/ for tools/summarise_profiles.py.
set -u
tag=${1:-vmf3}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/tools/bench_vmf3.py > $root/gpurun_out/bench_$tag.md 2> $root/gpurun_out/bench_$tag.err || exit 1
pass() {
  t=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $root/gpurun_out/pmc_${tag}_$t -- python3 $root/tools/bench_vmf3.py --reps 3 > $root/gpurun_out/pmc_${tag}_$t.log 2>&1 || exit 1
  echo "pmc $t ok"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
