#!/bin/bash
# usage (GPU box, from the repo root): bash tools/profile_narrow.sh <tag>
# The narrow general-m widths (tools/bench_narrow.py, m = 10, 12, 31, 2^24 rows): one --kernel-trace --stats run and
# separate --pmc passes; outputs under gpurun_out/ for tools/summarise_profiles.py.
set -u
tag=${1:-narrow}
root=$GRAFT_REPO_ROOT
S=${SHAPES:-10:24,12:24,31:24}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/tools/bench_narrow.py --shapes $S --s1 0 > $root/gpurun_out/bench_$tag.md 2> $root/gpurun_out/bench_$tag.err || exit 1
pass() {
  t=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $root/gpurun_out/pmc_${tag}_$t -- python3 $root/tools/bench_narrow.py --shapes $S --s1 0 > $root/gpurun_out/pmc_${tag}_$t.log 2>&1 || exit 1
  echo "pmc $t ok"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
