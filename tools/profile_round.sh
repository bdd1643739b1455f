#!/bin/bash
# usage (GPU box, from the repo root): bash tools/profile_round.sh <tag>
# One --kernel-trace --stats run of the default bench plus three separate --pmc passes (HBM fetch, HBM write, SQ
# activity); outputs land under gpurun_out/ for tools/summarise_profiles.py.
set -u
tag=${1:-cur}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/prof_$tag -- python3 $root/bench.py --cpu-rows 0 > $root/gpurun_out/bench_$tag.json 2> $root/gpurun_out/bench_$tag.err || exit 1
bash $root/tools/pmc_pass.sh ${tag}_fetch FETCH_SIZE || exit 1
bash $root/tools/pmc_pass.sh ${tag}_write WRITE_SIZE || exit 1
bash $root/tools/pmc_pass.sh ${tag}_sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE || exit 1
