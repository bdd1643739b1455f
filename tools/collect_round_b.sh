#!/bin/bash
# second half of tools/collect_round.sh: kernel stats and PMC passes of the m = 3 and KL kernels
set -u
tag=${1:-cur}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
run() { echo "== $*"; timeout -k 10 "$@" || exit 1; }
cd $root
run 400 bash tools/profile_vmf3.sh ${tag}_vmf3
cd /tmp && export TMPDIR=/tmp
run 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_kl -- python3 $root/tools/bench_kl.py --reps 3 --no-ive > $out/${tag}_kl_prof.log 2>&1
for pass in "fetch FETCH_SIZE" "sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  set -- $pass; t=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_${tag}_kl_$t -- python3 $root/tools/bench_kl.py --reps 3 --no-ive > $out/pmc_${tag}_kl_$t.log 2>&1 || exit 1
done
echo "collect b done"
