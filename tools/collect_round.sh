#!/bin/bash
# usage (GPU box, from the repo root): bash tools/collect_round.sh <tag>
# Everything the round's tables are made from, each step under its own time limit, outputs under gpurun_out/<tag>_*:
# the headline profile (kernel stats + PMC passes), the other configurations' bench lines, the secondary tables and the
# PMC / kernel-stats passes of the KL and m = 3 kernels.
set -u
tag=${1:-cur}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
run() { echo "== $*"; timeout -k 10 "$@" || exit 1; }
cd $root
run 300 bash tools/profile_round.sh $tag
run 200 python3 bench.py > $out/${tag}_bench_default.json
for c in 1 2 4 5; do run 300 python3 bench.py --config $c > $out/${tag}_bench_cfg$c.json; done
run 200 python3 tools/bench_vmf3.py > $out/${tag}_bench_vmf3.md
run 200 python3 tools/bench_kl.py > $out/${tag}_bench_kl.md
run 300 python3 tools/bench_narrow.py > $out/${tag}_bench_narrow.md
run 400 python3 tools/bench_extra.py > $out/${tag}_bench_extra.md
run 120 python3 tools/bench_latent.py > $out/${tag}_bench_latent.md
run 100 python3 tools/count_launches.py > $out/${tag}_launches.md
run 300 bash tools/profile_vmf3.sh ${tag}_vmf3
cd /tmp && export TMPDIR=/tmp
run 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_kl -- python3 $root/tools/bench_kl.py --reps 3 --no-ive > $out/${tag}_kl_prof.log 2>&1
for pass in "fetch FETCH_SIZE" "sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  set -- $pass; t=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_${tag}_kl_$t -- python3 $root/tools/bench_kl.py --reps 3 --no-ive > $out/pmc_${tag}_kl_$t.log 2>&1 || exit 1
done
echo "collect done"
