#!/bin/bash
# first half of tools/collect_round.sh (a gpurun call is limited to 20 minutes): headline profile, bench lines, tables
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
echo "collect a done"
