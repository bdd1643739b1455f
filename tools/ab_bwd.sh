#!/bin/bash
# usage (GPU box, repo root): tools/ab_bwd.sh "m:rows m:rows ..." name=lib [name=lib ...]
# interleaved in-process A/B (tools/ab_libs.py) of library builds on vmf_rsample_bwd (regenerated noise)
shapes=$1; shift
for sh in $shapes; do
  m=${sh%%:*}; rows=${sh##*:}
  echo "== bwd m=$m rows=$rows"
  timeout -k 10 120 python tools/ab_libs.py "$@" --m $m --rows $rows --rounds ${ROUNDS:-6} --steps 15 --kernel bwd || exit 1
done
