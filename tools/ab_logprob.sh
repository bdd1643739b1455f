#!/bin/bash
# usage (GPU box, repo root): tools/ab_logprob.sh "m:rows m:rows ..." name=lib [name=lib ...]
# interleaved in-process A/B (tools/ab_libs.py) of library builds on vmf_log_prob_fwd and vmf_log_prob_bwd
shapes=$1; shift
for sh in $shapes; do
  m=${sh%%:*}; rows=${sh##*:}
  for k in logprob logprobbwd; do
    echo "== $k m=$m rows=$rows"
    timeout -k 10 120 python tools/ab_libs.py "$@" --m $m --rows $rows --rounds ${ROUNDS:-6} --steps 20 --kernel $k || exit 1
  done
done
