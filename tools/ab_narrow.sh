#!/bin/bash
# usage (GPU box, repo root): tools/ab_narrow.sh "m m m ..." name=lib [name=lib ...]
# interleaved in-process A/B (tools/ab_libs.py) of library builds on the narrow general-m forward shapes, 2^24 rows
ms=$1; shift
for m in $ms; do
  echo "== m=$m"
  timeout -k 10 120 python tools/ab_libs.py "$@" --m $m --rows 16777216 --rounds ${ROUNDS:-8} --steps 30 --no-log-prob || exit 1
done
