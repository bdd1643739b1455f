#!/bin/bash
# usage (GPU box): tools/pmc_cmd.sh <tag> "<counters separated by spaces>" <python script and its arguments...>
# one rocprofv3 --pmc pass over the given command; output under gpurun_out/pmc_<tag>
set -u
tag=$1; ctrs=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
script=$GRAFT_REPO_ROOT/$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -- python3 $script "$@" > $out.log 2>&1
echo "pmc $tag exit $?"
