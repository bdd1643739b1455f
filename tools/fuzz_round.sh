#!/bin/bash
# usage (GPU box, repo root): tools/fuzz_round.sh <first seed>   -- every randomised sweep in chunks that each print a line
set -u
s0=${1:-7}
out=$GRAFT_REPO_ROOT/gpurun_out/fuzz_round_$s0.txt
: > $out
run() { echo "== $*" | tee -a $out; timeout -k 10 "$@" 2>&1 | grep -v amdgpu.ids | tee -a $out; [ ${PIPESTATUS[0]} -eq 0 ] || exit 1; }
for i in 0 1 2 3; do run 400 python3 -u tests/fuzz_parity.py --cases 6000 --seed $((s0 + i)); done
for i in 0 1; do run 400 python3 -u tests/fuzz_vmf3.py --cases 6000 --seed $((s0 + i)); done
for i in 0 1; do run 400 python3 -u tests/fuzz_kl_elbo.py --cases 6000 --seed $((s0 + i)); done
for i in 0 1; do run 400 python3 -u tests/fuzz_log_prob.py --cases 6000 --seed $((s0 + i)); done
run 400 python3 -u tests/fuzz_general_host.py --cases 1500 --seed $s0
run 400 python3 -u tests/fuzz_ive.py --orders 600 --seed $s0
echo "fuzz round done" | tee -a $out
