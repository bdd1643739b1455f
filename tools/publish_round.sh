#!/bin/bash
# usage (here, after tools/collect_round.sh TAG ran on the GPU box): bash tools/publish_round.sh TAG NAME
# Copies / summarises gpurun_out/TAG_* into the tracked profiles/NAME_* files.
set -eu
tag=$1; name=$2
cd "$(dirname "$0")/.."
g=gpurun_out
sed -i '/^== /d' $g/${tag}_*.json $g/${tag}_*.md
python3 tools/summarise_profiles.py stats $g/prof_$tag profiles/${name}_bench_kernel_stats_final.csv
python3 tools/summarise_profiles.py pmc $g/pmc_${tag}_fetch $g/pmc_${tag}_write $g/pmc_${tag}_sq --out profiles/traffic_latest.json --csv profiles/${name}_pmc_final.csv
python3 tools/summarise_profiles.py stats $g/prof_${tag}_vmf3 profiles/${name}_vmf3_kernel_stats.csv
python3 tools/summarise_profiles.py pmcall $g/pmc_${tag}_vmf3_fetch $g/pmc_${tag}_vmf3_write $g/pmc_${tag}_vmf3_sq profiles/${name}_pmc_vmf3.csv
python3 tools/summarise_profiles.py stats $g/prof_${tag}_kl profiles/${name}_kl_kernel_stats.csv
python3 tools/summarise_profiles.py pmcall $g/pmc_${tag}_kl_fetch $g/pmc_${tag}_kl_sq profiles/${name}_pmc_kl.csv
for c in 1 2 4 5; do cp $g/${tag}_bench_cfg$c.json profiles/${name}_bench_cfg$c.json; done
cp $g/${tag}_bench_default.json profiles/${name}_bench_final.json
grep '^{' $g/bench_$tag.json > profiles/${name}_bench_final_profiled.json
for t in vmf3 kl narrow extra latent; do cp $g/${tag}_bench_$t.md profiles/${name}_bench_$t.md; done
cp $g/${tag}_launches.md profiles/${name}_launches_per_step.md
[ -f $g/parity_margins.json ] && cp $g/parity_margins.json profiles/${name}_parity_margins.json
echo published
