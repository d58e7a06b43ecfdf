#!/bin/bash
# Profile set of one round for the bench command (GPU box): kernel-trace stats of `bench.py --profile-mode`, the PMC passes
# over the same command, their summary with provenance.  usage: bash tools/prof_round.sh TAG [f64|f32]
#   -> gpurun_out/<TAG>prof[_f32]/ ; copy into profiles/ with tools/collect_round.sh TAG [f32]
set -u
TAG=${1:-r03}; DT=${2:-f64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SUF=""; [ "$DT" = "f32" ] && SUF="_f32"
OUT=gpurun_out/${TAG}prof$SUF
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --profile-mode --steps 200 --warmup 20 --dtype $DT > $OUT/stats.log 2>&1 || { echo stats failed; tail -3 $OUT/stats.log; exit 1; }
PMC_OUT=$OUT/pmc PMC_PROG="bench.py --profile-mode --steps 4 --warmup 1 --dtype $DT" bash tools/pmc_passes.sh || exit 1
python3 tools/pmc_summary.py $OUT/pmc > $OUT/pmc_summary.json || exit 1
grep '^{' $OUT/stats.log | tail -1 > $OUT/profile_mode_line.json
ls $OUT
