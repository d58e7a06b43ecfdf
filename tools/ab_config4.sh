#!/bin/bash
# A/B of library variants on BASELINE config 4's per-GPU shard: per-kernel average durations from rocprofv3.
#   tools/ab_config4.sh NAME...   (NAME = product | a variant built by tools/build_variant.sh)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/abc4
mkdir -p $OUT
for v in "$@"; do
  if [ "$v" = product ]; then unset MP_HIP_LIB; else export MP_HIP_LIB=$PWD/ab/libmp_$v.so; fi
  rm -rf $OUT/$v
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -- python3 tools/run_config4.py --steps 4 > $OUT/$v.log 2>&1 || { echo "$v failed"; tail -3 $OUT/$v.log; exit 1; }
  f=$(ls -t $OUT/$v/*/*kernel_stats.csv | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "eval_kernel" in n or "clearance_kernel" in n:
        print(f'  {n[:70]:70s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
done
