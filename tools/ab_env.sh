#!/bin/bash
# A/B of environment knobs on tools/shape_bench.py: tools/ab_env.sh "SHAPES" "VAR=val VAR2=val" "VAR=val" ...
# (each quoted argument after the shapes is one configuration; "-" = no variables)
shapes=$1; shift
for cfg in "$@"; do
  echo "== $cfg"
  ( if [ "$cfg" != "-" ]; then export $cfg; fi; MP_SHAPE_REPS=${MP_SHAPE_REPS:-40} python3 tools/shape_bench.py "$shapes" 2>&1 | grep "grad=True" )
done
