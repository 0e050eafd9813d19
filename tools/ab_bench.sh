#!/bin/bash
# tools/ab_bench.sh <tag> <tag> ...: the headline bench line (value, kernel ms) of each variants/libmoe_hip_<tag>.so, twice, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2; do for tag in "$@"; do
  MOE_LIB_PATH=$PWD/variants/libmoe_hip_$tag.so python bench.py --no-traffic --no-cpu-baseline --no-batch1 --no-extras --no-determinism ${AB_ARGS} 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$tag', round(d['value'],1), 'evals/s  frac', round(d['roofline']['frac'],4), 'kernel ms', round(d['roofline']['avg_launch_ms'],3))"
done; done
