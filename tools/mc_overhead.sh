#!/bin/bash
# Where the MC kernel's VALU instructions go: SQ_INSTS_VALU per sample at {8, 16} tiles x {3, 6} inner GD steps
# (instructions = a + b steps + (c + d steps) tiles: b = per-step work outside the tile loops, a = per-sample set-up).
# usage: tools/mc_overhead.sh <outdir>
OUT="${1:-gpurun_out/mc_overhead}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/$OUT
export TMPDIR=/tmp
cd /tmp
for cfg in "C3:n=508,steps=3" "C3:n=508,steps=6" "C3:n=1020,steps=3" "C3:n=1020,steps=6"; do
  tag=$(echo $cfg | tr ':=,' '___')
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/$OUT/$tag -o p -- python $R/tools/prof_kg.py "$cfg" 8 2 > $R/$OUT/$tag.txt 2>&1
done
cd $R
python tools/mc_overhead.py $OUT | tee $OUT/summary.txt
