#!/bin/bash
# Round 5, GPU call H: the parity fuzzer on the final library (the lane-parked kernel is the default wherever it is built), the K^-1 y
# distance table, the math check over the widened range.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
(timeout 1500 python tools/fuzz_parity.py 160 5151 700 16 4 2>&1 | tail -6) > $O/r05_h_fuzz.txt
(timeout 900 python tools/fuzz_parity.py 60 5252 300 32 12 2>&1 | tail -4) >> $O/r05_h_fuzz.txt
timeout 300 python tools/kinvy_distance.py > $O/r05_h_kinvy.txt 2>&1
mkdir -p /tmp/mc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mathcheck.hip -o /tmp/mc/mathcheck 2>/dev/null && timeout 300 /tmp/mc/mathcheck > $O/r05_h_mathcheck.txt 2>&1
cat $O/r05_h_fuzz.txt $O/r05_h_kinvy.txt $O/r05_h_mathcheck.txt
