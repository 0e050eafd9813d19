#!/bin/bash
# the parity fuzzer on the FINAL library (pool, early inverse, fused finish), plain and with the pool poisoned; digests under poison
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
(timeout 1500 python tools/fuzz_parity.py 240 6161 700 16 4 2>&1 | tail -4) > $O/r05_t_fuzz.txt
(MOE_POOL_POISON=1 timeout 1500 python tools/fuzz_parity.py 120 6262 500 16 4 2>&1 | tail -3) >> $O/r05_t_fuzz.txt
(timeout 900 python tools/fuzz_parity.py 80 6363 300 32 12 2>&1 | tail -3) >> $O/r05_t_fuzz.txt
MOE_POOL_POISON=1 timeout 600 python tools/digest.py > $O/r05_t_digest_poisoned.txt 2>&1
timeout 600 python tools/digest.py > $O/r05_t_digest.txt 2>&1
diff $O/r05_t_digest_poisoned.txt $O/r05_t_digest.txt && echo "digests identical with and without MOE_POOL_POISON" >> $O/r05_t_fuzz.txt
diff $O/r05_t_digest.txt profiles/r05_r_digest_new.txt && echo "digests identical to r05_r (before the scratch / pool changes)" >> $O/r05_t_fuzz.txt
cat $O/r05_t_fuzz.txt
