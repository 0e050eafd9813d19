#!/bin/bash
# kernel fusions on the latency path (kg_y + kg_finish + kg_dir_sum + kg_zc_sum in one launch; counters cleared by build_xs_tab): same
# bits as the previous build, batch-1 latency before / after, suggestion before / after
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r05_r
mkdir -p $OUT
MOE_LIB_PATH=$PWD/variants/libmoe_hip_prev.so timeout 600 python tools/digest.py > $OUT/digest_prev.txt 2>&1
timeout 600 python tools/digest.py > $OUT/digest_new.txt 2>&1
MOE_KG_ZC_IN_FINISH=0 timeout 600 python tools/digest.py > $OUT/digest_new_zc0.txt 2>&1
diff $OUT/digest_prev.txt $OUT/digest_new.txt > $OUT/digest_diff.txt && echo "digests identical (prev vs new)" >> $OUT/digest_diff.txt
diff $OUT/digest_prev.txt $OUT/digest_new_zc0.txt >> $OUT/digest_diff.txt && echo "digests identical (prev vs new, MOE_KG_ZC_IN_FINISH=0)" >> $OUT/digest_diff.txt
cat $OUT/digest_new.txt $OUT/digest_diff.txt
for lib in prev new; do
  if [ $lib = prev ]; then export MOE_LIB_PATH=$PWD/variants/libmoe_hip_prev.so; else unset MOE_LIB_PATH; fi
  echo "== $lib"
  timeout 300 python tools/latency.py 2>&1 | grep -i "kg\|batch" | head -8
  timeout 600 python bench.py --config suggest --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('suggest  %.4f s' % d['value'])"
done > $OUT/latency_ab.txt 2>&1
unset MOE_LIB_PATH
cat $OUT/latency_ab.txt
