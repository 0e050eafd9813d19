#!/bin/bash
# finish stage reworked (multi-block): digests vs the pre-fusion library, batch-1 timeline, C5 / g=12 / suggest
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_u
mkdir -p $O
timeout 600 python tools/digest.py > $O/digest_new.txt 2>&1
diff $O/digest_new.txt profiles/r05_r_digest_prev.txt > $O/digest_diff.txt && echo "digests identical to the pre-fusion library" >> $O/digest_diff.txt
cat $O/digest_diff.txt
bash tools/kg1_timeline.sh 2>&1 | tail -14 > $O/kg1_timeline.txt
cat $O/kg1_timeline.txt
timeout 300 python tools/latency.py 2>&1 | grep "C3 KG\|last kernel" > $O/latency.txt; cat $O/latency.txt
timeout 900 python bench.py --config C5 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('C5', d['value'], d.get('kernel_ms_per_eval'))" > $O/c5.txt
timeout 900 python bench.py --config C5 --derivs 12 --steps 3 --warmup 1 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('C5 g12', d['value'], d.get('kernel_ms_per_eval'))" >> $O/c5.txt
timeout 600 python bench.py --config suggest --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('suggest', d['value'])" >> $O/c5.txt
cat $O/c5.txt
