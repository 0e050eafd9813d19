#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_w
mkdir -p $O
timeout 600 python tools/digest.py > $O/digest_new.txt 2>&1
diff $O/digest_new.txt profiles/r05_r_digest_prev.txt > $O/digest_diff.txt && echo "digests identical to the pre-fusion library" >> $O/digest_diff.txt
cat $O/digest_diff.txt
bash tools/kg1_timeline.sh 2>&1 | tail -26 > $O/kg1_timeline.txt
cat $O/kg1_timeline.txt
timeout 300 python tools/latency.py 2>&1 | grep "C3 KG\|last kernel" > $O/latency.txt; cat $O/latency.txt
for c in suggest suggest_c3; do timeout 600 python bench.py --config $c --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$c', d['value'])"; done > $O/suggest.txt
cat $O/suggest.txt
