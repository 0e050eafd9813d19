#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out/r05_z2
mkdir -p $O
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) > $O/pytest.txt; cat $O/pytest.txt
timeout 600 python tools/digest.py > $O/digest.txt 2>&1
diff $O/digest.txt profiles/r05_r_digest_prev.txt && echo "digests identical to the pre-fusion library" | tee -a $O/digest.txt
for c in suggest suggest_c3; do timeout 600 python bench.py --config $c --steps 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$c', d['value'], d['timeline']['ms_per_gradient_step']['median'])"; done | tee $O/suggest.txt
