#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r05_e
mkdir -p $OUT
(timeout 1200 python -m pytest tests/test_gpu_multistart.py tests/test_gpu_mcmc.py -x -q 2>&1 | tail -5) > $OUT/pytest.txt
for T in 16 8 4 1; do
MOE_MCMC_THREADS=$T timeout 900 python bench.py --config suggest --no-cpu-baseline > $OUT/suggest_t$T.json 2> $OUT/err.txt
MOE_MCMC_THREADS=$T timeout 900 python bench.py --config suggest_c3 --no-cpu-baseline > $OUT/suggest_c3_t$T.json 2>> $OUT/err.txt
done
cat $OUT/pytest.txt
python - <<PY
import json
for T in (16, 8, 4, 1):
    for f in ("suggest", "suggest_c3"):
        d = json.loads(open("$OUT/%s_t%d.json" % (f, T)).read().strip().splitlines()[-1])
        print(f, "threads", T, round(d["value"], 4), d["best_kg"], d["timeline"]["ms_per_gradient_step"]["median"], d["timeline"]["value_passes"])
PY
