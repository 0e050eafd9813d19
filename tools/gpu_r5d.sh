#!/bin/bash
# Round 5, GPU call D: full GPU suite on the current library, the default bench line, the suggest benchmarks, K(X,X) probe stats.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT="$PWD"
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r05_d
mkdir -p $OUT
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > $OUT/pytest_gpu.txt
timeout 900 python bench.py --config suggest > $OUT/suggest_bench.json 2> $OUT/suggest_err.txt
timeout 900 python bench.py --config suggest_c3 > $OUT/suggest_c3_bench.json 2> $OUT/suggest_c3_err.txt
MOE_BENCH_BACKEND=gloo MOE_BENCH_SHARE_GPU=1 timeout 900 python bench.py --config suggest --gpus 2 --no-cpu-baseline > $OUT/suggest_w2_bench.json 2> $OUT/suggest_w2_err.txt
MOE_BENCH_BACKEND=gloo MOE_BENCH_SHARE_GPU=1 timeout 900 python bench.py --config suggest --gpus 8 --no-cpu-baseline > $OUT/suggest_w8_bench.json 2> $OUT/suggest_w8_err.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench_err.txt
cat $OUT/pytest_gpu.txt
for f in suggest suggest_c3 suggest_w2 suggest_w8; do echo "== $f"; tail -3 $OUT/${f}_err.txt; python - <<PY
import json
try:
    d = json.loads(open("$OUT/${f}_bench.json").read().strip().splitlines()[-1])
    print({k: d[k] for k in ("value", "unit", "n_gpus", "found", "best_kg", "all_ranks_agree", "speedup_vs_cpu", "max_abs_diff_vs_reference_point") if k in d})
    print(d.get("cpu_baseline")); print(d["timeline"]["ms_per_gradient_step"], d["timeline"]["gradient_steps"], d["timeline"]["value_passes"], d.get("exchange"))
except Exception as e:
    print("no json:", e)
PY
done
python - <<PY
import json
d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["frac"], d["roofline"]["executed_frac"], d["roofline"]["kernel"], d["roofline"]["traffic"], d["roofline"]["in_kernel"])
print(d["roofline_cov_build"]["frac"], d["roofline_cov_build"]["traffic"], d.get("batch1"), d.get("batch8"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["one_core_evals_per_s"], [(e["threads"], round(e["evals_per_s"],3), round(e["linear_in_M_evals_per_s"],3)) for e in d["cpu_baseline"]["thread_sweep"]])
PY
tail -5 $OUT/bench_err.txt
