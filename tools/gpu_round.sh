#!/bin/bash
# One GPU evidence session (r6 form): full GPU suite + smoke, the default bench line (every BASELINE config inside it), the same
# bench under rocprofv3 kernel stats, C5 / its stretch point / a whole suggestion as their own lines, build times -> gpurun_out/<tag>_*
# usage: tools/gpu_round.sh <tag> [notests]      (copy what is to be judged from gpurun_out/ into profiles/)
TAG="${1:-r06_final}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$2" != "notests" ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.txt
  tail -4 gpurun_out/${TAG}_pytest_gpu.txt
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a gpurun_out/${TAG}_pytest_gpu.txt
fi
timeout 600 python bench.py 2> gpurun_out/${TAG}_bench.err > gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --config C5 --steps 5 --warmup 2 --no-batch1 2> /dev/null > gpurun_out/${TAG}_c5_bench.json
timeout 300 python bench.py --config C5 --derivs 12 --steps 3 --warmup 1 --no-batch1 --no-cpu-baseline 2> /dev/null > gpurun_out/${TAG}_c5g12_bench.json
timeout 300 python bench.py --config suggest 2> /dev/null > gpurun_out/${TAG}_suggest_bench.json
timeout 300 python bench.py --config suggest_c3 --no-cpu-baseline 2> /dev/null > gpurun_out/${TAG}_suggest_c3_bench.json
timeout 200 python tools/chol_time.py 3 12 2>&1 | grep "two-level" > gpurun_out/${TAG}_chol_time.txt
# the randomised parity sweep against the unmodified reference under the r6 rule (an end point off by > 1e-8 only where the two CPU codes
# themselves disagree): three seeds, and one on a poisoned pool
{ timeout 900 python tools/fuzz_parity.py 150 606 300 16 4 | tail -3; timeout 1500 python tools/fuzz_parity.py 60 607 700 32 12 | tail -3; timeout 600 python tools/fuzz_ensemble.py 80 2026 400 | tail -3; MOE_POOL_POISON=1 timeout 900 python tools/fuzz_parity.py 100 608 300 16 4 | tail -3; } > gpurun_out/${TAG}_fuzz.txt 2>&1; tail -3 gpurun_out/${TAG}_fuzz.txt
cd /tmp
rm -rf /tmp/prof_${TAG}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG} -o kg -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras --no-batch1 --no-determinism > $R/gpurun_out/${TAG}_bench_under_rocprof.json 2> /dev/null
cd $R
cp "$(find /tmp/prof_${TAG} -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_bench_kernel_stats.csv 2>/dev/null
head -4 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-150
python - <<PY
import json
def last(p):
    try: return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as e: return {"error": str(e)}
d=last("gpurun_out/${TAG}_bench.json")
print("headline", d.get("value"), (d.get("roofline") or {}).get("frac"), "batch1", (d.get("batch1") or {}).get("ms_per_eval"))
c=d.get("configs") or {}
print({k:(c.get(k,{}).get("evals_per_s") or c.get(k,{}).get("s_per_suggestion") or c.get(k,{}).get("value_grad_us") or c.get(k,{}).get("mean_1pt_us")) for k in ("C1","C2","C5","suggest")}, (c.get("C5") or {}).get("frac"), (c.get("C5") or {}).get("traffic_over_table"))
for name in ("c5","c5g12","suggest","suggest_c3"):
    x=last("gpurun_out/${TAG}_%s_bench.json" % name); print(name, x.get("value"), (x.get("roofline") or {}).get("frac"))
print(open("gpurun_out/${TAG}_chol_time.txt").read())
PY
