#!/bin/bash
# Round-5 evidence run: full GPU suite, smoke, bench (C3 with every extra; C5; C5 g = 12; suggest, suggest_c3), the same bench under
# rocprofv3 kernel stats, K(X,X) probe kernel stats per size, per-config table, latencies (API and GPP level), GP build / LL times.
#   usage: tools/gpu_round5.sh <tag>
TAG="${1:-r05_final}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"
O=$R/gpurun_out
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $O/${TAG}_pytest_gpu.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> $O/${TAG}_pytest_gpu.txt
# the same suite with every recycled device / pinned block poisoned (0xFF bytes) before it is pooled
MOE_POOL_POISON=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/${TAG}_pytest_gpu_poisoned.txt
timeout 300 python -c "from cornell_moe_amd import selftest; print('failures', selftest.run(verbose=True))" > $O/${TAG}_selftest.txt 2>&1
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
timeout 900 python bench.py --config C5 --steps 6 --warmup 2 --no-cpu-baseline > $O/${TAG}_c5_bench.json 2> $O/${TAG}_c5_bench.err
timeout 900 python bench.py --config C5 --derivs 12 --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_c5g12_bench.json 2> $O/${TAG}_c5g12_bench.err
timeout 900 python bench.py --config suggest > $O/${TAG}_suggest_bench.json 2> $O/${TAG}_suggest.err
timeout 900 python bench.py --config suggest_c3 > $O/${TAG}_suggest_c3_bench.json 2> $O/${TAG}_suggest_c3.err
MOE_BENCH_BACKEND=gloo MOE_BENCH_SHARE_GPU=1 timeout 900 python bench.py --config suggest --gpus 8 --no-cpu-baseline > $O/${TAG}_suggest_w8_bench.json 2> $O/${TAG}_suggest_w8.err
cd /tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o kg -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-extras --no-batch1 --no-determinism > $O/${TAG}_bench_under_rocprof.json 2> $O/prof.err
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/${TAG}_bench_kernel_stats.csv
rm -rf $O/prof
# K(X,X) probe alone, one size per process: the GP's constructor + `repeat` launches of the same shape
for spec in "3 8000" "12 26000"; do
  set -- $spec
  rm -rf $O/kxx
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kxx -o kxx -- python $R/tools/kxx_one.py $1 > $O/${TAG}_kxx_N$2.txt 2> $O/kxx.err
  python - "$(find $O/kxx -name '*kernel_stats.csv' | head -1)" >> $O/${TAG}_kxx_N$2.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "cov_build" in r["Name"]:
        print("%-70s calls %5s  avg %10.1f ns  min %10s  max %10s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r.get("MinNs"), r.get("MaxNs")))
PY
  rm -rf $O/kxx
done
cd $R
timeout 600 python tools/run_configs.py > $O/${TAG}_configs.json 2> $O/${TAG}_configs.err
timeout 300 python tools/latency.py > $O/${TAG}_latency.txt 2>&1
timeout 200 python tools/ei_loop.py 300 >> $O/${TAG}_latency.txt 2>&1
timeout 300 python tools/dkg_sweep.py > $O/${TAG}_dkg_sweep.txt 2>&1
timeout 200 python tools/chol_time.py > $O/${TAG}_chol_time.txt 2>&1
timeout 200 python tools/ll_time.py > $O/${TAG}_ll_time.txt 2>&1
bash tools/kg1_timeline.sh > $O/${TAG}_kg1_timeline.txt 2>&1
bash tools/build_timeline.sh 3 > $O/${TAG}_build_timeline.txt 2>&1
timeout 300 python tools/build_sweep.py 4000 8000 16000 > $O/${TAG}_build_sweep.txt 2>&1
tail -3 $O/${TAG}_pytest_gpu.txt; cut -c1-300 $O/${TAG}_bench.json; tail -3 $O/${TAG}_bench.err; head -5 $O/${TAG}_bench_kernel_stats.csv
cat $O/${TAG}_pytest_gpu_poisoned.txt $O/${TAG}_selftest.txt
cat $O/${TAG}_kxx_N8000.txt $O/${TAG}_kxx_N26000.txt $O/${TAG}_latency.txt $O/${TAG}_chol_time.txt $O/${TAG}_ll_time.txt $O/${TAG}_build_sweep.txt
for f in c5 c5g12 suggest suggest_c3 suggest_w8; do echo "== $f"; cut -c1-400 $O/${TAG}_${f}_bench.json; done
