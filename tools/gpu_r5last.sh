#!/bin/bash
# the library as committed last: the default bench line and the two suggestion benches
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=$PWD/gpurun_out
timeout 600 python bench.py > $O/r05_last_bench.json 2> $O/r05_last_bench.err
timeout 300 python bench.py --config suggest > $O/r05_last_suggest_bench.json 2> $O/r05_last_suggest.err
timeout 600 python bench.py --config suggest_c3 > $O/r05_last_suggest_c3_bench.json 2> $O/r05_last_suggest_c3.err
cut -c1-400 $O/r05_last_bench.json; cut -c1-300 $O/r05_last_suggest_bench.json; cut -c1-300 $O/r05_last_suggest_c3_bench.json
