set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee gpurun_out/r03_b_pytest_gpu.txt
python - << 'PY' 2>&1 | tee gpurun_out/r03_b_probes.txt
import sys; sys.path.insert(0, '.')
from cornell_moe_amd import api
print("fp64 sustained FMA rate: %.1f TFLOP/s" % api.fp64_rate())
print("fp64 sustained FMA rate: %.1f TFLOP/s" % api.fp64_rate())
print(api.kxx_build_probe(print))
PY
timeout 300 python tools/chol_time.py 2>&1 | tail -12 | tee gpurun_out/r03_b_chol_time.txt
bash tools/mc_overhead.sh gpurun_out/mc_overhead 2>&1 | tail -12
