cd "${GRAFT_REPO_ROOT:-/root/repo}"
for tr in "" 4; do
  echo "MOE_KG_TR=$tr"
  MOE_KG_TR=$tr timeout 600 python bench.py --config C5 --steps 6 --warmup 2 --no-cpu-baseline --no-traffic --no-extras --no-determinism 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('  value %.1f  frac %.4f  kernel ms/eval %s  batch1 %.1f' % (d['value'], d['roofline']['frac'], {k: round(v, 3) for k, v in d['kernel_ms_per_eval'].items()}, d['batch1']['value']))"
done | tee gpurun_out/r03_h_c5.txt
