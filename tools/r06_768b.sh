#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python bench.py --grid 768 --steps 3 --warmup 1 --no-cpu --no-secondary --pmc off > $O/bench768.json 2> $O/bench768.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06n/bench768.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['iters_per_solve'], d['true_rel_residual'], d['roofline']['solve']['frac'], d['counters'])
PY
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu --no-secondary --pmc off > $O/bench512.json 2> $O/bench512.err; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06n/bench512.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['iters_per_solve'], d['true_rel_residual'], d['roofline']['solve']['frac'], d['placement'])
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multirank_loopback.py tests/test_gpu_bicgstab_gmg.py tests/test_gpu_single_reduction.py -q -m gpu -x 2>&1 | tail -3
