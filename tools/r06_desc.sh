#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "describe or residual_update_inside" 2>&1 | tail -3
python bench.py --steps 2 --warmup 1 --no-cpu --no-secondary --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['runs'])"
