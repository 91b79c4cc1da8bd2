#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r06k; mkdir -p $O
timeout 900 python bench.py --grid 768 --steps 3 --warmup 1 --no-cpu --no-secondary --pmc off > $O/bench768.json 2> $O/bench768.err; echo "rc=$?"; tail -c 2500 $O/bench768.json; tail -3 $O/bench768.err
