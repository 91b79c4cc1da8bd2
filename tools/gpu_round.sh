#!/bin/bash
# one gpurun call: tests (optional selection), the default bench, rocprofv3 kernel trace + separate PMC passes
# usage: tools/gpu_round.sh <tag> [pytest-args...]
set -u
TAG=${1:-r02}; shift || true
mkdir -p gpurun_out/$TAG
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q -v -s "$@" > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/$TAG/pytest.log
  tail -5 gpurun_out/$TAG/pytest.log
fi
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  timeout 900 python bench.py --steps 10 --warmup 2 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err; echo "bench rc=$?"
  tail -c 3000 gpurun_out/$TAG/bench.json
fi
if [ "${SKIP_PROF:-0}" != "1" ]; then
  P=gpurun_out/$TAG/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- python bench.py --steps 3 --warmup 1 --no-cpu --no-secondary --pmc off > $P.trace.log 2>&1
  python tools/rocprof_summary.py $P/trace --out gpurun_out/$TAG/kernel_trace.md --title "$TAG bench.py --steps 3 --warmup 1 --no-cpu --no-secondary, rocprofv3 --kernel-trace --stats" || true
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $P/pmc_$C -o t -- python bench.py --steps 1 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2 > $P.$C.log 2>&1
    python tools/rocprof_summary.py $P/pmc_$C --out gpurun_out/$TAG/pmc_$C.md --title "$TAG --pmc $C (separate pass; KiB per dispatch; FETCH_SIZE to be doubled on gfx950)" || true
  done
  timeout 900 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VALU -d $P/pmc_SQ -o t -- python bench.py --steps 1 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2 > $P.SQ.log 2>&1
  python tools/rocprof_summary.py $P/pmc_SQ --out gpurun_out/$TAG/pmc_SQ.md --title "$TAG --pmc SQ counters (separate pass)" || true
  rm -rf $P   # raw CSVs are large; the summaries stay
  ls -la gpurun_out/$TAG
fi
