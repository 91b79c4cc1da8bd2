#!/bin/bash
# tools/velocity_profiles.sh TAG: the 256^3 velocity solves under rocprofv3 -- kernel trace, then FETCH_SIZE and WRITE_SIZE in passes of
# their own -- summaries in gpurun_out/TAG/velocity256_*.md
TAG=${1:-vel}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
CMD="python bench.py --system velocity --grid 256 --steps 2 --warmup 1 --kernel-reps 4 --no-cpu --pmc off"
P=/tmp/velprof_$TAG; rm -rf $P
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- $CMD > $P.trace.log 2>&1
python tools/rocprof_summary.py $P/trace --out gpurun_out/$TAG/velocity256_kernel_trace.md --title "$TAG: $CMD, rocprofv3 --kernel-trace --stats" > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv --pmc $C -d $P/pmc_$C -o t -- $CMD > $P.$C.log 2>&1
  python tools/rocprof_summary.py $P/pmc_$C --out gpurun_out/$TAG/velocity256_pmc_$C.md --title "$TAG: velocity 256^3 --pmc $C (separate pass; KiB per dispatch; FETCH_SIZE to be doubled on gfx950)" > /dev/null
done
tail -1 $P.trace.log | cut -c1-400
ls gpurun_out/$TAG
