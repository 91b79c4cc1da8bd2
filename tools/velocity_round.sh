#!/bin/bash
# tools/velocity_round.sh TAG: the velocity-heavy measurements of a round in one gpurun call -- bench.py --system velocity (256^3),
# its kernel trace and separate FETCH_SIZE / WRITE_SIZE passes, the config-5-size heaving plate, the first 300 steps of Taylor-Green 256^3
TAG=${1:-r03}
export TMPDIR=/tmp
O=gpurun_out/$TAG; mkdir -p $O
CMD="python bench.py --system velocity --grid 256 --steps 5 --warmup 2 --no-cpu"
$CMD > $O/velocity256.json 2> $O/velocity256.err
P=$O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -o t -- $CMD > $O/trace.log 2>&1
python tools/rocprof_summary.py $P/trace --out $O/velocity256_kernel_trace.md --title "$TAG $CMD, rocprofv3 --kernel-trace --stats" > /dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $C -d $P/pmc_$C -o t -- $CMD > $O/pmc_$C.log 2>&1
  python tools/rocprof_summary.py $P/pmc_$C --out $O/velocity256_pmc_$C.md --title "$TAG $CMD --pmc $C (separate pass; KiB per dispatch; FETCH_SIZE to be doubled on gfx950)" > /dev/null
done
rm -rf $P
timeout 900 python tools/config5_heaving_plate.py --steps 35 > $O/config5.txt 2>&1; tail -3 $O/config5.txt
timeout 900 python examples/python/taylor_green_3d.py --nt 300 --every 100 > $O/taylor_green_300.txt 2>&1; tail -3 $O/taylor_green_300.txt
tail -c 600 $O/velocity256.json
