cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python bench.py --system velocity --grid 256 --steps 2 --warmup 1 --kernel-reps 4 --no-cpu"
P=/tmp/velprof; rm -rf $P
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $P/b -o t -- $CMD > $P.b.log 2>&1
python tools/rocprof_summary.py $P/b --out gpurun_out/velb.md --title "velocity 256" > /dev/null
tail -2 $P.b.log | cut -c1-300
