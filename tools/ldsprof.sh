cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 1 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2"
P=/tmp/ldsprof; rm -rf $P
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $P/a -o t -- $CMD > $P.a.log 2>&1
python tools/rocprof_summary.py $P/a --out gpurun_out/lds_a.md --title "SQ activity" > /dev/null
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $P/b -o t -- $CMD > $P.b.log 2>&1
python tools/rocprof_summary.py $P/b --out gpurun_out/lds_b.md --title "SQ instruction counts" > /dev/null
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $P/c -o t -- $CMD > $P.c.log 2>&1
python tools/rocprof_summary.py $P/c --out gpurun_out/lds_c.md --title "SQ waits" > /dev/null
ls -la gpurun_out/lds_*.md
