export TMPDIR=/tmp
P=gpurun_out/velprof; rm -rf $P; mkdir -p $P
CMD="python bench.py --system velocity --grid 256 --steps 2 --warmup 1 --kernel-reps 10 --no-cpu"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -d $P/a -o t -- $CMD > $P/a.log 2>&1
python tools/rocprof_summary.py $P/a --out gpurun_out/velprof_sq_a.md --title "velocity 256 SQ a" > /dev/null
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES -d $P/b -o t -- $CMD > $P/b.log 2>&1
python tools/rocprof_summary.py $P/b --out gpurun_out/velprof_sq_b.md --title "velocity 256 SQ b" > /dev/null
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -d $P/c -o t -- $CMD > $P/c.log 2>&1
python tools/rocprof_summary.py $P/c --out gpurun_out/velprof_sq_c.md --title "velocity 256 SQ c" > /dev/null
rm -rf $P/a $P/b $P/c
grep "k_vel_product" gpurun_out/velprof_sq_a.md gpurun_out/velprof_sq_b.md gpurun_out/velprof_sq_c.md | cut -c1-900
head -12 gpurun_out/velprof_sq_a.md | tail -4 | cut -c1-600
tail -3 $P/c.log
