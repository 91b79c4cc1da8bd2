cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
cp petibm_amd/lib/var_$v.so petibm_amd/lib/libpetibm_amd.so
CMD="python bench.py --steps 1 --warmup 1 --no-cpu --no-secondary --pmc off --kernel-reps 2"
P=/tmp/ldsprof_$v; rm -rf $P
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $P/b -o t -- $CMD > $P.b.log 2>&1
python tools/rocprof_summary.py $P/b --out gpurun_out/ldsb_$v.md --title "$v" > /dev/null
done
