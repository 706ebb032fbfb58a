mkdir -p gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z0-9_]+" | sort -u > $R/gpurun_out/pmc/sq_counters.txt
for t in 1 0; do for k in conv640 big; do
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE"; do
    tag=$(echo $pass | cut -c4-12)
    rm -rf /tmp/p_$t_$k; VCX_GEMM_TUNE=$t timeout 120 rocprofv3 --kernel-trace --pmc $pass -d /tmp/p_${t}_${k}_$tag -o pmc -- python $R/tools/one_gemm.py $k 4 > /tmp/p.log 2>&1 || tail -3 /tmp/p.log
    echo "== tune $t $k $tag" >> $R/gpurun_out/pmc/summary.txt
    python $R/tools/pmc_summary.py $(find /tmp/p_${t}_${k}_$tag -name "*.db" | head -1) >> $R/gpurun_out/pmc/summary.txt 2>&1
  done; done; done
cat $R/gpurun_out/pmc/summary.txt | grep -v "^$" | head -150
