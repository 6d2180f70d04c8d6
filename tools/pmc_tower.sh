cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 python3 $R/tools/tower_ablation.py 2>&1 | tail -14
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_$n -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --roofline-iters 1 > $R/gpurun_out/pmc_$n.log 2>&1
  tail -2 $R/gpurun_out/pmc_$n.log | cut -c1-200
done
ls $R/gpurun_out/
