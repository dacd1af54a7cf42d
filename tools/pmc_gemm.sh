mkdir -p gpurun_out/pmc_r1; R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc_r1/counters_list.txt 2>&1 || true
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_r1/$tag -o g -- python $R/tools/gemm_bench.py --cfgs 0 --iters 3 --filter "L1 conv3x3 1280->640" > $R/gpurun_out/pmc_r1/$tag.out 2>&1
done
cd $R; find gpurun_out/pmc_r1 -name "*.csv" | head -20; du -sh gpurun_out/pmc_r1
