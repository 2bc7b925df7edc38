#!/bin/bash
# round 6: where do the waves of the GCFN backward middle kernel (and of the new contraction kernel) spend their cycles?  two PMC passes over one eager bf16 step
OUT=gpurun_out; mkdir -p $OUT
{
PMC_PREC=bf16 bash tools/pmc_train_model.sh "gcfn_bwd_mid_kernel|gemm_tnd_kernel" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE 2>&1 | grep -v "^rc=" | cut -c1-400
PMC_PREC=bf16 bash tools/pmc_train_model.sh "gcfn_bwd_mid_kernel|gemm_tnd_kernel" SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES 2>&1 | grep -v "^rc=" | cut -c1-400
} | tee $OUT/r6_pmc_mid.txt
