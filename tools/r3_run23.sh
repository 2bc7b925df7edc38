#!/bin/bash
# round 3, call 23: what bounds the two largest training kernels (PMC, two counter groups, kernel-trace only)
bash tools/pmc_train_model.sh "gcfn_bwd_mid_kernel|gemm_tn_kernel<1, false" GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU 2>&1 | tail -6 | cut -c1-600
bash tools/pmc_train_model.sh "gcfn_bwd_mid_kernel|gemm_tn_kernel<1, false" SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE 2>&1 | tail -6 | cut -c1-600
bash tools/pmc_train_model.sh "gcfn_bwd_mid_kernel|gemm_tn_kernel<1, false" SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM 2>&1 | tail -6 | cut -c1-600
