#!/bin/bash
# round 5: MFMA / VALU / LDS counters of the two dominant kernels (inference: fused GCFN; training, plain bf16: the GCFN backward middle kernel)
OUT=gpurun_out/r05; mkdir -p $OUT
{
echo "# rocprofv3 --pmc passes (kernel-trace only, one group per run), MI355X; sums over all launches of one bench forward (3 forwards: warm-up, step, gate) / one eager bf16 training step"
echo "== inference: gcfn_fused3_kernel<128, 2, 4, 0, false, false>"
SEPR_PIPELINES=1 bash tools/pmc_model.sh "gcfn_fused3_kernel<128, 2, 4, 0, false, false>" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE 2>&1 | grep -v "^rc=\|^{\|utt" | cut -c1-300
SEPR_PIPELINES=1 bash tools/pmc_model.sh "gcfn_fused3_kernel<128, 2, 4, 0, false, false>" SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS 2>&1 | grep -v "^rc=\|^{\|utt" | cut -c1-300
echo "== training (bf16, batch 16, eager step): gcfn_bwd_mid_kernel<1, 2, true>"
PMC_PREC=bf16 bash tools/pmc_train_model.sh "gcfn_bwd_mid_kernel" SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE 2>&1 | grep -v "^rc=" | cut -c1-300
PMC_PREC=bf16 bash tools/pmc_train_model.sh "gcfn_bwd_mid_kernel" SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS 2>&1 | grep -v "^rc=" | cut -c1-300
} | tee $OUT/pmc_dominant_kernels.txt
