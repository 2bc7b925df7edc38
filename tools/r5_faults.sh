#!/bin/bash
# Round-5 fault probes in ONE gpurun call (tools/probe/README.md): every configuration is its own process (the switches are latched once).
#   library variants (built here, shipped as .so): tnnoslp / tnwz = sepr_gemm_tn.hip with -fno-slp-vectorize / -mllvm -amdgpu-waitcnt-forcezero;
#   at1 / at2 = sepr_attention.hip with -DSEPR_AT_MASKPASS=1 / 2, at1noslp / at1wz = at1 + the two flags
OUT=gpurun_out/r05_faults; mkdir -p $OUT
export TMPDIR=/tmp
{
echo "== TN control: product configuration (pad 16384, !GEN loader)"; timeout 200 python tools/probe/tn_fault.py 2>&1 | grep wgrad_norm
echo "== TN general loader, pad 16384 (one workgroup per CU)"; SEPR_TN_FORCE_GEN=1 timeout 200 python tools/probe/tn_fault.py 2>&1 | grep wgrad_norm
echo "== TN general loader, pad 0 (two workgroups per CU)"; SEPR_TN_FORCE_GEN=1 SEPR_TN_GEN_PAD=0 timeout 200 python tools/probe/tn_fault.py 2>&1 | grep wgrad_norm
echo "== same, 256 workgroups per launch (SEPR_TN_WGS=256: at most one per CU although two would fit)"; SEPR_TN_WGS=256 SEPR_TN_FORCE_GEN=1 SEPR_TN_GEN_PAD=0 timeout 200 python tools/probe/tn_fault.py 2>&1 | grep wgrad_norm
for v in tnnoslp tnwz; do
  echo "== TN general loader, pad 0, library variant $v"; SEPR_LIB_VARIANT=$v SEPR_TN_FORCE_GEN=1 SEPR_TN_GEN_PAD=0 timeout 200 python tools/probe/tn_fault.py 2>&1 | grep wgrad_norm
done
} | tee $OUT/tn_fault.txt
{
for v in "" at1 at2 at1noslp at1wz; do
  echo "== attention forward determinism at B = 32 x 4 s, library variant '${v:-default}'"
  SEPR_LIB_VARIANT=$v DET_REPS=4 timeout 300 python tools/det_infer.py 2>&1 | grep -E "^rep|Error|error" | cut -c1-400
done
} | tee $OUT/attn_fault.txt
