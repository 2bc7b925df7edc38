#!/bin/bash
# Round-5 fault probes, second pass: which instruction FORM of the general loader's normalisation fails (tnp1..6 = -DSEPR_TN_PROBE=i, tnnopk /
# at1nopk = SLP on but the packed-fp32 instructions disabled), and the stand-alone packed-vs-scalar probe
OUT=gpurun_out/r05_faults; mkdir -p $OUT
export TMPDIR=/tmp
{
for v in tnnopk tnp1 tnp2 tnp3 tnp4 tnp5 tnp6; do
  echo "== TN general loader, pad 0, library variant $v"; SEPR_LIB_VARIANT=$v SEPR_TN_FORCE_GEN=1 SEPR_TN_GEN_PAD=0 timeout 200 python tools/probe/tn_fault.py 2>&1 | grep wgrad_norm | grep -v "x3=0"
done
} | tee $OUT/tn_fault2.txt
{
for v in at1 at1nopk; do
  echo "== attention forward determinism at B = 32 x 4 s, library variant '${v:-default}'"
  SEPR_LIB_VARIANT=$v DET_REPS=3 timeout 300 python tools/det_infer.py 2>&1 | grep -E "^rep|Error|error" | cut -c1-200
done
} | tee $OUT/attn_fault2.txt
{ for a in "76000 512" "76000 2048" "100000 256" "100000 2048" "40000 2048"; do timeout 60 tools/probe/pk_opsel $a; done; } 2>&1 | tee $OUT/pk_opsel.txt
