#!/bin/bash
# round-5 call: the fault fixes on the device - probe forms, general loader without the pad, attention with the mask pass, parity + speed
OUT=gpurun_out/r05; mkdir -p $OUT gpurun_out/r05_faults
export TMPDIR=/tmp
{ for a in "76000 512 96 0" "40000 2048 96 0"; do timeout 60 tools/probe/pk_opsel $a; done; } 2>&1 | tee gpurun_out/r05_faults/pk_opsel_forms2.txt
SEPR_TN_FORCE_GEN=1 timeout 200 python tools/probe/tn_fault.py 2>&1 | grep wgrad_norm | tee gpurun_out/r05_faults/tn_fault_fixed.txt
DET_REPS=6 timeout 300 python tools/det_infer.py 2>&1 | grep -E "^rep|Error|error" | cut -c1-300 | tee gpurun_out/r05_faults/attn_fixed.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "blocks or e2e_golden or pit_si_snr or full_size_batch or pipelines or ragged" 2>&1 | tail -4 | cut -c1-300
timeout 900 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -x -k "ega_train or general_loader or wgrad or front_and_heads or downconv_split" 2>&1 | tail -4 | cut -c1-300
bash tools/prof_kernel.sh "relattn_x3|gcfn_fused3_kernel<128, 2|cla_|dwconv_same" "" "SEPR_NOP=1" 2>&1 | tail -12
timeout 300 python bench.py --no-cpu-baseline --no-alt-precision --pmc off > $OUT/bench_infer.json 2> $OUT/bench_infer.err; python -c "
import json; r=json.loads(open('$OUT/bench_infer.json').read().strip().split(chr(10))[-1]); print('infer', r['value'], r['ms_per_step'], r['parity_ok'], r['pit_si_snr_max_abs_delta_db'], r['roofline']['avg_launch_ms'], (r.get('single_pipeline') or {}).get('value'))"
