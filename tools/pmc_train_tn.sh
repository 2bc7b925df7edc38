#!/bin/bash
# HBM traffic of the weight-gradient contraction (gemm_tn_kernel) in the training step: FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes (kernel-trace only), one eager batch-16 step per precision (a replayed hipGraph is opaque to the counters), summed over the
# contraction launches and divided by their count.  Writes gpurun_out/pmc_gemm_tn.json (copy to profiles/).
export TMPDIR=/tmp
OUT=$PWD/gpurun_out; mkdir -p $OUT
ROOT=$PWD
for prec in bf16x3 bf16; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${prec}_$c
    (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "gemm_tn_kernel" --output-format csv -d /tmp/pmc_${prec}_$c -o t -- \
       python $ROOT/bench.py --mode train --precision $prec --batch 16 --steps 1 --warmup 0 --no-train-graphs > /tmp/pmc_${prec}_$c.log 2>&1)
    tail -2 /tmp/pmc_${prec}_$c.log | cut -c1-600
  done
done
python3 - <<'PY' | tee $OUT/pmc_gemm_tn.json
import csv, glob, json
out = {"kernel": "gemm_tn_kernel (all instantiations)", "source": "tools/pmc_train_tn.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, kernel-trace only; bench.py --mode train --batch 16 --steps 1 --warmup 0 --no-train-graphs (eager: one timed step + the roofline step = 2 steps of launches)",
       "corrections": "FETCH_SIZE x2 (MI355X_MICROARCH.md HBM section: gfx950 reports half the bytes of 16 B/lane streaming reads) - upper bound for the 2-byte-source instantiations; WRITE_SIZE as reported; both counters are in KiB"}
for prec in ("bf16x3", "bf16"):
    rec = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fs = glob.glob(f"/tmp/pmc_{prec}_{c}/**/*counter_collection.csv", recursive=True)
        if not fs:
            continue
        rows = [r for r in csv.DictReader(open(fs[0])) if r["Counter_Name"] == c]
        rec[c + "_sum_KiB"] = sum(float(r["Counter_Value"]) for r in rows)
        rec["launches"] = len(rows)
    if "FETCH_SIZE_sum_KiB" in rec and "WRITE_SIZE_sum_KiB" in rec and rec["launches"]:
        n = rec["launches"]
        rec["fetch_bytes_per_launch"] = round(2 * 1024 * rec["FETCH_SIZE_sum_KiB"] / n)
        rec["write_bytes_per_launch"] = round(1024 * rec["WRITE_SIZE_sum_KiB"] / n)
        rec["hbm_bytes_per_launch"] = rec["fetch_bytes_per_launch"] + rec["write_bytes_per_launch"]
    out[prec] = rec
print(json.dumps(out, indent=1))
PY
