export TMPDIR=/tmp
for v in 30000 0 100000; do
SEPR_GF_SMALL_ROWS=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('small_rows=$v: %.1f utt/s  parity %.1f dB  gcfn avg %.3f ms  latency_b1 %s' % (r['value'], r['parity_db_vs_golden'], r['roofline']['avg_launch_ms'], r['latency_b1']))"
done
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "test_blocks or test_full_size or test_ragged" -p no:cacheprovider 2>&1 | tail -2
