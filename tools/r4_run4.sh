timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "blocks or e2e_golden or full_size_batch or pit_si_snr_gate_bench or ragged" 2>&1 | tail -4 | cut -c1-600
timeout 600 python -m pytest tests/test_train_gpu.py -m gpu -q -p no:cacheprovider -k "ega_train or cla_train_full_size or dropout_contract" 2>&1 | tail -4 | cut -c1-600
WHAT=infer,large bash tools/r4_profiles.sh
