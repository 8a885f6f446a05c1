#!/bin/bash
# round 4, call R: hardened parity evidence -- configs[1] end to end at full depth with 32 teacher-forced decode steps (decidable steps
# counted), the outlier-channel fixture at full width, tensor-parallel shards against the measured bf16 floor
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out; mkdir -p $O
( timeout 2400 python -m pytest tests/test_gpu_parity_full.py::test_outlier_channels_tower_stc_four_decoder_layers tests/test_gpu_parity_full.py::test_configs1_full_depth_end_to_end tests/test_gpu_tp.py -m gpu -q -p no:cacheprovider -s 2>&1 ) > $O/r04r_pytest_parity.log 2>&1
grep -E "parity-full|tp-local|passed|failed|Error" $O/r04r_pytest_parity.log | cut -c1-200 | tail -60
