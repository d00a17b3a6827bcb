#!/bin/bash
# round 4, first GPU call: new pair tests + encoder tests, then the pair-step probe over the small-graph thresholds
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pair.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -25 > gpurun_out/r4_first_tests.log
cat gpurun_out/r4_first_tests.log
: > gpurun_out/r4_pair_probe.jsonl
python tools/pair_probe.py --batched 1 --single >> gpurun_out/r4_pair_probe.jsonl 2>gpurun_out/r4_pair_probe.err
python tools/pair_probe.py --batched 0 >> gpurun_out/r4_pair_probe.jsonl 2>>gpurun_out/r4_pair_probe.err
for S in 32768 131072; do for U in 2 4; do
JMAC_SMALL_ITEMS=$S JMAC_FWD_U=$U python tools/pair_probe.py --batched 1 >> gpurun_out/r4_pair_probe.jsonl 2>>gpurun_out/r4_pair_probe.err
done; done
cat gpurun_out/r4_pair_probe.jsonl
tail -5 gpurun_out/r4_pair_probe.err
