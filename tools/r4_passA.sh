#!/bin/bash
cd /root/repo
python -m pytest tests/test_gpu_layer.py tests/test_gpu_ja_oracle.py tests/test_gpu_bf16.py tests/test_gpu_fullsize.py tests/test_gpu_encoder.py tests/test_gpu_graph.py tests/test_gpu_union_real.py -x -q 2>&1 | tail -2
bash tools/step_profile2.sh gpurun_out/r4_step_ja --ja > /dev/null 2>&1; grep -n "bwd_dst\|bwd_gather\|span\|rel_attn_fwd" gpurun_out/r4_step_ja/step_breakdown.txt
python tools/ja_sweep.py ja-real 2>&1 | tail -1
python tools/union_agg_probe.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('union f32 %.1f us %.3f  bf16 %.1f us %.3f  bwd %.1f us' % (d['fwd_f32_us'], d['fwd_f32_frac'], d['fwd_bf16_us'], d['fwd_bf16_frac'], d['bwd_f32_us']))"
