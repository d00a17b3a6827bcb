#!/bin/bash
# config-4 forward: the default (uniform two-deep) against the conditional two-deep form and the experiments
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_layer.py tests/test_gpu_fullsize.py tests/test_gpu_union_real.py -x -q 2>&1 | tail -2
run() { env "$@" python tools/closed/c4_probe.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['env'], 'f32 %.3f ms %.3f  bf16 %.3f ms %.3f' % (d['fwd_f32_ms'], d['fwd_f32_frac'], d['fwd_bf16_ms'], d['fwd_bf16_frac']), d['f32_checksum'][2], d['bf16_checksum'][2])
"; }
for r in 1 2; do
run JMAC_X=0
run JMAC_FWD_HW_DEPTH=2
done
