#!/bin/bash
cd /root/repo
for HW in 0 1; do JMAC_FWD_HW=$HW python tools/union_agg_probe.py 2>/dev/null; done
JMAC_FWD_HW_GP=2 python tools/union_agg_probe.py 2>/dev/null
for HW in 0 1; do JMAC_FWD_HW=$HW python tools/c4_probe.py 2>/dev/null; done
JMAC_FWD_HW_GP=1 python tools/c4_probe.py 2>/dev/null
JMAC_FWD_HW_GP=2 python tools/c4_probe.py 2>/dev/null
python tools/hw_debug.py 2>&1 | grep -v amdgpu | tail -12
