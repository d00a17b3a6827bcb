#!/bin/bash
cd /root/repo
python tools/hw_debug2.py 2>&1 | grep -v amdgpu | tail -4
for HW in 0 1; do JMAC_FWD_HW=$HW python tools/union_agg_probe.py 2>/dev/null; done
for HW in 0 1; do JMAC_FWD_HW=$HW python tools/c4_probe.py 2>/dev/null; done
