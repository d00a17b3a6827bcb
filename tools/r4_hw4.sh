#!/bin/bash
cd /root/repo
for D in 2 3; do JMAC_FWD_HW_DEPTH=$D python tools/c4_probe.py 2>/dev/null; done
for C in 4 6 12 16; do JMAC_FWD_HW=0 JMAC_COOP_MIN=$C python tools/union_agg_probe.py 2>/dev/null; done
JMAC_FWD_HW=1 JMAC_COOP_MIN=4 python tools/union_agg_probe.py 2>/dev/null
