#!/bin/bash
cd /root/repo
for C in 16 24 32 64; do JMAC_COOP_MIN=$C python tools/union_agg_probe.py 2>/dev/null; done
for C in 8 16 32; do JMAC_COOP_MIN=$C python tools/pair_probe.py --batched 1 2>/dev/null; done
