#!/bin/bash
cd /root/repo
python tools/union_agg_probe.py 2>/dev/null
JMAC_COOP_MIN_LARGE=24 python tools/union_agg_probe.py 2>/dev/null
JMAC_COOP_MIN_LARGE=32 python tools/union_agg_probe.py 2>/dev/null
python tools/c4_probe.py 2>/dev/null
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
