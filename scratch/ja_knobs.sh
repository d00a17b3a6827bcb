#!/bin/bash
# forward aggregation kernel on the ja shapes under the launch knobs (each in its own process: the knobs are read once)
for blk in 256 512 1024; do
  for u in 2 4; do
    echo "== JMAC_FWD_BLOCK=$blk JMAC_FWD_U=$u"
    JMAC_FWD_BLOCK=$blk JMAC_FWD_U=$u PROBE=none python scratch/lat_probe.py 2>&1 | grep -E "^(el|ja|en)"
  done
done
