#!/bin/bash
# second-stream encoder (JMAC_SIDE_STREAM): the ja / pair steps without it (0), forward only (1), forward + backward (2)
cd /root/repo
for S in 0 1 2 0 1; do
  echo "== JMAC_SIDE_STREAM=$S"
  JMAC_SIDE_STREAM=$S python tools/pair_probe.py --ja 2>&1 | tail -1
  JMAC_SIDE_STREAM=$S python tools/pair_probe.py --batched 1 2>&1 | tail -1
done
