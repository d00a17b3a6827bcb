#!/bin/bash
# round 6: cost of bench.py's CPU stand-in (oracle/jmac_oracle.py, kind "port") against the REFERENCE itself, five runs of
# tests/golden/crosstime_reference.py in the build container (the reference cannot travel to the GPU box).
#   -> profiles/r6_crosstime.txt (the five lines) + profiles/r6_crosstime.json (median ratio: bench.py's
#      cpu_baseline.port_vs_reference_cost_ratio).      usage (dev container, repo root): bash tools/r6_crosstime.sh [dim] [threads]
R="$(cd "$(dirname "$0")/.." && pwd)"; cd "$R"
DIM=${1:-300}; THREADS=${2:-8}
: > profiles/r6_crosstime.txt
for i in 1 2 3 4 5; do
  PYTHONDONTWRITEBYTECODE=1 python tests/golden/crosstime_reference.py $DIM $THREADS | tail -1 >> profiles/r6_crosstime.txt
done
python3 - <<'PY'
import json, re, statistics
lines = [l.strip() for l in open("profiles/r6_crosstime.txt") if "oracle/reference" in l]
ratios = [float(re.search(r"oracle/reference = ([0-9.]+)", l).group(1)) for l in lines]
ref = [float(re.search(r"reference ([0-9.]+) s/step", l).group(1)) for l in lines]
orc = [float(re.search(r"oracle ([0-9.]+) s/step", l).group(1)) for l in lines]
json.dump({"what": "oracle (bench.py cpu_baseline, kind 'port') / imported reference, seconds per training step of the ja-shaped workload, "
                   "build container (8 shared vCPUs), tests/golden/crosstime_reference.py x5 (tools/r6_crosstime.sh)",
           "ratios": ratios, "median_ratio": statistics.median(ratios), "reference_s_per_step": ref, "oracle_s_per_step": orc,
           "condition": "BASELINE.md section 2: equal cost +-10 %"}, open("profiles/r6_crosstime.json", "w"), indent=1)
print(open("profiles/r6_crosstime.json").read())
PY
