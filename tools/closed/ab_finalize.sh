#!/bin/bash
# A/B: which task of bwd_finalize_kernel holds its 10 us at ja size?  variants drop one task each (results wrong: timing only)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $R
for v in base noda nosp2 nosp1; do
  cp jmac_amd/csrc/aggregate.hip /tmp/agg_$v.hip
done
python3 - <<'PY'
import re
base = open("/tmp/agg_base.hip").read()
def var(name, repl):
    s = base
    for a, b in repl:
        assert a in s, a
        s = s.replace(a, b)
    open("/tmp/agg_%s.hip" % name, "w").write(s)
tail = "    const int nb = f.sp[0].nblocks + f.sp[1].nblocks + f.sp[2].nblocks + f.rd[0].nblocks + f.rd[1].nblocks;\n    if (nb > 0) hipLaunchKernelGGL(bwd_finalize_kernel, dim3((unsigned)nb), dim3(kBlock), 0, st, f);\n    return (int)hipGetLastError();"
var("noda", [(tail, "    f.rd[0].nblocks = 0;\n" + tail)])
var("nosp2", [(tail, "    f.sp[2].nblocks = 0;\n" + tail)])
var("nosp1", [(tail, "    f.sp[1].nblocks = 0;\n" + tail)])
PY
for v in base noda nosp2 nosp1; do
  mkdir -p /tmp/jv_$v
  for f in jmac_amd/csrc/*.hip; do b=$(basename $f .hip); if [ $b == aggregate ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -Ijmac_amd/csrc -c /tmp/agg_$v.hip -o /tmp/jv_$v/$b.o & else cp build/$b.o /tmp/jv_$v/$b.o; fi; done
done
wait
for v in base noda nosp2 nosp1; do /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/jv_$v/*.o -o /tmp/jmac_$v.so; done
cd /tmp && export TMPDIR=/tmp
for v in base noda nosp2 nosp1; do
  JMAC_LIB_PATH=/tmp/jmac_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abf_$v -o p -- python3 $R/tools/closed/overlap_probe3.py > /dev/null 2>&1
  echo "$v: $(grep -E 'bwd_finalize|rel_attn_bwd_gather|rel_attn_bwd_dst' /tmp/abf_$v/p_kernel_stats.csv | cut -d, -f1,4 | sed 's/(anonymous namespace):://g' | cut -c1-60 | tr '\n' ' ')"
done
