#!/bin/bash
# build an A/B variant of the library: tools/build_variant.sh <name> <extra hipcc flags...>  -> /tmp/jmac_<name>.so
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/jv_$name
for f in $R/jmac_amd/csrc/*.hip; do
  b=$(basename $f .hip)
  if [ "$b" == "${VARIANT_FILE:-aggregate}" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$R/jmac_amd/csrc "$@" -c $f -o /tmp/jv_$name/$b.o &
  else
    cp $R/build/$b.o /tmp/jv_$name/$b.o 2>/dev/null || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I$R/jmac_amd/csrc -c $f -o /tmp/jv_$name/$b.o &
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 /tmp/jv_$name/*.o -o /tmp/jmac_$name.so && echo built /tmp/jmac_$name.so
