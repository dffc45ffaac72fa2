#!/bin/bash
# Build diagnostic variants of the library (-DSPC_EXP=n, see spc_hip.hip) into build/variants/ (git-ignored; they
# travel to the GPU box with the snapshot).  usage: tools/exp_variants.sh "name:flags" ...
cd "$(dirname "$0")/.."
mkdir -p build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude $flags \
      sp_coupler_amd/csrc/spc_hip.hip -o build/variants/libspc_$name.so &
done
wait
ls -la build/variants/
