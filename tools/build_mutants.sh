#!/bin/bash
# Mutant libraries of the mutation control (tools/mutation_control.py): libspc_hip.so with ONE kernel line perturbed each
# (-DSPC_MUTANT=n, csrc/spc_hip.hip: SPC_MUT).  Built here (hipcc cross-compiles gfx950), four at a time; they travel to the
# GPU box with the snapshot (build/ is git-ignored, not gpurun-ignored).   usage: tools/build_mutants.sh [n ...]
cd "$(dirname "$0")/.."
mkdir -p build/mutants
MUTS=${@:-1 2 3 4 5 6 7 8 9 10 11 12 13 14 15 16 17 18 19 20 21 22 23 24 25 26 27}
printf "%s\n" $MUTS | xargs -P 4 -I{} sh -c '/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -ffp-contract=off -Iinclude -DSPC_MUTANT={} sp_coupler_amd/csrc/spc_hip.hip -o build/mutants/libspc_mutant{}.so 2>/dev/null && echo "mutant {} built" || echo "mutant {} FAILED"'
