#!/bin/bash
# usage: tools/build_variant.sh <name> [hipcc flags...]   -> build/lib<name>.so from the current sources (A/B arms for tools/ab_bench*.py)
#        SRC=<dir with csrc/ and include/> tools/build_variant.sh <name> ...   builds another source tree (e.g. a git worktree export)
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=${SRC:-$root}
csrc=$src/flash_attention_impls_amd/csrc
mkdir -p $root/build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Wno-unused-result "$@" \
    -o $root/build/lib$name.so $csrc/fa_capi.hip $csrc/fa_bwd_capi.hip $csrc/fa_diag.hip && echo built $root/build/lib$name.so
