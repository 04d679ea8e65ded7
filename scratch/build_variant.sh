#!/bin/bash
# build a variant of libmobrob_ppo.so with extra compiler flags:  scratch/build_variant.sh <name> [-DFLAG=VALUE ...]
# -> scratch/lib_<name>.so (git-ignored, travels to the GPU box); use it with MOBROB_PPO_LIB=scratch/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form "$@" \
  -o scratch/lib_${name}.so mobrob_amd/csrc/engine.hip
echo scratch/lib_${name}.so
