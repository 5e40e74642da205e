#!/bin/bash
# Builds an A/B variant of the library that differs from the shipped one ONLY in the persistent ping-pong GEMM, built from scripts/probe/gemm_bf16_pp_l2touch.hip (round 5's kernel WITH the L2 run-ahead switches):
#   scripts/build_pp_variant.sh <tag> <hipcc defines...>      e.g.  scripts/build_pp_variant.sh touch2 -DPP_L2_TOUCH=2
# -> mode_diffusion_policy_amd/libmode_hip_<tag>.so (git-ignored; travels to the GPU box with the snapshot).  Select it with MODE_HIP_LIB=<path>
# (mode_diffusion_policy_amd/_lib.py) or load it next to the shipped one (scripts/pp_l2touch_probe.py).  The shipped objects must be built first.
set -e
TAG=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/mode_diffusion_policy_amd/csrc
make -C $C -j16 > /dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -Wno-unused-result "$@" -c $R/scripts/probe/gemm_bf16_pp_l2touch.hip -o /tmp/gemm_bf16_pp_$TAG.o
OBJS=$(ls $C/*.o | grep -v gemm_bf16_pp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/gemm_bf16_pp_$TAG.o -o $R/mode_diffusion_policy_amd/libmode_hip_$TAG.so
echo built $R/mode_diffusion_policy_amd/libmode_hip_$TAG.so
