#!/bin/bash
# Builds libmode_hip.so WITH the round-6 wave-specialised weight-gradient + AdamW probe kernel linked in (scripts/probe/gemm_bf16_trws.hip; the shipped
# library has only weak hooks for it).  The variant then takes the kernel by default; A/B inside one process: mode_set_option("adamw_ws", 0 | 1),
# MODE_ADAMW_WS=0|1 for bench.py's training leg, scripts/fused_adamw_probe.py for the isolated launches.  `make -C mode_diffusion_policy_amd/csrc` restores
# the shipped library.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/mode_diffusion_policy_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I$C -c $R/scripts/probe/gemm_bf16_trws.hip -o $C/trws_probe.o
rm -f $R/mode_diffusion_policy_amd/libmode_hip.so
make -C $C -j8 EXTRA=trws_probe.o
