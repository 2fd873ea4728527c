#!/bin/bash
# ISA of the benchmark's chain-kernel instantiations only (seconds instead of minutes):
#   bash scripts/isa_probe.sh bwd|fwd [extra hipcc flags]   ->  /tmp/asm/<which>_probe.s
which=${1:-bwd}; shift
mkdir -p /tmp/asm
cd "$(dirname "$0")/../m2_mixer_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DM2M_ISA_PROBE -S --cuda-device-only "$@" -o /tmp/asm/${which}_probe.s tower_${which}.hip 2>&1 | grep -E "error|remark" | head -20
grep -E "^\s+\.(vgpr_count|vgpr_spill_count):|\.name:\s+_Z" /tmp/asm/${which}_probe.s | paste - - - | cut -c1-200
