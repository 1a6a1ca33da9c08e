#!/bin/bash
# Builds librover_step.so (HIP kernels + C ABI) for gfx950 next to this script.
# -ffp-contract=off: one IEEE rounding per operation, in the reference's evaluation order.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
exec "$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
  -Wall -Wno-unused-result ${ROVER_EXTRA_FLAGS:-} \
  -o "$here/librover_step.so" "$here/rover_capi.cpp" "$here/rover_kernels.hip" "$here/rover_cull.hip" "$here/rover_mlp.hip"
