#!/bin/bash
# Builds librover_step.so (HIP kernels + C ABI) for gfx950 next to this script.
# -ffp-contract=off: one IEEE rounding per operation, in the reference's evaluation order.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
# rover_version() carries a hash of the sources the library was built from: bench.py compares it with the hash recorded next to the
# PMC counts in profiles/ (a profile of another binary is reported as stale) — tools/src_hash.sh prints the same value
SRC_HASH="$(cd "$here" && cat rover_capi.cpp rover_kernels.hip rover_cull.hip rover_mlp.hip rover_internal.h rover_raymath.h ../../include/rover_step.h | sha256sum | cut -c1-12)"
exec "$HIPCC" -DROVER_SRC_HASH="\"$SRC_HASH\"" --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden \
  -Wall -Wno-unused-result ${ROVER_EXTRA_FLAGS:-} \
  -o "$here/librover_step.so" "$here/rover_capi.cpp" "$here/rover_kernels.hip" "$here/rover_cull.hip" "$here/rover_mlp.hip"
