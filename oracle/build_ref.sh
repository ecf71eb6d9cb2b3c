#!/usr/bin/env bash
# Builds oracle/_ref/*.so from the reference's own sources where they lie under
# /root/reference.  TEST INFRASTRUCTURE ONLY.  Outputs go to oracle/_ref/ (git-ignored,
# travels to the GPU box as a prebuilt binary).  No reference source enters the repo:
# the line ranges needed for the CPU Monte Carlo are streamed into a mktemp directory
# under /tmp that is removed before the script returns.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ref="${MCAMD_REFERENCE_DIR:-/root/reference}"
out="$here/_ref"
if [ ! -d "$ref/inc" ]; then
    echo "build_ref.sh: $ref/inc not present (GPU box?) - keeping prebuilt oracle/_ref" >&2
    exit 0
fi
mkdir -p "$out"
CXX="${CXX:-g++}"

# 1. closed form: the header compiles as it stands.
"$CXX" -O2 -std=c++17 -fPIC -shared -I"$ref/inc" "$here/ref_bs_driver.cpp" -o "$out/libref_bs.so"

# 2. CPU Monte Carlo + array-driven pricer: host-only line ranges of .cuh files.
tmp="$(mktemp -d /tmp/mcamd_ref.XXXXXX)"
trap 'rm -rf "$tmp"' EXIT
sed -n '13,26p'   "$ref/inc/tool.cuh"    > "$tmp/ref_optiondata.inc"
sed -n '104,173p' "$ref/inc/tool.cuh"    > "$tmp/ref_cpumc.inc"
sed -n '75,91p'   "$ref/inc/testing.cuh" > "$tmp/ref_arraycpu.inc"
# guard against the reference moving under our feet
grep -q '^struct OptionData {' "$tmp/ref_optiondata.inc"
grep -q '^void simulateBulletOptionPriceCPU' "$tmp/ref_cpumc.inc"
grep -q 'h_randomData' "$tmp/ref_arraycpu.inc"
"$CXX" -O2 -std=c++17 -fPIC -shared -I"$tmp" "$here/ref_cpumc_driver.cpp" -o "$out/libref_cpumc.so"
echo "built $out/libref_bs.so $out/libref_cpumc.so"
