#!/bin/bash
# tools/build_variant.sh <git-ref> [out.so] -- build the kernels of another revision into a second library
# (default java-sdr_amd/libjsdr_hip_b.so) for same-box A/B timing: `JSDR_LIB=<that .so> python bench.py ...`
# (boxes differ by several percent, so variants are only comparable within one gpurun session).
set -euo pipefail
REF=$1
OUT=${2:-java-sdr_amd/libjsdr_hip_b.so}
TMP=$(mktemp -d /tmp/jsdr_variant.XXXXXX)
mkdir -p "$TMP/java-sdr_amd" "$TMP/include"
git archive "$REF" java-sdr_amd/csrc java-sdr_amd/build.py include | tar -x -C "$TMP"
python "$TMP/java-sdr_amd/build.py" > /dev/null
cp "$TMP/java-sdr_amd/libjsdr_hip.so" "$OUT"
rm -rf "$TMP"
echo "$OUT  <- $REF"
