#!/bin/bash
# tools/build_define.sh <out.so> <-DNAME=VALUE ...> -- build the working tree's kernels with extra preprocessor
# definitions into a second library for same-box A/B timing: `JSDR_LIB=<out.so> python bench.py ...`
set -euo pipefail
OUT=$1; shift
TMP=$(mktemp -d /tmp/jsdr_define.XXXXXX)
mkdir -p "$TMP/java-sdr_amd" "$TMP/include"
cp -r java-sdr_amd/csrc "$TMP/java-sdr_amd/"
cp java-sdr_amd/build.py "$TMP/java-sdr_amd/"
cp include/jsdr_hip.h "$TMP/include/"
DEFS="$*"
python - "$TMP/java-sdr_amd/build.py" "$DEFS" <<'PY'
import sys
p, defs = sys.argv[1], sys.argv[2].split()
s = open(p).read()
s = s.replace('"-Wno-unused-result"]', '"-Wno-unused-result"] + ' + repr(defs))
open(p, "w").write(s)
PY
python "$TMP/java-sdr_amd/build.py" > /dev/null
cp "$TMP/java-sdr_amd/libjsdr_hip.so" "$OUT"
rm -rf "$TMP"
echo "$OUT  <- $DEFS"
