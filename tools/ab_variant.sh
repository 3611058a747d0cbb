#!/bin/bash
# Build the library of another git revision into saugns_amd/variants/lib_<name>.so
# for same-box A/B timing:  tools/ab_variant.sh <rev> <name>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=$1; NAME=$2
TMP=$(mktemp -d)
git -C "$ROOT" archive "$REV" saugns_amd include | tar -x -C "$TMP"
(cd "$TMP" && python saugns_amd/build.py >/dev/null)
mkdir -p "$ROOT/saugns_amd/variants"
cp "$TMP/saugns_amd/libsaugns_amd.so" "$ROOT/saugns_amd/variants/lib_$NAME.so"
rm -rf "$TMP"
echo "$ROOT/saugns_amd/variants/lib_$NAME.so"
