#!/bin/bash
# Disassembles the gfx950 code objects of vers_amd/lib/libvers_hip.so into $OUT (default /tmp/vers_isa): one .s per bundle.
# usage: scripts/isa.sh [kernel-name-substring]   -> prints the files / line ranges where the kernel is defined
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/vers_isa}
rm -rf "$OUT"; mkdir -p "$OUT"; cp "$ROOT/vers_amd/lib/libvers_hip.so" "$OUT/lib.so"
cd "$OUT"
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null 2>&1
rm -f lib.so.*.host-*
for f in lib.so.*gfx950; do /opt/rocm/lib/llvm/bin/llvm-objdump -d "$f" > "$f.s" 2>/dev/null; done
[ -n "$1" ] && grep -n "^[0-9a-f]* <.*$1" *.s | cut -c1-200
