#!/bin/bash
# Builds tests/simt/_build/libidist_emu.so: the real csrc sources compiled for the CPU
# lockstep emulator (TEST INFRASTRUCTURE; the product never loads this).
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
mkdir -p "$here/_build"
# (debug info doubles the compile time of this one large translation unit — 6 min instead of 3 — and is only wanted when a failure
#  is being chased: IDIST_EMU_DEBUG=1 adds it)
g++ -O2 ${IDIST_EMU_DEBUG:+-g} -std=c++17 -fPIC -shared -DIDIST_EMU -DIDIST_VARIANTS -mavx2 -mfma -ffp-contract=off -fno-fast-math \
    -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result -Wno-unknown-pragmas -Wno-unused-variable \
    -I"$here" -x c++ "$root/instant-distance_amd/csrc/idist_capi.hip" "$here/hip_emu.cpp" \
    -o "$here/_build/libidist_emu.so.tmp.$$"
mv -f "$here/_build/libidist_emu.so.tmp.$$" "$here/_build/libidist_emu.so"
echo "$here/_build/libidist_emu.so"
