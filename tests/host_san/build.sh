#!/bin/bash
# build.sh <out-dir> <sanitizer flags...> -- host-only build of every .hip file of libgsr + hip_stub.cpp + driver.cpp
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$1"; shift
SAN="$*"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
mkdir -p "$OUT"
CXXFLAGS="--offload-host-only -O1 -g -fno-omit-frame-pointer -std=c++17 -Wno-unused-function -Wno-unused-value $SAN"
pids=()
for f in "$ROOT"/mygauhuman_amd/csrc/*.hip; do
  o="$OUT/$(basename "${f%.hip}").o"
  $HIPCC $CXXFLAGS -c "$f" -o "$o" &
  pids+=($!)
done
$HIPCC $CXXFLAGS -x hip -c "$HERE/hip_stub.cpp" -o "$OUT/hip_stub.o" &
pids+=($!)
$HIPCC $CXXFLAGS -x hip -c "$HERE/driver.cpp" -o "$OUT/driver.o" &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
# each translation unit refers to its (absent) device image by a hashed symbol: give every one a dummy definition
nm -u "$OUT"/*.o | awk '/__hip_fatbin_/ {print $2}' | sort -u | awk '{print "char " $1 "[16];"}' > "$OUT/fatbin_dummies.c"
gcc -c "$OUT/fatbin_dummies.c" -o "$OUT/fatbin_dummies.o"
/opt/rocm/lib/llvm/bin/clang++ $SAN -o "$OUT/driver" "$OUT"/*.o -lpthread
