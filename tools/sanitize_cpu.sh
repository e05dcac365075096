#!/bin/bash
# Host-side AddressSanitizer + UndefinedBehaviorSanitizer pass over the C ABI.  GPU sanitizers are
# not available on this pool, so only the HOST half of every .hip file is instrumented (-fno-gpu-sanitize: argument validation,
# size queries, launch planning, error reporting; the device code is compiled as usual and never launched) and the CPU boundary tests run
# against that library with the ASan runtime preloaded into the (uninstrumented) python.
#   tools/sanitize_cpu.sh            build + run tests/test_boundary_cpu.py
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/npi_gnn_amd/build/asan"
mkdir -p "$OUT"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
objs=""
# every header a translation unit may include + this recipe: part of every object's key
HDR=$(cat "$ROOT"/npi_gnn_amd/csrc/*.h "$ROOT/include/npi_gnn.h" "$0" | sha256sum | cut -d" " -f1)
# the library's own source list (npi_gnn_amd/build.py), so that a new .hip file cannot be forgotten here
for f in $(cd "$ROOT" && python3 -c "from npi_gnn_amd.build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))"); do
  objs="$objs $OUT/$f.o"
  # an object is rebuilt only when its source or a header changed (content hash beside it): the instrumented build of all ten
  # files took 4-5 minutes of every CPU test run
  KEY="$HDR $(sha256sum "$ROOT/npi_gnn_amd/csrc/$f.hip" | cut -d" " -f1)"
  if [ -f "$OUT/$f.o" ] && [ "$(cat "$OUT/$f.key" 2>/dev/null)" = "$KEY" ]; then continue; fi
  ( $HIPCC -O1 -g -fno-gpu-sanitize -std=c++17 --offload-arch=gfx950 -fPIC -fno-gpu-rdc -fno-omit-frame-pointer \
         -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libasan \
         -c "$ROOT/npi_gnn_amd/csrc/$f.hip" -o "$OUT/$f.o" && echo "$KEY" > "$OUT/$f.key" ) &
done
wait
$HIPCC -shared -fPIC -fsanitize=address,undefined -shared-libasan -o "$OUT/libnpi_gnn_asan.so" $objs
echo "built $OUT/libnpi_gnn_asan.so"
cd "$ROOT"
LD_PRELOAD="$RT" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:verify_asan_link_order=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  NPI_GNN_LIB="$OUT/libnpi_gnn_asan.so" python3 -m pytest tests/test_boundary_cpu.py -x -q -p no:cacheprovider "$@"
