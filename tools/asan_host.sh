#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of libhello_mi355x.so (record stage, creation-time validation, shard hand-over,
# the shared scoring server's threads / sockets / slot checks):
# the whole library is rebuilt with the sanitizer on the host half of every translation unit (the gfx950 device code is
# compiled as usual; GPU ASan is not available on this pool), and the CPU tests that call into the library run against it.
#
#     tools/asan_host.sh [extra pytest arguments]        -> tools/_bin/asan/libhello_asan.so, report on stdout
#
# HELLO_LIB points hello_amd.engine.load_library at the instrumented build; python itself is not instrumented, so the
# sanitizer's runtime is preloaded and leak checking (python's own arenas) is off.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/tools/_bin/asan"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
RT="$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)"
mkdir -p "$OUT"
cd "$ROOT/hello_amd/csrc"
SRCS=$(sed -n 's/^SRCS = //p' Makefile)
for f in $SRCS; do
    "$HIPCC" -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off \
        -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-sanitize=vptr,function -Xarch_host -fno-omit-frame-pointer -c "$f" -o "$OUT/${f%.hip}.o" &
done
wait
OBJS=$(for f in $SRCS; do echo "$OUT/${f%.hip}.o"; done)
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-sanitize=vptr,function -shared-libsan -o "$OUT/libhello_asan.so" $OBJS
cd "$ROOT"
echo "# tools/asan_host.sh: $(basename "$RT"), HELLO_LIB=tools/_bin/asan/libhello_asan.so"
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1 LD_PRELOAD="$RT" HELLO_LIB="$OUT/libhello_asan.so" \
    python -m pytest tests/test_records.py tests/test_call_driver.py tests/test_loader_abi.py tests/test_shared_server.py \
    -q -m "not gpu" -p no:cacheprovider "$@"
