#!/bin/bash
# The kernel's device code (zj_device.h) compiled for the CPU emulator with AddressSanitizer + UBSan, then the
# emulator and golden suites run against it (GPU sanitizers are not available on the pool; this is the CPU build).
# Out-of-bounds LDS / plane / output indexing and signed-overflow UB in the shared code show up here.
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
cp tests/emu/libzjemu.so /tmp/libzjemu_backup.so 2>/dev/null || true
g++ -O1 -g -std=c++17 -fPIC -shared -fno-strict-aliasing -fsanitize=address,undefined -fno-omit-frame-pointer \
    -Wall -Wno-unknown-pragmas -o tests/emu/libzjemu.so tests/emu/zj_emu.cpp
touch tests/emu/libzjemu.so
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
    python -m pytest tests/test_emu.py tests/test_golden.py -q -s > /tmp/asan_emu.txt 2>&1 || true
echo "runtime errors (UBSan): $(grep -c 'runtime error' /tmp/asan_emu.txt)   ASan reports: $(grep -c 'AddressSanitizer' /tmp/asan_emu.txt)"
tail -1 /tmp/asan_emu.txt
rm -f tests/emu/libzjemu.so   # rebuilt without sanitizers on next use
