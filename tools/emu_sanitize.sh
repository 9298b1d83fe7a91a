#!/bin/bash
# The stepped emulation of the kernel bodies (tests/emu: the product's msm.hpp / poly.hpp / ipa.hpp / glv.hpp / fold_table.hpp / ntt-free
# host compilation, every lane's index arithmetic executed on the CPU) under AddressSanitizer + UndefinedBehaviorSanitizer.  GPU ASan is
# not available on the pool; this is the sanitizer run of the same indexing logic.  Takes a few times the plain suite's time.
#   tools/emu_sanitize.sh [pytest args]        e.g. tools/emu_sanitize.sh -k fold_table
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 PC_EMU_SANITIZE=1 python -c "import sys; sys.path[:0] = ['tests', 'oracle']; import test_emu_cpu as t; t.emu()" || exit 1      # built once, before any worker
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 PC_EMU_SANITIZE=1 \
  python -m pytest tests/test_emu_cpu.py -q -x -p no:cacheprovider "$@"
