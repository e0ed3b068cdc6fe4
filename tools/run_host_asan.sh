#!/bin/bash
# The host-side tests and a pattern fuzzer against the AddressSanitizer + UBSan build of libtgx (make -C term_amd/csrc asan).
# CPU only: the device code is not instrumented (no GPU sanitizer on this pool) and nothing here touches a GPU.
#   tools/run_host_asan.sh [fuzz seconds]
cd "$(dirname "$0")/.." || exit 1
make -C term_amd/csrc -j8 asan > /dev/null 2>&1 || { echo "asan build failed"; exit 1; }
rt=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export TGX_LIB=$PWD/build/tgx_asan/libtgx.so
# leaks: CPython never frees what it interns; ODR: the HIP runtime registers fat binaries twice under ASan
export ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD=$rt
python -m pytest tests/test_regex_host.py tests/test_regex_counted_classes.py tests/test_host_logic.py tests/test_analyzers_host.py tests/test_state_json_tokens.py tests/test_abi.py -x -q -m "not gpu" 2>&1 | tail -5 || exit 1
python tools/fuzz_patterns.py --seconds "${1:-30}" || exit 1
