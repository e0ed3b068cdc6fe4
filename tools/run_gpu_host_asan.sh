#!/bin/bash
# The differential tester, HOST buffers only, against the build whose HOST code carries AddressSanitizer + UBSan (make -C term_amd/csrc asan;
# the device code is compiled as usual -- no GPU sanitizer is involved).  Run on a GPU box:
#   gpurun --timeout 1500 -- 'bash tools/run_gpu_host_asan.sh'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
make -C term_amd/csrc -j32 asan > gpurun_out/asan_build.log 2>&1 || { tail -5 gpurun_out/asan_build.log; echo "asan build failed"; exit 1; }
rt=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export TGX_LIB=$PWD/build/tgx_asan/libtgx.so
export ASAN_OPTIONS=detect_leaks=0:detect_odr_violation=0:abort_on_error=1:halt_on_error=1:protect_shadow_gap=0
export UBSAN_OPTIONS=print_stacktrace=0:halt_on_error=0   # (every report of the run, not just the first)
export LD_PRELOAD=$rt
# (torch does not initialise under the preloaded ASan runtime: HOST buffers only, fed by the differential tester and
#  by the plain-C consumer -- the coalescing arenas, the copy pool, staging, the key-set bookkeeping, blobs and merges)
timeout 900 python tools/fuzz_device.py --host-only --first 0 --count "${1:-600}" --seed-timeout 120 > gpurun_out/asan_fuzz.log 2>&1
grep "runtime error" gpurun_out/asan_fuzz.log | sed 's/^.*csrc\///' | sort | uniq -c | sort -rn | head -20
grep -B2 -A12 "ERROR: AddressSanitizer" gpurun_out/asan_fuzz.log | head -40
grep "^FAIL\|^STUCK\|cases of" -A2 gpurun_out/asan_fuzz.log | cut -c1-300 | tail -12
