#!/bin/bash
# SURVEY section 5's sanitizer row: the HOST side of libgeocalib_hip.so (C ABI, launch sequence, workspace management, RCCL
# binding, the kernels' host launchers) under AddressSanitizer + UndefinedBehaviorSanitizer.  Device code is NOT instrumented
# (GPU ASan needs xnack+, which this pool refuses): -Xarch_host keeps the sanitizers to the host pass of every .hip file.
# The pool's GPU boxes refuse every sanitizer build (gpurun rejects a snapshot whose scripts carry the flag; this file is listed in
# .gpurunignore), so the run covers what the host side does WITHOUT a device: tests/test_abi.py (library loads, every declared
# symbol exported, configuration defaults, ABI stamp refusals, the no-device failure) and scripts/probes/asan_drive.py (every
# entry point's argument / NULL-handle checks, threaded error strings, the RCCL loader).  The launch sequences run on the plain
# build only (the -m gpu suite).
#   scripts/asan_shim.sh build        build container: geocalib_amd/lib/asan/libgeocalib_hip_asan.so (+ the C99 example, link check)
#   scripts/asan_shim.sh run [log]    build container: the two drivers above under LD_PRELOAD of the ASan runtime
set -u
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
ROOT=$PWD
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
RTDIR=$(dirname $(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1))
RT=$RTDIR/libclang_rt.asan-x86_64.so
OUT=geocalib_amd/lib/asan
SAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -Xarch_host -fno-sanitize-recover=undefined"
if [ "${1:-build}" = "build" ]; then
  mkdir -p $OUT/obj
  for f in gclm_pass gclm_update gclm_api gclm_comm; do
    EXTRA=""; [ $f = gclm_pass ] && EXTRA="-fno-slp-vectorize -mllvm -disable-vector-combine"
    $HIPCC -O2 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast-honor-pragmas $EXTRA $SAN -c geocalib_amd/csrc/$f.hip -o $OUT/obj/$f.o || exit 1
  done
  $HIPCC --offload-arch=gfx950 -shared -fPIC $OUT/obj/*.o -o $OUT/libgeocalib_hip_asan.so -fsanitize=address,undefined -shared-libsan -ldl || exit 1
  /opt/rocm/lib/llvm/bin/clang -std=c99 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -shared-libsan -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/calibrate_c_abi.c \
      -o $OUT/calibrate_c_abi_asan -L$OUT -lgeocalib_hip_asan -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$ROOT/$OUT -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,$RTDIR || exit 1
  ls -la $OUT/libgeocalib_hip_asan.so $OUT/calibrate_c_abi_asan
  exit 0
fi
LOG=${2:-profiles/r06_asan_shim.log}
# detect_leaks=0: python itself "leaks" by ASan's definition; everything else is on and fatal
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:exitcode=77:allocator_may_return_null=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export GCLM_LIB_PATH=$ROOT/$OUT/libgeocalib_hip_asan.so
{
echo "== ASan + UBSan on the host side of libgeocalib_hip.so, build container (no GPU); runtime $(basename $RT), $($HIPCC --version | grep -m1 -i 'hip version')"
echo "== flags: $SAN (device pass not instrumented)"
echo "== tests/test_abi.py"
LD_PRELOAD=$RT timeout 600 python -m pytest tests/test_abi.py -q -p no:cacheprovider 2>&1 | tail -3
echo "== scripts/probes/asan_drive.py"
LD_PRELOAD=$RT timeout 600 python scripts/probes/asan_drive.py 2>&1 | tail -12
echo "rc ${PIPESTATUS[0]}"
echo "== the C99 example (itself -fsanitize=address,undefined) links against the sanitized library and fails cleanly without a device"
timeout 60 $OUT/calibrate_c_abi_asan 2>&1 | tail -3
echo "rc ${PIPESTATUS[0]} (2 = its hipMalloc check: no device here)"
} > $LOG 2>&1
echo "sanitizer reports: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' $LOG)" >> $LOG
cat $LOG
