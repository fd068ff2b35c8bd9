#!/bin/bash
# The C ABI once under the AddressSanitizer/UBSan build of the HOST side of libtavb (make -C typeagent_py_amd/csrc debug).
# torch's CUDA init does not survive the ASan runtime, so the driver is tools/asan_exercise.py (hipMalloc through ctypes).
mkdir -p gpurun_out/asan; rm -f gpurun_out/asan/*
ASAN=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
E="env TAVB_LIBRARY=libtavb_debug.so LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1"
python tools/asan_exercise.py > gpurun_out/asan/plain.txt 2>&1; echo "plain rc=$?"; tail -n 4 gpurun_out/asan/plain.txt
timeout -k 5 600 $E python tools/asan_exercise.py > gpurun_out/asan/asan.txt 2>&1; echo "asan rc=$?"
if ! grep -q "asan exercise: all good" gpurun_out/asan/asan.txt; then  # (RCCL under the ASan runtime may refuse to initialise: the rest must still be clean)
  timeout -k 5 600 $E TAVB_ASAN_SKIP_RCCL=1 python tools/asan_exercise.py > gpurun_out/asan/asan_norccl.txt 2>&1; echo "asan (no rccl) rc=$?"; tail -n 6 gpurun_out/asan/asan_norccl.txt
fi
tail -n 12 gpurun_out/asan/asan.txt
grep -c "runtime error\|AddressSanitizer" gpurun_out/asan/asan.txt
