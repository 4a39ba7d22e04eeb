#!/bin/bash
# AddressSanitizer + UBSan over the kernel code and the host orchestration on the CPU (the GPU pool offers no sanitizer runs):
# the kernel-emulation build (tests/emu, -DMS_EMU) executes every kernel phase thread by thread on malloc'ed "device" memory, so
# out-of-bounds global/LDS indices, misaligned accesses and signed overflow in kernels and in the host units (csrc/*.cpp) show up here.
#   bash tools/asan_check.sh        (≈ 2 min)
set -e
cd "$(dirname "$0")/.."
OUT=${MS_EMU_LIB:-/tmp/libministark_emu_asan.so}
make -C tests/emu -j8 BUILD=/tmp/ms_emu_asan_obj EMUFLAGS="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer" libministark_emu.so LIBOUT="$OUT" > /dev/null
# (alloc_dealloc_mismatch=0: the emulation build replaces operator new / delete by counting malloc / free wrappers, local to the library (abi.cpp, the allocation-failure
#  test); a std::string grown inside libstdc++.so is then allocated by the sanitizer's operator new and released by the library's free - a mismatch by construction only)
export MS_EMU_LIB="$OUT" ASAN_OPTIONS=detect_leaks=0:alloc_dealloc_mismatch=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
python tools/asan_cases.py
( cd tests && python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29931 shard_worker.py 0 8 8 16 2>&1 | grep -E '^\{|ERROR|runtime error' )
( cd tests && python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29932 shard_worker.py 1 8 4 16 2>&1 | grep -E '^\{|ERROR|runtime error' )
( cd tests && MS_SHARD_SLICES=4 MS_SHARD_SLICE_MIN=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29933 shard_worker.py 0 9 8 16 2>&1 | grep -E '^\{|ERROR|runtime error' )   # sliced digest exchange
# r04: distributed round polynomials - proof assembled on rank 0, base-field DEEP points (gather fallback), half-empty coefficient ranges, multi-level scans with chunked proof gathers
( cd tests && python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29934 shard_worker.py 0 9 8 16 root-only 2>&1 | grep -E '^\{|ERROR|runtime error' )
( cd tests && python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29935 shard_worker.py 1 8 8 16 base-z 2>&1 | grep -E '^\{|ERROR|runtime error' )
( cd tests && python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29936 shard_worker.py 0 10 8 16 low-degree 2>&1 | grep -E '^\{|ERROR|runtime error' )
( cd tests && MS_SHARD_GATHER_CHUNK=2048 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29937 shard_worker.py 0 13 8 16 2>&1 | grep -E '^\{|ERROR|runtime error' )
echo "sanitizer run clean"
