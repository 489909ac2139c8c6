#!/bin/bash
# The library's HOST code (moc_host_rng.cpp: the mt19937 replay of torch.rand's mask stream, moc_host_max_kept) under
# AddressSanitizer + UndefinedBehaviorSanitizer, driven by its own CPU tests.  (GPU sanitizers are not available on this
# pool; the device code's memory discipline is covered by tests/test_isa_hazards_cpu.py and the parity suite.)
#   bash scripts/asan_host.sh        # needs the normal build first (make -C moc_amd/csrc): it relinks the device objects
set -e
cd "$(dirname "$0")/../moc_amd/csrc"
g++ -O1 -g -std=c++17 -fPIC -Wall -fsanitize=address,undefined -fno-omit-frame-pointer -c moc_host_rng.cpp -o /tmp/moc_host_rng_asan.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o /tmp/libmoc_hip_asan.so \
    moc_capi.o moc_scores.o moc_select.o moc_meta.o moc_p2p.o moc_attn.o /tmp/moc_host_rng_asan.o
cd ../..
ASAN=$(g++ -print-file-name=libasan.so); UBSAN=$(g++ -print-file-name=libubsan.so)
MOC_HIP_LIB=/tmp/libmoc_hip_asan.so LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_host_rng_cpu.py tests/test_abi_cpu.py -q
