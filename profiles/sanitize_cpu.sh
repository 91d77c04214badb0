#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer on everything that runs without a GPU (the GPU pool has no device sanitizer):
#   1. the oracle (oracle/Makefile: make sanitize) under the whole CPU suite;
#   2. the host side of libmi355nrphy.so (validators, derivations, metric arithmetic, plan bookkeeping that needs no device),
#      host objects rebuilt with -fsanitize=address,undefined -fno-gpu-sanitize, under tests/test_host.py;
#   3. both under profiles/fuzz_validators_cpu.py (validators and derivations on mutated descriptors).
# Build container, repository root:  bash profiles/sanitize_cpu.sh   -> prints the two pytest summaries.
set -eu
make -C oracle sanitize > /dev/null
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 NRPHY_ORACLE_SO=oracle/_san/liboracle.so \
  python3 -m pytest tests -q -m "not gpu" | tail -2
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 NRPHY_ORACLE_SO=oracle/_san/liboracle.so \
  python3 profiles/fuzz_validators_cpu.py 3000 | tail -4

C=srsran-edgeric-5g_amd/csrc
python3 srsran-edgeric-5g_amd/build.py > /dev/null
mkdir -p build/asan
for s in nrphy_host dl_control_host pdsch_async dl_slot_async; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-function -Iinclude -I$C -ffp-contract=off \
    -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer -x hip -c $C/$s.cpp -o build/asan/$s.o &
done
wait
OBJS=$(ls $C/*.o | grep -v "/nrphy_host.o\|/dl_control_host.o\|/pdsch_async.o\|/dl_slot_async.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan \
  -o build/asan/libmi355nrphy.so $OBJS build/asan/*.o
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 NRPHY_LIB_SO=$PWD/build/asan/libmi355nrphy.so \
  python3 -m pytest tests/test_host.py -q | tail -2
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 NRPHY_LIB_SO=$PWD/build/asan/libmi355nrphy.so \
  python3 profiles/fuzz_validators_cpu.py 3000 | tail -4
