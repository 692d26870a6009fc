#!/bin/bash
# CPU-only hygiene run (no GPU sanitizers exist on this pool): the C++ host shim and the oracle rebuilt with
# AddressSanitizer + UndefinedBehaviorSanitizer, their CPU tests run under them, the regular builds restored afterwards.
set -e
cd "$(dirname "$0")/.."
tmp=$(mktemp -d)
cp datum_amd/lib/libdatum_ocean_host.so $tmp/host.so
cp oracle/liboracle.so $tmp/oracle.so
restore() { cp $tmp/host.so datum_amd/lib/libdatum_ocean_host.so; cp $tmp/oracle.so oracle/liboracle.so; rm -rf $tmp; }
trap restore EXIT
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
g++ $SAN -std=c++14 -fPIC -ffp-contract=off -fno-fast-math -Wall -shared -o datum_amd/lib/libdatum_ocean_host.so datum_amd/host/ocean.cpp datum_amd/host/host_capi.cpp \
    -Ldatum_amd/lib -ldatum_ocean_hip -Wl,-rpath,'$ORIGIN'
g++ $SAN -std=c++14 -fPIC -fopenmp -ffp-contract=off -fno-fast-math -Wall -shared -o oracle/liboracle.so oracle/ocean_oracle.cpp
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=4
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest tests/test_host_shim.py tests/test_consumer_contract.py tests/test_oracle_pins.py tests/test_golden_and_abi.py -x -q -m "not gpu"
