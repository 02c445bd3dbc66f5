#!/bin/bash
# AddressSanitizer run of the host list builders (CPU build only; GPU ASan is not available on the pool).
#   tools/asan_prep.sh        -> builds tests/_build/libhost_shim_asan.so and runs the high-valence + list-builder tests against it
set -e
cd "$(dirname "$0")/.."
mkdir -p tests/_build
g++ -O1 -g -std=c++17 -fPIC -shared -fopenmp -fsanitize=address -fno-omit-frame-pointer -ffp-contract=off -Wno-unknown-pragmas \
  -o tests/_build/libhost_shim_asan.so tests/host_shim.cpp rdcfes_amd/csrc/rdc_meshprep.cpp rdcfes_amd/csrc/rdc_prep_ev.cpp rdcfes_amd/csrc/rdc_prep_cl.cpp
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 RDC_SHIM_SO=tests/_build/libhost_shim_asan.so \
  python -m pytest tests/test_host_highvalence.py tests/test_host_prep.py tests/test_host_ev.py tests/test_host_cl.py -x -q -p no:cacheprovider
