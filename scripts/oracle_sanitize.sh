#!/bin/bash
# The CPU suite with the oracle built under AddressSanitizer + UBSan (CPU build only; the GPU pool has no sanitizers).
# usage: bash scripts/oracle_sanitize.sh   (from the repo root; restores the normal oracle build afterwards)
set -e
cd "$(dirname "$0")/.."
trap 'make -s -C oracle -B libcm_oracle.so' EXIT
g++ -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -pthread -fsanitize=address,undefined \
    -fno-sanitize-recover=undefined -shared -o oracle/libcm_oracle.so oracle/cm_oracle.cpp
touch oracle/libcm_oracle.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider
