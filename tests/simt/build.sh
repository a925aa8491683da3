#!/bin/sh
# TEST INFRASTRUCTURE: builds the CPU-interpreted twin of the kernels (see simt_emu.h)
set -e
cd "$(dirname "$0")"
g++ -O2 -g -std=c++17 -shared -fPIC -o libemu_kernels.so emu_kernels.cpp simt_emu.cpp -lpthread
