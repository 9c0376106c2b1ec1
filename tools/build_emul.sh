#!/bin/bash
# Build the CPU emulation of the GPU entropy stage (test infrastructure only).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin/obj
gcc -std=gnu11 -O2 -fPIC -fvisibility=hidden -c jpeg_gpu_amd/csrc/entropy.c -o tools/bin/obj/entropy.o
gcc -std=gnu11 -O2 -fPIC -fvisibility=hidden -c jpeg_gpu_amd/csrc/layout.c -o tools/bin/obj/layout.o
g++ -std=c++17 -O2 -Wall -Wextra -fPIC -shared -fvisibility=hidden -Iinclude -o tools/bin/libhuff_emul.so \
  tools/huff_emul.cpp jpeg_gpu_amd/csrc/huff_prepare.cpp tools/bin/obj/entropy.o tools/bin/obj/layout.o
