#!/bin/bash
# developer aid: build the product library with extra -D flags next to it:  tools/build_variant.sh NAME -DFOO=1 ...  -> variants/liblz4_amd_NAME.so
set -e
cd "$(dirname "$0")/.."; name=$1; shift
mkdir -p variants build/obj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c lz4_amd/csrc/lz4amd_device.hip -o build/obj_$name/dev.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=lz4_amd/csrc/exports.map -o variants/liblz4_amd_$name.so build/obj_$name/dev.o build/obj/lz4amd_batch.o build/obj/lz4_api.o build/obj/lz4_stream_api.o build/obj/lz4hc_api.o build/obj/lz4frame_api.o build/obj/lz4frame_stream_api.o build/obj/lz4_compat_api.o build/obj/lz4file_api.o -lpthread
echo variants/liblz4_amd_$name.so
