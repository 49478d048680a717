#!/bin/bash
# A/B builds of the product library: recompile ONE source with extra flags (or from another git revision) and link it with the current
# objects into tools/ab/lib<name>.so; run the two libraries in the same gpurun call with SPEECHCLIP_HIP_LIB=tools/ab/lib<name>.so.
#   tools/build_ab.sh attention.hip "-DSC_ATTN_ABL=1" noexp        # working-tree source + flags
#   tools/build_ab.sh gemm.hip "" old HEAD~3                       # the file as of a revision
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; EXTRA=$2; NAME=${3:-B}; REV=$4
CS=$ROOT/speechclip_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -I$CS"
mkdir -p $ROOT/tools/ab
make -C $CS -j8 >/dev/null
IN=$CS/$SRC
if [ -n "$REV" ]; then IN=$ROOT/tools/ab/${NAME}_$SRC; git -C $ROOT show $REV:speechclip_amd/csrc/$SRC > $IN; fi
hipcc $FLAGS $EXTRA -c $IN -o $ROOT/tools/ab/${NAME}_${SRC%.hip}.o
OBJS=$(ls $CS/*.o | grep -v "/${SRC%.hip}.o$")
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/tools/ab/${NAME}_${SRC%.hip}.o -ldl -o $ROOT/tools/ab/lib$NAME.so
# the comparator library is looked up next to the library that loads it
cp -f $ROOT/speechclip_amd/libspeechclip_vendor_cmp.so $ROOT/tools/ab/ 2>/dev/null || true
ls -la $ROOT/tools/ab/lib$NAME.so
