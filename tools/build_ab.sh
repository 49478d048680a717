#!/bin/bash
# Build an A/B pair of the product library: A = gemm.hip of a git revision (default HEAD), B = the working tree.  Output: tools/ab/lib{A,B}.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
REV=${1:-HEAD}
CS=$ROOT/speechclip_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -I$CS"
mkdir -p $ROOT/tools/ab
make -C $CS -j8 >/dev/null
git -C $ROOT show $REV:speechclip_amd/csrc/gemm.hip > $ROOT/tools/ab/gemm_A.hip
git -C $ROOT show $REV:speechclip_amd/csrc/common.h > $ROOT/tools/ab/common.h
(cd $ROOT/tools/ab && hipcc $FLAGS -I$ROOT/tools/ab -c gemm_A.hip -o gemm_A.o 2>&1 | grep -E "error" || true)
OBJS=$(ls $CS/*.o | grep -v gemm.o)
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $ROOT/tools/ab/gemm_A.o -o $ROOT/tools/ab/libA.so
cp $ROOT/speechclip_amd/libspeechclip_hip.so $ROOT/tools/ab/libB.so
ls -la $ROOT/tools/ab/*.so
