#!/bin/bash
# tools/build_variant.sh <tag> <file.hip replacement> [extra flags]: an ablation-flagged copy of the library with ONE source file replaced:
#   openintel_amd/libopenintel_hip_ablation_<tag>.so   (tools load it with OI_LIB=ablation_<tag>)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; REPL=$2; shift 2
D=$R/openintel_amd/csrc/_obj_var_$TAG
mkdir -p $D
NAME=$(basename $REPL)
for f in $R/openintel_amd/csrc/*.hip; do
  o=$D/$(basename ${f%.hip}).o
  src=$f
  if [ "$(basename $f)" = "$NAME" ]; then src=$REPL; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DOI_ABLATION "$@" -Wno-unused-result -I$R/include -I$R/openintel_amd/csrc -c $src -o $o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/openintel_amd/libopenintel_hip_ablation_$TAG.so $D/*.o
echo built $R/openintel_amd/libopenintel_hip_ablation_$TAG.so
