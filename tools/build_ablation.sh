#!/bin/bash
# (OI_ABL_EXTRA=-DOI_NO_NT bash tools/build_ablation.sh --force  builds it with the default cache policy on the corpus stream.)
# Build an ABLATION copy of the library (A/B switches, diagnostic timing builds) next to the product one:
#   openintel_amd/libopenintel_hip_ablation.so ; use it with OI_LIB=ablation (tools only; never the tests/bench).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/openintel_amd/csrc/_obj_abl
cd $R/openintel_amd/csrc/_obj_abl
for f in $R/openintel_amd/csrc/*.hip; do
  o=$(basename ${f%.hip}).o
  if [ "$1" = "--force" ] || [ ! -f $o ] || [ $f -nt $o ] || [ $R/openintel_amd/csrc/oi_internal.h -nt $o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DOI_ABLATION ${OI_ABL_EXTRA:-} -Wno-unused-result -I$R/include -I$R/openintel_amd/csrc -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/openintel_amd/libopenintel_hip_ablation.so *.o
echo built $R/openintel_amd/libopenintel_hip_ablation.so
