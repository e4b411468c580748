#!/bin/bash
# builds everything that travels to the GPU box (product library, emulator build, oracle, x2bench), then runs gpurun
set -e
R=$(cd $(dirname $0)/.. && pwd)
make -C $R/gmat_amd/csrc -j8 > /dev/null
make -C $R/tests/hipemu -j8 > /dev/null
make -C $R/oracle > /dev/null
bash $R/tools/build_layout_variants.sh > /dev/null
mkdir -p $R/tools/bin
g++ -O2 -std=c++17 $R/tools/x2bench.cpp -o $R/tools/bin/x2bench -L$R/gmat_amd/lib -lgmat_hip -Wl,-rpath,'$ORIGIN/../../gmat_amd/lib'
exec /usr/local/graft/bin/gpurun "$@"
