#!/bin/bash
# Build the default library and the per-dimension libraries the tests use (hipcc cross-compiles without a GPU); in parallel, ~4 min.
#   bash tools/build_all.sh
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
( cd mpc-code_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o libmpc_amd.so.tmp mpc_amd.hip 2>&1 | grep -v "argument unused" ; mv libmpc_amd.so.tmp libmpc_amd.so ) &
python3 - <<PY &
import sys; sys.path.insert(0, "$ROOT")
from mpc_code_amd import capi
for dims in ((3, 2, 3, 3, 3, 1, 0), (5, 2, 2, 2, 5, 0, 0)):
    print(capi.build_library(dims=dims, force=True))
PY
wait
ls -la mpc-code_amd/csrc/libmpc_amd.so mpc-code_amd/csrc/jit/
