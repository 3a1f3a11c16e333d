#!/bin/bash
# Build the default library and the per-dimension libraries the tests use (hipcc cross-compiles without a GPU); in parallel, ~3.5 min.
#   bash tools/build_all.sh
cd "$(dirname "$0")/.." || exit 1
ROOT=$(pwd)
python3 - <<PY &
import sys; sys.path.insert(0, "$ROOT")
from mpc_code_amd import capi
print(capi.build_library(force=True))      # the default library: two objects compiled side by side, then linked
PY

python3 - <<PY &
import sys; sys.path.insert(0, "$ROOT")
from mpc_code_amd import capi
for dims in ((3, 2, 3, 3, 3, 1, 0), (5, 2, 2, 2, 5, 0, 0)):
    print(capi.build_library(dims=dims, force=True))
PY
wait
ls -la mpc-code_amd/csrc/libmpc_amd.so mpc-code_amd/csrc/jit/
