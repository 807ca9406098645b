#!/usr/bin/env bash
# Round 6, call 73: smoke() and the library-identity tests on the final library.
set -u
cd "$(dirname "$0")/.."; OUT=gpurun_out/r06c73; mkdir -p $OUT
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
timeout 60 python -m pytest tests -q -m gpu -k "native_library or config1_gpu" 2>&1 | tail -3 | tee $OUT/pytest_identity.txt
