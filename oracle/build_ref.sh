#!/usr/bin/env bash
# Build the reference's two native extensions from the sources where they lie
# under /root/reference into oracle/_ref/ ($ANNLITE_REF_BUILD overrides): git-ignored binaries only -- no reference source or
# generated code stays anywhere; the directory is the repository's own (not a shared world-writable /tmp path).
#
# TEST INFRASTRUCTURE ONLY.  The outputs are used in THIS container to
#   (1) pin oracle/pq_oracle.c + oracle/pq_oracle.py against the real reference
#       (tests/test_oracle_vs_reference.py, skipped when /root/reference is absent), and
#   (2) generate the committed golden fixtures (tests/golden/make_golden.py).
# Nothing of it is imported by the product (annlite_amd/), and the GPU
# tests / smoke() / bench.py never touch it.
#
# Recipe mirrors the reference's own build flags (setup.py:51-55 compiler
# directives, setup.py:125-144 "-O3 -march=native -fopenmp") so the LUT loops
# FMA-contract exactly like a wheel built by the reference's setup.py on an FMA host.
set -euo pipefail
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
# oracle/_ref/: .gitignore'd (stays out of history) and .gpurunignore'd (stays in this container: the GPU box pins the oracle
# through the committed golden fixtures); only extension modules land here, generated C/C++ is deleted right after compiling
OUT="${ANNLITE_REF_BUILD:-$HERE/_ref}"
if [ ! -d "$REF/bindings" ]; then
  echo "build_ref: $REF not present, skipping (GPU box uses committed golden fixtures)"; exit 0
fi
mkdir -p "$OUT"
chmod 700 "$OUT"
rm -f "$OUT"/*.cpp "$OUT"/*.c  # never keep generated sources (they quote the reference)
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")

# 1) annlite.pq_bind  <- bindings/pq_bindings.pyx (Cython -> C++); the generated C++ (it quotes the .pyx
#    lines) is a temporary and is deleted: only the .so stays
if [ ! -f "$OUT/pq_bind$EXT" ] || [ "$REF/bindings/pq_bindings.pyx" -nt "$OUT/pq_bind$EXT" ]; then
  TMPCPP=$(mktemp -d)/pq_bind.cpp
  cython -+ -3 --module-name annlite.pq_bind \
    -X language_level=3 -X embedsignature=True -X annotation_typing=False \
    -o "$TMPCPP" "$REF/bindings/pq_bindings.pyx"
  g++ -O3 -march=native -fopenmp -std=c++14 -shared -fPIC -w \
    -I"$PYINC" -I"$NPINC" "$TMPCPP" -o "$OUT/pq_bind$EXT"
  rm -rf "$(dirname "$TMPCPP")"
fi

# 2) annlite.hnsw_bind <- bindings/hnsw_bindings.cpp (pybind11).  pybind11 3.x asserts the GIL
#    on every inc/dec-ref; the reference creates a py::array_t inside gil_scoped_release
#    (hnsw_bindings.cpp:312-326), so the two -D flags below are required or knn_query aborts.
if [ ! -f "$OUT/hnsw_bind$EXT" ] || [ "$REF/bindings/hnsw_bindings.cpp" -nt "$OUT/hnsw_bind$EXT" ]; then
  g++ -O3 -march=native -fopenmp -std=c++14 -shared -fPIC -w \
    -DPYBIND11_NO_ASSERT_GIL_HELD_INCREF_DECREF -DNDEBUG \
    $(python3 -m pybind11 --includes) -I"$REF/include/hnswlib" \
    "$REF/bindings/hnsw_bindings.cpp" -o "$OUT/hnsw_bind$EXT" -pthread || \
    echo "build_ref: hnsw_bind failed to build (only needed for the HNSW-over-PQ fixtures)"
fi
ls -la "$OUT"
