#!/bin/bash
# Fixtures of the full-size GPU parity tests' HOST side (tests/conftest.py: host_solve_cached): run the three tests whose
# oracle solve is 40-60 s of host LAPACK with CCZ_WRITE_FULLSIZE_GOLDEN set, on a GPU box; copy the results into
# tests/golden/fullsize/.  The fixtures hold the oracle's output for moments that a deterministic device generator reproduces;
# each test re-checks a probe of its own moments before it trusts one.
#   gpurun -- 'bash tools/gen_golden_fullsize.sh'  &&  cp gpurun_out/golden_fullsize/*.npz tests/golden/fullsize/
set -e
OUT=${1:-$PWD/gpurun_out/golden_fullsize}
mkdir -p "$OUT"
CCZ_WRITE_FULLSIZE_GOLDEN="$OUT" python -m pytest -q -p no:cacheprovider \
  "tests/test_gpu_round2.py::test_ns_shape_against_oracle" \
  "tests/test_gpu_round2.py::test_c3_mcca_shape_against_oracle_and_certificate" \
  "tests/test_gpu_round3.py::test_ns_dimensions_per_column_on_a_separated_spectrum"
ls -la "$OUT"
