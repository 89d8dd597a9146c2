#!/bin/bash
# TEST INFRASTRUCTURE (this container only).  Compiles the UNMODIFIED reference
# PyPore/cparsers.pyx and PyPore/calignment.pyx into a scratch directory OUTSIDE the repo so that
# tests/golden/make_golden*.py can import them and record golden vectors.
# Nothing derived from the reference source (generated C, .so, bytecode) enters the repo
# or travels to the GPU box; only the recorded outputs (tests/golden/*.npz) do.
# Recipe: SURVEY.md Appendix A.
set -euo pipefail
REF=${REF:-/root/reference/PyPore}
OUT=${OUT:-/tmp/pypore_oracle}
mkdir -p "$OUT"
cp "$REF/cparsers.pyx" "$OUT/cparsers.pyx"
cp "$REF/calignment.pyx" "$OUT/calignment.pyx"
cd "$OUT"
cython -2 cparsers.pyx -o cparsers.c
cython -2 calignment.pyx -o calignment.c
PYINC=$(python3 -c "import sysconfig;print(sysconfig.get_paths()['include'])")
NPINC=$(python3 -c "import numpy;print(numpy.get_include())")
EXT=$(python3 -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
gcc -O2 -fPIC -shared -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
    -I"$PYINC" -I"$NPINC" cparsers.c -o "cparsers$EXT" -lm
gcc -O2 -fPIC -shared -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION \
    -I"$PYINC" -I"$NPINC" calignment.c -o "calignment$EXT" -lm
echo "built $OUT/cparsers$EXT $OUT/calignment$EXT"
