#!/usr/bin/env bash
# tools/r05_ab.sh -- same box, alternating: the library in lib_prev/ (a build of another revision) against the product, block cadence
set -u
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for rep in 1 2; do
  for which in prev new; do
    if [ $which = prev ]; then export MSDR_LIB=$PWD/minimal-sdr_amd/lib_prev/libmsdr.so; else unset MSDR_LIB; fi
    echo "== $which (pass $rep)"
    bash tools/r05_blk2.sh
  done
done
