#!/bin/bash
# Attribution of the single-term joiner forward: rebuilds the library with each experiment macro set and times it.
set -e
for flags in "-DWR_X_NOSTORE" "-DWR_X_NOEPI" "-DWR_X_NOLOAD" "-DWR_X_NOLOAD -DWR_X_NOEPI"; do
  WR_EXTRA_HIPCC_FLAGS="$flags" python -c "import sys; sys.path.insert(0,'.'); from wenet_celoss_amd import _lib; _lib.build(force=True)"
  WR_EXTRA_HIPCC_FLAGS="$flags" python tools/amp_x.py
done
