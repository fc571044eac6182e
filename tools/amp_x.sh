#!/bin/bash
# Experiments on the single-term joiner forward: rebuilds the library with each macro set and times it.
set -e
for flags in "$@"; do
  WR_EXTRA_HIPCC_FLAGS="$flags" python -c "import sys; sys.path.insert(0,'.'); from wenet_celoss_amd import _lib; _lib.build(force=True)"
  WR_EXTRA_HIPCC_FLAGS="$flags" python tools/amp_x.py 2>/dev/null | head -1
done
