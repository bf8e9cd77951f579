#!/bin/bash
# lab: the one-launch chain scan at C2 for tile shapes (CX_CHAIN_SHAPE), time per sweep and the stamps of one launch
for s in ${SHAPES:-1 0 4 2 3}; do
  echo "== CX_CHAIN_SHAPE=$s"
  CX_CHAIN_SHAPE=$s python3 tools/lab/c2_onepass.py 2>&1 | grep "one launch" | tail -2
  CX_CHAIN_SHAPE=$s CX_CHAIN_ONEPASS_STAMPS=1 python3 tools/lab/c2_onepass.py 2>&1 | grep "onepass" | tail -8
done
