#!/bin/bash
echo "== suite"; timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5
bash tools/r3_ab.sh libstm_hip.so
