#!/bin/bash
echo "== parity (K=50 paths)"; timeout 600 python -m pytest tests -m gpu -q -x -k "k50 or full_size or late or c2 or content or shapes or edge or stale or two_ranks" 2>&1 | grep -E "passed|failed" | tail -3
bash tools/r3_ab.sh libstm_hip.so
