#!/bin/bash
python tools/r3_diag2.py 2>&1 | tail -14
STM_POST_IMPL=1 python tools/r3_diag2.py 2>&1 | tail -14
