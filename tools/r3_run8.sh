#!/bin/bash
for cfg in "1024 8 8" "512 8 8" "2048 8 8" "1024 4 8" "1024 16 8" "1024 8 16" "512 16 16" "256 8 8"; do set -- $cfg; echo "group_kb $1 rows $2 depth $3"; STM_BETASS_GROUP_KB=$1 STM_BETASS_ROWS=$2 STM_BETASS_DEPTH=$3 bash tools/r3_prof.sh 2>&1 | grep -E "beta_ss_part" | cut -c1-120; done
