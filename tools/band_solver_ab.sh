#!/bin/bash
# usage (on the GPU box): tools/band_solver_ab.sh  -> the LM step with the round-5 band-solver path and with its parts switched off
echo "LM step, tools/step_breakdown.py <config>: python wall / C++ solve ms per step, three repetitions on one box"
echo "  default            : interiors of 16 (<= 128 columns) or 32 control points, no right-hand-side copy, no back-correction"
echo "  MVUS_DIRECT_RHS=0  : with the right-hand-side copy (k_cholesky_and_rhs)"
echo "  MVUS_PART_BACK=1   : with the copy and the back-correction of the interiors' columns (k_part_back) -- the path of rounds 2-4 and of time shards"
echo "  MVUS_PART_LEN=other: the other interior length (32 where 16 is the default and vice versa)"
for c in 1 4 2 3; do
  other=16; if [ $c = 1 ] || [ $c = 4 ]; then other=32; fi
  for r in 1 2 3; do
    a=$(python3 tools/step_breakdown.py $c 2>&1 | tail -1 | awk '{print $3" / "$7}')
    b=$(MVUS_DIRECT_RHS=0 python3 tools/step_breakdown.py $c 2>&1 | tail -1 | awk '{print $3" / "$7}')
    d=$(MVUS_PART_BACK=1 python3 tools/step_breakdown.py $c 2>&1 | tail -1 | awk '{print $3" / "$7}')
    e=$(MVUS_PART_LEN=$other python3 tools/step_breakdown.py $c 2>&1 | tail -1 | awk '{print $3" / "$7}')
    echo "configs[$c] rep $r: default $a | copy $b | copy + back-correction $d | interiors of $other: $e"
  done
done
