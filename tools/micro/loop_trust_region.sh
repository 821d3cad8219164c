#!/bin/bash
# usage (GPU box): tools/micro/loop_trust_region.sh "<seeds>" "<radii>" -> the incremental loop with ba_solver=lm on EVERY BA (no hand-off
# of the wide-band BAs to TRF, the library's own damping floor 3e-3, not the loop's 0.3); radius -1 = no trust region, 0 = scipy's |x0|
for s in $1; do
  for tr in ${2:--1 0}; do
    echo "== seed $s lm everywhere, lambda_min 3e-3, trust radius $tr"
    python tools/incremental_loop.py --obs 79000 --solver lm --lm-wide lm --lambda-min 3e-3 --trust-radius $tr --seed $s 2>&1 | grep -E "trajectory:|loop finished|Error|error" | head -4
  done
done
