for cfg in 2 3; do
for wg in "10 1" "16 1" "20 2" "16 2" "21 2" "13 4" "21 4" "11 2" "8 2"; do
set -- $wg
MVUS_WIN=$1 MVUS_WIN_GROUPS=$2 python tools/micro/time_win.py $cfg 2>&1 | tail -1 | sed "s/lib default/G=$2/"
done; done
