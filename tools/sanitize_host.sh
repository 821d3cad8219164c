#!/bin/bash
# AddressSanitizer + UBSan over the host build of the shared device / solver headers (tests/hostcheck), driven by the CPU tests.
# GPU sanitizers are not available on the pool; this covers ba_math.h, ba_solver.h, ba_schur.h, ba_partition.h, triangulate.hip.h
# and the host side of spline_fit.hip.h.       tools/sanitize_host.sh        (build container; ~1.5 min)
set -e
cd "$(dirname "$0")/.."
H=tests/hostcheck
cp $H/libhostcheck.so /tmp/libhostcheck_plain.so 2>/dev/null || true
g++ -O1 -g -std=c++17 -shared -fPIC -DMVUS_WITH_SCHUR -fsanitize=address,undefined -fno-omit-frame-pointer -o $H/libhostcheck.so $H/hostcheck.cpp $H/host_backend.cpp
ASAN_OPTIONS=detect_leaks=0:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
  python -m pytest tests/test_solver_host.py tests/test_device_math_host.py tests/test_fd_mode_host.py tests/test_traj_to_spline.py \
                   tests/test_triangulate.py tests/test_host_logic.py -q -m "not gpu" 2>&1 | grep -E "passed|failed|runtime error|AddressSanitizer|SUMMARY" || true
rm -f $H/libhostcheck.so
[ -f /tmp/libhostcheck_plain.so ] && cp /tmp/libhostcheck_plain.so $H/libhostcheck.so
