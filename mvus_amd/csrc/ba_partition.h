// Partition of the control-point chain for the band solver of ba_schur_hip.hip.h: plain host C++ (no HIP), so that the
// test-only host build can check its invariants.
#pragma once
#include <algorithm>
#include <vector>

namespace mvus {

#ifndef MVUS_PART_L
#define MVUS_PART_L 32
#endif
constexpr int kPartL = MVUS_PART_L;                  // control points per interior
constexpr int kPartRowsMax = 3 * (kPartL + 6);       // scalar rows of the longest interior (merged tails included)

// Partition of a chain of `n` control points starting at local control point `c0`: interiors of `len` (<= kPartL) control points
// separated by separators of sctrl (every interior BETWEEN two separators is at least sctrl long, so that separators
// never couple directly); `close` = the chain must END with a separator (the cut towards the next time shard),
// otherwise a tail too short for another interior is merged into the last one.
struct ChainPart { std::vector<int> i0, i1, sep; };     // scalar rows; sep[k] = separator right of interior k
inline ChainPart partition_chain(int c0, int n, int sctrl, bool close, int len = kPartL) {
  len = std::max(2 * sctrl, std::min(len, kPartL));       // the kernels' arrays are sized for kPartL
  ChainPart cp;
  const int end = c0 + n;
  for (int g = c0; g < end;) {
    int e = std::min(g + len, end);
    if (close) {
      e = std::min(g + len, end - sctrl);
      // control points between this interior and the closing separator: none, or a separator plus an interior of at
      // least sctrl control points -- a shorter interior would let its two separators couple directly through the band,
      // which the reduced (block tridiagonal) separator system cannot express.  Shorten this interior to leave exactly that.
      const int rem = end - sctrl - e;
      if (rem > 0 && rem < 2 * sctrl) e = end - 3 * sctrl;
    }
    cp.i0.push_back(3 * g); cp.i1.push_back(3 * e); g = e;
    if (g < end) {
      const int e2 = std::min(g + sctrl, end);
      if (!close && end - e2 < 1) { cp.i1.back() = 3 * end; g = end; }          // tail too short for another interior: merge
      else { cp.sep.push_back(3 * g); g = e2; }
    }
  }
  return cp;
}


}  // namespace mvus
