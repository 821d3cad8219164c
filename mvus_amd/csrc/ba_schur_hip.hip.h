// Normal-equation assembly + Schur-complement LM solver on the HIP backend (placeholder until ba_schur lands).
#pragma once
#include "ba_solver.h"
namespace mvus {
struct HipBackend;
template <class BE> int schur_export(BE& be, double*, double*, double*, double*, int32_t*) { be.err = "normal equations: not built in this revision"; return MVUS_E_INVALID; }
template <class BE> SolveResult lm_schur_hip(BE& be, std::vector<double>&, const std::vector<double>&, const std::vector<double>&, const SolveOptions&, double*) {
  SolveResult r; r.error = -1; be.err = "LM_SCHUR: not built in this revision"; return r; }
template <class BE> void schur_time_assembly(BE&) {}
}  // namespace mvus
