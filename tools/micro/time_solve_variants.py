"""LM step time (20 two-evaluation solves) for every variants/libmvusba_*.so on a BASELINE config."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, time; sys.path.insert(0, %r)
import torch
from mvus_amd import synth, problem as mp, ba
sc = synth.baseline_scene(int(sys.argv[1])); prob, x0 = mp.problem_from_scene(sc)
h = ba.BAHandle(prob); x = x0.copy()
for _ in range(5): x = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False).x
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): r = h.solve(x, solver=ba.SOLVER_LM_SCHUR, jac_mode=ba.JAC_ANALYTIC, max_nfev=2, return_fun=False); x = r.x
torch.cuda.synchronize(); print('%%.4f ms/step, cost %%.10g' %% ((time.perf_counter() - t0) * 1e3 / 40, r.cost))
''' % ROOT
cfg = sys.argv[1] if len(sys.argv) > 1 else '2'
for so in sorted(glob.glob(os.path.join(ROOT, 'variants', 'libmvusba_*.so'))):
    out = subprocess.run([sys.executable, '-c', code, cfg], env=dict(os.environ, MVUS_LIB_PATH=so), capture_output=True, text=True)
    print(os.path.basename(so), out.stdout.strip() or out.stderr.strip()[-300:])
