"""Time the residual+Jacobian kernel (rotating outputs) of every variants/libmvusba_*.so on BASELINE configs[2]."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys; sys.path.insert(0, %r)
from mvus_amd import synth, problem as mp, ba
sc = synth.baseline_scene(int(sys.argv[1])); prob, x0 = mp.problem_from_scene(sc)
h = ba.BAHandle(prob); h.set_x(x0)
t = [h.time_kernel(ba.KERNEL_RESIDUAL_JACOBIAN, 100) for _ in range(3)]
t1 = h.time_kernel(ba.KERNEL_RESIDUAL_JACOBIAN_ONE_BUFFER, 50)
ta = h.time_kernel(ba.KERNEL_ASSEMBLY, 20)
print('rotating %%s us, one buffer %%.1f us, assembly %%.1f us' %% (' '.join('%%.1f' %% (1e3 * v) for v in t), 1e3 * t1, 1e3 * ta))
''' % ROOT
cfg = sys.argv[1] if len(sys.argv) > 1 else '2'
for so in sorted(glob.glob(os.path.join(ROOT, 'variants', 'libmvusba_*.so'))):
    env = dict(os.environ, MVUS_LIB_PATH=so)
    out = subprocess.run([sys.executable, '-c', code, cfg], env=env, capture_output=True, text=True)
    print(os.path.basename(so), out.stdout.strip() or out.stderr.strip()[-400:])
