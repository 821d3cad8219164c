"""Time the fused assembly kernel of every variants/libmvusba_*.so on BASELINE configs[2] (phase probes)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys; sys.path.insert(0, %r)
from mvus_amd import synth, problem as mp, ba
sc = synth.baseline_scene(int(sys.argv[1])); prob, x0 = mp.problem_from_scene(sc)
h = ba.BAHandle(prob); h.set_x(x0)
print('fused %%.1f us, from J %%.1f us' %% (1e3 * h.time_kernel(ba.KERNEL_FUSED_ASSEMBLY, 30), 1e3 * h.time_kernel(ba.KERNEL_ASSEMBLY, 30)))
''' % ROOT
cfg = sys.argv[1] if len(sys.argv) > 1 else '2'
for so in sorted(glob.glob(os.path.join(ROOT, 'variants', 'libmvusba_*.so'))):
    out = subprocess.run([sys.executable, '-c', code, cfg], env=dict(os.environ, MVUS_LIB_PATH=so), capture_output=True, text=True)
    print(os.path.basename(so), out.stdout.strip() or out.stderr.strip()[-300:])
