import subprocess, sys, torch
print('cuda', torch.cuda.is_available(), torch.cuda.device_count())
x = torch.zeros(4, device='cuda'); torch.cuda.synchronize()
try:
    r = subprocess.run([sys.executable, '-c', 'print("child ok")'], capture_output=True, text=True, timeout=60)
    print('subprocess rc', r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
except Exception as e:
    print('subprocess failed:', type(e).__name__, e)
import multiprocessing as mp
def w(q): q.put('spawned child ok')
if __name__ == '__main__':
    try:
        ctx = mp.get_context('spawn'); q = ctx.Queue(); p = ctx.Process(target=w, args=(q,)); p.start(); p.join(60); print('spawn exit', p.exitcode, q.get(timeout=5) if p.exitcode == 0 else '')
    except Exception as e:
        print('spawn failed:', type(e).__name__, e)
    try:
        ctx = mp.get_context('forkserver'); q = ctx.Queue(); p = ctx.Process(target=w, args=(q,)); p.start(); p.join(60); print('forkserver exit', p.exitcode)
    except Exception as e:
        print('forkserver failed:', type(e).__name__, e)
