"""bench.py's own multi-rank flow, end to end, on whatever box runs the GPU tests: the exact command line the driver uses on
an 8-GPU node (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
--gpus N ...`), as a FRESH child process, with MVUS_BENCH_ONE_DEVICE=1 so that both ranks share cuda:0 and gloo carries the sums
(RCCL refuses two ranks on one device).  Everything else is the product path: time shards of BASELINE configs[3], the
all-reduce callback on the library's device buffers, barrier + max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_bench(nproc, extra, timeout=900):
    env = dict(os.environ, MVUS_BENCH_ONE_DEVICE='1', HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', str(nproc)] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, 'bench.py failed (rc %d):\n%s\n%s' % (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line from rank 0, got %d:\n%s' % (len(lines), r.stdout[-2000:])
    return json.loads(lines[0])


def test_two_rank_bench_on_config3_time_shards():
    """Strong scaling leg: the fixed configs[3] problem (64 cams x 2M obs) cut into two time slices."""
    # (from this start LM rejects its first trials while the damping grows: enough steps for an accepted one)
    out = _run_bench(2, ['--config', '3', '--steps', '8', '--warmup', '2'])
    assert out['n_gpus'] == 2 and out['steps'] == 8 and out['warmup'] == 2
    assert out['scaling'] == 'strong' and out['metric'] == 'residuals/sec' and out['unit'] == 'residuals/s'
    assert out['dtype'] == 'f64' and out['higher_is_better'] is True and out['vs_baseline'] is None
    cfg = out['config']
    assert 'configs[3]' in cfg['workload'] and '64 cams' in cfg['workload']
    assert cfg['parallelism'] == 'time-shard x2' and cfg['solver'] == 'lm'
    assert cfg['cost_last'] < cfg['cost_first']                                  # the sharded LM descends
    assert out['value'] > 0 and abs(out['value'] * out['ms_per_step'] * 1e-3 - 2_000_000) < 50_000   # value = all ranks' detections / time
    assert 'roofline' in out and out['roofline']['obs_per_launch'] < 1_100_000   # each rank times its own slice's kernel
    assert 'cpu_baseline' not in out                                             # an N=1 report


def test_two_rank_bench_weak_scaling_default_config():
    """The driver's SCALE runs use the default workload: configs[2]-shaped, ~500k observations per rank (weak scaling)."""
    out = _run_bench(2, ['--steps', '2', '--warmup', '1'])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak'
    assert 'configs[2]' in out['config']['workload'] and out['config']['parallelism'] == 'time-shard x2'
    assert out['config']['cost_last'] < out['config']['cost_first']
    assert abs(out['value'] * out['ms_per_step'] * 1e-3 - 1_000_000) < 50_000      # ~500k per rank, summed over the ranks
    # ... and the same line carries the node's STRONG-scaling figure on the configuration BASELINE.json lists for it (configs[3])
    s3 = out['strong_config3']
    assert s3['n_gpus'] == 2 and 'configs[3]' in s3['workload'] and '64 cams' in s3['workload']
    assert s3['ms_per_step'] > 0 and abs(s3['residuals_per_sec'] * s3['ms_per_step'] * 1e-3 - 2_000_000) < 50_000
    assert s3['cost_last'] <= s3['cost_first']
    ls = out['long_solve']                                                        # the same trials inside one call, beside the headline
    assert ls['trials'] >= 1 and ls['ms_per_trial'] > 0 and ls['linearisations'] <= ls['trials'] + 1
