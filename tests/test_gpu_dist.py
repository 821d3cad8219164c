"""GPU check of the all-reduce hook: RCCL through torch.distributed on buffers aliased from the library's device
pointers (world_size 1 on the single test GPU; the N>1 arithmetic is covered by tests/test_dist_gloo.py)."""
import os
import socket

import numpy as np
import pytest

from golden_util import load_case
from mvus_amd import _lib, problem as mp

pytestmark = pytest.mark.gpu


def test_allreduce_hook_with_rccl_world1():
    import torch
    import torch.distributed as dist
    from mvus_amd.ba import BAHandle
    from mvus_amd.dist import sharded_handle
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        scene, g = load_case('rs_F_2int_3cam')
        prob, x0 = mp.problem_from_scene(scene)
        opts = _lib.default_opts(_lib.SOLVER_TRF_LSMR, _lib.JAC_PATTERN, 6)
        opts.lsmr_maxiter = 4
        with BAHandle(prob) as h0:
            r0 = h0.solve(g['x0'], opts=opts)
        h, keep = sharded_handle(prob, 0, 1, 0, group=dist.group.WORLD)
        r1 = h.solve(g['x0'], opts=opts)
        assert h.allreduce_stats['calls'] > 10                      # the hook really ran (J^T u, dot products)
        assert keep.size == prob.M
        np.testing.assert_allclose(r1.cost, r0.cost, rtol=1e-12)
        np.testing.assert_allclose(r1.x, r0.x, rtol=0, atol=1e-10)
        h.close()
    finally:
        dist.destroy_process_group()
