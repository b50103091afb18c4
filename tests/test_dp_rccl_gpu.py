"""The data-parallel step through real RCCL on one GPU (world size 1): every collective call of
distributed.DataParallelStep runs on the `nccl` backend and the losses equal the single-GPU step's."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('card', ['smoke_multi.yaml'])
def test_dp_step_over_rccl_world_one(card):
    env = dict(os.environ, MASTER_PORT='29541', RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'dp_selftest.py'),
                        os.path.join(ROOT, 'model_cards', card), '16'], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and 'dp_selftest ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
