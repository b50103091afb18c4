"""The whole data-parallel step with TWO ranks holding DIFFERENT data (gloo rehearsal: both ranks on the one GPU of the
test box, collectives staged through the host): FusedSGD's flat gradient buffer, the "skip parameters without a gradient"
flags, the two-stream encode, the weight-gradient side stream and its join ahead of the collective, the gradient buckets
(synchronous under gloo) -- against a single-process run with the same semantics (per-rank BatchNorm statistics, loss over
the global batch).  SURVEY 8e; reference entry script /root/reference/scripts/train_multi.py:77-107 is single-GPU."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, 'scripts', 'dp_world2_check.py')


def _run(tmp_path, method, port, steps, accumulate=1, precision='-', collective='all_reduce'):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', MPR_DIST_BACKEND='gloo',
               MPR_DP_COLLECTIVE=collective)
    tail = [str(accumulate), precision]
    procs = [subprocess.Popen([sys.executable, SCRIPT, 'rank', str(tmp_path), method, str(steps)] + tail, cwd=ROOT,
                              env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-1500:] for o in outs)
    r = subprocess.run([sys.executable, SCRIPT, 'ref', str(tmp_path), method, str(steps)] + tail, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    return torch.load(tmp_path / 'ref.pt'), [torch.load(tmp_path / f'rank{k}.pt') for k in range(2)]


@pytest.mark.parametrize('method,port', [('clip', 29561), ('siglipplus', 29563)])
def test_dp_step_world2_equals_single_process_reference(tmp_path, method, port):
    """ONE optimizer step: summed per-rank gradients == gradients of the global loss, parameter by parameter (the only
    differences are the summation orders of fp32 atomics and of the all-reduce)."""
    ref, ranks = _run(tmp_path, method, port, 1)
    for got in ranks:
        assert abs(got['losses'][0] - ref['losses'][0]) <= 1e-4 * max(1.0, abs(ref['losses'][0]))
        for k, v in ref['params'].items():
            err = float((got['params'][k] - v).abs().max())
            assert err <= 5e-3 * float(v.abs().max()) + 3e-5, (k, err, float(v.abs().max()))
    for k in ranks[0]['params']:                         # replicas stay bit-identical
        assert torch.equal(ranks[0]['params'][k], ranks[1]['params'][k]), k


def test_dp_training_world2_tracks_reference_over_steps(tmp_path):
    """Three steps: the loss curves stay together (parameters drift apart at the rate any two summation orders do in a
    train-mode BatchNorm net) and the replicas stay bit-identical."""
    ref, ranks = _run(tmp_path, 'clip', 29565, 3)
    for got in ranks:
        for a, b in zip(got['losses'], ref['losses']):
            assert abs(a - b) <= 5e-3 * max(1.0, abs(b)), (got['losses'], ref['losses'])
    for k in ranks[0]['params']:
        assert torch.equal(ranks[0]['params'][k], ranks[1]['params'][k]), k


def test_dp_accumulation_fp32_mode_world2_is_tight(tmp_path):
    """Two optimizer steps of two micro-batches each (accumulate_grad_batches 2, as the reference's own card accumulates:
    /root/reference/model_cards/example_multi.yaml:36-42), gradient sum by reduce-scatter + all-gather, conv stacks in the
    fp32 parity mode: no atomics anywhere, so sharded and single-process runs agree to fp32 summation order."""
    ref, ranks = _run(tmp_path, 'clip', 29569, 2, accumulate=2, precision='32', collective='rs_ag')
    assert len(ref['losses']) == 4
    for got in ranks:
        for a, b in zip(got['losses'], ref['losses']):
            assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (got['losses'], ref['losses'])
        for k, v in ref['params'].items():
            err = float((got['params'][k] - v).norm() / v.norm().clamp_min(1e-12))
            assert err <= 2e-4, (k, err)
    for k in ranks[0]['params']:
        assert torch.equal(ranks[0]['params'][k], ranks[1]['params'][k]), k


@pytest.mark.parametrize('accumulate,interval,collective,port', [(1, None, 'all_reduce', 29567), (2, 0.5, 'rs_ag', 29571)])
def test_train_multi_entrypoint_data_parallel_world2(tmp_path, accumulate, interval, collective, port):
    """scripts/train_multi.py launched as two ranks (what torchrun does: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*):
    sharded sampler, DataParallelStep inside Trainer.fit, validation loss averaged over ranks, rank 0 alone writes the run
    directory with Lightning's checkpoint naming (reference: /root/reference/scripts/train_multi.py:86-107)."""
    import json
    import yaml
    card = yaml.safe_load(open(os.path.join(ROOT, 'model_cards', 'smoke_multi.yaml')))
    card['trainer_args']['accumulate_grad_batches'] = accumulate
    card['trainer_args']['val_check_interval'] = interval
    card['bs'] = 8
    cpath = tmp_path / 'dp_smoke.yaml'
    cpath.write_text(yaml.safe_dump(card))
    logdir = tmp_path / 'logs'
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', MPR_DIST_BACKEND='gloo',
               MPR_DP_COLLECTIVE=collective)
    cmd = [sys.executable, 'train_multi.py', '-m', str(cpath), '--synthetic', '64', '--max-epochs', '1',
           '--logdir', str(logdir)]
    procs = [subprocess.Popen(cmd, cwd=os.path.join(ROOT, 'scripts'), env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-2000:] for o in outs)
    assert 'Training from model card' in outs[0] and 'Training from model card' not in outs[1]
    runs = list(logdir.glob('*/version_*'))
    assert len(runs) == 1                                          # rank 0 only
    metrics = [json.loads(l) for l in (runs[0] / 'metrics.jsonl').read_text().splitlines()]
    assert any('valid_loss' in m for m in metrics) and any('train_loss' in m for m in metrics)
    ckpts = list((runs[0] / 'checkpoints').glob('epoch=0_valid_loss=*.ckpt'))
    assert len(ckpts) == (2 if interval else 1)          # save_top_k: 2
    # val_check_interval 0.5: two validation runs inside the one epoch (64 samples / 2 ranks / batch 8 = 4 batches)
    assert sum('valid_loss' in m for m in metrics) == (2 if interval else 1)
