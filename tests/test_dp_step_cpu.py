"""CPU / gloo, world_size 2: the WHOLE distributed.DataParallelStep (zero_grad -> encode -> sharded contrastive loss ->
backward -> flat-buffer SUM all-reduce -> optimizer step) against a single-process step on the concatenated global batch.

The HIP encoders need a GPU, so the model here is a test double with the same surface (encode(), loss with the package's
CLIPLoss / SigLIPLoss parameters, train_loss) built from plain torch layers WITHOUT batch coupling -- then the sharded step
must reproduce the global step exactly: same loss, same parameters after every step, on both ranks.  The per-rank loss
arithmetic is the torch double of tests/test_distributed_cpu.py.  (The FusedSGD flat-buffer path of the same class runs
on the GPU box: tests/test_dp_world2_gpu.py.)"""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp
from torch import nn

from oracle import coordination as OC
from test_distributed_cpu import TorchMath, _free_port


class _Double(nn.Module):
    def __init__(self, method):
        super().__init__()
        from multimodal_plankton_recognition_amd.coordination import CLIPLoss, CLIPPlus, SigLIPLoss, SigLIPPlus
        torch.manual_seed(3)
        self.image_encoder = nn.Sequential(nn.Linear(12, 16), nn.Tanh(), nn.Linear(16, 8, bias=False))
        self.profile_encoder = nn.Sequential(nn.Linear(7, 16), nn.Tanh(), nn.Linear(16, 8, bias=False))
        self.unused = nn.Parameter(torch.ones(3))          # never receives a gradient: must be left alone
        self.loss = {'clip': CLIPLoss, 'siglip': SigLIPLoss, 'clipplus': lambda: CLIPPlus(beta=.25),
                     'siglipplus': lambda: SigLIPPlus(beta=.25)}[method]()
        self.train_loss = []

    def encode(self, image, profile, **kw):
        return {'image_emb': self.image_encoder(image), 'profile_emb': self.profile_encoder(profile)}


def _data(world, b, step):
    rs = np.random.RandomState(100 + step)
    return (torch.from_numpy(rs.standard_normal((world * b, 12)).astype(np.float32)),
            torch.from_numpy(rs.standard_normal((world * b, 7)).astype(np.float32)))


def _oracle_loss(method, core, a, p):
    if method == 'clip':
        return OC.clip_loss(a, p, core.logit_scale, 1)
    if method == 'clipplus':
        return OC.clip_plus(a, p, core.logit_scale, 1, .25)
    if method == 'siglip':
        return OC.siglip_loss(a, p, core.logit_scale, core.bias, 1)
    return OC.siglip_plus(a, p, core.logit_scale, core.bias, 1, .25)


def _worker(rank, world, port, b, method, steps, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from multimodal_plankton_recognition_amd import distributed as D
    torch.set_num_threads(1)
    D.init(backend='gloo')
    model = _Double(method)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
    stepper = D.DataParallelStep(model, opt, world, math=TorchMath())
    losses = []
    for s in range(steps):
        img, prof = _data(world, b, s)
        sl = slice(rank * b, (rank + 1) * b)
        losses.append(float(stepper.step({'image': img[sl], 'profile': prof[sl], 'buckets': 1})))
    out[rank] = (losses, {k: v.detach().numpy().copy() for k, v in model.state_dict().items()})
    D.barrier()
    D.shutdown()


@pytest.mark.parametrize('method', ['clip', 'siglip', 'clipplus', 'siglipplus'])
def test_whole_dp_step_world2_equals_global_step(method):
    world, b, steps = 2, 6, 3
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), b, method, steps, out), nprocs=world, join=True)
    # single process, global batch
    model = _Double(method)
    core = model.loss if method in ('clip', 'siglip') else (model.loss.clip if method == 'clipplus' else model.loss.siglip)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
    ref_losses = []
    for s in range(steps):
        img, prof = _data(world, b, s)
        opt.zero_grad()
        emb = model.encode(img, prof)
        loss = _oracle_loss(method, core, emb['image_emb'], emb['profile_emb'])
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    ref_sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    for rank in range(world):
        losses, sd = out[rank]
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
        for k in ref_sd:
            np.testing.assert_allclose(sd[k], ref_sd[k], rtol=2e-4, atol=2e-6, err_msg=f'rank {rank} {k}')
        assert np.array_equal(sd['unused'], np.ones(3, np.float32))


def test_dp_step_rejects_buckets_and_rank_loss():
    from multimodal_plankton_recognition_amd import distributed as D
    from multimodal_plankton_recognition_amd.coordination import RankLoss
    m = _Double('clip')
    m.loss = RankLoss(margin=0.25)
    with pytest.raises(NotImplementedError):
        D.DataParallelStep(m, torch.optim.SGD(m.parameters(), lr=0.1), 1, comm=object(), math=TorchMath())


# ---------------------------------------------------------------------------------------------------------------
# Round 3: gradient accumulation under DP (the reference's own card sets accumulate_grad_batches: 4,
# model_cards/example_multi.yaml:36-42), the reduce-scatter + all-gather form of the gradient sum, the GLOBAL-batch
# validation loss, and the replica check of the first steps.
def _worker_acc(rank, world, port, b, method, windows, of, collective, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from multimodal_plankton_recognition_amd import distributed as D
    torch.set_num_threads(1)
    D.init(backend='gloo')
    model = _Double(method)
    if rank == 1:                                   # replicas that did NOT start equal: the broadcast must repair it
        with torch.no_grad():
            model.image_encoder[0].weight.add_(0.5)
    with pytest.raises(ValueError):
        D.Comm(collective='ring')
    comm = D.Comm(collective=collective)
    D.broadcast_module(model, comm)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
    stepper = D.DataParallelStep(model, opt, world, comm=comm, math=TorchMath())
    losses, s = [], 0
    for _ in range(windows):
        for micro in range(of):
            img, prof = _data(world, b, s)
            sl = slice(rank * b, (rank + 1) * b)
            losses.append(float(stepper.step({'image': img[sl], 'profile': prof[sl], 'buckets': 1}, micro=micro, of=of)))
            s += 1
    img, prof = _data(world, b, 999)
    sl = slice(rank * b, (rank + 1) * b)
    model.valid_loss = []
    with torch.no_grad():
        val = float(stepper.validation_step({'image': img[sl], 'profile': prof[sl], 'buckets': 1}))
    ok_before = list(stepper.verified)
    # a replica that drifts (what a gradient written after its bucket was reduced would cause) is detected and repaired
    if rank == 1:
        with torch.no_grad():
            model.profile_encoder[0].bias.add_(1e-3)
    detected = not stepper.verify_replicas()
    resynced = stepper.verify_replicas()
    out[rank] = (losses, {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}, val, ok_before, detected,
                 resynced)
    D.barrier()
    D.shutdown()


@pytest.mark.parametrize('method,collective', [('clip', 'rs_ag'), ('siglipplus', 'all_reduce'), ('clipplus', 'rs_ag')])
def test_dp_accumulation_validation_and_replica_check_world2(method, collective):
    import warnings
    world, b, windows, of = 2, 5, 2, 3
    mgr = mp.Manager()
    out = mgr.dict()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        mp.spawn(_worker_acc, args=(world, _free_port(), b, method, windows, of, collective, out), nprocs=world, join=True)
    model = _Double(method)
    core = model.loss if method in ('clip', 'siglip') else (model.loss.clip if method == 'clipplus' else model.loss.siglip)
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-3, nesterov=True)
    ref_losses, s = [], 0
    for _ in range(windows):
        opt.zero_grad()
        for _ in range(of):                          # Lightning: every micro-batch's loss / accumulate_grad_batches
            img, prof = _data(world, b, s)
            emb = model.encode(img, prof)
            loss = _oracle_loss(method, core, emb['image_emb'], emb['profile_emb'])
            (loss / of).backward()
            ref_losses.append(float(loss))
            s += 1
        opt.step()
    img, prof = _data(world, b, 999)
    with torch.no_grad():
        emb = model.encode(img, prof)
        ref_val = float(_oracle_loss(method, core, emb['image_emb'], emb['profile_emb']))
    ref_sd = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    for rank in range(world):
        losses, sd, val, ok_before, detected, resynced = out[rank]
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
        assert abs(val - ref_val) <= 2e-5 * abs(ref_val), (val, ref_val)       # the GLOBAL-batch loss, on every rank
        assert ok_before == [True, True]              # the first two optimizer steps were checked on the "hardware"
        assert detected and resynced
        for k in ref_sd:
            np.testing.assert_allclose(sd[k], ref_sd[k], rtol=2e-4, atol=2e-6, err_msg=f'rank {rank} {k}')
    for k in ref_sd:                                  # after the re-synchronisation the replicas are bit-identical again
        assert np.array_equal(out[0][1][k], out[1][1][k]), k
