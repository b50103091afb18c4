"""CPU / gloo, world_size 2 and 4: the exchange logic of the sharded CLIP loss (distributed.dp_clip) equals the
oracle's single-process CLIPLoss on the concatenated global batch -- loss, per-rank embedding gradients, and the
SUM over ranks of d(logit_scale).  The per-rank arithmetic is injected (torch on CPU, from the oracle); on the GPU
box the same dp_clip runs with distributed.HipClipMath (covered by tests/test_model_gpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import coordination as OC


class TorchMath:
    """Reference arithmetic for the math interface of distributed.dp_clip / dp_siglip (test double of HipClipMath):
    `gathered` [world, 2, b, D] normalised embeddings; role 0 = my image rows against all profiles, role 1 = my profile
    rows against all images."""

    def normalize(self, a, p):
        x = torch.stack((a.detach(), p.detach()))
        nrm = x.norm(dim=2).clamp_min(1e-12)
        return x / nrm[..., None], 1.0 / nrm

    @staticmethod
    def _blocks(gathered, rank):
        world, _, b, D = gathered.shape
        u_all, v_all = gathered[:, 0].reshape(world * b, D), gathered[:, 1].reshape(world * b, D)
        return gathered[rank, 0] @ v_all.T, gathered[rank, 1] @ u_all.T, rank * b, b

    def clip_fwd(self, gathered, logit_scale, rank, mul):
        s_img, s_prof, off, b = self._blocks(gathered, rank)
        scale = logit_scale.detach().exp()
        idx = torch.arange(b)
        lse = torch.stack((torch.logsumexp(s_img * scale, 1), torch.logsumexp(s_prof * scale, 1)))
        diag = torch.stack(((s_img * scale)[idx, off + idx], (s_prof * scale)[idx, off + idx]))
        return lse, (lse - diag).sum() * mul

    @staticmethod
    def _norm_bwd(du, u, inv, x, other, mse_coef):
        dx = inv[:, None] * (du - u * (u * du).sum(1, keepdim=True))
        return dx + mse_coef * (x - other) if mse_coef else dx

    def clip_bwd(self, gathered, logit_scale, lse, lse_all, rank, coef, uv, inv, image_emb=None, profile_emb=None,
                 mse_coef=0.0):
        world, _, b, D = gathered.shape
        s_img, s_prof, off, _ = self._blocks(gathered, rank)
        scale = logit_scale.detach().exp()
        idx = torch.arange(b)
        out, dls = [], None
        for z, S in enumerate((s_img, s_prof)):
            l = S * scale
            g = torch.exp(l - lse[z][:, None]) + torch.exp(l - lse_all[:, 1 - z].reshape(-1)[None, :])
            g[idx, off + idx] -= 2
            g = g * coef
            if z == 0:
                dls = (g * l).sum()
            y_all = gathered[:, 1 - z].reshape(world * b, D)
            x, other = (image_emb, profile_emb) if z == 0 else (profile_emb, image_emb)
            out.append(self._norm_bwd((g * scale) @ y_all, uv[z], inv[z], x, other, mse_coef))
        return out[0], out[1], dls

    @staticmethod
    def _signs(S, off):
        sg = -torch.ones_like(S)
        idx = torch.arange(S.shape[0])
        sg[idx, off + idx] = 1
        return sg

    def siglip_fwd(self, gathered, logit_scale, bias, rank, mul):
        s_img, _, off, _ = self._blocks(gathered, rank)
        z = s_img * logit_scale.detach().exp() + bias.detach()
        return -torch.nn.functional.logsigmoid(self._signs(z, off) * z).sum() * mul

    def siglip_bwd(self, gathered, logit_scale, bias, rank, coef, uv, inv, image_emb=None, profile_emb=None,
                   mse_coef=0.0):
        world, _, b, D = gathered.shape
        s_img, s_prof, off, _ = self._blocks(gathered, rank)
        scale = logit_scale.detach().exp()
        out, dls, db = [], None, None
        for z, S in enumerate((s_img, s_prof)):
            l = S * scale
            sg = self._signs(S, off)
            g = -sg * coef * torch.sigmoid(-sg * (l + bias.detach()))
            if z == 0:
                dls, db = (g * l).sum(), g.sum()
            y_all = gathered[:, 1 - z].reshape(world * b, D)
            x, other = (image_emb, profile_emb) if z == 0 else (profile_emb, image_emb)
            out.append(self._norm_bwd((g * scale) @ y_all, uv[z], inv[z], x, other, mse_coef))
        return out[0], out[1], dls, db

    def sqdiff_sum(self, a, b):
        return ((a - b) ** 2).sum()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, b, d, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from multimodal_plankton_recognition_amd import distributed as D
    torch.set_num_threads(1)
    D.init(backend='gloo')
    rs = np.random.RandomState(7)
    a_all = torch.from_numpy(rs.standard_normal((world * b, d)).astype(np.float32))
    p_all = torch.from_numpy((rs.standard_normal((world * b, d)) * 1.5 + 0.1).astype(np.float32))
    ls = torch.tensor(1.3)
    sl = slice(rank * b, (rank + 1) * b)
    loss, d_img, d_prof, dls = D.dp_clip(a_all[sl], p_all[sl], ls, D.Comm(), TorchMath())
    dls_total = D.Comm().all_reduce_sum(dls.reshape(1).clone())
    assert abs(D.max_over_ranks(float(rank)) - (world - 1)) < 1e-9
    out[rank] = (loss.item(), d_img.numpy(), d_prof.numpy(), dls_total.item())
    D.barrier()
    D.shutdown()


@pytest.mark.parametrize('world,b,d', [(2, 8, 16), (4, 5, 12)])
def test_sharded_clip_equals_global_oracle(world, b, d):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), b, d, out), nprocs=world, join=True)
    rs = np.random.RandomState(7)
    a = torch.from_numpy(rs.standard_normal((world * b, d)).astype(np.float32)).requires_grad_(True)
    p = torch.from_numpy((rs.standard_normal((world * b, d)) * 1.5 + 0.1).astype(np.float32)).requires_grad_(True)
    ls = torch.tensor(1.3, requires_grad=True)
    ref = OC.clip_loss(a, p, ls, 1)
    ref.backward()
    for rank in range(world):
        loss, d_img, d_prof, dls_total = out[rank]
        sl = slice(rank * b, (rank + 1) * b)
        assert abs(loss - ref.item()) < 1e-5
        np.testing.assert_allclose(d_img, a.grad[sl].numpy(), rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(d_prof, p.grad[sl].numpy(), rtol=1e-4, atol=1e-7)
        assert abs(dls_total - ls.grad.item()) < 1e-5


def _worker_general(rank, world, port, b, d, method, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from multimodal_plankton_recognition_amd import distributed as D
    torch.set_num_threads(1)
    D.init(backend='gloo')
    rs = np.random.RandomState(11)
    a_all = torch.from_numpy(rs.standard_normal((world * b, d)).astype(np.float32))
    p_all = torch.from_numpy((rs.standard_normal((world * b, d)) * 1.5 + 0.1).astype(np.float32))
    ls, bias = torch.tensor(1.3), torch.tensor(-4.0)
    sl = slice(rank * b, (rank + 1) * b)
    beta = .25 if method.endswith('plus') else 0.0
    comm = D.Comm()
    if method.startswith('siglip'):
        loss, d_img, d_prof, dls, db = D.dp_siglip(a_all[sl], p_all[sl], ls, bias, comm, TorchMath(), beta)
        db_total = comm.all_reduce_sum(db.reshape(1).clone()).item()
    else:
        loss, d_img, d_prof, dls = D.dp_clip(a_all[sl], p_all[sl], ls, comm, TorchMath(), beta)
        db_total = 0.0
    dls_total = comm.all_reduce_sum(dls.reshape(1).clone()).item()
    out[rank] = (loss.item(), d_img.numpy(), d_prof.numpy(), dls_total, db_total)
    D.barrier()
    D.shutdown()


@pytest.mark.parametrize('method', ['siglip', 'siglipplus', 'clipplus'])
@pytest.mark.parametrize('world,b,d', [(2, 6, 16), (4, 3, 8)])
def test_sharded_siglip_and_plus_losses_equal_global_oracle(world, b, d, method):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_general, args=(world, _free_port(), b, d, method, out), nprocs=world, join=True)
    rs = np.random.RandomState(11)
    a = torch.from_numpy(rs.standard_normal((world * b, d)).astype(np.float32)).requires_grad_(True)
    p = torch.from_numpy((rs.standard_normal((world * b, d)) * 1.5 + 0.1).astype(np.float32)).requires_grad_(True)
    ls = torch.tensor(1.3, requires_grad=True)
    bias = torch.tensor(-4.0, requires_grad=True)
    if method == 'siglip':
        ref = OC.siglip_loss(a, p, ls, bias, 1)
    elif method == 'siglipplus':
        ref = OC.siglip_plus(a, p, ls, bias, 1, .25)
    else:
        ref = OC.clip_plus(a, p, ls, 1, .25)
    ref.backward()
    for rank in range(world):
        loss, d_img, d_prof, dls_total, db_total = out[rank]
        sl = slice(rank * b, (rank + 1) * b)
        assert abs(loss - ref.item()) < 1e-4 * max(1.0, abs(ref.item()))
        np.testing.assert_allclose(d_img, a.grad[sl].numpy(), rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(d_prof, p.grad[sl].numpy(), rtol=2e-4, atol=1e-6)
        assert abs(dls_total - ls.grad.item()) < 1e-4 * max(1.0, abs(ls.grad.item()))
        if method.startswith('siglip'):
            assert abs(db_total - bias.grad.item()) < 1e-4 * max(1.0, abs(bias.grad.item()))
