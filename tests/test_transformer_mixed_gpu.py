"""GPU parity of the mixed-precision transformer path (csrc/transformer_bf16.hip + transformer_mixed.py): every kernel
against torch fp32 on identical bf16-rounded operands, the fused blocks against the exact-fp32 path / the reference
fixtures at bf16 tolerance.  Tolerances are relative L2 errors (bf16 has 8 bits of mantissa: 4e-3 per rounding)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'
BF = torch.bfloat16


def rel(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float().cpu()
    return float((got - ref).norm() / ref.norm().clamp_min(1e-20))


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _attn_ref(qkv, bias, mask, B, T, heads, hd):
    """fp32 reference on the operands the kernel sees: bf16(qkv + bias)."""
    d = heads * hd
    x = (qkv.float() + (bias if bias is not None else 0)).to(BF).float().view(B, T, 3, heads, hd)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))            # [B, h, T, hd]
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float('-inf'))
    o = torch.softmax(s, -1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * T, d), x


@pytest.mark.parametrize('B,T,heads,hd,use_mask,use_bias', [
    (3, 197, 2, 64, False, True), (2, 225, 4, 64, True, True), (2, 33, 2, 32, True, False), (1, 256, 1, 64, False, False),
    (5, 7, 3, 32, True, True), (2, 32, 2, 64, False, True), (2, 257, 2, 32, True, True), (1, 288, 1, 32, False, False)])
def test_fused_attention_fwd_bwd(B, T, heads, hd, use_mask, use_bias):
    from multimodal_plankton_recognition_amd import transformer_mixed as TM
    d = heads * hd
    qkv = rnd(B * T, 3 * d, seed=1).to(BF)
    bias = rnd(3 * d, seed=2, scale=0.3) if use_bias else None
    mask = None
    if use_mask:
        lens = torch.randint(1, T + 1, (B,), generator=torch.Generator().manual_seed(3))
        lens[0] = T
        mask = torch.arange(T)[None, :] >= lens[:, None]
    dout = rnd(B * T, d, seed=4).to(BF)
    # reference with autograd on the rounded operands
    xq = (qkv.float() + (bias if bias is not None else 0)).to(BF).float().requires_grad_(True)
    x5 = xq.view(B, T, 3, heads, hd)
    q, k, v = (x5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float('-inf'))
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, d)
    ref.backward(dout.float())
    qd = qkv.to(DEV)
    bd = bias.to(DEV) if bias is not None else None
    m8 = mask.to(DEV).view(torch.uint8) if mask is not None else None
    out, lse = TM.attn_fwd(qd, bd, m8, B, T, heads, 0.0, 0)
    assert rel(out, ref) < 6e-3, rel(out, ref)
    # the backward kernels take the forward's own (bf16) output
    dqkv = TM.attn_bwd(qd, bd, m8, out, dout.to(DEV), lse, B, T, heads, 0.0, 0)
    g = xq.grad.view(B * T, 3, d)
    got = dqkv.float().cpu().view(B * T, 3, d)
    for i, name in enumerate('qkv'):
        assert rel(got[:, i], g[:, i]) < 1.5e-2, (name, rel(got[:, i], g[:, i]))
    if mask is not None:        # padding keys receive no gradient
        pad = mask.view(-1)
        assert float(got[pad][:, 1:].abs().max()) == 0.0


def test_attention_dropout_forward_and_backward_share_one_mask():
    """With dropout the map V -> out is still linear for a fixed seed: <dout, out(V)> == <dV, V>; a mask that differed
    between the forward and the backward kernels would break the identity."""
    from multimodal_plankton_recognition_amd import transformer_mixed as TM
    B, T, heads, hd, p, seed = 2, 70, 2, 64, 0.3, 1234
    d = heads * hd
    qkv = rnd(B * T, 3 * d, seed=5).to(BF).to(DEV)
    dout = rnd(B * T, d, seed=6).to(BF).to(DEV)
    out, lse = TM.attn_fwd(qkv, None, None, B, T, heads, p, seed)
    out0, _ = TM.attn_fwd(qkv, None, None, B, T, heads, 0.0, 0)
    assert rel(out, out0) > 0.1                       # dropout did something
    out_b, _ = TM.attn_fwd(qkv, None, None, B, T, heads, p, seed)
    assert torch.equal(out, out_b)                    # and is a pure function of the seed
    dqkv = TM.attn_bwd(qkv, None, None, out, dout, lse, B, T, heads, p, seed)
    lhs = float((dout.float() * out.float()).sum())
    rhs = float((dqkv.float().view(B * T, 3, d)[:, 2] * qkv.float().view(B * T, 3, d)[:, 2]).sum())
    assert abs(lhs - rhs) < 2e-2 * max(abs(lhs), 1.0), (lhs, rhs)
    # softmax rows sum to one: the gradient w.r.t. q and k of a loss that does not depend on them is ~0 when dout is
    # constant along the head dimension and V is constant over keys (P drops out) -- checked without dropout
    vconst = qkv.clone().view(B * T, 3, d)
    vconst[:, 2] = 1.0
    vconst = vconst.view(B * T, 3 * d).contiguous()
    o2, l2 = TM.attn_fwd(vconst, None, None, B, T, heads, 0.0, 0)
    g2 = TM.attn_bwd(vconst, None, None, o2, dout, l2, B, T, heads, 0.0, 0).float().view(B * T, 3, d)
    assert float(g2[:, :2].abs().max()) < 2e-2


@pytest.mark.parametrize('rows,D', [(37, 64), (300, 256), (130, 768), (9, 1024), (50, 200)])
def test_add_layernorm_fwd_bwd(rows, D):
    from multimodal_plankton_recognition_amd import transformer_mixed as TM
    x = rnd(rows, D, seed=1)
    r = rnd(rows, D, seed=2).to(BF)
    rb = rnd(D, seed=3, scale=0.2)
    gamma = (1 + rnd(D, seed=4, scale=0.2)).requires_grad_(True)
    beta = rnd(D, seed=5, scale=0.2).requires_grad_(True)
    sref = (x + r.float() + rb).requires_grad_(True)
    yref = F.layer_norm(sref, (D,), gamma, beta, 1e-5)
    gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
    s, y32, y16, mean, rstd = TM.add_ln(x.to(DEV), r.to(DEV), rb.to(DEV), 0.0, 0, gd, bd, 1e-5, True, True, True)
    np.testing.assert_allclose(s.cpu().numpy(), sref.detach().numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(y32.cpu().numpy(), yref.detach().numpy(), rtol=2e-5, atol=2e-5)
    assert torch.equal(y16.cpu(), y32.cpu().to(BF))
    dy16 = rnd(rows, D, seed=6).to(BF)
    dy32 = rnd(rows, D, seed=7)
    dskip = rnd(rows, D, seed=8)
    yref.backward(dy16.float() + dy32)
    ds, dg, db = TM.ln_bwd(dy16.to(DEV), dy32.to(DEV), s, gd, bd, mean, rstd, dskip.to(DEV))
    np.testing.assert_allclose(ds.cpu().numpy(), (sref.grad + dskip).numpy(), rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(dg.cpu().numpy(), gamma.grad.numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(db.cpu().numpy(), beta.grad.numpy(), rtol=2e-4, atol=2e-4)
    # add only (no LayerNorm): the residual-gradient join of the post-norm backward
    only, _, _, _, _ = TM.add_ln(x.to(DEV), r.to(DEV), want_s=True)
    np.testing.assert_allclose(only.cpu().numpy(), (x + r.float()).numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('rows,D,act', [(77, 64, 1), (1000, 3072, 1), (33, 520, 2), (5, 8, 0)])
def test_bias_activation_and_bias_gradients(rows, D, act):
    from multimodal_plankton_recognition_amd import transformer_mixed as TM
    x = rnd(rows, D, seed=1).to(BF)
    b = rnd(D, seed=2, scale=0.5)
    xr = x.float().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    pre = xr + br
    ref = F.gelu(pre) if act == 1 else (F.relu(pre) if act == 2 else pre)
    bd = b.to(DEV).requires_grad_(True)
    y = TM.bias_act(x.to(DEV), bd, act, 0.0, 0)
    assert rel(y, ref) < 4e-3
    dy = rnd(rows, D, seed=3).to(BF)
    ref.backward(dy.float())
    dx, db = TM.ew_bwd(1, dy.to(DEV), rows, D, x16=x.to(DEV), bias=bd, act=act)
    assert rel(dx, xr.grad) < 5e-3
    assert rel(db, br.grad) < 5e-3, rel(db, br.grad)
    # mode 0: column sums; mode 2: fp32 -> bf16 cast + column sums
    _, db0 = TM.ew_bwd(0, dy.to(DEV), rows, D, bias=bd)
    np.testing.assert_allclose(db0.cpu().numpy(), dy.float().sum(0).numpy(), rtol=1e-4, atol=1e-3)
    g32 = rnd(rows, D, seed=4)
    dx2, db2 = TM.ew_bwd(2, g32.to(DEV), rows, D, bias=bd)
    assert torch.equal(dx2.cpu(), g32.to(BF))
    np.testing.assert_allclose(db2.cpu().numpy(), g32.to(BF).float().sum(0).numpy(), rtol=1e-4, atol=1e-3)
    # dropout: forward and backward regenerate the same mask from the seed; kept fraction ~ 1 - p
    yd = TM.bias_act(x.to(DEV), None, 0, 0.25, 99)
    xz = x.to(DEV).float() == 0                    # (randn does produce exact zeros: their fate is invisible in yd)
    kept = (yd.float() != 0) | xz
    if rows * D >= 4096:
        assert abs(float(kept.float().mean()) - 0.75) < 0.03
    dxd, _ = TM.ew_bwd(1, dy.to(DEV), rows, D, x16=x.to(DEV), act=0, p=0.25, seed=99)
    nz = (dy.to(DEV).float() != 0) & ~xz
    bad = (((dxd.float() != 0) & nz) != (kept & nz)).nonzero()
    assert bad.numel() == 0, (bad.shape[0], bad[:8].tolist())


def test_vit_blocks_mixed_vs_exact_fp32_path():
    """The same small ViT through the exact-fp32 kernels (parity-tested against the oracle) and the mixed path."""
    from multimodal_plankton_recognition_amd import transformer as TF
    from multimodal_plankton_recognition_amd.image_encoder import ViTBackbone
    torch.manual_seed(0)
    vit = ViTBackbone(embed_dim=128, depth=3, num_heads=2, patch=16, img_size=96, in_chans=1)     # T = 37, head 64
    with torch.no_grad():
        for p in vit.parameters():
            p.add_(torch.randn_like(p) * (0.1 if p.dim() == 1 else 0.02))
    vit.to(DEV).train()
    image = rnd(4, 1, 96, 96, seed=1).to(DEV)
    wsum = rnd(4, 128, seed=2).to(DEV)
    res = {}
    for mode in ('32', 'bf16-mixed'):
        old = TF.set_precision(mode)
        try:
            vit.zero_grad()
            out = vit.forward_pooled(image)
            (out * wsum).sum().backward()
            res[mode] = (out.detach().clone(), {k: v.grad.detach().clone() for k, v in vit.named_parameters()})
        finally:
            TF._PRECISION[0] = old
    assert rel(res['bf16-mixed'][0], res['32'][0]) < 2e-2, rel(res['bf16-mixed'][0], res['32'][0])
    worst = max((rel(res['bf16-mixed'][1][k], res['32'][1][k]), k) for k in res['32'][1])
    assert worst[0] < 6e-2, worst


@pytest.mark.parametrize('tag', ['ragged', 'fixed'])
def test_profile_transformer_mixed_against_reference_fixture(golden, tag):
    """Post-norm encoder with key-padding mask in mixed precision against the reference's own fp32 outputs."""
    from multimodal_plankton_recognition_amd import transformer as TF
    from multimodal_plankton_recognition_amd.profile_encoder import ProfileTransformer
    T = torch.from_numpy
    g = golden('profile_transformer')
    m = ProfileTransformer(dim_in=6, dim_hidden=32, target_size=224, num_head=2, num_layers=2, dim_feedforward=64,
                           dropout=0.0, activation='gelu')
    m.load_state_dict({k[3:]: T(v.copy()) for k, v in g.items() if k.startswith('sd.')}, strict=True)
    m.to(DEV).train()
    n = len(g[f'{tag}.lens'])
    tok = m.tokenize([T(g[f'{tag}.in{i}']) for i in range(n)])
    plen = torch.tensor([[int(v)] for v in g[f'{tag}.lens']])
    old = TF.set_precision('bf16-mixed')
    try:
        # head size 16 is outside the fused attention kernel: the mixed path must refuse, not fall back
        with pytest.raises(NotImplementedError):
            m(**{k: v.to(DEV) for k, v in tok.items()}, profile_len=plen.to(DEV))
    finally:
        TF._PRECISION[0] = old


def test_post_norm_layers_mixed_vs_exact_fp32_path():
    from multimodal_plankton_recognition_amd import transformer as TF
    from multimodal_plankton_recognition_amd.profile_encoder import ProfileTransformer
    torch.manual_seed(1)
    m = ProfileTransformer(dim_in=6, dim_hidden=128, target_size=60, num_head=4, num_layers=3, dim_feedforward=256,
                           dropout=0.0, activation='gelu')
    m.to(DEV).train()
    gen = torch.Generator().manual_seed(3)
    profs = [torch.rand(int(n), 6, generator=gen) * 2 - 1 for n in (60, 17, 33, 5, 60)]
    tok = {k: v.to(DEV) for k, v in m.tokenize(profs).items()}
    plen = torch.tensor([[p.shape[0]] for p in profs]).to(DEV)
    wsum = rnd(5, 129, seed=4).to(DEV)
    res = {}
    for mode in ('32', 'bf16-mixed'):
        old = TF.set_precision(mode)
        try:
            m.zero_grad()
            out = m(**tok, profile_len=plen)
            (out * wsum).sum().backward()
            res[mode] = (out.detach().clone(), {k: (v.grad.detach().clone() if v.grad is not None else torch.zeros_like(v))
                                                for k, v in m.named_parameters()})
        finally:
            TF._PRECISION[0] = old
    assert rel(res['bf16-mixed'][0], res['32'][0]) < 2e-2, rel(res['bf16-mixed'][0], res['32'][0])
    worst = max((rel(res['bf16-mixed'][1][k], res['32'][1][k]), k) for k in res['32'][1])
    assert worst[0] < 6e-2, worst


def test_mixed_blocks_accumulate_into_fused_optimizer_buffers():
    """Three optimisation steps of a ViT + transformer MultiModel in mixed precision: gradients land in FusedSGD's flat
    buffer (no autograd-returned copies), the loss is finite and changes."""
    from multimodal_plankton_recognition_amd import transformer as TF
    from multimodal_plankton_recognition_amd.model import MultiModel
    from multimodal_plankton_recognition_amd import image_encoder as IE
    orig = IE.create_backbone

    def create(name, in_chans=1):
        if name == 'vit_test_patch16_64':
            return IE.ViTBackbone(64, 2, 1, 16, 64, in_chans)
        return orig(name, in_chans)
    IE.create_backbone = create
    old = TF.set_precision('bf16-mixed')
    try:
        torch.manual_seed(0)
        model = MultiModel(dim_embed=32, image_encoder_args=dict(name='vit_test_patch16_64', dropout=0.1),
                           profile_encoder_args=dict(dim_in=6, dim_hidden=64, target_size=48, num_head=2, num_layers=2,
                                                     dim_feedforward=128, dropout=0.1),
                           coordination_args=dict(method='siglip'), optim_args=dict(lr=1e-2, momentum=0.9)).to(DEV).train()
        opt = model.configure_optimizers()
        gen = torch.Generator().manual_seed(5)
        profs = [torch.rand(int(n), 6, generator=gen) * 2 - 1 for n in (48, 20, 48, 9, 30, 48, 2, 40)]
        batch = {k: v.to(DEV) for k, v in model.profile_encoder.tokenize(profs).items()}
        batch.update(image=torch.randn(8, 1, 64, 64, generator=gen).to(DEV),
                     image_shape=torch.randint(32, 400, (8, 2), generator=gen).to(DEV),
                     profile_len=torch.tensor([[p.shape[0]] for p in profs]).to(DEV), buckets=1)
        losses = []
        for _ in range(3):
            opt.zero_grad()
            loss = model.training_step(batch, 0)
            loss.backward()
            w = model.image_encoder.backbone.blocks[0].mlp.fc1.weight
            assert w.grad is not None and float(w.grad.abs().sum()) > 0
            opt.step()
            losses.append(float(loss))
        assert all(math.isfinite(v) for v in losses) and losses[0] != losses[-1], losses
    finally:
        TF._PRECISION[0] = old
        IE.create_backbone = orig


def test_mixed_training_tracks_exact_fp32_training():
    """Same initialisation, same batch, dropout off: five SGD steps through the exact-fp32 kernels and through the
    mixed-precision path give the same loss curve to bf16 accuracy."""
    from multimodal_plankton_recognition_amd import transformer as TF
    from multimodal_plankton_recognition_amd.model import MultiModel
    from multimodal_plankton_recognition_amd import image_encoder as IE
    orig = IE.create_backbone

    def create(name, in_chans=1):
        if name == 'vit_test_patch16_64':
            return IE.ViTBackbone(64, 2, 1, 16, 64, in_chans)
        return orig(name, in_chans)
    IE.create_backbone = create
    curves = {}
    try:
        for mode in ('32', 'bf16-mixed'):
            old = TF.set_precision(mode)
            try:
                torch.manual_seed(0)
                model = MultiModel(dim_embed=32, image_encoder_args=dict(name='vit_test_patch16_64', dropout=0.0),
                                   profile_encoder_args=dict(dim_in=6, dim_hidden=64, target_size=48, num_head=2,
                                                             num_layers=2, dim_feedforward=128, dropout=0.0),
                                   coordination_args=dict(method='clip'), optim_args=dict(lr=5e-2, momentum=0.9)).to(DEV).train()
                opt = model.configure_optimizers()
                gen = torch.Generator().manual_seed(5)
                profs = [torch.rand(int(n), 6, generator=gen) * 2 - 1 for n in (48, 20, 48, 9, 30, 48, 2, 40)]
                batch = {k: v.to(DEV) for k, v in model.profile_encoder.tokenize(profs).items()}
                batch.update(image=torch.randn(8, 1, 64, 64, generator=gen).to(DEV),
                             image_shape=torch.randint(32, 400, (8, 2), generator=gen).to(DEV),
                             profile_len=torch.tensor([[p.shape[0]] for p in profs]).to(DEV), buckets=1)
                losses = []
                for _ in range(5):
                    opt.zero_grad()
                    loss = model.training_step(batch, 0)
                    loss.backward()
                    opt.step()
                    losses.append(float(loss.detach()))
                curves[mode] = losses
            finally:
                TF._PRECISION[0] = old
    finally:
        IE.create_backbone = orig
    a, b = np.array(curves['32']), np.array(curves['bf16-mixed'])
    assert a[0] != a[-1]
    np.testing.assert_allclose(b, a, rtol=2e-2, err_msg=str(curves))


def test_patch_embedding_trains_after_a_no_grad_forward():
    """A no_grad forward (export / sanity validation) before training must not freeze the cached view of the patch
    embedding's filter: the weight receives a gradient in the next training pass, and the backbone stays deep-copyable."""
    import copy
    from multimodal_plankton_recognition_amd import transformer as TF
    from multimodal_plankton_recognition_amd.image_encoder import ViTBackbone
    torch.manual_seed(0)
    vit = ViTBackbone(embed_dim=64, depth=1, num_heads=1, patch=16, img_size=64, in_chans=1).to(DEV)
    x = torch.randn(4, 1, 64, 64, device=DEV)
    old = TF.set_precision('bf16-mixed')
    try:
        with torch.no_grad():
            vit.eval()
            vit.forward_pooled(x)
        twin = copy.deepcopy(vit)
        for m in (vit, twin):
            m.train()
            m.forward_pooled(x).sum().backward()
            g = m.patch_embed.proj.weight.grad
            assert g is not None and float(g.abs().sum()) > 0
    finally:
        TF._PRECISION[0] = old
