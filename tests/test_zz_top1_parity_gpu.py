"""Top-1 parity on a synthetic-class task (SURVEY 8d "parity reported with the numbers"; north star: "matching reference
top-1 within +-0.1 %").  Collected LAST on purpose (zz): a statistical statement must never hide deterministic tests.

The GPU path and the CPU oracle train the same MultiModel (ResNet-18 + ProfileCNN + CLIP, src/model.py:93-101) from the
same initialisation on the same batches (dropout 0), then both embed 1024 held-out pairs of 12 synthetic classes
(tests/top1_task.py); cross-modal retrieval top-1 must agree.

What makes this well-posed (round 3; measured, tests/tools/top1_oracle_spread.py, profiles/r03_top1_spread.md):
  * a short, high-learning-rate run is a CHAOTIC observable: the fp32 oracle ITSELF, started from weights that differ by
    <= 1 fp32 ulp, lands anywhere in 0.934 .. 0.956 after the 40 steps at lr 2e-2 the round-2 test used (per-step loss apart
    by 1e-3 after 6 steps, 7 % after 40) -- no tolerance on that number means anything, for any implementation;
  * trained to the plateau (100 steps, lr 5e-3, noise 0.4) the same perturbations give 0.999 .. 1.000: the plateau is the
    reproducible quantity, and the oracle's own spread there -- one sample in 1024, 0.1 % -- is the tolerance unit.
So: both paths train to the plateau; top-1 of the bf16 throughput path and of the fp32 parity path must be within 0.3 % of
the oracle's (3 x its own spread).  The fp32 parity path (no atomics: bit-reproducible) additionally follows the oracle's
per-step loss for the first steps, before chaos amplifies summation-order differences (1e-3 is reached at step ~12 by
1-ulp perturbations of the oracle itself), and two runs of it are bit-identical (nothing leaks from one run to the next)."""
import pytest
import torch

import top1_task as TT

pytestmark = pytest.mark.gpu
DEV = 'cuda'
STEPS, LR, BATCH, NOISE = 100, 5e-3, 48, 0.4
ORACLE_SPREAD = 1.0 / 1024          # measured: 0.9990 .. 1.0000 over 1-ulp perturbed starts


def _train_gpu(model, init_sd, batches, test, test_labels):
    model.load_state_dict(init_sd)
    model.to(DEV).train()
    opt = model.configure_optimizers()
    losses = []
    for b in batches:
        opt.zero_grad()
        loss = model.training_step({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}, 0)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    model.eval()
    with torch.no_grad():
        out = model.encode(**{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in test.items()})
    final = torch.cat([v.detach().float().flatten() for v in model.state_dict().values() if v.is_floating_point()]).cpu()
    return TT.top1(out['image_emb'], out['profile_emb'], test_labels), [float(v) for v in torch.stack(losses).cpu()], final


def test_top1_at_the_plateau_matches_cpu_oracle_training():
    from multimodal_plankton_recognition_amd import layers_f32
    from multimodal_plankton_recognition_amd.model import MultiModel
    from oracle import model as OM
    cfg = dict(TT.CFG, optim_args=dict(TT.CFG['optim_args'], lr=LR))
    init_sd = TT.init_state(cfg)
    batches, test, test_labels = TT.make_task(STEPS, BATCH, NOISE)

    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd = {k: v.clone() for k, v in init_sd.items()}
    bufs, ref_losses = {}, []
    for b in batches:
        ref_losses.append(float(OM.train_step(sd, b, cfg, bufs)[0]))
    with torch.no_grad():
        emb = OM.encode(sd, test, cfg, train=False)
    acc_cpu = TT.top1(emb['image_emb'], emb['profile_emb'], test_labels)
    assert acc_cpu >= 0.99, f'the oracle did not reach the plateau ({acc_cpu})'

    torch.manual_seed(0)
    model = MultiModel(dim_embed=TT.DIM_EMBED, **cfg)
    acc_bf16, loss_bf16, _ = _train_gpu(model, init_sd, batches, test, test_labels)
    old = layers_f32.set_conv_precision('32')
    try:
        acc_f32, loss_f32, fin_a = _train_gpu(model, init_sd, batches, test, test_labels)
        _, loss_f32_b, fin_b = _train_gpu(model, init_sd, batches[:10], test, test_labels)
        _, loss_f32_c, fin_c = _train_gpu(model, init_sd, batches[:10], test, test_labels)
    finally:
        layers_f32._PRECISION[0] = old
    dev = [abs(a - b) / abs(b) for a, b in zip(loss_f32, ref_losses)]
    print(f'retrieval top-1 at the plateau: CPU oracle {acc_cpu:.4f}, bf16 path {acc_bf16:.4f}, fp32 path {acc_f32:.4f}; '
          f'fp32-path loss vs oracle: step 0 {dev[0]:.1e}, step 4 {dev[4]:.1e}, step 7 {dev[7]:.1e}, last {dev[-1]:.1e}; '
          f'bf16-path loss vs oracle: step 0 {abs(loss_bf16[0] - ref_losses[0]) / ref_losses[0]:.1e}, '
          f'last {abs(loss_bf16[-1] - ref_losses[-1]) / ref_losses[-1]:.1e}')
    tol = 3 * ORACLE_SPREAD + 1e-9
    assert acc_bf16 >= 0.99 and abs(acc_bf16 - acc_cpu) <= tol, (acc_cpu, acc_bf16)
    assert acc_f32 >= 0.99 and abs(acc_f32 - acc_cpu) <= tol, (acc_cpu, acc_f32)
    assert max(dev[:5]) < 2e-4 and max(dev[:8]) < 1e-3, dev[:8]
    assert loss_f32_b == loss_f32_c == loss_f32[:10] and torch.equal(fin_b, fin_c), 'fp32 path: runs are not bit-identical'
    # both paths learn the task the way the oracle does (the trajectories themselves are chaotic: a few per cent apart)
    assert abs(loss_bf16[-1] - ref_losses[-1]) < 0.05 * ref_losses[-1] and abs(loss_f32[-1] - ref_losses[-1]) < 0.05 * ref_losses[-1]
