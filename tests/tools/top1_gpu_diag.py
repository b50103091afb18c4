"""(test infrastructure: runs the oracle next to the HIP path)
Diagnosis of the round-2 top-1 failure, ONE GPU invocation (VERDICT r02 item 1b / ADVICE r02 high):

  * the oracle's trajectory of the task (per-step loss, final top-1);
  * `reps` runs of the default bf16 path from one initialisation on identical batches: per-step loss of every rep, the first
    step at which two reps differ at all / by more than 1e-3, final top-1 of each;
  * the same with the fp32 parity mode (`precision: 32`): no atomics on that path, so the reps must be BIT-identical -- if
    they are not, process-global state leaks from one rep into the next (slice arena, panel registry, scratch loans);
  * optionally the bf16 path with the reproducible BatchNorm sums (MPR_BWD_ATOMIC_SLICES=0 in the environment +
    mpr_conv_set_stat_slices(0)), to see how much of the rep-to-rep difference those atomics explain.

  python tests/tools/top1_gpu_diag.py --steps 40 --lr 2e-2 --noise 0.6 --reps 5 --out gpurun_out/top1_diag_old.json
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def run_gpu(model, init_sd, batches, test, test_labels, top1):
    dev = 'cuda'
    model.load_state_dict(init_sd)
    model.to(dev).train()
    opt = model.configure_optimizers()
    losses = []
    for b in batches:
        opt.zero_grad()
        loss = model.training_step({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}, 0)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    losses = [float(v) for v in torch.stack(losses).cpu()]
    model.eval()
    with torch.no_grad():
        out = model.encode(**{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in test.items()})
    acc = top1(out['image_emb'], out['profile_emb'], test_labels)
    final = torch.cat([v.detach().float().flatten().cpu() for v in model.state_dict().values() if v.is_floating_point()])
    return losses, acc, final


def summarize(tag, runs, ref_losses):
    L = torch.tensor([r[0] for r in runs], dtype=torch.float64)
    accs = [r[1] for r in runs]
    any_diff = next((i for i in range(L.shape[1]) if not bool((L[:, i] == L[0, i]).all())), None)
    dev = (L - L[0]).abs().max(0).values / L[0].abs()
    big = next((i for i, d in enumerate(dev.tolist()) if d > 1e-3), None)
    ref = torch.tensor(ref_losses, dtype=torch.float64)
    vs = ((L - ref).abs() / ref.abs()).max(0).values
    vs_first = next((i for i, d in enumerate(vs.tolist()) if d > 1e-3), None)
    bit = all(torch.equal(runs[0][2], r[2]) for r in runs[1:])
    print(f'[{tag}] top-1 {["%.4f" % a for a in accs]}; reps bit-identical after training: {bit}; first step with any loss '
          f'difference between reps: {any_diff}; first > 1e-3 between reps: {big}; first > 1e-3 against the oracle: {vs_first} '
          f'(step 0: {float(vs[0]):.2e})', flush=True)
    return dict(tag=tag, top1=accs, bit_identical=bit, first_any=any_diff, first_1e3=big, first_vs_oracle=vs_first,
                losses=[r[0] for r in runs])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--lr', type=float, default=2e-2)
    ap.add_argument('--batch', type=int, default=48)
    ap.add_argument('--noise', type=float, default=0.6)
    ap.add_argument('--reps', type=int, default=5)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    from top1_task import CFG, DIM_EMBED, make_task, top1, init_state
    from oracle import model as OM
    from multimodal_plankton_recognition_amd import _native as N
    from multimodal_plankton_recognition_amd import layers_f32
    from multimodal_plankton_recognition_amd.model import MultiModel
    cfg = dict(CFG, optim_args=dict(CFG['optim_args'], lr=a.lr))
    init_sd = init_state(cfg)
    batches, test, test_labels = make_task(a.steps, a.batch, a.noise)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    sd = {k: v.clone() for k, v in init_sd.items()}
    bufs, ref_losses = {}, []
    for b in batches:
        ref_losses.append(float(OM.train_step(sd, b, cfg, bufs)[0]))
    with torch.no_grad():
        emb = OM.encode(sd, test, cfg, train=False)
    acc_ref = top1(emb['image_emb'], emb['profile_emb'], test_labels)
    print(f'[oracle] top-1 {acc_ref:.4f}, loss {ref_losses[0]:.5f} -> {ref_losses[-1]:.5f}', flush=True)
    torch.manual_seed(0)
    model = MultiModel(dim_embed=DIM_EMBED, **cfg)
    res = dict(args=vars(a), oracle=dict(top1=acc_ref, losses=ref_losses), modes=[])
    runs = [run_gpu(model, init_sd, batches, test, test_labels, top1) for _ in range(a.reps)]
    res['modes'].append(summarize('bf16 default', runs, ref_losses))
    old = N.query('mpr_conv_set_stat_slices', 0)
    runs = [run_gpu(model, init_sd, batches, test, test_labels, top1) for _ in range(a.reps)]
    res['modes'].append(summarize('bf16, conv stat slices off' + (' + MPR_BWD_ATOMIC_SLICES=0' if os.environ.get(
        'MPR_BWD_ATOMIC_SLICES') == '0' else ''), runs, ref_losses))
    N.query('mpr_conv_set_stat_slices', old)
    layers_f32.set_conv_precision('32')
    runs = [run_gpu(model, init_sd, batches, test, test_labels, top1) for _ in range(min(a.reps, 3))]
    res['modes'].append(summarize('fp32 parity mode', runs, ref_losses))
    if a.out:
        os.makedirs(os.path.dirname(a.out) or '.', exist_ok=True)
        json.dump(res, open(a.out, 'w'))


if __name__ == '__main__':
    main()
