"""(test infrastructure: lives under tests/ because it runs the oracle)
How well-posed is "top-1 after a short training run"?  

Runs the CPU oracle's trajectory of the top-1 task several times from initial weights that differ by <= 1 fp32 ulp
(and, optionally, with the bf16-storage emulation of oracle/rounding.py), and prints the spread of the retrieval
top-1 and of the per-step loss.  If the oracle's own spread is several points, the end-of-training top-1 of a short,
high-learning-rate, train-mode-BatchNorm run is a chaotic observable and no tolerance on it means anything.

  python tests/tools/top1_oracle_spread.py [--steps 40] [--lr 2e-2] [--batch 48] [--runs 6] [--bf16] [--noise 0.6]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--lr', type=float, default=2e-2)
    ap.add_argument('--batch', type=int, default=48)
    ap.add_argument('--runs', type=int, default=6)
    ap.add_argument('--noise', type=float, default=0.6)
    ap.add_argument('--test', type=int, default=1024)
    ap.add_argument('--bf16', action='store_true')
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    from top1_task import CFG, make_task, top1, init_state
    from oracle import model as OM
    from oracle.rounding import emulate_bf16
    cfg = dict(CFG, optim_args=dict(CFG['optim_args'], lr=a.lr))
    init_sd = init_state(cfg)
    batches, test, test_labels = make_task(a.steps, a.batch, a.noise, a.test)
    torch.set_num_threads(8)
    res = []
    for run in range(a.runs):
        sd = {k: v.clone() for k, v in init_sd.items()}
        if run:                                     # run 0 = unperturbed
            g = torch.Generator().manual_seed(100 + run)
            for k, v in sd.items():
                if v.is_floating_point() and not k.endswith(('running_mean', 'running_var')):
                    sign = (torch.randint(0, 3, v.shape, generator=g) - 1).to(v.dtype)       # -1, 0, +1 ulp
                    v.copy_(torch.where(sign > 0, torch.nextafter(v, v + 1), torch.where(sign < 0, torch.nextafter(v, v - 1), v)))
        t0 = time.time()
        bufs, losses = {}, []
        with emulate_bf16(a.bf16):
            for b in batches:
                loss, _ = OM.train_step(sd, b, cfg, bufs)
                losses.append(float(loss))
            with torch.no_grad():
                emb = OM.encode(sd, test, cfg, train=False)
        acc = top1(emb['image_emb'], emb['profile_emb'], test_labels)
        res.append(dict(run=run, top1=acc, losses=losses))
        print(f'run {run}: top-1 {acc:.4f}  loss[0] {losses[0]:.5f} loss[-1] {losses[-1]:.5f}  ({time.time() - t0:.0f} s)', flush=True)
    accs = [r['top1'] for r in res]
    L = torch.tensor([r['losses'] for r in res])
    dev = (L - L[0]).abs().max(0).values / L[0].abs()
    first = next((i for i, d in enumerate(dev.tolist()) if d > 1e-3), None)
    print(f'top-1 spread over {a.runs} runs: min {min(accs):.4f} max {max(accs):.4f}; '
          f'first step with a relative loss deviation > 1e-3: {first}; max dev at last step {float(dev[-1]):.3e}')
    if a.out:
        json.dump(dict(args=vars(a), runs=res), open(a.out, 'w'))


if __name__ == '__main__':
    main()
