"""Embedding export / inference path: `MultiModel.predict_step` + L2-normalisation over whole folds,
written in the pickle schema the reference's benchmark scripts read (scripts/benchmark_raw.py:71-104, produced by
experiments.ipynb cells 4 and 6):

    {model_name: {fold_name: {'image': [N, D] float32, 'profile': [N, D] float32, 'label': [N] str, 'classes': [...]}}}

    cd scripts && python3 export_embeddings.py -m ../model_cards/<card>.yaml -c <checkpoint.ckpt> -d <fold dir> [-d ...] -o out.pkl

Forward only: the same gfx950 kernels as training (eval-mode BatchNorm, no dropout), the normalisation on mpr_l2norm_fwd.
`--synthetic N` replaces the folds by N synthetic pairs each (no dataset on disk); without -c the weights are random.
"""
import argparse
import pickle
import sys
from pathlib import Path

import numpy as np
import torch
import yaml
from torch.utils.data import DataLoader

sys.path.append(str(Path(__file__).resolve().parent.parent))
from multimodal_plankton_recognition_amd import _native as N  # noqa: E402
from multimodal_plankton_recognition_amd.data import (ImageTransformTest, MultiSet, ProfileTransformTest,  # noqa: E402
                                                      SyntheticMultiSet)
from multimodal_plankton_recognition_amd.model import MultiModel  # noqa: E402


def l2_normalize(x):
    """F.normalize(x, dim=-1) (src/coordination.py:26-27) on the native kernel."""
    x = x.detach().contiguous().float()
    u = torch.empty_like(x)
    inv = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    N.call('mpr_l2norm_fwd', x, u, inv, x.shape[0], x.shape[1])
    return u


def make_collate(model):
    def collate(batch):
        image, profile, label, image_shape, profile_len = zip(*(sample.values() for sample in batch))
        out = {'image': torch.stack(image)}
        out.update(model.profile_encoder.tokenize(profile))
        out['image_shape'] = torch.stack(image_shape)
        out['profile_len'] = torch.stack(profile_len)
        return out, list(label)
    return collate


@torch.no_grad()
def export_fold(model, dataset, bs, device, num_workers=0):
    loader = DataLoader(dataset, batch_size=bs, shuffle=False, num_workers=num_workers, collate_fn=make_collate(model))
    img, prof, labels = [], [], []
    for i, (batch, label) in enumerate(loader):
        batch = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
        out = model.predict_step(batch, i)
        img.append(l2_normalize(out['image_emb']).cpu().numpy())
        prof.append(l2_normalize(out['profile_emb']).cpu().numpy())
        labels += label
    return {'image': np.concatenate(img), 'profile': np.concatenate(prof), 'label': np.asarray(labels),
            'classes': np.asarray(dataset.class_names)}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('-m', '--modelcard', required=True)
    ap.add_argument('-c', '--checkpoint', default=None, help='Lightning-style .ckpt written by train_multi.py')
    ap.add_argument('-d', '--dataset', action='append', default=[], help='fold directory with test.csv (repeatable)')
    ap.add_argument('-o', '--output', required=True)
    ap.add_argument('--synthetic', type=int, default=0, help='[new] N synthetic pairs per fold instead of datasets')
    ap.add_argument('--batch', type=int, default=None)
    args = ap.parse_args(argv)
    card = yaml.safe_load(open(args.modelcard))
    device = torch.device('cuda', 0)
    model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                       card['coordination_args'], card['optim_args'])
    if args.checkpoint:
        model.load_state_dict(torch.load(args.checkpoint, map_location='cpu')['state_dict'])
    model.to(device).eval()
    T, bs = card['target_size'], args.batch or card['bs']
    folds = {}
    if args.synthetic:
        for name, seed in (('fold_0', 11), ('fold_1', 12)):
            folds[name] = SyntheticMultiSet(args.synthetic, T, seed=seed)
    for d in args.dataset:
        d = Path(d)
        folds['_'.join(d.parts[-2:])] = MultiSet(d / 'test.csv', ImageTransformTest(T), ProfileTransformTest(T))
    name = Path(args.modelcard).name.split('.')[0]
    result = {name: {fold: export_fold(model, ds, bs, device, card.get('num_workers', 0)) for fold, ds in folds.items()}}
    with open(args.output, 'wb') as f:
        pickle.dump(result, f)
    for fold, r in result[name].items():
        print(f'{name}/{fold}: image {r["image"].shape} profile {r["profile"].shape} labels {len(r["label"])} '
              f'classes {len(r["classes"])}')
    return result


if __name__ == '__main__':
    main()
