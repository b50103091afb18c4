"""Few-shot cross-modal benchmark on exported embeddings -- the GPU counterpart of the reference's
scripts/benchmark_cross.py:24-96 (same sampling, the same 8 gallery/query setups, the same result layout), with the
exact-search ANNClassifier of multimodal_plankton_recognition_amd/ann.py.

    python scripts/benchmark_fewshot.py embeddings.pkl --model <name> [--fold 0] [--shots 5 10] [--repeats 3] [--k 1 5 10]

`embeddings.pkl` has the reference's schema {model: {fold: {'image', 'profile', 'label', 'classes'}}} (written by
scripts/export_embeddings.py).  Prints mean accuracy per (shots, k, setup).
"""
import argparse
import pickle
import random
import sys
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from multimodal_plankton_recognition_amd.ann import ANNClassifier          # noqa: E402

ANN_KW = dict(n_neighbors=32, metric='euclidean', diversify_prob=0.0, pruning_degree_multiplier=3.0, low_memory=False,
              random_state=0)                                              # benchmark_cross.py:30-37


def sample(y, n):
    """n random members of every class (benchmark_cross.py:14-21)."""
    idx = []
    orig = np.arange(len(y))
    for label in np.unique(y):
        idx.extend(random.sample(list(orig[y == label]), n))
    return np.array(idx)


def threshold(images, profiles, labels, th):
    """classes with at least th members (benchmark_cross.py:99-109)."""
    uniq, counts = np.unique(labels, return_counts=True)
    keep = np.isin(labels, uniq[counts >= th])
    return images[keep], profiles[keep], labels[keep]


def benchmark(images, profiles, labels, n, repeats, K):
    acc = {}
    for run in range(repeats):
        tr = sample(labels, n)
        te = np.setdiff1d(np.arange(len(labels)), tr)
        it, pt, lt = images[tr], profiles[tr], labels[tr]
        iq, pq, lq = images[te], profiles[te], labels[te]
        setups = [(ANNClassifier(it, lt, **ANN_KW), ('I - I', 'I - P', 'I - I+P'), ((iq,), (pq,), (iq, pq))),
                  (ANNClassifier(pt, lt, **ANN_KW), ('P - I', 'P - P', 'P - I+P'), ((iq,), (pq,), (iq, pq))),
                  (ANNClassifier(np.concatenate((it, pt)), np.tile(lt, 2), **ANN_KW), ('I+P - I', 'I+P - P'), ((iq,), (pq,)))]
        for clf, keys, queries in setups:
            for k in K:
                for key, X in zip(keys, queries):
                    pred = clf.predict(*X, k=min(k, len(clf.y_)), epsilon=.3)
                    acc.setdefault((k, key), []).append(float((pred == lq).mean()))
    return {kk: float(np.mean(v)) for kk, v in acc.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('embeddings')
    ap.add_argument('--model', default=None)
    ap.add_argument('--fold', default=None)
    ap.add_argument('--shots', type=int, nargs='+', default=[5])
    ap.add_argument('--repeats', type=int, default=3)
    ap.add_argument('--k', type=int, nargs='+', default=[1, 5, 10])
    ap.add_argument('--seed', type=int, default=0)
    args = ap.parse_args()
    random.seed(args.seed)
    data = pickle.load(open(args.embeddings, 'rb'))
    model = args.model or next(iter(data))
    fold = args.fold if args.fold is not None else next(iter(data[model]))
    fold = fold if fold in data[model] else int(fold)
    d = data[model][fold]
    images, profiles = np.asarray(d['image'], np.float32), np.asarray(d['profile'], np.float32)
    names = np.asarray(d['label'])
    _, labels = np.unique(names, return_inverse=True)                      # LabelEncoder: sorted unique strings
    for n in args.shots:
        im, pr, lb = threshold(images, profiles, labels, n + 1)
        res = benchmark(im, pr, lb, n, args.repeats, args.k)
        for (k, key), v in sorted(res.items()):
            print(f'model {model} fold {fold} shots {n:3d} k {k:3d} {key:8s} accuracy {v:.4f}', flush=True)


if __name__ == '__main__':
    main()
