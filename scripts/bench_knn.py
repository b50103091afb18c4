"""Few-shot evaluation throughput: exact kNN + weighted vote (ann.ANNClassifier) on synthetic embeddings of the size the
reference's benchmark handles (50 classes, n shots per class in the gallery, the rest of ~100 K samples as queries)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from multimodal_plankton_recognition_amd.ann import ANNClassifier
rng = np.random.default_rng(0)
D, n_cls, nq = 512, 50, 100_000
centers = rng.normal(size=(n_cls, D)).astype(np.float32)
for shots in (5, 32):
    y = np.repeat(np.arange(n_cls), shots)
    G = (centers[y] + 2.0 * rng.normal(size=(len(y), D))).astype(np.float32)
    yq = rng.integers(0, n_cls, nq)
    Xi = torch.from_numpy((centers[yq] + 2.0 * rng.normal(size=(nq, D))).astype(np.float32)).cuda()
    Xp = torch.from_numpy((centers[yq] + 2.5 * rng.normal(size=(nq, D))).astype(np.float32)).cuda()
    clf = ANNClassifier(G, y, metric='euclidean')
    for k in (1, 10):
        clf.predict(Xi[:1000], k=min(k, len(y)))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pred = clf.predict(Xi, Xp, k=min(k, len(y)))
        dt = time.perf_counter() - t0
        print(f'gallery {len(y):5d} ({shots} shots), {nq} queries x 2 modalities, k={k:2d}: {dt * 1e3:7.1f} ms '
              f'({nq / dt / 1e6:.2f} M queries/s), accuracy {float((pred == yq).mean()):.4f}', flush=True)
