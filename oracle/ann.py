"""Oracle (test infrastructure): CPU restatement of the few-shot classifier /root/reference/src/ann.py:6-34.

The reference's neighbour search is pynndescent.NNDescent (third party, unpinned, not installed in this image): an
approximate index the author configured to behave like exact search.  The restatement searches exactly (numpy, fp32,
distances evaluated as sqrt(sum (x - g)^2) / 1 - cos, ties to the smaller index); weights and vote follow the reference
line by line and use the same sklearn ``weighted_mode``.  PARITY UNPINNED for the search (no reference fixture exists and
the dependency is absent); the vote is pinned by sklearn's own function.
"""
import numpy as np
from sklearn.utils.extmath import weighted_mode


def kneighbors(G, X, k, metric='euclidean'):
    G = np.asarray(G, np.float32)
    X = np.asarray(X, np.float32)
    idx = np.empty((X.shape[0], k), np.int64)
    dist = np.empty((X.shape[0], k), np.float32)
    for i, x in enumerate(X):
        if metric == 'euclidean':
            d = np.sqrt(((G - x) ** 2).sum(1, dtype=np.float32))
        else:
            d = 1 - (G @ x) / np.sqrt((G * G).sum(1) * (x * x).sum())
            d = np.maximum(d, 0).astype(np.float32)
        order = np.lexsort((np.arange(len(d)), d))[:k]
        idx[i], dist[i] = order, d[order]
    return idx, dist


def get_weights(dist):
    """src/ann.py:28-34."""
    with np.errstate(divide='ignore'):
        dist = 1.0 / dist
    inf_mask = np.isinf(dist)
    inf_row = np.any(inf_mask, axis=1)
    dist[inf_row] = inf_mask[inf_row]
    return dist


def predict(G, y, Xs, k, metric='euclidean'):
    """src/ann.py:19-25 for a tuple of query modalities Xs."""
    parts = [kneighbors(G, X, k, metric) for X in Xs]
    idx = np.hstack([p[0] for p in parts])
    dist = np.hstack([p[1] for p in parts])
    predictions, _ = weighted_mode(np.asarray(y)[idx], get_weights(dist), axis=1)
    return predictions.astype(int).ravel()
