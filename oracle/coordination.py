"""Oracle (test infrastructure): cross-modal coordination losses, fp32 CPU.

Functional restatement of /root/reference/src/coordination.py.  Each function
takes the loss parameters explicitly (``logit_scale``/``bias`` are 0-d tensors)
so that gradients w.r.t. them can be read back by the parity tests.
"""
import torch
import torch.nn.functional as F


def _unit_rows(x):
    # F.normalize default: x / max(||x||_2, 1e-12) over dim 1  (coordination.py:33-34, 82-83)
    return x / x.norm(dim=1, keepdim=True).clamp_min(1e-12)


def _bucketed_logits(image_emb, profile_emb, logit_scale, buckets):
    # coordination.py:29-38 / 79-89: contiguous sub-batches, per-bucket all-pairs cosine * exp(scale)
    n_total, dim = image_emb.shape
    assert n_total % buckets == 0, "Batch size must be divisible by number of buckets!"
    n = n_total // buckets
    u = _unit_rows(image_emb).reshape(buckets, n, dim)
    v = _unit_rows(profile_emb).reshape(buckets, n, dim)
    return torch.bmm(u, v.transpose(1, 2)) * logit_scale.exp(), n


def clip_loss(image_emb, profile_emb, logit_scale, buckets=1):
    """CLIPLoss.forward, coordination.py:26-47."""
    logits, n = _bucketed_logits(image_emb, profile_emb, logit_scale, buckets)
    idx = torch.arange(n)
    lse_rows = torch.logsumexp(logits, dim=2)          # image -> profile
    lse_cols = torch.logsumexp(logits, dim=1)          # profile -> image
    diag = logits[:, idx, idx]
    row_ce = (lse_rows - diag).mean(dim=1).mean()      # mean over n, then over buckets (:43)
    col_ce = (lse_cols - diag).mean(dim=1).mean()      # (:44)
    return (row_ce + col_ce) / 2                       # (:45)


def siglip_loss(image_emb, profile_emb, logit_scale, bias, buckets=1):
    """SigLIPLoss.forward, coordination.py:76-95: +z on the diagonal, -z elsewhere."""
    logits, n = _bucketed_logits(image_emb, profile_emb, logit_scale, buckets)
    z = logits + bias
    sign = 2 * torch.eye(n).unsqueeze(0) - 1
    per_bucket = -F.logsigmoid(sign * z).sum(dim=(1, 2)) / n   # (:93)
    return per_bucket.mean()                                    # (:95)


def mse(image_emb, profile_emb):
    """nn.MSELoss on the un-normalised embeddings, coordination.py:55,62 / 103,110."""
    return (image_emb - profile_emb).pow(2).mean()


def clip_plus(image_emb, profile_emb, logit_scale, buckets=1, beta=0.25):
    """CLIPPlus.forward, coordination.py:60-64."""
    return clip_loss(image_emb, profile_emb, logit_scale, buckets) + beta * mse(image_emb, profile_emb)


def siglip_plus(image_emb, profile_emb, logit_scale, bias, buckets=1, beta=0.25):
    """SigLIPPlus.forward, coordination.py:108-112."""
    return siglip_loss(image_emb, profile_emb, logit_scale, bias, buckets) + beta * mse(image_emb, profile_emb)


def rank_loss(image_emb, profile_emb, margin):
    """RankLoss.forward, coordination.py:123-135 (plain division, no eps; no buckets)."""
    u = image_emb / image_emb.norm(dim=1, keepdim=True)
    v = profile_emb / profile_emb.norm(dim=1, keepdim=True)
    s = u @ v.T
    n = s.shape[0]
    s = s * (1 - 2 * torch.eye(n))                      # diagonal negated (:129)
    l0 = F.relu(margin + s.sum(0)).mean()
    l1 = F.relu(margin + s.sum(1)).mean()
    return (l0 + l1) / 2


def retrieval_top1(image_emb, profile_emb):
    """argmax_j cos(u_i, v_j) -- the int64 'class indices' that must be bit-exact
    on margin-controlled fixtures (SURVEY 8d 'Parity reported with the numbers')."""
    s = _unit_rows(image_emb) @ _unit_rows(profile_emb).T
    return s.argmax(dim=1), s.argmax(dim=0)
