"""Oracle (test infrastructure): MultiModel wiring + one optimisation step, fp32 CPU.

Restates /root/reference/src/model.py:19-148 (encode -> bias-free projections ->
coordination loss; SGD over *all* parameters incl. the loss's) on a flat
state_dict whose keys are MultiModel's own (``image_encoder.backbone.*``,
``image_projection.weight``, ``profile_encoder.*``, ``profile_projection.weight``,
``loss.*``).  ``cfg`` mirrors the YAML card (scripts/train_multi.py:58-64).
"""
import torch
import torch.nn.functional as F
from . import coordination as C
from .profile_encoder import profile_cnn_forward, profile_transformer_forward, profile_lstm_forward
from .image_encoder import image_encoder_forward

_BUFFER_SUFFIXES = ('running_mean', 'running_var', 'num_batches_tracked')


def is_param(key):
    return not key.endswith(_BUFFER_SUFFIXES)


def sub(sd, prefix):
    return _View(sd, prefix)


class _View(dict):
    """A prefixed window onto a parent state_dict (shares tensors, so in-place
    BatchNorm buffer updates land in the parent)."""
    def __init__(self, parent, prefix):
        super().__init__({k[len(prefix):]: v for k, v in parent.items() if k.startswith(prefix)})


def encode(sd, batch, cfg, train=False, apply_dropout=False):
    """MultiModel.encode, model.py:72-85.  apply_dropout: also run the encoders' output dropout (image_encoder.py:29,
    profile_encoder.py:240) in train mode -- off by default because parity runs use p = 0 (RNG streams cannot match);
    the timed CPU baseline switches it on so that it does the work of the benchmarked step."""
    ie, pe = cfg['image_encoder_args'], cfg['profile_encoder_args']
    img_feat = image_encoder_forward(sub(sd, 'image_encoder.'), batch['image'], batch['image_shape'],
                                     arch=ie['name'], train=train, metadata=ie.get('metadata', True),
                                     **cfg.get('image_arch_kw', {}))
    p = sub(sd, 'profile_encoder.')
    if 'num_head' in pe:                                   # model.py:34-39
        prof_feat = profile_transformer_forward(p, batch['profile'], batch['time'], batch['padding_mask'],
                                                batch['profile_len'], pe['num_head'], pe.get('num_layers', 6),
                                                pe.get('activation', 'gelu'), pe.get('metadata', True))
    elif 'blocks' in pe:
        prof_feat = profile_cnn_forward(p, batch['profile'], batch['profile_len'], pe['blocks'], train,
                                        pe.get('metadata', True))
    else:
        prof_feat = profile_lstm_forward(p, batch['profile'], batch['last_idx'], batch['profile_len'],
                                         pe['num_layers'], pe.get('metadata', True))
    if apply_dropout and train:
        img_feat = F.dropout(img_feat, ie.get('dropout', 0.1), True)
        prof_feat = F.dropout(prof_feat, pe.get('dropout', 0.1), True)
    return {'image_emb': F.linear(img_feat, sd['image_projection.weight']),       # model.py:80
            'profile_emb': F.linear(prof_feat, sd['profile_projection.weight'])}  # model.py:82


def coordination(sd, emb, cfg, buckets):
    """Loss dispatch of model.py:44-56 applied as in training_step model.py:95-98."""
    ca = cfg['coordination_args']
    m = ca.get('method')
    a, b = emb['image_emb'], emb['profile_emb']
    if m == 'clip':
        return C.clip_loss(a, b, sd['loss.logit_scale'], buckets)
    if m == 'siglip':
        return C.siglip_loss(a, b, sd['loss.logit_scale'], sd['loss.bias'], buckets)
    if m == 'clipplus':
        return C.clip_plus(a, b, sd['loss.clip.logit_scale'], buckets, ca.get('beta', .25))
    if m == 'siglipplus':
        return C.siglip_plus(a, b, sd['loss.siglip.logit_scale'], sd['loss.siglip.bias'], buckets, ca.get('beta', .25))
    if m == 'rank':
        return C.rank_loss(a, b, ca.get('margin', .25))
    raise Exception("Coordination loss not found.")


def sgd_update(params, grads, bufs, lr, momentum=0.0, weight_decay=0.0, nesterov=False, dampening=0.0):
    """torch.optim.SGD single step (model.py:147-148); ``bufs`` maps key -> momentum buffer."""
    with torch.no_grad():
        for k, p in params.items():
            g = grads[k]
            if g is None:
                continue
            if weight_decay:
                g = g + weight_decay * p
            if momentum:
                if k not in bufs:
                    bufs[k] = g.clone()
                else:
                    bufs[k].mul_(momentum).add_(g, alpha=1 - dampening)
                g = g + momentum * bufs[k] if nesterov else bufs[k]
            p.add_(g, alpha=-lr)


def train_step(sd, batch, cfg, bufs, apply_dropout=False):
    """training_step (model.py:93-101) + backward + SGD step.  Mutates ``sd``/``bufs``; returns
    (loss, grads)."""
    params = {k: v for k, v in sd.items() if is_param(k) and v.is_floating_point()}
    for v in params.values():
        v.requires_grad_(True)
    emb = encode(sd, batch, cfg, train=True, apply_dropout=apply_dropout)
    loss = coordination(sd, emb, cfg, batch.get('buckets', 1))
    grads = dict(zip(params, torch.autograd.grad(loss, list(params.values()), allow_unused=True)))
    for v in params.values():
        v.requires_grad_(False)
    sgd_update(params, grads, bufs, **cfg['optim_args'])
    return loss.detach(), grads
