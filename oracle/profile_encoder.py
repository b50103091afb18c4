"""Oracle (test infrastructure): 1-D profile encoders, fp32 CPU, functional.

Restates /root/reference/src/profile_encoder.py on plain state_dicts (the keys
are the reference module's own ``state_dict()`` keys).  ``train=True`` means
BatchNorm uses batch statistics and updates the running buffers in ``sd`` in
place (torch defaults: momentum 0.1, eps 1e-5, unbiased running_var), exactly
as ``nn.BatchNorm1d`` does at profile_encoder.py:126,129,168.  Dropout is not
applied (parity runs use p=0 / eval, SURVEY section 7 'Dropout parity').
"""
import math
import torch
import torch.nn.functional as F
from .rounding import r


def _bn(sd, name, x, train):
    rm, rv = sd[name + '.running_mean'], sd[name + '.running_var']
    y = F.batch_norm(x, rm, rv, sd[name + '.weight'], sd[name + '.bias'],
                     training=train, momentum=0.1, eps=1e-5)
    if train and (name + '.num_batches_tracked') in sd:
        sd[name + '.num_batches_tracked'] += 1
    return y


def _basic_block_1d(sd, p, x, stride, train):
    # _BasicBlock.forward, profile_encoder.py:132-148
    out = r(F.conv1d(x, r(sd[p + 'conv1.weight']), None, stride, 1))
    out = r(F.relu(_bn(sd, p + 'bn1', out, train)))
    out = r(F.conv1d(out, r(sd[p + 'conv2.weight']), None, 1, 1))
    out = _bn(sd, p + 'bn2', out, train)
    if (p + 'downsample.0.weight') in sd:
        identity = r(F.conv1d(x, r(sd[p + 'downsample.0.weight']), None, stride, 0))
        identity = r(_bn(sd, p + 'downsample.1', identity, train))
    else:
        identity = x
    return r(F.relu(out + identity))


def profile_cnn_features(sd, profile, blocks, train=False, prefix=''):
    """ProfileCNN.forward_features, profile_encoder.py:213-226.  profile: [B, L, dim_in]."""
    x = profile.transpose(1, 2)
    x = r(F.conv1d(x, sd[prefix + 'conv1.weight'], None, 2, 1))     # :167 (stem: fp32 operands)
    x = r(F.relu(_bn(sd, prefix + 'bn1', x, train)))
    x = F.max_pool1d(x, 3, 2, 1)                                     # :170
    for li, reps in enumerate(blocks, start=1):
        for bi in range(reps):
            stride = 2 if (li > 1 and bi == 0) else 1               # :172-175, :197
            x = _basic_block_1d(sd, f'{prefix}layer{li}.{bi}.', x, stride, train)
    return x


def profile_cnn_forward(sd, profile, profile_len, blocks, train=False, metadata=True, prefix=''):
    """ProfileCNN.forward, profile_encoder.py:229-240 (AdaptiveMaxPool1d named 'avgpool' :177)."""
    x = profile_cnn_features(sd, profile, blocks, train, prefix)
    x = F.adaptive_max_pool1d(x, 1).flatten(1)      # :177,232-233 (single arg-max gets the gradient)
    if metadata:
        meta = profile_len.to(profile.dtype) / profile.shape[1]     # :236-237
        x = torch.cat((x, meta), 1)
    return x


def transformer_tokenize(profiles, padding_idx):
    """ProfileTransformer.tokenize, profile_encoder.py:38-54: time = 0..len, padded with
    padding_idx; zero-padded profiles with one zero CLS row prepended; mask = padded slots."""
    if not isinstance(profiles, (list, tuple)):
        profiles = [profiles]
    n = max(p.shape[0] for p in profiles) + 1
    time = torch.full((len(profiles), n), padding_idx, dtype=torch.long)
    prof = torch.zeros(len(profiles), n, profiles[0].shape[1], dtype=profiles[0].dtype)
    for i, p in enumerate(profiles):
        time[i, :p.shape[0] + 1] = torch.arange(p.shape[0] + 1)
        prof[i, 1:p.shape[0] + 1] = p
    return {'profile': prof, 'time': time, 'padding_mask': time == padding_idx}


def _encoder_layer(sd, p, x, mask, nhead, activation):
    # nn.TransformerEncoderLayer defaults (post-norm, LN eps 1e-5, batch_first) -- profile_encoder.py:23-28
    b, t, d = x.shape
    hd = d // nhead
    qkv = F.linear(x, sd[p + 'self_attn.in_proj_weight'], sd[p + 'self_attn.in_proj_bias'])
    q, k, v = (z.reshape(b, t, nhead, hd).transpose(1, 2) for z in qkv.chunk(3, dim=-1))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(hd)
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :], float('-inf'))
    a = (s.softmax(dim=-1) @ v).transpose(1, 2).reshape(b, t, d)
    a = F.linear(a, sd[p + 'self_attn.out_proj.weight'], sd[p + 'self_attn.out_proj.bias'])
    x = F.layer_norm(x + a, (d,), sd[p + 'norm1.weight'], sd[p + 'norm1.bias'], 1e-5)
    act = F.gelu if activation == 'gelu' else F.relu
    h = F.linear(act(F.linear(x, sd[p + 'linear1.weight'], sd[p + 'linear1.bias'])),
                 sd[p + 'linear2.weight'], sd[p + 'linear2.bias'])
    return F.layer_norm(x + h, (d,), sd[p + 'norm2.weight'], sd[p + 'norm2.bias'], 1e-5)


def profile_transformer_forward(sd, profile, time, padding_mask, profile_len, num_head,
                                num_layers, activation='gelu', metadata=True, prefix=''):
    """ProfileTransformer.forward, profile_encoder.py:57-68.  The embedding row at padding_idx is
    whatever ``sd`` holds (nn.Embedding zeroes it at init, profile_encoder.py:19)."""
    x = F.linear(profile, sd[prefix + 'expand.weight']) + sd[prefix + 'position.weight'][time]
    for i in range(num_layers):
        x = _encoder_layer(sd, f'{prefix}encoder.layers.{i}.', x, padding_mask, num_head, activation)
    x = x[:, 0]
    if metadata:
        x = torch.cat((x, profile_len.to(profile.dtype) / profile.shape[1]), 1)   # :64-66 (denominator incl. CLS)
    return x


def profile_lstm_forward(sd, profile, last_idx, profile_len, num_layers, metadata=True, prefix=''):
    """ProfileLSTM.forward, profile_encoder.py:98-108 (nn.LSTM gate order i, f, g, o)."""
    x = F.linear(profile, sd[prefix + 'expand.weight'])
    b, t, d = x.shape
    for l in range(num_layers):
        wi, wh = sd[f'{prefix}lstm.weight_ih_l{l}'], sd[f'{prefix}lstm.weight_hh_l{l}']
        bi, bh = sd[f'{prefix}lstm.bias_ih_l{l}'], sd[f'{prefix}lstm.bias_hh_l{l}']
        h = x.new_zeros(b, d)
        c = x.new_zeros(b, d)
        outs = []
        for s in range(t):
            g = F.linear(x[:, s], wi, bi) + F.linear(h, wh, bh)
            i_, f_, g_, o_ = g.chunk(4, dim=1)
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
            h = torch.sigmoid(o_) * torch.tanh(c)
            outs.append(h)
        x = torch.stack(outs, 1)
    x = x[torch.arange(b), last_idx]
    if metadata:
        x = torch.cat((x, profile_len.to(profile.dtype) / profile.shape[1]), 1)
    return x
