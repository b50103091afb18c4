"""1-D profile encoders -- drop-in counterparts of /root/reference/src/profile_encoder.py
(same class names, constructor arguments, ``tokenize`` outputs, ``forward`` keyword contract and
``state_dict`` keys), running on the hand-written gfx950 kernels.

ProfileCNN (src/profile_encoder.py:151-240) is fully native: sequences stay channels-last
``[B, L, C]`` bf16 (the reference's ``transpose(1, 2)`` disappears -- the input already is
channels-last), every conv / BN / pool is a C-ABI kernel.
"""
from typing import Dict, Iterable

import torch
from torch import Tensor, nn

from . import layers, ops
from .layers import BasicBlock, BatchNormParams, PoolTailFn, StemFn
from .ops import ConvGeom


class ProfileCNN(nn.Module):
    """ResNet-style 1-D CNN.  Reference: src/profile_encoder.py:151-240 (+ _BasicBlock :111-148)."""

    def __init__(self, dim_in, blocks, groups: int = 1, block_type=BasicBlock, base_channels: int = 32,
                 dropout=0.1, metadata: bool = True) -> None:
        super().__init__()
        if base_channels % 8:
            raise ValueError('base_channels must be a multiple of 8 (16-byte channel groups in the kernels)')
        self.in_channels = self.base_channels = base_channels
        self.dilation = 1
        self.groups = groups            # stored, never used -- as in the reference (:125)

        self.conv1 = nn.Conv1d(dim_in, base_channels, 3, 2, 1, bias=False)
        self.bn1 = BatchNormParams(base_channels)
        self.geom = ConvGeom(tuple(self.conv1.weight.shape), 2, 1)

        self.layer1 = self._make_layer(blocks[0], base_channels, 1)
        self.layer2 = self._make_layer(blocks[1], base_channels * 2, 2)
        self.layer3 = self._make_layer(blocks[2], base_channels * 4, 2)
        self.layer4 = self._make_layer(blocks[3], base_channels * 8, 2)

        self.p_drop = float(dropout)
        self.dim_out = base_channels * 8 + metadata
        self.metadata = metadata
        ops.to_krsc_(self)              # block filters: [K][S][C] memory (state_dict / shapes unchanged)

    def _make_layer(self, repeat_times, channels, stride):
        layers_ = [BasicBlock(1, self.in_channels, channels, stride,
                              downsample=(stride != 1 or self.in_channels != channels))]
        self.in_channels = channels
        for _ in range(1, repeat_times):
            layers_.append(BasicBlock(1, channels, channels, 1, downsample=False))
        return nn.Sequential(*layers_)

    def tokenize(self, profile) -> Dict[str, Tensor]:
        # src/profile_encoder.py:206-210
        if not isinstance(profile, (list, tuple)):
            profile = [profile]
        return {'profile': torch.stack(list(profile))}

    def forward_features(self, profile: Tensor) -> Tensor:
        """-> channels-last bf16 feature map [B, L/32, 8*base] (the reference returns [B, C, L])."""
        x = profile.contiguous().float()
        from . import layers_f32
        if layers_f32.conv_f32():            # `precision: 32`: fp32 maps on the exact-fp32 kernels (parity mode)
            out = layers_f32.stem(self, x)
        else:
            out = StemFn.apply(x, self.conv1.weight, self.bn1.weight, self.bn1.bias, self)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                out = blk(out)
        return out

    def forward(self, profile: Tensor, **kwargs) -> Tensor:
        fmap = self.forward_features(profile)
        meta = kwargs['profile_len'].contiguous() if self.metadata else None
        p = self.p_drop if self.training else 0.0
        return PoolTailFn.apply(fmap, meta, 'max', profile.shape[1], p)


class ProfileTransformer(nn.Module):
    """Transformer profile encoder, reference src/profile_encoder.py:9-68.  The torch modules below are
    parameter containers only (same ``state_dict`` keys and default initialisation as the reference);
    ``forward`` runs LayerNorm / attention / GELU-MLP / embedding on the gfx950 kernels in exact fp32
    (``transformer.py``)."""

    def __init__(self, dim_in: int, dim_hidden: int, target_size: int, num_head: int, num_layers: int = 6,
                 dim_feedforward: int = 2024, dropout: float = 0.1, activation: str = 'gelu',
                 metadata: bool = True) -> None:
        super().__init__()
        self.expand = nn.Linear(dim_in, dim_hidden, bias=False)
        self.position = nn.Embedding(target_size + 2, dim_hidden, padding_idx=-1)
        self.padding_idx = self.position.padding_idx
        self.encoder = nn.TransformerEncoder(
            nn.TransformerEncoderLayer(d_model=dim_hidden, nhead=num_head, dim_feedforward=dim_feedforward,
                                       dropout=dropout, activation=activation, batch_first=True),
            num_layers=num_layers, enable_nested_tensor=False)
        self.p_drop = float(dropout)
        self.dim_out = dim_hidden + metadata
        self.metadata = metadata

    def tokenize(self, profile) -> Dict[str, Tensor]:
        # src/profile_encoder.py:38-54
        if not isinstance(profile, (list, tuple)):
            profile = [profile]
        n = max(p.shape[0] for p in profile) + 1
        time = torch.full((len(profile), n), self.padding_idx, dtype=torch.long)
        prof = torch.zeros(len(profile), n, profile[0].shape[1], dtype=profile[0].dtype)
        for i, p in enumerate(profile):
            time[i, :p.shape[0] + 1] = torch.arange(p.shape[0] + 1)
            prof[i, 1:p.shape[0] + 1] = p
        return {'profile': prof, 'time': time, 'padding_mask': time == self.padding_idx}

    def forward(self, profile: Tensor, time: Tensor, padding_mask: Tensor, **kwargs) -> Tensor:
        # src/profile_encoder.py:57-68: expand + position(time) -> post-norm encoder (key-padding mask) -> CLS
        from . import transformer as TF
        from .layers import TailFn, linear
        B, T, _ = profile.shape
        d = self.expand.weight.shape[0]
        x = linear(profile.reshape(B * T, -1).float().contiguous(), self.expand.weight)
        x = TF.EmbeddingAddFn.apply(x, self.position.weight, time, self.padding_idx).view(B, T, d)
        x = TF.post_norm_stack(self.encoder.layers, x, padding_mask, self.p_drop, self.training)
        cls = x[:, 0].contiguous()
        meta = kwargs['profile_len'].contiguous() if self.metadata else None
        return TailFn.apply(cls, meta, profile.shape[1], self.p_drop if self.training else 0.0)


class ProfileLSTM(nn.Module):
    """LSTM profile encoder, reference src/profile_encoder.py:71-108: bias-free expand -> nn.LSTM -> output at
    ``last_idx`` -> metadata concat -> dropout.  ``self.lstm`` is a parameter container (reference ``state_dict`` keys and
    initialisation); the arithmetic runs on the gfx950 kernels in exact fp32 (``lstm.py``)."""

    def __init__(self, dim_in: int, dim_hidden: int, num_layers: int, dropout: float = 0.1,
                 metadata: bool = True) -> None:
        super().__init__()
        self.expand = nn.Linear(dim_in, dim_hidden, bias=False)
        self.lstm = nn.LSTM(dim_hidden, dim_hidden, num_layers, batch_first=True, dropout=dropout)
        self.p_drop = float(dropout)
        self.dim_out = dim_hidden + metadata
        self.metadata = metadata

    def tokenize(self, profile) -> Dict[str, Tensor]:
        if not isinstance(profile, (list, tuple)):
            profile = [profile]
        last = torch.tensor([p.shape[0] - 1 for p in profile]).long()
        return {'profile': nn.utils.rnn.pad_sequence(list(profile), batch_first=True), 'last_idx': last}

    def forward(self, profile: Tensor, last_idx: Tensor, **kwargs) -> Tensor:
        from .layers import TailFn, linear
        from .lstm import lstm_stack
        B, T, _ = profile.shape
        d = self.expand.weight.shape[0]
        x = linear(profile.reshape(B * T, -1).float().contiguous(), self.expand.weight).view(B, T, d)      # :99
        feat = lstm_stack(self.lstm, x, last_idx, self.training)                                            # :100-101
        meta = kwargs['profile_len'].contiguous() if self.metadata else None                               # :102-105
        return TailFn.apply(feat, meta, profile.shape[1], self.p_drop if self.training else 0.0)
