"""GPU side of the input pipeline: per-step random augmentation of PRE-DECODED batches.

The reference does everything per sample on the host (PIL decode, Lanczos resize, torchvision v2 transforms:
/root/reference/src/data.py:73-157,198-204) -- at the tens of thousands of samples per second this path trains at, that
loader is the bottleneck.  Split: what is deterministic per sample is done ONCE when the dataset is cached
(``cache_image`` / ``cache_profile``: scale-bar crop, Lanczos resize of the long side to ceil(1.05 T) with edge padding,
grayscale bytes; raw profile counts as they are), what is random per step -- crop offsets, vertical flip, the paired
horizontal-flip / time-reversal, profile noise -- runs on the device in two kernels (``csrc/augment.hip``) that also do the
value transforms (bytes -> [-1, 1]; log1p / ceiling, linear resize).  Random decisions are drawn on the host from a
``torch.Generator`` and handed to the kernels, so a batch is reproducible and is tested against the host transforms of
``data.py`` decision by decision.
"""
import math

import numpy as np
import torch

from . import _native as N
from .data import PROFILE_CEIL, resize_pil

F32 = torch.float32


def cache_image(img, target_size=224, side=None):
    """PIL image -> uint8 [S, S] (S = ceil(1.05 T) for training, `side` = T for the test transform): src/data.py:77-80 /
    :98-101 up to the grayscale bytes (PIL 'L' luma)."""
    img = img.crop((0, 25, img.width, img.height))
    img = resize_pil(img.convert('L'), math.ceil(1.05 * target_size) if side is None else side, edge=True)
    return torch.from_numpy(np.asarray(img).copy())


def cache_profile(prof):
    """[L, C] raw counts -> fp32 tensor (the log / resize happen on the device)."""
    return torch.as_tensor(np.asarray(prof), dtype=F32).contiguous()


class GpuAugment:
    """``images, profiles = aug(u8_images, raw_profiles, lengths)`` on device tensors; train mode draws the random
    decisions, ``decisions=...`` replays given ones (tests).  ``side`` = edge of the cached images / length the profiles
    are resized to (default ceil(1.05 T); T for the test transforms: then there is nothing to crop)."""

    def __init__(self, target_size=224, noise=1e-3, seed=0, side=None):
        self.T = int(target_size)
        self.S = math.ceil(1.05 * self.T) if side is None else int(side)
        self.noise = float(noise)
        self.gen = torch.Generator().manual_seed(seed)
        self._ceil = {}

    def identity(self, B):
        """The decisions of the test transforms: no crop offset, no flips, no noise."""
        z = torch.zeros(B, dtype=torch.int32)
        return {'top': z, 'left': z, 'vflip': z.to(torch.uint8), 'pair_flip': z.to(torch.uint8), 'prof_left': z, 'seed': 0}

    def draw(self, B):
        g, span = self.gen, self.S - self.T + 1
        r = lambda hi: torch.randint(0, hi, (B,), generator=g, dtype=torch.int32)
        return {'top': r(span), 'left': r(span), 'vflip': r(2).to(torch.uint8), 'pair_flip': r(2).to(torch.uint8),
                'prof_left': r(span), 'seed': int(torch.randint(0, 2 ** 31 - 1, (1,), generator=g))}

    def images(self, u8, d):
        B, S, S2 = u8.shape
        assert S == self.S and S2 == self.S and u8.dtype == torch.uint8, 'cached images must be uint8 [B, ceil(1.05 T), ceil(1.05 T)]'
        dev = u8.device
        out = torch.empty(B, 1, self.T, self.T, dtype=F32, device=dev)
        N.call('mpr_aug_image', u8.contiguous(), d['top'].to(dev), d['left'].to(dev), d['vflip'].to(dev), d['pair_flip'].to(dev),
               out, B, self.S, self.T)
        return out

    def profiles(self, raw, lengths, d, noise=None):
        B, Lmax, C = raw.shape
        dev = raw.device
        key = (C, dev)
        if key not in self._ceil:
            self._ceil[key] = torch.tensor(PROFILE_CEIL[:C], dtype=F32, device=dev)
        out = torch.empty(B, self.T, C, dtype=F32, device=dev)
        N.call('mpr_aug_profile', raw.contiguous().float(), lengths.to(device=dev, dtype=torch.int32), d['prof_left'].to(dev),
               d['pair_flip'].to(dev), self._ceil[key], out, B, Lmax, C, self.S, self.T,
               self.noise if noise is None else float(noise), d['seed'])
        return out

    def __call__(self, u8_images, raw_profiles, lengths, decisions=None):
        d = decisions if decisions is not None else self.draw(u8_images.shape[0])
        return self.images(u8_images, d), self.profiles(raw_profiles, lengths, d)


class DevicePipeline:
    """Trainer hook (``Trainer(batch_transform=...)``): turns a ``data.cached_collate`` batch that is already on the device
    into the batch dict ``MultiModel.training_step`` expects (scripts/train_multi.py:66-76), with the random part of
    ImageTransformTrain / ProfileTransformTrain / PairAugmentation done by the two augmentation kernels (training) or the
    test transforms' fixed decisions (evaluation; the validation set is cached at T).  After the device-side resize every
    profile has T samples, so the tokenizers' work is fixed-shape: CNN = the tensor itself, transformer = zero CLS row +
    arange time + all-valid mask, LSTM = last index T - 1."""

    def __init__(self, model, target_size, buckets, noise=1e-3, seed=0):
        self.train_aug = GpuAugment(target_size, noise, seed)
        self.eval_aug = GpuAugment(target_size, 0.0, seed, side=target_size)
        self.T, self.buckets = int(target_size), buckets
        pe = type(model.profile_encoder).__name__
        self.kind = 'transformer' if pe == 'ProfileTransformer' else ('lstm' if pe == 'ProfileLSTM' else 'cnn')

    def __call__(self, batch, training):
        if 'image_u8' not in batch:
            return batch
        aug = self.train_aug if training else self.eval_aug
        B = batch['image_u8'].shape[0]
        d = aug.draw(B) if training else aug.identity(B)
        image, profile = aug(batch['image_u8'], batch['profile_raw'], batch['raw_len'], decisions=d)
        out = {'image': image, 'profile': profile}
        dev = image.device
        if self.kind == 'transformer':
            out['profile'] = torch.cat((torch.zeros(B, 1, profile.shape[2], device=dev), profile), 1)
            out['time'] = torch.arange(self.T + 1, device=dev).repeat(B, 1)
            out['padding_mask'] = torch.zeros(B, self.T + 1, dtype=torch.bool, device=dev)
        elif self.kind == 'lstm':
            out['last_idx'] = torch.full((B,), self.T - 1, dtype=torch.long, device=dev)
        out.update(image_shape=batch['image_shape'], profile_len=batch['profile_len'], buckets=self.buckets)
        return out
