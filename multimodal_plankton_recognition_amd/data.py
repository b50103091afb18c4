"""Datasets feeding the train_multi path.

* ``SyntheticMultiSet`` -- on-the-fly image/profile pairs with the statistics of SURVEY.md 8(d); produces
  exactly the sample dict of the reference's ``MultiSet.__getitem__`` (src/data.py:57-59), so that
  ``multi_collate`` and the training loop run with no dataset on disk (``train_multi.py --synthetic``).
* ``MultiSet`` + transforms -- CPU-side counterpart of src/data.py:19-59,73-157,198-204,267-306 built on PIL /
  numpy / torch only (torchvision and cv2 are not available here): 25-px scale-bar crop, Lanczos resize of the
  long side with edge-replicate padding to a square, grayscale, [-1, 1] range, random crop / vertical flip;
  profile log1p / per-channel ceiling / [-1, 1], bilinear resize, random crop, 1e-3 noise; paired horizontal
  flip <-> time reversal.  This is host-side I/O, not part of the GPU hot path.
"""
import math
import random
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

PROFILE_CEIL = [9.6058, 8.9211, 8.9211, 8.9211, 8.9211, 8.9211]        # src/data.py:127,146


class SyntheticMultiSet(Dataset):
    def __init__(self, length=2048, target_size=224, num_classes=50, seed=1234):
        self.length, self.T, self.num_classes, self.seed = length, target_size, num_classes, seed
        self.class_names = np.array([f'class_{i:02d}' for i in range(num_classes)])

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        g = torch.Generator().manual_seed(self.seed * 1000003 + index)
        T = self.T
        image = (torch.randn(1, T, T, generator=g) * 0.0938 + 0.6136).clamp_(0, 1) * 2 - 1
        profile = torch.rand(T, 6, generator=g) * 2 - 1
        return {'image': image, 'profile': profile,
                'label': self.class_names[int(torch.randint(0, self.num_classes, (1,), generator=g))],
                'image_shape': torch.randint(32, 401, (2,), generator=g),
                'profile_length': torch.randint(8, 1025, (1,), generator=g)}


# ------------------------------------------------------------------------------------------------ image pipeline
def resize_pil(img, target_res=224, edge=True):
    """src/data.py:267-306 (edge=True branch): Lanczos-resize the long side to target_res, replicate-pad the
    short side to a square."""
    from PIL import Image
    w, h = img.size
    if h <= w:
        img = img.resize((target_res, int(np.around(target_res * h / w))), Image.Resampling.LANCZOS)
        arr = np.asarray(img)
        top = (target_res - arr.shape[0]) // 2
        pad = [(top, target_res - arr.shape[0] - top), (0, 0)] + [(0, 0)] * (arr.ndim - 2)
    else:
        img = img.resize((int(np.around(target_res * w / h)), target_res), Image.Resampling.LANCZOS)
        arr = np.asarray(img)
        left = (target_res - arr.shape[1]) // 2
        pad = [(0, 0), (left, target_res - arr.shape[1] - left)] + [(0, 0)] * (arr.ndim - 2)
    return Image.fromarray(np.pad(arr, pad_width=pad, mode='edge'))


def _to_gray_pm1(img):
    arr = torch.from_numpy(np.asarray(img).copy()).float()
    if arr.ndim == 3:                                   # ITU-R 601-2 luma, as torchvision's Grayscale
        arr = arr[..., 0] * 0.2989 + arr[..., 1] * 0.587 + arr[..., 2] * 0.114
    return (arr / 255.0).unsqueeze(0) * 2 - 1            # [1, H, W] in [-1, 1]  (src/data.py:80-82)


class ImageTransformTrain:
    """src/data.py:73-91."""

    def __init__(self, target_size=224):
        self.T = target_size

    def __call__(self, img):
        img = img.crop((0, 25, img.width, img.height))
        x = _to_gray_pm1(resize_pil(img, math.ceil(1.05 * self.T), edge=True))
        top = random.randint(0, x.shape[1] - self.T)
        left = random.randint(0, x.shape[2] - self.T)
        x = x[:, top:top + self.T, left:left + self.T]
        if random.random() < 0.5:
            x = x.flip(1)                                # RandomVerticalFlip
        return x.contiguous()


class ImageTransformTest:
    """src/data.py:94-107."""

    def __init__(self, target_size=224):
        self.T = target_size

    def __call__(self, img):
        img = img.crop((0, 25, img.width, img.height))
        return _to_gray_pm1(resize_pil(img, self.T, edge=True))


# ------------------------------------------------------------------------------------------------ profile pipeline
def _profile_base(prof, length):
    """log1p / ceiling * 2 - 1, then the reference's ``v2.Resize((1, length))`` on the [C, 1, L] tensor
    (src/data.py:129-133,149-152): torchvision resizes tensors with ``interpolate(mode='bilinear', antialias=True)`` --
    a triangle filter whose support grows with the downscaling factor (real profiles are mostly LONGER than the target, so
    this is an area-weighted average, not a two-point sample); upscaling is plain bilinear.  float64 like the reference
    (``torch.tensor(ndarray)``), cast to float32 by the caller at the end."""
    x = torch.as_tensor(np.asarray(prof), dtype=torch.float64).add(1).log()
    x = x.div(torch.tensor(PROFILE_CEIL[:x.shape[1]], dtype=torch.float64)).mul(2).add(-1)      # src/data.py:129
    x = F.interpolate(x.t()[None, :, None, :], size=(1, length), mode='bilinear', align_corners=False, antialias=True)
    return x[0, :, 0, :]                                                         # [C, length]


class ProfileTransformTrain:
    """src/data.py:124-141."""

    def __init__(self, target_size=224):
        self.T = target_size

    def __call__(self, prof):
        x = _profile_base(prof, math.ceil(1.05 * self.T))
        left = random.randint(0, x.shape[1] - self.T)
        x = x[:, left:left + self.T]
        x = x + 1e-3 * torch.randn_like(x)
        return x.t().float().contiguous()


class ProfileTransformTest:
    """src/data.py:144-157."""

    def __init__(self, target_size=224):
        self.T = target_size

    def __call__(self, prof):
        return _profile_base(prof, self.T).t().float().contiguous()


class PairAugmentation:
    """src/data.py:198-204: horizontal image flip <-> profile time reversal, p = 0.5."""

    def __call__(self, image, profile):
        if random.randint(0, 1) == 0:
            image = image.flip(-1)
            profile = profile.flip(0)
        return image, profile


class MultiSet(Dataset):
    """src/data.py:19-59: CSV-indexed image + profile pairs (paths relative to the CSV's directory)."""

    def __init__(self, annotation_path, image_transforms, profile_transform, pair_augmentation=None):
        import pandas as pd
        annotation_path = Path(annotation_path)
        self.parent = annotation_path.parent
        self.table = pd.read_csv(annotation_path)
        self.class_names = np.unique(self.table['class'])
        self.image_transforms = image_transforms
        self.profile_transform = profile_transform
        self.pair_augmentation = pair_augmentation

    def __len__(self):
        return len(self.table)

    def __getitem__(self, index):
        from PIL import Image
        image = Image.open(self.parent / self.table.image[index]).convert('RGB')
        profile = np.loadtxt(self.parent / self.table.profile[index], delimiter=',', skiprows=1)
        image_shape = torch.tensor(image.size[::-1])                 # pre-crop (H, W), src/data.py:46
        profile_length = torch.tensor([profile.shape[0]])
        image = self.image_transforms(image)
        profile = self.profile_transform(profile)
        label = self.table['class'][index]
        if self.pair_augmentation:
            image, profile = self.pair_augmentation(image, profile)
        return {'image': image, 'profile': profile, 'label': label, 'image_shape': image_shape,
                'profile_length': profile_length}


def make_multi_collate(model, buckets):
    """scripts/train_multi.py:66-76: unzip the five sample fields by dict order, drop the label, stack, tokenize."""
    def multi_collate(batch):
        image, profile, _, image_shape, profile_len = zip(*(sample.values() for sample in batch))
        out = {'image': torch.stack(image)}
        out.update(model.profile_encoder.tokenize(profile))
        out['image_shape'] = torch.stack(image_shape)
        out['profile_len'] = torch.stack(profile_len)
        out['buckets'] = buckets
        return out
    return multi_collate


# ------------------------------------------------------------------------------------------------ cached dataset (GPU augmentation)
class CachedMultiSet(Dataset):
    """MultiSet whose samples carry only the DETERMINISTIC part of the transforms (``augment.cache_image`` /
    ``cache_profile``: scale-bar crop, Lanczos resize to the cached square, grayscale bytes; raw profile counts), computed
    on first access and kept: the random part runs on the device per step (``augment.DevicePipeline``).
    ``train=True`` caches images at ceil(1.05 T) (room for the random crop), ``train=False`` at T (test transform)."""

    def __init__(self, annotation_path, target_size=224, train=True):
        import pandas as pd
        annotation_path = Path(annotation_path)
        self.parent = annotation_path.parent
        self.table = pd.read_csv(annotation_path)
        self.class_names = np.unique(self.table['class'])
        self.T, self.train = target_size, train
        self._cache = {}

    def __len__(self):
        return len(self.table)

    def __getitem__(self, index):
        hit = self._cache.get(index)
        if hit is None:
            from PIL import Image
            from .augment import cache_image, cache_profile
            image = Image.open(self.parent / self.table.image[index]).convert('RGB')
            profile = np.loadtxt(self.parent / self.table.profile[index], delimiter=',', skiprows=1)
            side = self.T if not self.train else None
            hit = self._cache[index] = {
                'image_u8': cache_image(image, self.T) if side is None else cache_image(image, self.T, side=side),
                'profile_raw': cache_profile(profile), 'label': self.table['class'][index],
                'image_shape': torch.tensor(image.size[::-1]),                     # pre-crop (H, W), src/data.py:46
                'profile_length': torch.tensor([profile.shape[0]])}
        return hit


def cached_collate(batch):
    """Stack the cached fields: image bytes [B, S, S], zero-padded raw profiles [B, Lmax, C] + their lengths."""
    raws = [b['profile_raw'] for b in batch]
    lmax = max(r.shape[0] for r in raws)
    raw = torch.zeros(len(batch), lmax, raws[0].shape[1])
    for i, r in enumerate(raws):
        raw[i, :r.shape[0]] = r
    return {'image_u8': torch.stack([b['image_u8'] for b in batch]), 'profile_raw': raw,
            'raw_len': torch.tensor([r.shape[0] for r in raws], dtype=torch.int32),
            'image_shape': torch.stack([b['image_shape'] for b in batch]),
            'profile_len': torch.stack([b['profile_length'] for b in batch])}
