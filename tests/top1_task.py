"""The synthetic-class retrieval task shared by tests/test_zz_top1_parity_gpu.py and scripts/top1_oracle_spread.py.

No real dataset is available offline (the reference's data directories are git-ignored), so classes are synthetic:
each class has a smooth random image pattern and a random profile curve; a sample is its class prototype plus noise.
Everything is generated on the CPU from fixed seeds, so the oracle and the HIP path see identical tensors."""
import torch

N_CLASSES, T = 12, 64

CFG = dict(image_encoder_args=dict(name='resnet18', num_classes=0, pretrained=False, dropout=0.0, in_chans=1, metadata=True),
           profile_encoder_args=dict(dim_in=6, blocks=[1, 1, 1, 1], base_channels=16, dropout=0.0, metadata=True),
           coordination_args=dict(method='clip'),
           optim_args=dict(lr=2e-2, momentum=0.9, weight_decay=1e-3, nesterov=True))
DIM_EMBED = 64


def prototypes(gen):
    img = torch.nn.functional.interpolate(torch.randn(N_CLASSES, 1, 8, 8, generator=gen), size=(T, T), mode='bilinear',
                                          align_corners=False)
    prof = torch.nn.functional.interpolate(torch.randn(N_CLASSES, 6, 12, generator=gen), size=T, mode='linear',
                                           align_corners=False).transpose(1, 2)
    return img * 0.5, prof * 0.6


def batch_of(protos, labels, gen, noise):
    img, prof = protos
    B = labels.shape[0]
    return {'image': (img[labels] + noise * torch.randn(B, 1, T, T, generator=gen)).clamp(-1, 1),
            'profile': (prof[labels] + noise * torch.randn(B, T, 6, generator=gen)).clamp(-1, 1),
            'image_shape': torch.full((B, 2), 100), 'profile_len': torch.full((B, 1), 200), 'buckets': 1}


def make_task(steps, B, noise, n_test=1024, seed=7):
    gen = torch.Generator().manual_seed(seed)
    protos = prototypes(gen)
    batches = [batch_of(protos, torch.randint(0, N_CLASSES, (B,), generator=gen), gen, noise) for _ in range(steps)]
    test_labels = torch.arange(n_test) % N_CLASSES
    test = batch_of(protos, test_labels, gen, noise)
    return batches, test, test_labels


def top1(img_emb, prof_emb, labels):
    """Cross-modal retrieval: does the nearest profile embedding belong to the image's class?"""
    u = torch.nn.functional.normalize(img_emb.float().cpu())
    v = torch.nn.functional.normalize(prof_emb.float().cpu())
    nearest = (u @ v.T).argmax(1)
    return float((labels[nearest] == labels).float().mean())


def init_state(cfg=CFG, seed=0):
    """The product module's own initialisation (timm / torch defaults restated there) as a CPU state_dict."""
    from multimodal_plankton_recognition_amd.model import MultiModel
    torch.manual_seed(seed)
    model = MultiModel(dim_embed=DIM_EMBED, **cfg)
    return {k: v.detach().clone() for k, v in model.state_dict().items()}
