"""Top-1 parity on a synthetic-class task (SURVEY 8d "parity reported with the numbers"): the GPU path and the CPU oracle
train the same MultiModel from the same initialisation on the same batches (dropout 0), then both embed a held-out set;
cross-modal retrieval top-1 (does the nearest profile embedding belong to the image's class?) must agree.

No real dataset is available (the reference's data directories are git-ignored), so classes are synthetic: each class has a
smooth random image pattern and a random profile curve; samples are the class prototypes plus noise, sized so that the task
is learnable but not trivial."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
N_CLASSES, T = 12, 64


def _prototypes(gen):
    img = torch.nn.functional.interpolate(torch.randn(N_CLASSES, 1, 8, 8, generator=gen), size=(T, T), mode='bilinear',
                                          align_corners=False)
    prof = torch.nn.functional.interpolate(torch.randn(N_CLASSES, 6, 12, generator=gen), size=T, mode='linear',
                                           align_corners=False).transpose(1, 2)
    return img * 0.5, prof * 0.6


def _batch(protos, labels, gen, noise):
    img, prof = protos
    B = labels.shape[0]
    return {'image': (img[labels] + noise * torch.randn(B, 1, T, T, generator=gen)).clamp(-1, 1),
            'profile': (prof[labels] + noise * torch.randn(B, T, 6, generator=gen)).clamp(-1, 1),
            'image_shape': torch.full((B, 2), 100), 'profile_len': torch.full((B, 1), 200), 'buckets': 1}


def _top1(img_emb, prof_emb, labels):
    u = torch.nn.functional.normalize(img_emb.float().cpu())
    v = torch.nn.functional.normalize(prof_emb.float().cpu())
    nearest = (u @ v.T).argmax(1)
    return float((labels[nearest] == labels).float().mean())


def test_top1_of_gpu_training_matches_cpu_oracle_training():
    from multimodal_plankton_recognition_amd.model import MultiModel
    from oracle import model as OM
    cfg = dict(image_encoder_args=dict(name='resnet18', num_classes=0, pretrained=False, dropout=0.0, in_chans=1, metadata=True),
               profile_encoder_args=dict(dim_in=6, blocks=[1, 1, 1, 1], base_channels=16, dropout=0.0, metadata=True),
               coordination_args=dict(method='clip'),
               optim_args=dict(lr=2e-2, momentum=0.9, weight_decay=1e-3, nesterov=True))
    torch.manual_seed(0)
    model = MultiModel(dim_embed=64, **cfg)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    init_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    gen = torch.Generator().manual_seed(7)
    protos = _prototypes(gen)
    steps, B, noise = 40, 48, 0.6
    batches = [_batch(protos, torch.randint(0, N_CLASSES, (B,), generator=gen), gen, noise) for _ in range(steps)]
    test_labels = torch.arange(1024) % N_CLASSES
    test = _batch(protos, test_labels, gen, noise)

    # CPU oracle
    torch.set_num_threads(min(16, torch.get_num_threads()))
    bufs = {}
    for b in batches:
        OM.train_step(sd, b, cfg, bufs)
    with torch.no_grad():
        emb = OM.encode(sd, test, cfg, train=False)
    acc_cpu = _top1(emb['image_emb'], emb['profile_emb'], test_labels)

    # GPU path: fp32 atomics make two runs of the SAME build differ in the last bits, and 40 steps of a train-mode
    # BatchNorm net amplify that to a spread of ~2 points of top-1 (measured over 16 runs: 0.898 .. 0.963, mean 0.938,
    # with or without the round-2 fusions) -- so the statement is about the MEAN of five runs (three left a few per cent
    # of the runs of this test outside 3 points: one failure in ~20 full-suite runs)
    accs = []
    for rep in range(5):
        model.load_state_dict(init_sd)
        model.to(DEV).train()
        opt = model.configure_optimizers()
        for b in batches:
            opt.zero_grad()
            loss = model.training_step({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in b.items()}, 0)
            loss.backward()
            opt.step()
        model.eval()
        with torch.no_grad():
            out = model.encode(**{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in test.items()})
        accs.append(_top1(out['image_emb'], out['profile_emb'], test_labels))
    acc_gpu = sum(accs) / len(accs)
    print(f'synthetic-class retrieval top-1: CPU oracle {acc_cpu:.4f}, GPU path {accs} (mean {acc_gpu:.4f})')
    assert acc_cpu > 3.0 / N_CLASSES, f'the task was not learned by the oracle ({acc_cpu})'
    assert abs(acc_gpu - acc_cpu) <= 0.03 + 1e-9, (acc_cpu, accs)
    assert min(accs) > acc_cpu - 0.09, (acc_cpu, accs)
