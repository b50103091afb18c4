"""Drop-in counterpart of /root/reference/scripts/train_multi.py (same CLI: -d/--dataset, -m/--modelcard, same
YAML keys, same run naming and checkpoint layout), driving the gfx950-native MultiModel.

    cd scripts && python3 train_multi.py -d <fold dir with train.csv/test.csv> -m ../model_cards/<card>.yaml

New, clearly flagged options: --synthetic N (no dataset on disk: N synthetic pairs per epoch), --max-epochs,
--limit-batches (smoke runs).

[new] Multi-GPU data parallel (the reference is single-GPU): one process per GPU under torchrun,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        train_multi.py -d <fold> -m <card>

`bs` is then the PER-GPU batch; every rank trains on its shard of each global batch (DistributedSampler), the contrastive
loss is taken over the global batch (buckets = 1), gradients are all-reduced over RCCL, rank 0 writes logs / checkpoints.
"""
import argparse
import os
import sys
from pathlib import Path

import torch
import yaml
from torch.utils.data import DataLoader

sys.path.append('../')
sys.path.append(str(Path(__file__).resolve().parent.parent))
from multimodal_plankton_recognition_amd.data import (ImageTransformTest, ImageTransformTrain, MultiSet,  # noqa: E402
                                                      PairAugmentation, ProfileTransformTest,
                                                      ProfileTransformTrain, SyntheticMultiSet, make_multi_collate)
from multimodal_plankton_recognition_amd.data import CachedMultiSet, cached_collate  # noqa: E402
from multimodal_plankton_recognition_amd.augment import DevicePipeline  # noqa: E402
from multimodal_plankton_recognition_amd.model import MultiModel  # noqa: E402
from multimodal_plankton_recognition_amd.trainer import EarlyStopping, ModelCheckpoint, TensorBoardLogger, Trainer  # noqa: E402

parser = argparse.ArgumentParser()
parser.add_argument("-d", "--dataset", help="Location to dataset tables.")
parser.add_argument("-m", "--modelcard", help="Path to model card (yaml file).")
parser.add_argument("--synthetic", type=int, default=0, help="[new] use N synthetic pairs per epoch instead of a dataset")
parser.add_argument("--max-epochs", type=int, default=None, help="[new] override trainer_args.max_epochs")
parser.add_argument("--limit-batches", type=int, default=None, help="[new] cap batches per epoch (smoke runs)")
parser.add_argument("--logdir", default="../logs/", help="[new] where runs are written (reference: ../logs/)")
parser.add_argument("--gpu-augment", action="store_true",
                    help="[new] cache the deterministic transforms per sample and run crop / flips / resize / noise on the GPU")
args = parser.parse_args()

world, rank = int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('RANK', 0))
if world > 1:
    # one process per GPU: pick this rank's device BEFORE anything touches the GPU
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % max(torch.cuda.device_count(), 1))

card = Path(args.modelcard)
with open(card, 'r') as stream:
    card_dict = yaml.safe_load(stream)

torch.set_float32_matmul_precision(card_dict.get('precision', 'highest'))
target_size = card_dict.get('target_size')
bs = card_dict['bs']

if args.synthetic:
    data_path = Path('synthetic/data')
    train_set = SyntheticMultiSet(args.synthetic, target_size, seed=1234)
    test_set = SyntheticMultiSet(max(bs, args.synthetic // 4), target_size, seed=4321)
elif args.gpu_augment:
    data_path = Path(f'{args.dataset}')
    train_set = CachedMultiSet(data_path / 'train.csv', target_size, train=True)
    test_set = CachedMultiSet(data_path / 'test.csv', target_size, train=False)
else:
    data_path = Path(f'{args.dataset}')
    train_set = MultiSet(annotation_path=data_path / 'train.csv', image_transforms=ImageTransformTrain(target_size),
                         profile_transform=ProfileTransformTrain(target_size), pair_augmentation=PairAugmentation())
    test_set = MultiSet(annotation_path=data_path / 'test.csv', image_transforms=ImageTransformTest(target_size),
                        profile_transform=ProfileTransformTest(target_size))

if world > 1:
    torch.manual_seed(0)                        # identical initial weights on every rank (parameters are replicated)
model = MultiModel(
    dim_embed=card_dict['dim_embedding'],
    image_encoder_args=card_dict['image_encoder_args'],
    profile_encoder_args=card_dict['profile_encoder_args'],
    coordination_args=card_dict['coordination_args'],
    optim_args=card_dict['optim_args'],
)
if world > 1:
    # the replicas share their parameters (same seed above, and Trainer.fit broadcasts rank 0's before the first step), NOT
    # their randomness: loader workers, augmentation, crops / flips and dropout draw from per-rank streams from here on
    import random
    import numpy as np
    torch.manual_seed(1 + rank)
    random.seed(1 + rank)
    np.random.seed(1 + rank)
multi_collate = make_multi_collate(model, card_dict['buckets'])
batch_transform = None
if args.gpu_augment and not args.synthetic:
    multi_collate = cached_collate
    batch_transform = DevicePipeline(model, target_size, card_dict['buckets'])

if world > 1:
    from torch.utils.data.distributed import DistributedSampler
    train_sampler = DistributedSampler(train_set, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
    valid_sampler = DistributedSampler(test_set, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
    train_loader = DataLoader(dataset=train_set, batch_size=bs, sampler=train_sampler, num_workers=card_dict['num_workers'],
                              drop_last=True, collate_fn=multi_collate)
    valid_loader = DataLoader(dataset=test_set, batch_size=bs, sampler=valid_sampler, num_workers=card_dict['num_workers'],
                              drop_last=True, collate_fn=multi_collate)
else:
    train_loader = DataLoader(dataset=train_set, batch_size=bs, shuffle=True, num_workers=card_dict['num_workers'],
                              drop_last=True, collate_fn=multi_collate)
    valid_loader = DataLoader(dataset=test_set, batch_size=bs, shuffle=True, num_workers=card_dict['num_workers'],
                              drop_last=True, collate_fn=multi_collate)

name = card.name.split('.')[0] + '_' + '_'.join(str(data_path).split('/')[-2:])
logger = TensorBoardLogger(save_dir=args.logdir, name=name) if rank == 0 else None
checkpoint = ModelCheckpoint(filename="{epoch}_{valid_loss:.5f}", monitor="valid_loss",
                             save_top_k=card_dict.get('save_top_k', 1), mode="min")
stopper = EarlyStopping(monitor='valid_loss', min_delta=0.0, patience=card_dict['patience'], check_finite=False,
                        mode='min')
trainer_args = dict(card_dict['trainer_args'])
if args.max_epochs is not None:
    trainer_args['max_epochs'] = args.max_epochs
    trainer_args['min_epochs'] = min(trainer_args.get('min_epochs') or 0, args.max_epochs)
trainer = Trainer(log_every_n_steps=len(train_loader), logger=logger, callbacks=[checkpoint, stopper],
                  limit_train_batches=args.limit_batches, limit_val_batches=args.limit_batches,
                  batch_transform=batch_transform, **trainer_args)

if rank == 0:
    print(f'Training from model card {args.modelcard}')
trainer.fit(model, train_loader, valid_loader)
if world > 1:
    from multimodal_plankton_recognition_amd import distributed as D
    D.barrier()
    D.shutdown()
