"""Minimal stand-in for the slice of ``lightning`` the reference's scripts use
(scripts/train_multi.py:86-107): Trainer.fit with gradient accumulation, per-epoch validation,
ModelCheckpoint (top-k by a monitored metric, Lightning's ``{epoch}_{valid_loss:.5f}`` naming and
``state_dict`` / ``hyper_parameters`` checkpoint keys), EarlyStopping, and a logger that writes
``<save_dir>/<name>/version_N/metrics.jsonl`` where TensorBoardLogger would write event files.
``lightning`` itself is not installed in this image.
"""
import json
import math
import os
import re

import torch

from . import ops


class TensorBoardLogger:
    """Directory layout of lightning's TensorBoardLogger; metrics go to metrics.jsonl (tensorboard is absent)."""

    def __init__(self, save_dir, name):
        root = os.path.join(save_dir, name)
        os.makedirs(root, exist_ok=True)
        versions = [int(m.group(1)) for d in os.listdir(root) if (m := re.fullmatch(r'version_(\d+)', d))]
        self.version = max(versions) + 1 if versions else 0
        self.log_dir = os.path.join(root, f'version_{self.version}')
        os.makedirs(self.log_dir, exist_ok=True)

    def log_metrics(self, metrics):
        with open(os.path.join(self.log_dir, 'metrics.jsonl'), 'a') as f:
            f.write(json.dumps(metrics) + '\n')


class ModelCheckpoint:
    def __init__(self, filename='{epoch}', monitor='valid_loss', save_top_k=1, mode='min'):
        self.filename, self.monitor, self.save_top_k, self.mode = filename, monitor, save_top_k, mode
        self.best = []          # [(score, path)]

    def format_name(self, epoch, metrics):
        # lightning renders "{epoch}_{valid_loss:.5f}" as "epoch=86_valid_loss=0.92302"
        def sub(m):
            key, fmt = m.group(1), m.group(2) or ''
            val = epoch if key == 'epoch' else metrics[key]
            return f'{key}=' + format(val, fmt[1:] if fmt else '')
        return re.sub(r'\{(\w+)(:[^}]*)?\}', sub, self.filename) + '.ckpt'

    def on_validation_end(self, trainer, model, metrics):
        if self.monitor not in metrics or self.save_top_k == 0:
            return
        score = float(metrics[self.monitor])
        sign = 1.0 if self.mode == 'min' else -1.0
        worst = max(self.best, key=lambda t: sign * t[0]) if self.best else None
        if self.save_top_k > 0 and len(self.best) >= self.save_top_k and sign * score >= sign * worst[0]:
            return
        ckpt_dir = os.path.join(trainer.logger.log_dir, 'checkpoints')
        os.makedirs(ckpt_dir, exist_ok=True)
        path = os.path.join(ckpt_dir, self.format_name(trainer.current_epoch, metrics))
        v = 0
        while os.path.exists(path):          # a second validation run in one epoch (val_check_interval): lightning's -vN suffix
            v += 1
            path = os.path.join(ckpt_dir, self.format_name(trainer.current_epoch, metrics)[:-5] + f'-v{v}.ckpt')
        torch.save({'state_dict': model.state_dict(), 'hyper_parameters': getattr(model, 'hparams', {}),
                    'epoch': trainer.current_epoch, 'global_step': trainer.global_step,
                    'optimizer_states': [trainer.optimizer.state_dict()] if trainer.optimizer else []}, path)
        self.best.append((score, path))
        if self.save_top_k > 0 and len(self.best) > self.save_top_k:
            drop = max(self.best, key=lambda t: sign * t[0])
            self.best.remove(drop)
            if os.path.exists(drop[1]):
                os.remove(drop[1])


class EarlyStopping:
    def __init__(self, monitor='valid_loss', min_delta=0.0, patience=3, check_finite=True, mode='min'):
        self.monitor, self.min_delta, self.patience, self.check_finite, self.mode = monitor, min_delta, patience, check_finite, mode
        self.best, self.wait = None, 0

    def on_validation_end(self, trainer, model, metrics):
        if self.monitor not in metrics:
            return
        score = float(metrics[self.monitor])
        if self.check_finite and not math.isfinite(score):
            trainer.should_stop = True
            return
        sign = 1.0 if self.mode == 'min' else -1.0
        if self.best is None or sign * score < sign * self.best - self.min_delta:
            self.best, self.wait = score, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                trainer.should_stop = True


def load_from_checkpoint(model_cls, path, map_location='cpu', **override):
    """LightningModule.load_from_checkpoint: rebuild from saved hyper-parameters, then load the weights."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model = model_cls(**{**ckpt['hyper_parameters'], **override})
    model.load_state_dict(ckpt['state_dict'])
    return model


class Trainer:
    """fit() also drives data-parallel training (the reference is single-GPU): launched under torchrun
    (WORLD_SIZE > 1, one process per GPU) every rank runs this loop on ITS shard of each global batch through
    distributed.DataParallelStep -- global contrastive loss, bucketed gradient all-reduce over RCCL, gradient accumulation
    with one reduction per optimizer step -- validation computes the same GLOBAL-batch contrastive loss (every rank logs
    the same number: checkpointing and early stopping decide identically everywhere), and only rank 0 writes logs and
    checkpoints.  Parameters and buffers are broadcast from rank 0 before the first step.  The loaders must shard the data
    (train_multi.py: DistributedSampler)."""

    def __init__(self, logger=None, callbacks=(), max_epochs=1000, min_epochs=0, accumulate_grad_batches=1,
                 precision=None, val_check_interval=None, check_val_every_n_epoch=1, log_every_n_steps=50,
                 max_steps=-1, limit_train_batches=None, limit_val_batches=None, device=None, batch_transform=None,
                 **unused):
        self.logger, self.callbacks = logger, list(callbacks)
        self.max_epochs, self.min_epochs = max_epochs, min_epochs or 0
        self.accumulate = max(1, int(accumulate_grad_batches or 1))
        self.check_val_every_n_epoch = check_val_every_n_epoch or 1
        self.val_check_interval = val_check_interval
        self.dp_global_validation = False
        self.max_steps = max_steps
        self.limit_train_batches, self.limit_val_batches = limit_train_batches, limit_val_batches
        # conv stacks: bf16 storage / fp32 accumulate unless the card asks for `precision: 32` (fp32 maps on the exact-fp32
        # kernels: the parity mode, layers_f32.py); transformer stacks: exact fp32 unless the card asks for mixed precision
        # ('16-mixed' / 'bf16-mixed'), then bf16 GEMM operands + fused attention (transformer_mixed.py)
        self.precision = precision
        if precision is not None:
            from . import layers_f32, transformer
            transformer.set_precision(precision)
            layers_f32.set_conv_precision(precision)
        self.device = torch.device(device) if device is not None else torch.device('cuda', torch.cuda.current_device())
        self.current_epoch, self.global_step, self.should_stop = 0, 0, False
        self.optimizer = None
        self.callback_metrics = {}
        # [new] optional device-side step between the loader and the model: callable(batch_on_device, training) -> batch
        # (augment.DevicePipeline: random crop / flips / resize / noise on cached batches)
        self.batch_transform = batch_transform
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        self.rank = int(os.environ.get('RANK', 0))
        self._comm = None

    @property
    def is_global_zero(self):
        return self.rank == 0

    def log(self, metrics):
        if self._comm is not None:
            # epoch means of the per-rank losses -> mean over ranks: one number on every rank
            keys = sorted(k for k, v in metrics.items() if k.endswith('_loss') and isinstance(v, (int, float)))
            if keys:
                t = torch.tensor([float(metrics[k]) for k in keys], dtype=torch.float64,
                                 device=self.device if self._comm.overlaps else 'cpu')
                self._comm.all_reduce_sum(t)
                metrics = dict(metrics, **{k: float(v) / self.world for k, v in zip(keys, t.tolist())})
        self.callback_metrics.update(metrics)
        if self.logger is not None and self.is_global_zero:
            self.logger.log_metrics(metrics)

    def _to_device(self, batch, training=True):
        batch = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
        return self.batch_transform(batch, training) if self.batch_transform is not None else batch

    def _val_interval(self, train_loader):
        """Lightning's `val_check_interval` (scripts/train_multi.py:103 of the reference passes the card's value through):
        a float in (0, 1] = that fraction of a training epoch, an int = that many training batches; None / 1.0 = once at
        the end of the epoch.  -> number of training batches between validation runs, or None for end-of-epoch only."""
        v = self.val_check_interval
        if v is None or (isinstance(v, float) and v == 1.0):
            return None
        if isinstance(v, bool) or not isinstance(v, (int, float)):
            raise ValueError(f'val_check_interval must be an int or a float (got {v!r})')
        if isinstance(v, int):
            if v < 1:
                raise ValueError(f'val_check_interval (int) must be >= 1 (got {v})')
            n = len(train_loader) if hasattr(train_loader, '__len__') else None
            if n is not None and self.limit_train_batches is not None:
                n = min(n, int(self.limit_train_batches))
            if n is not None and v > n:
                raise ValueError(f'val_check_interval ({v}) must be less than or equal to the number of training batches ({n})')
            return v
        if not 0.0 < v <= 1.0:
            raise ValueError(f'val_check_interval (float) must be in (0, 1] (got {v})')
        if not hasattr(train_loader, '__len__'):
            raise ValueError('a fractional val_check_interval needs a training loader with a length')
        n = len(train_loader)
        if self.limit_train_batches is not None:
            n = min(n, int(self.limit_train_batches))
        return max(1, int(n * v))

    def _validate(self, model, valid_loader, stepper):
        model.eval()
        with torch.no_grad():
            for i, batch in enumerate(valid_loader):
                if self.limit_val_batches is not None and i >= self.limit_val_batches:
                    break
                batch = self._to_device(batch, training=False)
                if stepper is not None and self.dp_global_validation:
                    batch['buckets'] = 1              # the ranks' shards of a validation batch = one global bucket
                    stepper.validation_step(batch)
                else:
                    model.validation_step(batch, i)
        model.on_validation_epoch_end()
        for cb in self.callbacks:
            if self.is_global_zero or not isinstance(cb, ModelCheckpoint):     # checkpoints: rank 0 only
                cb.on_validation_end(self, model, self.callback_metrics)
        model.train()

    def fit(self, model, train_loader, valid_loader=None):
        model.trainer = self
        model.to(self.device)
        self.optimizer = model.configure_optimizers()
        stepper = None
        if self.world > 1:
            from . import distributed as D
            D.init(self.device)
            stepper = D.DataParallelStep(model, self.optimizer, self.world)
            self._comm = stepper.comm
            # replica equality must not rest on every rank having drawn the same initial values from the same seed
            D.broadcast_module(model, stepper.comm)
            # contrastive models validate on the GLOBAL batch (comparable with a single-GPU run at the same global batch);
            # anything else (classifier heads) validates per rank and the epoch means are averaged over the ranks (log())
            self.dp_global_validation = hasattr(model, 'loss') and hasattr(model, 'encode')
        every = self._val_interval(train_loader)
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            if hasattr(getattr(train_loader, 'sampler', None), 'set_epoch'):
                train_loader.sampler.set_epoch(epoch)       # DistributedSampler: a new shuffle, the same on every rank
            model.train()
            self.optimizer.zero_grad()
            pending = 0                                    # micro-batches accumulated since the last optimizer step
            n_batches = None
            if stepper is not None and self.accumulate > 1:
                # the data-parallel step must know which micro-batch closes a window BEFORE it runs it (the gradient
                # buckets cross the links during that backward): Lightning closes a short window on the epoch's last batch
                n_batches = len(train_loader)
                if self.limit_train_batches is not None:
                    n_batches = min(n_batches, int(self.limit_train_batches))
            validated_at_end = False
            for i, batch in enumerate(train_loader):
                if self.limit_train_batches is not None and i >= self.limit_train_batches:
                    break
                validated_at_end = False
                if stepper is not None:
                    batch = self._to_device(batch)
                    batch['buckets'] = 1                  # the GLOBAL batch is one contrastive bucket (distributed.py)
                    if self.accumulate == 1:
                        stepper.step(batch)
                        self.global_step += 1
                    else:
                        window = min(self.accumulate, n_batches - (i - pending))      # (the epoch's last window may be short)
                        stepper.step(batch, micro=pending, of=window)
                        pending += 1
                        if pending == window:
                            pending = 0
                            self.global_step += 1
                else:
                    loss = model.training_step(self._to_device(batch), i)
                    ops.backward(loss / self.accumulate if self.accumulate > 1 else loss)
                    pending += 1
                    if pending == self.accumulate:
                        self.optimizer.step()
                        self.optimizer.zero_grad()
                        pending = 0
                        self.global_step += 1
                if (every is not None and valid_loader is not None and (i + 1) % every == 0
                        and (epoch + 1) % self.check_val_every_n_epoch == 0):
                    self._validate(model, valid_loader, stepper)
                    validated_at_end = True
                    if self.should_stop and epoch + 1 >= self.min_epochs:
                        break
                if 0 < self.max_steps <= self.global_step:
                    self.should_stop = True
                    break
            if pending and stepper is None:
                # Lightning steps on the last batch of an epoch even when the accumulation window is not full
                self.optimizer.step()
                self.optimizer.zero_grad()
                self.global_step += 1
            model.on_train_epoch_end()
            if (valid_loader is not None and (epoch + 1) % self.check_val_every_n_epoch == 0 and every is None
                    and not validated_at_end):
                self._validate(model, valid_loader, stepper)
            if self.should_stop and epoch + 1 >= self.min_epochs:
                break
        return model
