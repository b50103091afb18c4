"""MultiModel / ImageModel / ProfileModel -- drop-in counterparts of /root/reference/src/model.py.

Same constructor arguments, attribute names, ``state_dict`` keys and step methods as the
reference's LightningModules (``lightning`` itself is absent here; ``trainer.Trainer`` drives the
same hooks).  All arithmetic below the module boundary runs on the gfx950 kernels.
"""
import os
from typing import Any, Callable, Dict

import torch
from torch import Tensor, nn

from . import ops
from .coordination import CLIPLoss, CLIPPlus, RankLoss, SigLIPLoss, SigLIPPlus
from .image_encoder import ImageEncoder
from .layers import linear
from .profile_encoder import ProfileCNN, ProfileLSTM, ProfileTransformer


class _BiasFreeLinear(nn.Linear):
    """nn.Linear(bias=False) parameters (same init, same key ``weight``), exact-fp32 MFMA GEMM forward."""

    def forward(self, input: Tensor) -> Tensor:
        return linear(input, self.weight, self.bias)


class _StepModule(nn.Module):
    """The slice of LightningModule the reference relies on."""

    def save_hyperparameters(self, **hp):
        self.hparams = dict(hp)

    def log_dict(self, metrics):
        self.logged = {k: (float(v) if torch.is_tensor(v) else v) for k, v in metrics.items()}
        if getattr(self, 'trainer', None) is not None:
            self.trainer.log(self.logged)

    @property
    def current_epoch(self):
        tr = getattr(self, 'trainer', None)
        return tr.current_epoch if tr is not None else 0

    def configure_optimizers(self):
        # optim.SGD(self.parameters(), **optim_args) -- src/model.py:147-148; one fused launch here
        return ops.FusedSGD(self.parameters(), **self.optim_args)


class MultiModel(_StepModule):
    """Reference: src/model.py:19-148."""

    def __init__(self, dim_embed, image_encoder_args: Dict[str, Any], profile_encoder_args: Dict[str, Any],
                 coordination_args: Dict[str, Any], optim_args: Dict[str, Any]) -> None:
        super().__init__()
        self.save_hyperparameters(dim_embed=dim_embed, image_encoder_args=image_encoder_args,
                                  profile_encoder_args=profile_encoder_args, coordination_args=coordination_args,
                                  optim_args=optim_args)
        self.image_encoder = ImageEncoder(**image_encoder_args)
        self.image_projection = _BiasFreeLinear(self.image_encoder.dim_out, dim_embed, bias=False)

        if 'num_head' in profile_encoder_args:                          # src/model.py:34-39
            self.profile_encoder = ProfileTransformer(**profile_encoder_args)
        elif 'blocks' in profile_encoder_args:
            self.profile_encoder = ProfileCNN(**profile_encoder_args)
        else:
            self.profile_encoder = ProfileLSTM(**profile_encoder_args)
        self.profile_projection = _BiasFreeLinear(self.profile_encoder.dim_out, dim_embed, bias=False)

        method = coordination_args.get('method')                       # src/model.py:44-56
        if method == 'clip':
            self.loss = CLIPLoss()
        elif method == 'siglip':
            self.loss = SigLIPLoss()
        elif method == 'clipplus':
            self.loss = CLIPPlus(beta=coordination_args.get('beta', .25))
        elif method == 'siglipplus':
            self.loss = SigLIPPlus(beta=coordination_args.get('beta', .25))
        elif method == 'rank':
            self.loss = RankLoss(margin=coordination_args.get('margin', 0.25))
        else:
            raise Exception("Coordination loss not found.")

        self.optim_args = optim_args
        self.train_loss = []
        self.valid_loss = []
        # overlap the profile branch with the image branch (see encode); MPR_TWO_STREAMS=0 serialises them
        # (per-kernel profiles are only attributable without the overlap)
        self.two_streams = os.environ.get('MPR_TWO_STREAMS', '1') != '0'
        self._side_stream = None

    def safe_forward(self, model: Callable, **kwargs):
        return model(**kwargs) if not any(v is None for v in kwargs.values()) else None

    def tokenize(self, profile: Tensor) -> Dict[str, Tensor]:
        return self.profile_encoder.tokenize(profile)

    def encode(self, image, profile, **kwargs) -> Dict[str, Tensor]:
        # every other batch key is forwarded to BOTH encoders (src/model.py:72-85)
        both = image is not None and profile is not None and getattr(image, 'is_cuda', False)
        if both and self.two_streams:
            # the two branches are independent until the loss: the (small, latency-bound) profile kernels run on
            # a second HIP stream underneath the image branch; autograd replays each branch's backward on the
            # stream its forward ran on, so the overlap carries over to the backward pass
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream()
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                profile_emb = self.safe_forward(self.profile_encoder, profile=profile, **kwargs)
                profile_emb = self.safe_forward(self.profile_projection, input=profile_emb)
            image_emb = self.safe_forward(self.image_encoder, image=image, **kwargs)
            image_emb = self.safe_forward(self.image_projection, input=image_emb)
            main.wait_stream(side)
            if profile_emb is not None:
                profile_emb.record_stream(main)
            return {'image_emb': image_emb, 'profile_emb': profile_emb}
        image_emb = self.safe_forward(self.image_encoder, image=image, **kwargs)
        profile_emb = self.safe_forward(self.profile_encoder, profile=profile, **kwargs)
        image_emb = self.safe_forward(self.image_projection, input=image_emb)
        profile_emb = self.safe_forward(self.profile_projection, input=profile_emb)
        return {'image_emb': image_emb, 'profile_emb': profile_emb}

    def forward(self, **kwargs):
        return self.encode(**kwargs)

    def training_step(self, batch: Dict[str, Tensor], batch_idx: int) -> Tensor:
        embeddings = self.encode(**batch)
        embeddings['buckets'] = batch['buckets']
        loss = self.loss(**embeddings)
        self.train_loss.append(loss.detach())
        return loss

    def on_train_epoch_end(self) -> None:
        loss = torch.stack(self.train_loss).mean()
        self.log_dict({'train_loss': loss, 'step': self.current_epoch})
        self.train_loss.clear()

    def validation_step(self, batch: Dict[str, Tensor], batch_idx: int):
        embeddings = self.encode(**batch)
        embeddings['buckets'] = batch['buckets']
        loss = self.loss(**embeddings)
        self.valid_loss.append(loss.detach())

    def on_validation_epoch_end(self) -> None:
        loss = torch.stack(self.valid_loss).mean()
        self.log_dict({'valid_loss': loss, 'step': self.current_epoch})
        self.valid_loss.clear()

    def predict_step(self, batch: Dict[str, Tensor], batch_idx: int, dataloader_idx: int = 0) -> Any:
        label = batch.get('label')
        embeddings = self.encode(**{k: v for k, v in batch.items() if k != 'label'})
        return embeddings | {'label': label} if label is not None else embeddings


class _CeFn(torch.autograd.Function):
    """CrossEntropyLoss (mean) + argmax in one kernel; src/model.py:167,197,227."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, argmax, dlogits = ops.softmax_ce(logits.contiguous(), labels.contiguous(), want_grad=True)
        ctx.save_for_backward(dlogits)
        ctx.mark_non_differentiable(argmax)
        return loss, argmax

    @staticmethod
    def backward(ctx, gout, _):
        (dlogits,) = ctx.saved_tensors
        return ops.scale_by_scalar(dlogits, gout.contiguous().float()), None


class _Classifier(_StepModule):
    """Shared body of ImageModel / ProfileModel (src/model.py:151-295, 298-451): encoder -> fc -> CE,
    argmax predictions.  Label strings map to ids by sorted order (LabelEncoder().fit, :170)."""

    def _init_head(self, encoder, class_names, optim_args):
        self.class_names = sorted(set(class_names))
        self.fc = nn.Linear(encoder.dim_out, len(self.class_names))
        self.optim_args = optim_args
        self.train_loss, self.valid_loss = [], []
        self.valid_pred, self.valid_true = [], []

    def _logits(self, x):
        return linear(x, self.fc.weight, self.fc.bias)

    def _label_ids(self, label, device):
        if torch.is_tensor(label):
            return label.to(device=device, dtype=torch.long)
        lut = {c: i for i, c in enumerate(self.class_names)}
        return torch.tensor([lut[l] for l in label], dtype=torch.long, device=device)

    def training_step(self, batch, batch_idx):
        logits = self.forward(**{k: v for k, v in batch.items() if k != 'label'})['logits']
        loss, _ = _CeFn.apply(logits, self._label_ids(batch['label'], logits.device))
        self.train_loss.append(loss.detach())
        return loss

    def validation_step(self, batch, batch_idx):
        logits = self.forward(**{k: v for k, v in batch.items() if k != 'label'})['logits']
        y = self._label_ids(batch['label'], logits.device)
        loss, pred = _CeFn.apply(logits, y)
        self.valid_loss.append(loss.detach())
        self.valid_pred.append(pred)
        self.valid_true.append(y)

    def on_train_epoch_end(self):
        self.log_dict({'train_loss': torch.stack(self.train_loss).mean(), 'step': self.current_epoch})
        self.train_loss.clear()

    def on_validation_epoch_end(self):
        pred, true = torch.cat(self.valid_pred), torch.cat(self.valid_true)
        acc = (pred == true).float().mean()
        self.log_dict({'valid_loss': torch.stack(self.valid_loss).mean(), 'valid_acc': acc,
                       'step': self.current_epoch})
        self.valid_loss.clear(); self.valid_pred.clear(); self.valid_true.clear()

    def predict_step(self, batch, batch_idx, dataloader_idx=0):
        logits = self.forward(**{k: v for k, v in batch.items() if k != 'label'})['logits']
        _, argmax, _ = ops.softmax_ce(logits.contiguous())
        return {'logits': logits, 'pred': argmax}


class ImageModel(_Classifier):
    """Reference: src/model.py:151-295."""

    def __init__(self, image_encoder_args: Dict[str, Any], optim_args: Dict[str, Any], class_names) -> None:
        super().__init__()
        self.save_hyperparameters(image_encoder_args=image_encoder_args, optim_args=optim_args,
                                  class_names=list(class_names))
        self.image_encoder = ImageEncoder(**image_encoder_args)
        self._init_head(self.image_encoder, class_names, optim_args)

    def forward(self, image, **kwargs):
        return {'logits': self._logits(self.image_encoder(image=image, **kwargs))}      # src/model.py:194-197


class ProfileModel(_Classifier):
    """Reference: src/model.py:298-451."""

    def __init__(self, profile_encoder_args: Dict[str, Any], optim_args: Dict[str, Any], class_names) -> None:
        super().__init__()
        self.save_hyperparameters(profile_encoder_args=profile_encoder_args, optim_args=optim_args,
                                  class_names=list(class_names))
        if 'num_head' in profile_encoder_args:
            self.profile_encoder = ProfileTransformer(**profile_encoder_args)
        elif 'blocks' in profile_encoder_args:
            self.profile_encoder = ProfileCNN(**profile_encoder_args)
        else:
            self.profile_encoder = ProfileLSTM(**profile_encoder_args)
        self._init_head(self.profile_encoder, class_names, optim_args)

    def forward(self, profile, **kwargs):
        return {'logits': self._logits(self.profile_encoder(profile=profile, **kwargs))}  # src/model.py:350-353
