"""Data-parallel train_multi over the 8 GPUs of one node: one process per GPU, RCCL over xGMI.

The reference is single-GPU; this is the build-side extension BASELINE config C4 asks
for.  Samples are sharded over ranks (rank r owns rows [r*b, (r+1)*b) of the global batch), encoders
and BatchNorm statistics are per rank, and the ONLY coupling is the global contrastive matrix:

  1. all-gather of the L2-normalised embeddings  (2 x [b, D] fp32 per rank, one collective)
  2. every rank walks only ITS row blocks  U_loc V_all^T  and  V_loc U_all^T  (b x n each, tile by tile on the fp32
     MFMA -- neither the n x n matrix nor the blocks are stored: csrc/loss_fused.hip), whose row log-sum-exps are local
  3. all-gather of the two LSE vectors (2 x [b] fp32) -- the softmax over the other axis needs them
  4. dL/dU_loc, dL/dV_loc are then computed locally against U_all / V_all (no reduce-scatter of
     embedding gradients); the loss value and d(logit_scale) are summed over ranks
  5. one SUM all-reduce of the flattened parameter gradients (the loss is already a mean over the
     GLOBAL batch, so per-rank contributions add).

`dp_clip` is written against a small "math" interface so that the exchange logic is exercised on CPU
with gloo in tests (tests inject a torch-based math object); the product default is the HIP one and
raises without the native library.
"""
import os

import torch
import torch.distributed as dist

from . import _native as N
from . import ops

F32 = torch.float32


# ------------------------------------------------------------------------------------------------ process group
def init(device=None, backend=None):
    if dist.is_initialized():
        return
    if backend is None:
        backend = os.environ.get('MPR_DIST_BACKEND')       # 'gloo': rehearsal of the N > 1 path on a box with fewer GPUs
    if backend is None:
        backend = 'nccl' if (device is not None and torch.device(device).type == 'cuda') else 'gloo'
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    kw = {}
    if backend == 'nccl' and device is not None:
        kw['device_id'] = torch.device(device)
    dist.init_process_group(backend=backend, rank=int(os.environ.get('RANK', 0)),
                            world_size=int(os.environ.get('WORLD_SIZE', 1)), **kw)


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()


def barrier():
    dist.barrier()


def max_over_ranks(value: float) -> float:
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class Comm:
    """The collectives the path needs.  `collective` selects how a slice of the gradient buffer is summed over the ranks:
    'all_reduce' (RCCL's own choice of algorithm) or 'rs_ag' -- an explicit reduce-scatter into this rank's 1/world share
    followed by an all-gather of the shares, both in place (SURVEY 8e "Collectives (4)": on point-to-point xGMI the two
    halves are the direct, per-link-bound form of the ring all-reduce; kept switchable, MPR_DP_COLLECTIVE, until a node
    run has priced one against the other)."""

    def __init__(self, group=None, collective=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.collective = collective or os.environ.get('MPR_DP_COLLECTIVE', 'all_reduce')
        if self.collective not in ('all_reduce', 'rs_ag'):
            raise ValueError(f"Comm: collective must be 'all_reduce' or 'rs_ag' (got {self.collective!r})")

    def _staged(self, x):
        # gloo rehearsal with device tensors (several ranks sharing one GPU): stage through the host
        return x.is_cuda and dist.get_backend(self.group) == 'gloo'

    def all_gather(self, x):
        """[...] -> [world, ...] (rank-major)."""
        x = x.contiguous()
        if self._staged(x):
            out = torch.empty(self.world * x.numel(), dtype=x.dtype)
            dist.all_gather_into_tensor(out, x.reshape(-1).cpu(), group=self.group)
            return out.to(x.device).view((self.world,) + tuple(x.shape))
        out = torch.empty(self.world * x.numel(), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.reshape(-1), group=self.group)
        return out.view((self.world,) + tuple(x.shape))

    def all_reduce_sum(self, x):
        if self._staged(x):
            h = x.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            x.copy_(h)
            return x
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)
        return x

    @property
    def overlaps(self):
        """Can a collective run beside compute (RCCL: its own stream)?  The gloo rehearsal is synchronous."""
        return dist.get_backend(self.group) == 'nccl'

    def all_reduce_sum_async(self, x):
        """Enqueue the all-reduce behind torch's CURRENT stream and return the work handle (work.wait() makes the then
        current stream wait for it)."""
        return dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def sum_grads(self, x, async_op=False):
        """Sum a contiguous 1-D slice of the gradient buffer over the ranks, in place, by the configured collective.
        -> list of work handles (async_op) or x."""
        works = []
        n = x.numel()
        m = n // self.world * self.world if self.collective == 'rs_ag' else 0
        if m and self.world > 1:
            if self._staged(x):
                h = x[:m].cpu()
                mine = h.view(self.world, -1)[self.rank].clone()
                dist.reduce_scatter_tensor(mine, h, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_gather_into_tensor(h, mine, group=self.group)
                x[:m].copy_(h)
            else:
                mine = x[:m].view(self.world, -1)[self.rank]          # this rank's share, reduced in place
                w1 = dist.reduce_scatter_tensor(mine, x[:m], op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
                w2 = dist.all_gather_into_tensor(x[:m], mine, group=self.group, async_op=async_op)
                works += [w1, w2] if async_op else []
        if m < n:                                                    # (rs_ag: the < world elements that do not divide)
            if async_op:
                works.append(self.all_reduce_sum_async(x[m:]))
            else:
                self.all_reduce_sum(x[m:])
        return works if async_op else x


# ------------------------------------------------------------------------------------------------ math back end
class HipClipMath:
    """The per-rank arithmetic of the sharded CLIP / SigLIP loss on the gfx950 kernels of csrc/loss_fused.hip (no
    fallback): the row blocks  U_loc V_all^T  and  V_loc U_all^T  are formed tile by tile on the fp32 MFMA and never stored.
    `gathered` is the all-gathered [world, 2, b, D] tensor of normalised embeddings (image, profile per rank)."""

    def _ws(self, gathered):
        world, _, b, D = gathered.shape
        return torch.empty(N.query('mpr_clipf_workspace_floats', world, b, D, 1), dtype=F32, device=gathered.device)

    def normalize(self, image_emb, profile_emb):
        """-> uv [2, b, D] (F.normalize of both modalities), inv [2, b] (1 / max(|x|, 1e-12))."""
        b, D = image_emb.shape
        uv = torch.empty(2, b, D, dtype=F32, device=image_emb.device)
        inv = torch.empty(2, b, dtype=F32, device=image_emb.device)
        N.call('mpr_clipf_norm', image_emb, profile_emb, uv, inv, b, D)
        return uv, inv

    def clip_fwd(self, gathered, logit_scale, rank, mul):
        """-> lse [2, b] (row log-sum-exps of my image rows / my profile rows), mul * sum (lse - positive logit)."""
        world, _, b, D = gathered.shape
        lse = torch.empty(2, b, dtype=F32, device=gathered.device)
        share = torch.empty((), dtype=F32, device=gathered.device)
        N.call('mpr_clipf_fwd', gathered, logit_scale.detach(), None, lse, share, float(mul), self._ws(gathered),
               world, rank, b, D, 1)
        return lse, share

    def clip_bwd(self, gathered, logit_scale, lse, lse_all, rank, coef, uv, inv, image_emb=None, profile_emb=None,
                 mse_coef=0.0):
        """-> dL/d image_emb, dL/d profile_emb [b, D], this rank's share of d logit_scale."""
        world, _, b, D = gathered.shape
        dev = gathered.device
        d_img, d_prof = torch.empty(b, D, dtype=F32, device=dev), torch.empty(b, D, dtype=F32, device=dev)
        dls = torch.empty((), dtype=F32, device=dev)
        N.call('mpr_clipf_bwd', gathered, logit_scale.detach(), None, lse, lse_all, float(coef), uv, inv, image_emb,
               profile_emb, float(mse_coef), None, d_img, d_prof, dls, None, self._ws(gathered), world, rank, b, D, 1)
        return d_img, d_prof, dls

    def siglip_fwd(self, gathered, logit_scale, bias, rank, mul):
        """-> mul * sum over my image rows x all profiles of -logsigmoid(+-z)."""
        world, _, b, D = gathered.shape
        share = torch.empty((), dtype=F32, device=gathered.device)
        N.call('mpr_clipf_fwd', gathered, logit_scale.detach(), bias.detach(), None, share, float(mul),
               self._ws(gathered), world, rank, b, D, 1)
        return share

    def siglip_bwd(self, gathered, logit_scale, bias, rank, coef, uv, inv, image_emb=None, profile_emb=None,
                   mse_coef=0.0):
        """-> dL/d image_emb, dL/d profile_emb, this rank's shares of d logit_scale and d bias."""
        world, _, b, D = gathered.shape
        dev = gathered.device
        d_img, d_prof = torch.empty(b, D, dtype=F32, device=dev), torch.empty(b, D, dtype=F32, device=dev)
        dls, db = torch.empty((), dtype=F32, device=dev), torch.empty((), dtype=F32, device=dev)
        N.call('mpr_clipf_bwd', gathered, logit_scale.detach(), bias.detach(), None, None, float(coef), uv, inv,
               image_emb, profile_emb, float(mse_coef), None, d_img, d_prof, dls, db, self._ws(gathered), world, rank,
               b, D, 1)
        return d_img, d_prof, dls, db

    def sqdiff_sum(self, a, b):
        out = torch.empty((), dtype=F32, device=a.device)
        ws = torch.empty(N.query('mpr_loss_workspace_floats'), dtype=F32, device=a.device)
        N.call('mpr_sqdiff_sum', a.contiguous(), b.contiguous(), out, ws, a.numel())
        return out


def _mse_term(a32, p32, beta, n, math):
    """beta * MSELoss(image_emb, profile_emb) over the GLOBAL batch (src/coordination.py:60-64,108-112):
    -> (this rank's share of the loss value, coefficient of (x - other) in the local embedding gradients)."""
    if not beta:
        return None, 0.0
    D = a32.shape[1]
    return math.sqdiff_sum(a32, p32) * (float(beta) / (n * D)), 2.0 * float(beta) / (n * D)


def _global_loss(comm, share, mse_share):
    """The loss VALUE (logging only -- no gradient depends on it): one scalar all-reduce of the ranks' shares, enqueued
    AFTER the gradient kernels so that its latency is not between the forward and backward halves of the loss stage."""
    local = share if mse_share is None else share + mse_share
    return comm.all_reduce_sum(local.reshape(1).clone()).reshape(())


def dp_siglip(image_emb, profile_emb, logit_scale, bias, comm, math, beta=0.0, grad_scale=1.0, want_grad=True):
    """Sharded SigLIP loss (src/coordination.py:76-95 over the GLOBAL batch, buckets = 1; + beta * MSE for SigLIPPlus).

    Every (image i, profile j) pair is owned by the rank that owns row i: the loss and the parameter gradients are sums of
    the row-block shares.  The profile gradients need column j against ALL images, so each rank also walks its column block
    V_loc U_all^T (the pair function is symmetric in its two roles) -- twice the tiny product instead of a reduce-scatter
    of [n, D] gradients.  Returns (loss, dL/d image_emb, dL/d profile_emb, d logit_scale share, d bias share)."""
    b = image_emb.shape[0]
    n = b * comm.world
    a32, p32 = image_emb.detach().float().contiguous(), profile_emb.detach().float().contiguous()
    uv, inv = math.normalize(a32, p32)
    gathered = comm.all_gather(uv)                                   # [world, 2, b, D]
    share = math.siglip_fwd(gathered, logit_scale, bias, comm.rank, 1.0 / n)
    mse, mse_coef = _mse_term(a32, p32, beta, n, math)
    if not want_grad:
        return _global_loss(comm, share, mse), None, None, None, None
    d_img, d_prof, dls, db = math.siglip_bwd(gathered, logit_scale, bias, comm.rank, grad_scale / n, uv, inv,
                                             a32 if beta else None, p32 if beta else None, mse_coef * grad_scale)
    return _global_loss(comm, share, mse), d_img, d_prof, dls, db


def dp_clip(image_emb, profile_emb, logit_scale, comm, math, beta=0.0, grad_scale=1.0, want_grad=True):
    """Sharded CLIP loss (src/coordination.py:26-47 over the GLOBAL batch, buckets = 1; + beta * MSE for CLIPPlus).

    Returns (loss [global value, identical on every rank], dL/d image_emb, dL/d profile_emb,
    d logit_scale [this rank's partial: the total is the SUM over ranks])."""
    b = image_emb.shape[0]
    n = b * comm.world
    a32, p32 = image_emb.detach().float().contiguous(), profile_emb.detach().float().contiguous()
    uv, inv = math.normalize(a32, p32)
    gathered = comm.all_gather(uv)                                   # [world, 2, b, D]
    # lse[0]: my images over all profiles, lse[1]: my profiles over all images; the softmax over the other axis of a
    # row block needs every rank's vector of the other role
    lse, share = math.clip_fwd(gathered, logit_scale, comm.rank, 1.0 / (2.0 * n))
    mse, mse_coef = _mse_term(a32, p32, beta, n, math)
    if not want_grad:                                                # validation: the value only
        return _global_loss(comm, share, mse), None, None, None
    lse_all = comm.all_gather(lse)                                   # [world, 2, b]
    # grad_scale: 1 / accumulate_grad_batches (the VALUE returned stays the unscaled global-batch loss)
    d_img, d_prof, dls = math.clip_bwd(gathered, logit_scale, lse, lse_all, comm.rank, grad_scale / (2.0 * n), uv, inv,
                                       a32 if beta else None, p32 if beta else None, mse_coef * grad_scale)
    return _global_loss(comm, share, mse), d_img, d_prof, dls


# ------------------------------------------------------------------------------------------------ gradient buckets
class GradBuckets:
    """Bucketed all-reduce of FusedSGD's flat gradient buffer, overlapped with the rest of the backward pass.

    The buffer is laid out in parameter order (image encoder: stem, layer1..4, then projections, profile encoder, loss);
    backward finishes it back to front.  Each bucket is a contiguous slice; the moment the last parameter of a bucket has
    had its gradient kernels ENQUEUED (ops.grad_ready_observers: the fused backward Functions report through
    ops.grad_target, autograd-accumulated gradients through the post-accumulate hook), the bucket's all-reduce is enqueued
    behind the streams that write it -- the weight-gradient side stream and, through an event, the backward stream -- on
    RCCL's own stream, so it runs underneath the remaining data-gradient chain.  ResNet-18's layer4 alone is 75 % of the
    gradient bytes and the FIRST thing backward finishes: its 33.6 MB cross xGMI while layers 3..1 are still computing.
    Every bucket remembers the compute streams its parameters reported on (the two encoders run on different streams, each
    with its own weight-gradient side stream) and its collective waits for exactly those.  Complete buckets without a later
    report (the branch enqueued last) are launched at the top of finish(), still asynchronously; buckets that are not
    complete when backward ends (parameters without a gradient this step) are reduced after every gradient stream has
    been joined.  Without overlap support
    (gloo rehearsal) everything happens in finish()."""

    def __init__(self, opt, comm, groups):
        """groups: lists of parameters, each list contiguous in the optimizer's flat buffer, in any order."""
        opt._install()
        self.opt, self.comm = opt, comm
        self.buckets = []                    # [lo, hi, n_params, ids]
        covered = set()
        for params in groups:
            params = [p for p in params if id(p) in opt.offsets]
            if not params:
                continue
            spans = sorted(opt.offsets[id(p)] for p in params)
            lo, hi = spans[0][0], (spans[-1][0] + spans[-1][1] + 3) // 4 * 4
            ids = {id(p) for p in params}
            inside = {i for i, (o, n) in opt.offsets.items() if lo <= o < hi}
            if inside != ids:
                raise ValueError('GradBuckets: a bucket must be a contiguous run of the optimizer\'s parameters')
            self.buckets.append([lo, hi, len(ids), ids])
            covered |= ids
        rest = sorted((o, n) for i, (o, n) in opt.offsets.items() if i not in covered)
        self.rest = self._runs(rest)
        self.owner = {i: k for k, b in enumerate(self.buckets) for i in b[3]}
        self.count, self.sent, self.works, self.pending, self.streams = [], [], [], [], []
        self._observer = self._on_ready
        self.active = False

    @staticmethod
    def _runs(spans):
        runs = []
        for o, n in spans:
            hi = (o + n + 3) // 4 * 4
            if runs and runs[-1][1] == o:
                runs[-1][1] = hi
            else:
                runs.append([o, hi])
        return runs

    def begin(self):
        self.count = [set() for _ in self.buckets]
        self.sent = [False] * len(self.buckets)
        self.streams = [{} for _ in self.buckets]      # per bucket: the compute streams its gradient kernels were enqueued on
        self.works, self.pending = [], []
        self.active = True
        if self._observer not in ops.grad_ready_observers:
            ops.grad_ready_observers.append(self._observer)

    def _on_ready(self, param):
        # A report means "this parameter's gradient kernels are ABOUT to be enqueued" (ops.grad_target hands out the
        # pointer first).  A bucket whose last parameter has reported is therefore launched at the next report from
        # OUTSIDE it: backward Functions run one after the other and each touches one bucket only, so by then the
        # Function that completed the bucket has returned and all its launches are in the queues.
        if not self.active:
            return
        k = self.owner.get(id(param))
        if self.pending and self.comm.overlaps:
            for j in [j for j in self.pending if j != k]:
                self._launch(j)
                self.pending.remove(j)
        if k is None:
            return
        if self.sent[k]:
            # a parameter that receives gradient from a second backward node (a shared / tied weight, a module called twice)
            # would be written while or after its slice is being reduced: ranks would apply different gradients, silently
            raise RuntimeError('GradBuckets: a gradient was reported for a parameter whose bucket has already been sent to '
                               'the all-reduce (a parameter used by more than one backward node?); set MPR_DP_BUCKETS=0 for '
                               'this model (one all-reduce after backward)')
        self.count[k].add(id(param))
        if self.comm.overlaps:
            cur = torch.cuda.current_stream()          # (autograd runs a node on the stream of its forward: the two encoders
            self.streams[k][cur.cuda_stream] = cur     #  differ, and each has its own weight-gradient side stream)
        if len(self.count[k]) == self.buckets[k][2] and k not in self.pending:
            self.pending.append(k)

    def _launch(self, k):
        lo, hi = self.buckets[k][0], self.buckets[k][1]
        srcs = list(self.streams[k].values()) or [torch.cuda.current_stream()]
        # The bucket's gradients were written on `srcs` (the backward streams of the Functions that own its parameters) and on
        # their weight-gradient side streams: the collective goes behind ALL of them, carried by one of the side streams
        # (which lags anyway) so that the data-gradient chain itself never waits for RCCL.
        carrier = ops.wgrad_side_stream_of(srcs[-1]) or srcs[-1]
        for s in srcs:
            if s is not carrier:
                carrier.wait_stream(s)
            sd = ops.wgrad_side_stream_of(s)
            if sd is not None and sd is not carrier:
                carrier.wait_stream(sd)
        with torch.cuda.stream(carrier):
            self.works += self.comm.sum_grads(self.opt.flat_grad[lo:hi], async_op=True)
        self.sent[k] = True

    def finish(self):
        """After backward: join the gradient streams, reduce what is left, wait for what is in flight."""
        self.active = False
        if self.comm.overlaps:
            # complete buckets nobody launched (the branch the host enqueued LAST -- the profile encoder: autograd walks the
            # image branch first -- has no later report to trigger it): still asynchronous, each behind its own streams only,
            # BEFORE the join below makes this stream wait for everything.  The profile branch is done on the GPU a few
            # milliseconds before the image chain: its gradients cross the links underneath that chain.
            for k, b in enumerate(self.buckets):
                if not self.sent[k] and len(self.count[k]) == b[2]:
                    self._launch(k)
        self.pending = []
        ops.join_gradient_streams()
        g = self.opt.flat_grad
        for k, b in enumerate(self.buckets):
            if not self.sent[k]:
                self.comm.sum_grads(g[b[0]:b[1]])
        for lo, hi in self.rest:
            self.comm.all_reduce_sum(g[lo:hi])
        for w in self.works:
            w.wait()
        self.works = []

    def close(self):
        if self._observer in ops.grad_ready_observers:
            ops.grad_ready_observers.remove(self._observer)


def default_bucket_groups(model):
    """Buckets in the order backward completes them: layer4 (+ the image projection, next to it in the buffer: 75 % of the
    gradient bytes, done first) | layer3 | the profile branch (its own stream: done ~a quarter into the image branch's
    backward) | layer2 | layer1 + stem -- only that last one, 0.65 MB at ResNet-18, is reduced after backward has ended.
    Whatever is left over (the loss's scalars) is reduced in finish()."""
    bb = getattr(getattr(model, 'image_encoder', None), 'backbone', None)
    if bb is None or not hasattr(bb, 'layer4'):
        return []
    groups = [list(bb.layer4.parameters()) + [p for p in getattr(model, 'image_projection', torch.nn.Module()).parameters()],
              list(bb.layer3.parameters())]
    prof = [p for n in ('profile_encoder', 'profile_projection') for p in getattr(model, n, torch.nn.Module()).parameters()]
    if prof:
        groups.append(prof)
    groups.append(list(bb.layer2.parameters()))
    groups.append([p for n, p in bb.named_parameters() if not n.startswith(('layer2.', 'layer3.', 'layer4.'))])
    return groups


# ------------------------------------------------------------------------------------------------ DP step
class DataParallelStep:
    """zero_grad -> encode (local) -> sharded CLIP / SigLIP (+ MSE) -> backward -> flat SUM all-reduce -> fused SGD."""

    def __init__(self, model, optimizer, world, comm=None, math=None):
        from .coordination import CLIPLoss, CLIPPlus, SigLIPLoss, SigLIPPlus
        loss = model.loss
        if isinstance(loss, (CLIPLoss, CLIPPlus)):
            self.kind, self.core = 'clip', (loss if isinstance(loss, CLIPLoss) else loss.clip)
        elif isinstance(loss, (SigLIPLoss, SigLIPPlus)):
            self.kind, self.core = 'siglip', (loss if isinstance(loss, SigLIPLoss) else loss.siglip)
        else:
            raise NotImplementedError('data-parallel step: clip / siglip (+ Plus) have sharded losses; "rank" cannot train '
                                      'in the reference either')
        self.beta = float(getattr(loss, 'beta', 0.0)) if isinstance(loss, (CLIPPlus, SigLIPPlus)) else 0.0
        self.model, self.opt = model, optimizer
        self.comm = comm or Comm()
        self.math = math or HipClipMath()
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.steps_done, self.verified = 0, []
        self.verify_steps = int(os.environ.get('MPR_DP_VERIFY_STEPS', '2'))
        self._flat = None
        self._views = None
        self.buckets = None
        if hasattr(optimizer, 'flat_grad') and os.environ.get('MPR_DP_BUCKETS', '1') != '0':
            try:
                self.buckets = GradBuckets(optimizer, self.comm, default_bucket_groups(model))
            except ValueError:       # a model whose parameter order splits a group: backbone stages only
                bb = model.image_encoder.backbone
                self.buckets = GradBuckets(optimizer, self.comm, [list(bb.layer4.parameters()), list(bb.layer3.parameters())])

    def _flat_views(self):
        if self._flat is None:
            sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]        # 16-byte aligned slots
            self._flat = torch.zeros(sum(sizes), dtype=F32, device=self.params[0].device)
            views, o = [], 0
            for p, s in zip(self.params, sizes):
                views.append(self._flat[o:o + p.numel()].view_as(p))
                o += s
            self._views = views
        return self._flat, self._views

    def _sharded_loss(self, emb, grad_scale=1.0, want_grad=True):
        core = self.core
        if self.kind == 'clip':
            loss, d_img, d_prof, dls = dp_clip(emb['image_emb'], emb['profile_emb'], core.logit_scale, self.comm,
                                               self.math, self.beta, grad_scale, want_grad)
            return loss, d_img, d_prof, [(core.logit_scale, dls)]
        loss, d_img, d_prof, dls, db = dp_siglip(emb['image_emb'], emb['profile_emb'], core.logit_scale, core.bias,
                                                 self.comm, self.math, self.beta, grad_scale, want_grad)
        return loss, d_img, d_prof, [(core.logit_scale, dls), (core.bias, db)]

    def validation_step(self, batch):
        """The GLOBAL-batch loss of one validation batch (the ranks' shards of it together are one contrastive bucket, as in
        the training step): the same number on every rank, comparable with a single-GPU run at the same global batch.
        Call under torch.no_grad() with the model in eval mode."""
        if batch.get('buckets', 1) != 1:
            raise NotImplementedError('data-parallel validation: buckets must be 1 (the global batch is one bucket)')
        emb = self.model.encode(**batch)
        loss = self._sharded_loss(emb, want_grad=False)[0]
        self.model.valid_loss.append(loss.detach())
        return loss

    def step(self, batch, micro=0, of=1):
        """One micro-batch of an optimisation step; `of` = accumulate_grad_batches (Lightning semantics,
        scripts/train_multi.py:99-104 of the reference: every micro-batch's loss is divided by `of`, the optimizer steps
        after the last one).  Every micro-batch's loss is the GLOBAL contrastive loss over the ranks' shards of it (two
        small all-gathers each); the gradient buckets cross the links once, during the LAST micro-batch's backward."""
        model = self.model
        first, last = micro == 0, micro == of - 1
        if first:
            self.opt.zero_grad()
        if batch.get('buckets', 1) != 1:
            raise NotImplementedError('data-parallel step: buckets must be 1 (the global batch is one bucket)')
        if self.buckets is not None and last:
            self.buckets.begin()
        emb = model.encode(**batch)
        loss, d_img, d_prof, pgrads = self._sharded_loss(emb, grad_scale=1.0 / of)
        torch.autograd.backward([emb['image_emb'], emb['profile_emb']], [d_img, d_prof])
        arena = getattr(self.opt, 'flat_grad', None)
        fused = arena is not None and all(getattr(p, '_mpr_grad', None) is not None for p, _ in pgrads)
        if fused:
            # FusedSGD: every gradient already sits in the optimizer's flat buffer (the fused backward Functions
            # accumulate into it) -- ONE all-reduce of that buffer, no gather copies
            for p, g in pgrads:
                if p.grad is not p._mpr_grad:
                    p.grad = p._mpr_grad
                p.grad.add_(g.reshape(p.shape))
                p._mpr_touched = True
        else:
            for p, g in pgrads:
                p.grad = g.reshape(p.shape) if p.grad is None or first else p.grad + g.reshape(p.shape)
        model.train_loss.append(loss.detach())
        if not last:
            return loss
        if fused:
            if self.buckets is not None:
                self.buckets.finish()          # (layer4 / layer3 / ... went out during backward; the rest goes now)
            else:
                ops.join_gradient_streams()
                self.comm.sum_grads(arena)
        else:
            flat, views = self._flat_views()
            # a parameter that received no gradient keeps grad None (the optimizer then skips it, weight decay included,
            # as in the single-process step); the graph is the same on every rank, so "has a gradient" is too
            has = [p.grad is not None for p in self.params]
            flat.zero_()
            live = [(v, p.grad) for v, p, h in zip(views, self.params, has) if h]
            if live:
                torch._foreach_copy_([v for v, _ in live], [g for _, g in live])
            self.comm.sum_grads(flat)
            for p, v, h in zip(self.params, views, has):
                p.grad = v if h else None
        self.opt.step()
        self.steps_done += 1
        if self.steps_done <= self.verify_steps:
            self.verify_replicas()
        return loss

    # ---- replicas must stay bit-identical: checked on the hardware in the first steps -------------------------------
    def _replica_fingerprint(self):
        parts = [p.detach().double().sum() for p in self.params] + [p.detach().double().abs().sum() for p in self.params]
        return torch.stack(parts)

    def verify_replicas(self):
        """After an optimizer step every rank must hold bit-identical parameters (same initial values, same summed
        gradients).  A gradient written after its bucket went to the all-reduce, or a collective that ran ahead of a
        producer, shows up as ranks that differ: compared here (MIN and MAX over the ranks of per-parameter sums, two tiny
        all-reduces) in the first `verify_steps` steps of a run -- the overlap bookkeeping is then verified on the actual
        machine, under RCCL, before training relies on it.  On a mismatch the bucketed overlap is switched off (one
        reduction after backward), the replicas are re-synchronised from rank 0 and a warning is printed; it never passes
        silently."""
        if self.comm.world < 2:
            return True
        fp = self._replica_fingerprint()
        if self.comm._staged(fp):
            fp = fp.cpu()
        lo, hi = fp.clone(), fp.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.comm.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.comm.group)
        same = bool(torch.equal(lo, hi))
        self.verified.append(same)
        if not same:
            import warnings
            bad = int((lo != hi).sum())
            warnings.warn(f'data-parallel step {self.steps_done}: replicas differ after the optimizer step ({bad} parameter '
                          f'sums): switching the bucketed gradient overlap OFF and re-synchronising from rank 0')
            if self.buckets is not None:
                self.buckets.close()
                self.buckets = None
            broadcast_module(self.model, self.comm)
            self.verify_steps = self.steps_done + 2
        return same


def broadcast_module(model, comm=None, src=0):
    """Every parameter and buffer of `model` := rank `src`'s (replica equality then does not rest on the seed alone)."""
    group = comm.group if comm is not None else None
    staged = comm._staged if comm is not None else (lambda t: t.is_cuda and dist.get_backend(group) == 'gloo')
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            if staged(t):
                h = t.detach().cpu()
                dist.broadcast(h, src=src, group=group)
                t.copy_(h)
            else:
                dist.broadcast(t.detach(), src=src, group=group)
    if hasattr(ops, 'pack_registry'):
        for p in model.parameters():               # bf16 filter panels are caches of the fp32 masters: rebuild on next use
            if hasattr(p, '_mpr_packed'):
                del p._mpr_packed
