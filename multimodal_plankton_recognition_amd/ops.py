"""Thin tensor-level wrappers over the C-ABI kernels (libmpr_hip.so).

Everything here takes / returns device tensors and enqueues work on torch's current HIP stream.
Feature maps are channels-last bf16: ``[B, H, W, C]`` (2-D) or ``[B, L, C]`` (1-D == H of 1).
PyTorch supplies memory (caching allocator) and streams only -- no torch arithmetic runs here.
"""
import os

import torch

from . import _native as N

BF16 = torch.bfloat16
F32 = torch.float32


def _geom(x):
    """[B,L,C] | [B,H,W,C] -> (B, H, W, C)."""
    if x.dim() == 3:
        return x.shape[0], 1, x.shape[1], x.shape[2]
    return tuple(x.shape)


def _like_spatial(x, B, P, Q, K):
    return (B, Q, K) if x.dim() == 3 else (B, P, Q, K)


class ConvGeom:
    """Static description of one convolution (torch OIHW / OIW weight)."""

    def __init__(self, weight_shape, stride=1, pad=0):
        if len(weight_shape) == 2:       # nn.Linear [K, C] == a 1x1 convolution over the token axis
            self.K, self.C = weight_shape
            self.R = self.S = self.sh = self.sw = 1
            self.ph = self.pw = 0
        elif len(weight_shape) == 3:     # Conv1d [K, C, S]
            self.K, self.C, self.S = weight_shape
            self.R, self.sh, self.sw, self.ph, self.pw = 1, 1, stride, 0, pad
        else:
            self.K, self.C, self.R, self.S = weight_shape
            self.sh = self.sw = stride
            self.ph = self.pw = pad

    def out_hw(self, H, W):
        return (H + 2 * self.ph - self.R) // self.sh + 1, (W + 2 * self.pw - self.S) // self.sw + 1

    @property
    def tail(self):
        return (self.R, self.S, self.sh, self.sw, self.ph, self.pw)


# ------------------------------------------------------------------------------------------ weights
_pack_cache = {}


def _kcrs_strides(weight):
    """Element strides (sk, sc, sr, ss) of a conv weight viewed as [K, C, R, S] (Conv1d: R == 1)."""
    st = weight.stride()
    if weight.dim() == 2:
        return (st[0], st[1], 0, 0)
    return (st[0], st[1], 0, st[2]) if weight.dim() == 3 else tuple(st)


def is_krsc(weight):
    """True when the weight's MEMORY is [K][R][S][C] (channels-last): what conv_wgrad produces natively."""
    if weight.dim() == 2:
        return weight.is_contiguous()
    if weight.dim() == 3:
        K, C, S = weight.shape
        return tuple(weight.stride()) == (S * C, 1, C)
    K, C, R, S = weight.shape
    return tuple(weight.stride()) == (R * S * C, 1, S * C, C)


def to_krsc_(module):
    """Re-lay every Conv1d/Conv2d weight under `module` as [K][R][S][C] in memory (logical shape, values and
    state_dict unchanged): the weight gradient then needs no permute and the bf16 panels are a plain cast."""
    for m in module.modules():
        if isinstance(m, (torch.nn.Conv1d, torch.nn.Conv2d)) and m.weight.shape[1] % 8 == 0:
            w = m.weight.data
            perm = (0, 2, 1) if w.dim() == 3 else (0, 2, 3, 1)
            inv = (0, 2, 1) if w.dim() == 3 else (0, 3, 1, 2)
            m.weight.data = w.permute(*perm).contiguous().permute(*inv)
    return module


class _PackRegistry:
    """All filters that have bf16 panels: lets the optimizer refresh every panel in ONE launch after its step."""

    def __init__(self):
        self.entries = []        # [weakref(weight), data_ptr, wf, wd, geom]
        self.table = None
        # The refresh runs on its own stream behind the optimizer step; a stream that is about to READ panels waits for it
        # once (packed_weights -> ready()).  The ResNet stem reads no panel (the fused stem builds its own from the fp32
        # filter), so the 55-us refresh sits underneath the stem's first kernels instead of in front of them.
        self.stream = None
        self.event = None
        self.gen = 0
        self.seen = {}           # raw stream handle -> generation it has waited for

    def ready(self, dev):
        if self.event is None:
            return
        h = torch._C._cuda_getCurrentRawStream(dev)
        if self.seen.get(h) != self.gen:
            torch.cuda.current_stream().wait_event(self.event)
            self.seen[h] = self.gen

    def add(self, weight, wf, wd, geom):
        import weakref
        self.entries = [e for e in self.entries if e[0]() is not None and e[0]() is not weight]
        self.entries.append([weakref.ref(weight), weight.data_ptr(), wf, wd, geom])
        self.table = None

    def repack_all(self):
        live = [e for e in self.entries if e[0]() is not None and e[0]().data_ptr() == e[1]]
        if len(live) != len(self.entries):
            self.entries, self.table = live, None
        if not live:
            return
        if self.table is None:
            rows = [[e[1], e[2].data_ptr(), e[3].data_ptr() if e[3] is not None else 0, e[4].K, e[4].C, e[4].R, e[4].S,
                     *_kcrs_strides(e[0]()), 0] for e in live]
            self.table = torch.tensor(rows, dtype=torch.int64).pin_memory().to(live[0][2].device, non_blocking=True)
        if OVERLAP_REPACK:
            cur = torch.cuda.current_stream()
            if self.stream is None:
                self.stream = torch.cuda.Stream()
                self.event = torch.cuda.Event()
            self.stream.wait_stream(cur)
            with torch.cuda.stream(self.stream):
                N.call('mpr_conv_pack_weights_multi', self.table, len(live))
                self.event.record(self.stream)
            self.gen += 1
            # the panels and the table were allocated on other streams: tell the allocator that this one uses them too (once
            # per tensor), or memory freed with a dying model could be handed out again while the refresh still writes it
            if getattr(self.table, '_mpr_on_pack_stream', None) is not self.stream:
                self.table.record_stream(self.stream)
                self.table._mpr_on_pack_stream = self.stream
            for e in live:
                for t in (e[2], e[3]):
                    if t is not None and getattr(t, '_mpr_on_pack_stream', None) is not self.stream:
                        t.record_stream(self.stream)
                        t._mpr_on_pack_stream = self.stream
        else:
            N.call('mpr_conv_pack_weights_multi', self.table, len(live))
        for e in live:
            w = e[0]()
            w._mpr_packed = ((w.data_ptr(), w._version), e[2], e[3])


OVERLAP_REPACK = os.environ.get('MPR_OVERLAP_REPACK', '1') != '0'
pack_registry = _PackRegistry()


def packed_weights(weight, geom, need_dgrad=True):
    """bf16 GEMM panels of an fp32 conv weight (any dense layout), cached on (storage, version)."""
    # the cache entry lives ON the tensor object (dies with it: no id()/address aliasing between tensors)
    key = (weight.data_ptr(), weight._version)
    hit = getattr(weight, '_mpr_packed', None)
    pack_registry.ready(weight.device.index)
    if hit is not None and hit[0] == key and (hit[2] is not None or not need_dgrad):
        return hit[1], hit[2]
    nf = (geom.K + 127) // 128 * 128 * ((geom.R * geom.S * geom.C + 63) // 64 * 64)
    nd = (geom.C + 127) // 128 * 128 * ((geom.R * geom.S * geom.K + 63) // 64 * 64)
    # keep the panel buffers of an earlier pack (stable pointers for the registry's table)
    wf = hit[1] if hit is not None and hit[0][0] == key[0] else torch.empty(nf, dtype=BF16, device=weight.device)
    wd = hit[2] if hit is not None and hit[0][0] == key[0] and hit[2] is not None else (
        torch.empty(nd, dtype=BF16, device=weight.device) if need_dgrad else None)
    N.call('mpr_conv_pack_weights_strided', weight.detach(), *_kcrs_strides(weight), wf, wd, geom.K, geom.C, geom.R,
           geom.S)
    if hit is None or hit[1] is not wf or hit[2] is not wd:
        pack_registry.add(weight, wf, wd, geom)
    weight._mpr_packed = (key, wf, wd)
    return wf, wd


# Observers of "the gradient of this parameter has been enqueued" (data-parallel training launches the all-reduce of a
# bucket of the flat gradient buffer as soon as its last parameter reports in: distributed.GradBuckets).  Called with the
# parameter, on the host, from inside the backward pass.
grad_ready_observers = []


def grad_target(param):
    """The optimizer-owned gradient memory of `param` (FusedSGD: a view of its flat buffer with the parameter's
    strides, zeroed by zero_grad) or None.  A backward that finds one ACCUMULATES into it and hands autograd None."""
    tgt = getattr(param, '_mpr_grad', None) if torch.is_tensor(param) else None
    if tgt is not None:
        param._mpr_touched = True
        _note_arena_stream(tgt.device.index)
        for obs in grad_ready_observers:
            obs(param)
    return tgt


# ---- streams that write gradient memory behind autograd's back -------------------------------------------
# A Function that accumulates into grad_target() returns None, so no AccumulateGrad node (and none of the engine's
# end-of-backward stream synchronisation) covers that write.  Every stream that did such a write -- the stream of
# the backward node, or the weight-gradient side stream below -- is joined into the caller's stream by ONE
# end-of-backward callback.
ASYNC_WGRAD = os.environ.get('MPR_ASYNC_WGRAD', '1') != '0'
_arena_streams = {}        # raw stream handle -> torch stream object
_wgrad_side = {}           # raw handle of a compute stream -> its weight-gradient side stream
_join_queued = [False]
KEEPALIVE_WGRAD_OPERANDS = os.environ.get('MPR_WGRAD_KEEPALIVE', '1') != '0'
_wgrad_keepalive = []      # operands of weight gradients still queued on a side stream (released by the join)


def _note_arena_stream(dev, stream=None):
    h = stream.cuda_stream if stream is not None else torch._C._cuda_getCurrentRawStream(dev)
    if h not in _arena_streams:
        _arena_streams[h] = stream if stream is not None else torch.cuda.current_stream()
    if not _join_queued[0]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_gradient_streams)
            _join_queued[0] = True
        except RuntimeError:        # not inside a backward pass: the caller reads the result on its own stream
            pass


def join_gradient_streams():
    """Make torch's current stream wait for every stream that wrote optimizer-owned gradient memory."""
    _join_queued[0] = False
    cur = torch.cuda.current_stream()
    for h, s in _arena_streams.items():
        if h != cur.cuda_stream:
            cur.wait_stream(s)
    _wgrad_keepalive.clear()


def _wgrad_stream(dev):
    """Side stream for weight gradients: they are leaves of the backward graph (nothing downstream reads them
    before the optimizer), so they run beside the dgrad -> BatchNorm-backward chain -- MFMA-bound work filling the
    tails and the HBM-bound stretches of that chain."""
    h = torch._C._cuda_getCurrentRawStream(dev)
    side = _wgrad_side.get(h)
    if side is None:
        # lowest priority: the side stream must only FILL what the dependent chain leaves free -- at equal priority its
        # 200-us kernels hold every CU while the chain's 5-us BatchNorm finalize kernels queue behind them
        try:
            low = int(os.environ.get('MPR_WGRAD_PRIORITY', torch.cuda.Stream.priority_range()[0]))
        except Exception:
            low = 0
        side = _wgrad_side[h] = (torch.cuda.current_stream(), torch.cuda.Stream(priority=low))
    return side


def wgrad_side_stream_of(stream):
    """The weight-gradient side stream attached to `stream` (None when none has been created)."""
    pair = _wgrad_side.get(stream.cuda_stream)
    return pair[1] if pair is not None else None


# ------------------------------------------------------------------------------------------ conv
def conv_fwd(x, wf, g, want_stats):
    B, H, W, C = _geom(x)
    assert C == g.C, f'conv_fwd: input has {C} channels, weight expects {g.C}'
    P, Q = g.out_hw(H, W)
    y = torch.empty(_like_spatial(x, B, P, Q, g.K), dtype=BF16, device=x.device)
    stats = None
    if want_stats:
        rows = N.query('mpr_conv_fwd_stat_rows', B, P, Q, g.K, C, *g.tail)
        zeroed = False
        if rows == FIN_SLICES and N.query('mpr_conv_set_stat_slices', -1) == FIN_SLICES:     # slice rows (atomics)
            stats, zeroed = _slice_rows(g.K, x.device)
        else:
            stats = torch.empty(rows, 2, g.K, dtype=F32, device=x.device)
        if zeroed:
            N.query('mpr_conv_stats_prezeroed', 1)
    N.call('mpr_conv_fwd', x, wf, y, stats, B, H, W, C, g.K, *g.tail)
    return y, stats


def conv_dgrad(dy, wd, g, x_shape, add=None):
    dx = torch.empty(x_shape, dtype=BF16, device=dy.device)
    B, H, W, C = _geom(dx)
    N.call('mpr_conv_dgrad', dy, wd, dx, add, B, H, W, C, g.K, *g.tail)
    return dx


S2_BN_FUSION = os.environ.get('MPR_S2_BN_FUSION', '1') != '0'      # (A/B: the downsampling blocks' share of the BatchNorm-backward fusion)


def conv_dgrad_shortcut(dy, wd, g, x_shape, dy_sc, wd_sc, g_sc, note=None, mask_y=None):
    """Block-input gradient of a downsampling BasicBlock: data gradient of the stride-2 3x3 conv (dy, wd, g) + data gradient
    of the parallel 1x1 / stride-2 shortcut conv (dy_sc, wd_sc, g_sc).  The shortcut's gradient only reaches the even
    pixels: it is formed on the half-resolution grid (a plain GEMM, a quarter of the rows) and added there by the
    parity-class kernel, instead of a full-resolution 3/4-zero map written and read back.
    `note` = (bn_x, BNState) of the BatchNorm that produced the block input (the previous block's bn2; mask_y = that
    block's output = this block's input): the epilogue also applies the ReLU mask and leaves bn2's backward sums.
    -> (dx, slices or None), or None when the geometry is not served."""
    if len(x_shape) != 4 or (g_sc.R, g_sc.S, g_sc.sh, g_sc.sw, g_sc.ph, g_sc.pw) != (1, 1, 2, 2, 0, 0):
        return None
    B, H, W, C = x_shape
    if not N.query('mpr_conv_dgrad_add_even_supported', B, H, W, C, g.K, *g.tail) or tuple(dy_sc.shape[1:3]) != (H // 2, W // 2):
        return None
    half = torch.empty((B, H // 2, W // 2, C), dtype=BF16, device=dy.device)
    N.call('mpr_conv_dgrad', dy_sc, wd_sc, half, None, B, H // 2, W // 2, C, g_sc.K, 1, 1, 1, 1, 0, 0)
    dx = torch.empty(x_shape, dtype=BF16, device=dy.device)
    if note is not None and DGRAD_BN_FUSION and S2_BN_FUSION and note[1].mean is not None and C <= 512:
        slices, zeroed = _slice_rows(C, dy.device)
        N.call('mpr_conv_dgrad_s2_bn', dy, wd, dx, half, mask_y, note[0], note[1].mean, note[1].invstd, slices, FIN_SLICES,
               int(zeroed), B, H, W, C, g.K, *g.tail)
        return dx, slices
    N.call('mpr_conv_dgrad_s2', dy, wd, dx, half, B, H, W, C, g.K, *g.tail)
    return dx, None


AUTOTUNE = os.environ.get('MPR_AUTOTUNE', '1') != '0'
_FIXED_WGRAD_WGS = int(os.environ.get('MPR_WGRAD_WGS', '0'))      # experiments: one target for every geometry
_wgrad_split = {}          # geometry -> (window kernel?, workgroup-count target of the split over pixels)
# MPR_WGRAD_PLAN=<file>: the tuner's choices are loaded from / appended to this JSON file, so that a run whose timings
# cannot be trusted (rocprofv3 --pmc serialises kernels and the tuner then prefers other kernels than the real step
# runs) executes the plan of an undisturbed run: scripts/prof_pmc.sh, scripts/prof_mfma.sh
_WGRAD_PLAN = os.environ.get('MPR_WGRAD_PLAN')
if _WGRAD_PLAN and os.path.exists(_WGRAD_PLAN):
    import json as _json
    with open(_WGRAD_PLAN) as _f:
        _wgrad_split.update({tuple(int(v) for v in k.split(',')): tuple(c) for k, c in _json.load(_f).items()})


WGRAD_SCRATCH_FLOATS = 20 * 1024 * 1024      # 80 MB per stream: 256 workgroups x 128 x 576 accumulators (+ margin)
_wgrad_scratch = {}                          # raw stream handle -> scratch tensor


def _wgrad_call(x, dy, out, dw, accumulate, B, H, W, C, K, R, S, sh, sw, ph, pw, stream_handle=None, choice=None):
    """mpr_conv_wgrad_ex: every per-launch choice travels as an argument (nothing process-global is armed "for the next
    call").  `choice` = (kernel, workgroup target) of the tuner, None = the library's defaults.  The sliding-window kernel is
    lent per-stream scratch for its partial tiles (plain stores + one reduction pass instead of 75 MB of fp32 atomics per
    launch): consecutive launches on one stream reuse the buffer in stream order, different streams have different buffers."""
    buf = None
    if (R, S, sh, sw, ph, pw) == (3, 3, 1, 1, 1, 1) and C % 64 == 0 and K % 64 == 0 and USE_WGRAD_SCRATCH:
        h = stream_handle if stream_handle is not None else N.stream(x.device.index)
        buf = _wgrad_scratch.get(h)
        if buf is None:
            buf = _wgrad_scratch[h] = torch.empty(WGRAD_SCRATCH_FLOATS, dtype=F32, device=x.device)
    kernel, target = (-1, 0) if choice is None else (int(choice[0]), int(choice[1]))
    N.call('mpr_conv_wgrad_ex', x, dy, out, dw, accumulate, B, H, W, C, K, R, S, sh, sw, ph, pw, buf,
           buf.numel() if buf is not None else 0, target, kernel, stream_handle=stream_handle)


USE_WGRAD_SCRATCH = os.environ.get('MPR_WGRAD_SCRATCH', '1') != '0'


def _tune_wgrad(x, dy, g, key):
    """First use of a large geometry: time the weight-gradient kernels -- the sliding-window kernel (3x3 stride 1 only)
    and the gather kernel, each at a few workgroup-count targets of the split over pixels -- on scratch output, and keep
    the candidate with the smallest CU-TIME: time alone x the share of the chip its workgroups hold (one window workgroup
    or two gather workgroups fill a CU: their registers leave no room for anything else on it).  Inside the step the weight
    gradients run beside the data-gradient chain, which takes whatever CUs they leave: a launch on 160 CUs that runs 30 %
    longer costs the step LESS than one that holds all 256 (in-process A/B on C3: fixed targets of 160 / 192 workgroups
    10.37 / 10.40 ms per step, 256: 10.55, the alone-fastest choice per shape: 10.66)."""
    B, H, W, C = _geom(x)
    ws = torch.empty(g.K * g.R * g.S * g.C, dtype=F32, device=x.device)
    win_ok = (g.R, g.S, g.sh, g.sw, g.ph, g.pw) == (3, 3, 1, 1, 1, 1)
    # (no candidate below ~5/8 of the chip: the score keeps falling with the workgroup count, but the side stream must not
    #  become the critical path -- 64 workgroups everywhere: 12.1 ms per step)
    cands = ([(1, 160), (1, 192), (1, 256), (1, 512)] if win_ok else []) + \
            [(0, 256), (0, 384), (0, 512), (0, 768), (0, 1024)]
    best, best_s = (0, 768), None
    for win, tg in cands:
        _wgrad_call(x, dy, ws, None, 0, B, H, W, C, g.K, *g.tail, choice=(win, tg))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            _wgrad_call(x, dy, ws, None, 1, B, H, W, C, g.K, *g.tail, choice=(win, tg))
        b.record()
        b.synchronize()
        slots = 256 if win else 512
        score = a.elapsed_time(b) * min(tg, slots) / slots
        if best_s is None or score < best_s:
            best, best_s = (win, tg), score
    _wgrad_split[key] = best
    if _WGRAD_PLAN:
        import json
        with open(_WGRAD_PLAN, 'w') as f:
            json.dump({','.join(str(int(v)) for v in k): list(c) for k, c in _wgrad_split.items()}, f, indent=0)
    return best


def conv_wgrad(x, dy, g, weight):
    """Gradient of `weight` (a tensor: its layout and, if the optimizer installed one, its gradient memory are
    used; or a plain shape -> contiguous OIHW result).  Returns None when it accumulated into grad_target(weight)."""
    B, H, W, C = _geom(x)
    choice = None                                      # (kernel, workgroup target) of THIS launch: an argument, not a setter
    if _FIXED_WGRAD_WGS:
        choice = (-1, _FIXED_WGRAD_WGS)
    elif AUTOTUNE and C % 64 == 0 and g.K % 64 == 0 and dy.numel() // g.K >= 16384:
        key = (B, H, W, C, g.K, *g.tail)
        choice = _wgrad_split.get(key)
        if choice is None:
            choice = _tune_wgrad(x, dy, g, key)
    return _conv_wgrad(x, dy, g, weight, B, H, W, C, choice)


def _conv_wgrad(x, dy, g, weight, B, H, W, C, choice=None):
    if not torch.is_tensor(weight):
        ws = torch.empty(g.K * g.R * g.S * g.C, dtype=F32, device=x.device)
        dw = torch.empty(weight, dtype=F32, device=x.device)
        _wgrad_call(x, dy, ws, dw, 0, B, H, W, C, g.K, *g.tail, choice=choice)
        return dw
    tgt = grad_target(weight)
    if is_krsc(weight):          # the kernel's native [K][R][S][C] result IS the gradient's memory
        if tgt is not None:
            if ASYNC_WGRAD:
                cur, side = _wgrad_stream(x.device.index)
                side.wait_stream(cur)
                _wgrad_call(x, dy, tgt, None, 1, B, H, W, C, g.K, *g.tail, stream_handle=side.cuda_stream, choice=choice)
                _note_arena_stream(x.device.index, side)
                # the allocator must not recycle the operands under the side stream.  Tensor.record_stream would do, but a
                # block freed with a pending foreign-stream use sits in limbo until that (low-priority, lagging) stream
                # passes it, and the allocator hands out fresh memory meanwhile: 52 GB reserved for a 6 GB working set,
                # growing slowly.  Holding the two tensors until the end-of-backward join (every later user of their
                # memory is then ordered behind the side stream) costs ~2.7 GB of gradients kept a little longer:
                # 7.8 GB reserved.  (Outside a backward pass no join is queued: record_stream.)
                if KEEPALIVE_WGRAD_OPERANDS and _join_queued[0]:
                    _wgrad_keepalive.append((x, dy))
                else:
                    x.record_stream(side)
                    dy.record_stream(side)
            else:
                _wgrad_call(x, dy, tgt, None, 1, B, H, W, C, g.K, *g.tail, choice=choice)
            return None
        dw = torch.empty_strided(weight.shape, weight.stride(), dtype=F32, device=x.device)
        _wgrad_call(x, dy, dw, None, 0, B, H, W, C, g.K, *g.tail, choice=choice)
        return dw
    ws = torch.empty(g.K * g.R * g.S * g.C, dtype=F32, device=x.device)
    if tgt is not None and tgt.is_contiguous():
        _wgrad_call(x, dy, ws, tgt, 1, B, H, W, C, g.K, *g.tail, choice=choice)
        return None
    dw = torch.empty(weight.shape, dtype=F32, device=x.device)
    _wgrad_call(x, dy, ws, dw, 0, B, H, W, C, g.K, *g.tail, choice=choice)
    return dw


def stem_fwd(x, weight, g, want_stats):
    """x: fp32 [B,H,W,Cin] / [B,L,Cin] (Cin == 1 may also come as NCHW [B,1,H,W])."""
    B, H, W, C = _geom(x)
    P, Q = g.out_hw(H, W)
    y = torch.empty(_like_spatial(x, B, P, Q, g.K), dtype=BF16, device=x.device)
    stats = None
    if want_stats:
        rows = N.query('mpr_stem_fwd_stat_rows', B, P, Q, g.K)
        stats = torch.empty(rows, 2, g.K, dtype=F32, device=x.device)
    N.call('mpr_stem_fwd', x, weight.detach(), y, stats, B, H, W, C, g.K, *g.tail)
    return y, stats


def stem_s2d_operands(x, weight):
    """ResNet stem as space-to-depth (csrc/stem.hip): x fp32 [B,H,W,1] -> xs bf16 [B,H/2+3,W/2+3,8];
    weight [K,1,7,7] -> packed panel of the equivalent [K,8,4,4] stride-1 pad-0 filter (cached)."""
    B, H, W, _ = x.shape
    K = weight.shape[0]
    xs = torch.empty(B, H // 2 + 3, W // 2 + 3, 8, dtype=BF16, device=x.device)
    N.call('mpr_stem_s2d', x, xs, B, H, W)
    g2 = ConvGeom((K, 8, 4, 4), 1, 0)
    key = (weight.data_ptr(), weight._version)
    hit = getattr(weight, '_mpr_s2d', None)
    if hit is None or hit[0] != key:
        # (buffers persist across steps; the panel is packed here directly -- the temporary filter must not enter
        # the optimizer's repack registry)
        w2 = hit[2] if hit is not None else torch.empty(K, 8, 4, 4, dtype=F32, device=x.device)
        wf2 = hit[1] if hit is not None else torch.empty((K + 127) // 128 * 128 * 128, dtype=BF16, device=x.device)
        N.call('mpr_stem_w_s2d', weight.detach(), w2, K)
        N.call('mpr_conv_pack_weights', w2, wf2, None, K, 8, 4, 4)
        weight._mpr_s2d = (key, wf2, w2)
        hit = weight._mpr_s2d
    return xs, g2, hit[1]


def stem_s2d_wgrad(xs, dy, g2, weight):
    dw2 = conv_wgrad(xs, dy, g2, (g2.K, 8, 4, 4))
    tgt = grad_target(weight)
    if tgt is not None and tgt.is_contiguous():
        N.call('mpr_stem_dw_gather', dw2, tgt, g2.K, 1)
        return None
    dw = torch.empty(weight.shape if torch.is_tensor(weight) else weight, dtype=F32, device=xs.device)
    N.call('mpr_stem_dw_gather', dw2, dw, g2.K, 0)
    return dw


def stem_wgrad(x, dy, g, weight):
    B, H, W, C = _geom(x)
    tgt = grad_target(weight)
    if tgt is not None and tgt.is_contiguous():
        N.call('mpr_stem_wgrad', x, dy, tgt, 1, B, H, W, C, g.K, *g.tail)
        return None
    dw = torch.empty(weight.shape if torch.is_tensor(weight) else weight, dtype=F32, device=x.device)
    N.call('mpr_stem_wgrad', x, dy, dw, 0, B, H, W, C, g.K, *g.tail)
    return dw


# ------------------------------------------------------------------------------------------ fused ResNet stem
STEM_FUSED = os.environ.get('MPR_STEM_FUSED', '1') != '0'
STEMF_SLICES = 64


def stemf_ok(x, g):
    """Is (input, stem geometry) served by the fused / recomputed stem of csrc/stem_fused.hip?"""
    return (STEM_FUSED and x.dim() == 4 and x.shape[3] == 1 and g.C == 1 and g.K == 64
            and (g.R, g.S, g.sh, g.sw, g.ph, g.pw) == (7, 7, 2, 2, 3, 3)
            and bool(N.query('mpr_stemf_supported', x.shape[1], x.shape[2], g.K)))


def stemf_forward(x, weight, bn, train, want_bwd):
    """x fp32 [B,H,W,1] -> pooled bf16 [B,H/4,W/4,64] (= maxpool(relu(bn(conv7x7/2(x))))), BNState, saved tensors for
    stemf_backward (xb, wp, idx).  The full-resolution conv output is never stored."""
    B, H, W, _ = x.shape
    dev = x.device
    xb = torch.empty(B, H + 6, W + 8, dtype=BF16, device=dev)
    wp = torch.empty(64, 64, dtype=BF16, device=dev)
    N.call('mpr_stemf_prep', x, weight.detach(), xb, wp, B, H, W)
    st = BNState()
    st.pending = None
    st.eval = not train
    st.scale = torch.empty(64, dtype=F32, device=dev)
    st.shift = torch.empty(64, dtype=F32, device=dev)
    pooled = torch.empty(B, H // 4, W // 4, 64, dtype=BF16, device=dev)
    idx = torch.empty(pooled.shape, dtype=torch.uint8, device=dev) if want_bwd else None
    count = B * (H // 2) * (W // 2)
    if train:
        # 64 slice rows: same-address global atomics serialise, and 64 x 128 floats are nothing for the consumer to sum
        stats, zeroed = _slice_rows(64, dev, STEMF_SLICES)
        N.call('mpr_stemf_stats', xb, wp, stats, STEMF_SLICES, int(zeroed), B, H, W)
        st.mean = torch.empty(64, dtype=F32, device=dev)
        st.invstd = torch.empty(64, dtype=F32, device=dev)
        N.call('mpr_stemf_pool', xb, wp, stats, STEMF_SLICES, count, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
               bn.running_var, float(bn.momentum), float(bn.eps), st.scale, st.shift, st.mean, st.invstd, pooled, idx,
               B, H, W)
    else:
        st.mean = st.invstd = None
        N.call('mpr_bn_eval_coefs', bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
               float(bn.eps), st.scale, st.shift, 64)
        N.call('mpr_stemf_pool', xb, wp, None, 0, count, None, None, None, None, 0.0, float(bn.eps), st.scale, st.shift,
               None, None, pooled, idx, B, H, W)
    return pooled, st, (xb, wp, idx)


def stemf_backward(dpooled, xb, wp, idx, weight, gamma, beta, st, bn, train):
    """-> (dw, dgamma, dbeta); None where accumulated into the optimizer's gradient memory."""
    B, H, W = xb.shape[0], xb.shape[1] - 6, xb.shape[2] - 8
    dev = xb.device
    parts = N.query('mpr_stemf_bwd_parts', B, H, W)
    partial = torch.empty(parts, 7168, dtype=F32, device=dev)
    scratch = torch.empty(7168, dtype=torch.float64, device=dev)
    N.call('mpr_stemf_bwd', xb, dpooled, idx, partial, B, H, W)
    dgamma, dbeta, acc_bn, ret = _bn_grad_targets(gamma, beta, 64, dev)
    tgt = grad_target(weight)
    if tgt is not None and tgt.is_contiguous():
        dw, acc_dw = tgt, 1
    else:
        dw, acc_dw, tgt = torch.empty(weight.shape, dtype=F32, device=dev), 0, None
    count = B * (H // 2) * (W // 2)
    if train:
        N.call('mpr_stemf_bwd_finalize', partial, parts, scratch, wp, count, gamma.detach(), st.mean, st.invstd, 0.0, 0,
               dw, acc_dw, dgamma, dbeta, acc_bn)
    else:
        N.call('mpr_stemf_bwd_finalize', partial, parts, scratch, wp, count, gamma.detach(), bn.running_mean,
               bn.running_var, float(bn.eps), 1, dw, acc_dw, dgamma, dbeta, acc_bn)
    return (None if tgt is not None else dw), (dgamma if ret else None), (dbeta if ret else None)


# ------------------------------------------------------------------------------------------ batch norm
_ticket_pool = {}
# Measured WORSE and therefore off: the last-workgroup pattern needs agent-scope release/acquire fences, and a release
# fence writes back the XCD's whole dirty L2 -- which at that point holds the conv output just produced (15.3 vs 12.9 ms
# per C3 step).  Kept as an option for the record.
FUSED_FINALIZE = os.environ.get('MPR_FUSED_FINALIZE', '0') != '0'


def _ticket(dev):
    """One zeroed int of device memory for a last-workgroup-finishes kernel (it leaves the int zero again).  Slots
    rotate through a pool so that launches in flight on different streams never share one."""
    pool = _ticket_pool.get(dev)
    if pool is None:
        pool = _ticket_pool[dev] = [torch.zeros(4096, dtype=torch.int32, device=dev), 0]
    pool[1] = (pool[1] + 1) % 4096
    return pool[0][pool[1]:pool[1] + 1]


def _prereduce(parts, nsplit=64, limit=512):
    """Long per-workgroup partial lists are folded to `nsplit` rows first (spreads the read over the chip)."""
    n, _, C = parts.shape
    if n <= limit:
        return parts
    out = torch.empty(nsplit, 2, C, dtype=F32, device=parts.device)
    N.call('mpr_bn_reduce_partials', parts, n, out, nsplit, C)
    return out


# BatchNorm finalize folded into the consuming apply kernel (mpr_bn_apply_fin / mpr_bn_bwd_apply_fin): the partial sums
# are pre-reduced to FIN_SLICES rows and every workgroup of the apply kernel finishes them itself.
FIN_IN_CONSUMER = os.environ.get('MPR_FIN_IN_CONSUMER', '1') != '0'
# (4 = the library's default; round 3, one step in ms with 1 / 2 / 4 / 8 / 16 / 32 / 64 slice rows: 9.56 / 9.33 / 9.33 / 9.40 /
#  9.57 / 9.79 / 10.3 -- every workgroup of an apply pass sums the rows itself; the contention of the atomics does not show)
FIN_SLICES = int(os.environ.get('MPR_FIN_SLICES', '4'))
# short partial lists of narrow layers (rows x C floats per sum <= this) are handed to the consumer as they are: the
# pre-reduction launch costs more than every apply workgroup summing <= 16 KB itself (the profile branch's 1-D layers:
# 14..112 partial rows of 256..32 channels -- ~45 fewer launches per step on that latency-bound stream)
FIN_DIRECT_FLOATS = int(os.environ.get('MPR_FIN_DIRECT_FLOATS', '4096'))
BWD_ATOMIC_SLICES = os.environ.get('MPR_BWD_ATOMIC_SLICES', '1') != '0'
# The slice rows come from a per-device arena that is zeroed ONCE per step (FusedSGD.zero_grad -> reset_slice_arena): a
# memset per BatchNorm is a ~5 us fill launch, 77 of them per C3 step.  When the arena runs out (no optimizer resetting it,
# or more BatchNorm calls per step than slots) the launcher zeroes a fresh buffer itself.
SLICE_ARENA = os.environ.get('MPR_SLICE_ARENA', '1') != '0'
_SLICE_SLOTS, _SLICE_FLOATS = 512, FIN_SLICES * 2 * 512
if 'MPR_FIN_SLICES' in os.environ:      # experiment only: the forward kernels' slice count is a process-wide library setting
    N.query('mpr_conv_set_stat_slices', FIN_SLICES)
_slice_arena = {}          # device -> [tensor [slots][floats], next free slot]


def _slice_rows(C, dev, nslices=FIN_SLICES):
    """-> (zero-able tensor [nslices, 2, C], already zero?)."""
    if SLICE_ARENA and nslices * 2 * C <= _SLICE_FLOATS:
        a = _slice_arena.get(dev)
        if a is None:
            a = _slice_arena[dev] = [torch.zeros(_SLICE_SLOTS, _SLICE_FLOATS, dtype=F32, device=dev), 0]
        if a[1] < _SLICE_SLOTS:
            t = a[0][a[1]][:nslices * 2 * C].view(nslices, 2, C)
            a[1] += 1
            return t, True
    return torch.empty(nslices, 2, C, dtype=F32, device=dev), False


def reset_slice_arena():
    """Start of an optimisation step (every stream has been joined): zero the used part of each arena in one go."""
    for a in _slice_arena.values():
        if a[1]:
            a[0][:a[1]].zero_()
            a[1] = 0


def _consumer_slices(parts):
    n, _, C = parts.shape
    if n * C <= FIN_DIRECT_FLOATS:
        return parts
    return _prereduce(parts, FIN_SLICES, FIN_SLICES)


class BNState:
    """Per-call BatchNorm coefficients: scale/shift always, mean/invstd in train mode.  `pending`: the statistics are
    still (pre-reduced) partial sums -- the first bn_apply finalizes them inside its own kernel."""
    __slots__ = ('scale', 'shift', 'mean', 'invstd', 'pending', 'eval')


def bn_coefs(stats, count, bn, train, x=None, defer=False, want_bwd=False):
    """bn: object with weight, bias, running_mean, running_var, num_batches_tracked, momentum, eps.
    defer=True: the caller promises that the next use of the result is ops.bn_apply (which then finalizes).
    want_bwd (eval mode only): a backward pass will follow -- keep what the affine map's backward needs."""
    C = bn.weight.shape[0]
    dev = bn.weight.device
    st = BNState()
    st.pending = None
    st.eval = not train
    st.scale = torch.empty(C, dtype=F32, device=dev)
    st.shift = torch.empty(C, dtype=F32, device=dev)
    if train:
        if stats is None:
            rows = x.numel() // C
            stats = torch.empty(N.query('mpr_bn_reduce_rows', rows, C), 2, C, dtype=F32, device=dev)
            N.call('mpr_bn_stats', x, stats, rows, C)
        st.mean = torch.empty(C, dtype=F32, device=dev)
        st.invstd = torch.empty(C, dtype=F32, device=dev)
        if defer and FIN_IN_CONSUMER and C <= 512:
            st.pending = (_consumer_slices(stats), count, bn)
        elif FUSED_FINALIZE and stats.shape[0] > 512:
            # long partial list: pre-reduction and finalize in one launch (the last workgroup finalizes)
            slices = torch.empty(64, 2, C, dtype=F32, device=dev)
            N.call('mpr_bn_reduce_finalize_stats', stats, stats.shape[0], slices, 64, _ticket(dev), count,
                   bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, float(bn.momentum),
                   float(bn.eps), st.scale, st.shift, st.mean, st.invstd, C)
        else:
            stats = _prereduce(stats)
            N.call('mpr_bn_finalize_stats', stats, stats.shape[0], count, bn.weight.detach(), bn.bias.detach(),
                   bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps), st.scale, st.shift,
                   st.mean, st.invstd, C)
    else:
        st.mean = st.invstd = None
        N.call('mpr_bn_eval_coefs', bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
               float(bn.eps), st.scale, st.shift, C)
        if want_bwd:
            # eval-mode BatchNorm inside a differentiated graph: its backward is the affine map's (count 0 below)
            st.mean = bn.running_mean
            st.invstd = torch.empty(C, dtype=F32, device=dev)
            N.call('mpr_bn_eval_invstd', bn.running_var, float(bn.eps), st.invstd, C)
    return st


def _bwd_count(st, rows):
    """Element count for the BatchNorm-backward finalize: 0 marks an eval-mode layer (running statistics)."""
    return 0 if st.eval else rows


def bn_apply(x, st, residual=None, relu=True):
    C = x.shape[-1]
    y = torch.empty_like(x)
    if st.pending is not None:
        slices, count, bn = st.pending
        st.pending = None
        N.call('mpr_bn_apply_fin', x, slices, slices.shape[0], count, bn.weight.detach(), bn.bias.detach(),
               bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps), st.scale, st.shift, st.mean,
               st.invstd, residual, int(relu), y, x.numel() // C, C)
        return y
    N.call('mpr_bn_apply', x, st.scale, st.shift, residual, int(relu), y, x.numel() // C, C)
    return y


DUAL_BN_APPLY = os.environ.get('MPR_DUAL_BN_APPLY', '1') != '0'


def bn_apply_dual(x, st, xr, st_r, relu=True):
    """act(BN(x) + bf16(BN_r(xr))) in one pass -- a projection-shortcut block's output without storing the normalised
    shortcut map (bit-identical to bn_apply(xr, st_r, None, False) followed by bn_apply(x, st, that, relu)); None when the
    kernel does not cover the case (C > 512)."""
    C = x.shape[-1]
    if not DUAL_BN_APPLY or C > 512 or C % 8 or xr.shape != x.shape:
        return None
    y = torch.empty_like(x)

    def side(s):
        if s.pending is None:
            return [None, 0, 0, None, None, None, None, 0.0, 0.0, s.scale, s.shift, None, None]
        slices, count, bn = s.pending
        s.pending = None
        return [slices, slices.shape[0], count, bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                float(bn.momentum), float(bn.eps), s.scale, s.shift, s.mean, s.invstd]
    a, b = side(st), side(st_r)
    N.call('mpr_bn_apply_dual', x, *a, xr, *b, int(relu), y, x.numel() // C, C)
    return y


MASK_NONE, MASK_Y, MASK_RECOMPUTE = 0, 1, 2


def _bn_grad_targets(gamma, beta, C, dev):
    """(dgamma, dbeta, accumulate, handed_to_autograd?) -- the optimizer's gradient memory when both have one."""
    tg, tb = grad_target(gamma), grad_target(beta)
    if tg is not None and tb is not None:
        return tg, tb, 1, False
    return torch.empty(C, dtype=F32, device=dev), torch.empty(C, dtype=F32, device=dev), 0, True


def _bwd_finalize(parts, rows, gamma, st, dgamma, dbeta, acc, coef, C, dev):
    if FUSED_FINALIZE and parts.shape[0] > 512:
        slices = torch.empty(64, 2, C, dtype=F32, device=dev)
        N.call('mpr_bn_reduce_bwd_finalize', parts, parts.shape[0], slices, 64, _ticket(dev), rows, gamma.detach(),
               st.mean, st.invstd, dgamma, dbeta, acc, coef, C)
    else:
        parts = _prereduce(parts)
        N.call('mpr_bn_bwd_finalize', parts, parts.shape[0], rows, gamma.detach(), st.mean, st.invstd, dgamma, dbeta,
               acc, coef, C)


def bn_bwd(dy, y, x, gamma, st, mask_mode, want_dz=False, beta=None):
    """-> dx (bf16), dgamma, dbeta (fp32; None when accumulated into the optimizer's buffers), dz (bf16 | None)."""
    C = x.shape[-1]
    rows = x.numel() // C
    dev = x.device
    nparts = N.query('mpr_bn_reduce_rows', rows, C)
    direct = FIN_IN_CONSUMER and C <= 512
    if direct and BWD_ATOMIC_SLICES and nparts * C > FIN_DIRECT_FLOATS:
        # long partial list: the reduce pass adds into FIN_SLICES rows itself (fp32 atomics), no pre-reduction launch
        parts, zeroed = _slice_rows(C, dev)
        N.call('mpr_bn_bwd_reduce_slices', dy, y, x, st.mean, st.invstd, st.scale, st.shift, mask_mode, parts, FIN_SLICES,
               int(zeroed), rows, C)
    else:
        parts = torch.empty(nparts, 2, C, dtype=F32, device=dev)
        N.call('mpr_bn_bwd_reduce', dy, y, x, st.mean, st.invstd, st.scale, st.shift, mask_mode, parts, rows, C)
    dgamma, dbeta, acc, ret = _bn_grad_targets(gamma, beta, C, dev)
    dx = torch.empty_like(x)
    dz = torch.empty_like(x) if want_dz else None
    if st.eval and not direct:
        raise N.NativeLibraryError('backward through an eval-mode BatchNorm with more than 512 channels is not implemented')
    if direct:
        slices = _consumer_slices(parts)
        N.call('mpr_bn_bwd_apply_fin', dy, y, x, slices, slices.shape[0], _bwd_count(st, rows), gamma.detach(), st.mean, st.invstd,
               dgamma, dbeta, acc, st.scale, st.shift, mask_mode, dx, dz, rows, C)
        return dx, (dgamma if ret else None), (dbeta if ret else None), dz
    coef = torch.empty(3, C, dtype=F32, device=dev)
    _bwd_finalize(parts, rows, gamma, st, dgamma, dbeta, acc, coef, C, dev)
    N.call('mpr_bn_bwd_apply', dy, y, x, coef, st.scale, st.shift, mask_mode, dx, dz, rows, C)
    return dx, (dgamma if ret else None), (dbeta if ret else None), dz


# ---- BatchNorm-backward reduction fused into the producing data gradient (csrc/conv_win.hip, template BNB) --------
# The gradient w.r.t. a BatchNorm(+ReLU) output is produced by a data-gradient convolution; its epilogue can apply the
# ReLU mask and accumulate (sum dz, sum dz * xhat) itself, so bn_bwd's reduce pass over (dy, y, x) disappears and the
# apply pass reads the already masked dz.  Inside a BasicBlock (conv2's data gradient -> bn1) this is local; ACROSS
# blocks (conv1's data gradient + skip of block k+1 -> bn2 of block k) the producer block's forward leaves a note for the
# consumer's backward and the consumer leaves the sums for the producer's backward -- both keyed by the tensor's storage
# and valid for one backward pass of one forward pass only.
DGRAD_BN_FUSION = os.environ.get('MPR_DGRAD_BN_FUSION', '1') != '0'


class BlockChain:
    """Per-backbone hand-off between the autograd Functions of consecutive BasicBlocks.  Entries hold strong references
    to the tensors they describe (so a matching data_ptr IS that tensor) and are dropped at the backbone's next forward."""

    def __init__(self):
        self.notes = {}      # data_ptr of a block output -> (out, x2, BNState): "whoever computes d(out) may fuse my bn2 sums"
        self.sums = {}       # data_ptr of a gradient -> (dz, slices): "this gradient is already masked, here are its sums"

    def clear(self):
        self.notes.clear()
        self.sums.clear()

    # The entries are non-leaf tensors of ONE forward / backward pass: a copy or a pickle of the owning module
    # (copy.deepcopy(model), torch.save(model)) gets an EMPTY table -- the blocks of the copy share the one new object
    # through deepcopy's memo, exactly as the originals share this one.
    def __deepcopy__(self, memo):
        new = BlockChain()
        memo[id(self)] = new
        return new

    def __getstate__(self):
        return {'notes': {}, 'sums': {}}

    def note_bn2(self, out, x2, st2):
        if DGRAD_BN_FUSION:
            self.notes[out.data_ptr()] = (out, x2, st2)

    def take_note(self, x):
        n = self.notes.pop(x.data_ptr(), None)
        if n is None or n[0].shape != x.shape:
            return None
        return n[1], n[2]

    def offer_sums(self, dz, slices):
        self.sums[dz.data_ptr()] = (dz, slices)

    def take_sums(self, dout):
        n = self.sums.pop(dout.data_ptr(), None)
        if n is None or n[0].shape != dout.shape:
            return None
        return n[1]


def conv_dgrad_bn(dy, wd, g, x_shape, bn_x, st, mask_mode, mask_y=None, add=None):
    """Data gradient with the ReLU mask and the BatchNorm-backward sums of the layer that produced its input fused into
    the epilogue.  -> (dz, slices [FIN_SLICES, 2, C]) or None when the geometry is not served."""
    if not DGRAD_BN_FUSION or st.mean is None:
        return None
    B, H, W, C = (x_shape[0], 1, x_shape[1], x_shape[2]) if len(x_shape) == 3 else tuple(x_shape)
    if C > 512 or not N.query('mpr_conv_dgrad_bn_supported', B, H, W, C, g.K, *g.tail):
        return None
    dz = torch.empty(x_shape, dtype=BF16, device=dy.device)
    slices, zeroed = _slice_rows(C, dy.device)
    N.call('mpr_conv_dgrad_bn', dy, wd, dz, add, mask_mode, mask_y, bn_x, st.mean, st.invstd, st.scale, st.shift, slices,
           FIN_SLICES, int(zeroed), B, H, W, C, g.K, *g.tail)
    return dz, slices


def bn_bwd_from_sums(dz, x, gamma, st, slices, beta=None):
    """BatchNorm backward when dz (already ReLU-masked) and its sums are given: ONE pass, dx = k1 dz + k2 x + k3.
    -> dx, dgamma, dbeta (None when accumulated into the optimizer's buffers)."""
    C = x.shape[-1]
    rows = x.numel() // C
    dgamma, dbeta, acc, ret = _bn_grad_targets(gamma, beta, C, x.device)
    dx = torch.empty_like(x)
    N.call('mpr_bn_bwd_apply_fin', dz, None, x, slices, slices.shape[0], _bwd_count(st, rows), gamma.detach(), st.mean,
           st.invstd, dgamma, dbeta, acc, st.scale, st.shift, MASK_NONE, dx, None, rows, C)
    return dx, (dgamma if ret else None), (dbeta if ret else None)


# ------------------------------------------------------------------------------------------ pooling
def _pool_geom(x, k, s, p):
    B, H, W, C = _geom(x)
    if x.dim() == 3:
        return B, H, W, C, 1, k, 1, s, 0, p
    return B, H, W, C, k, k, s, s, p, p


def bn_relu_maxpool_fwd(x, st, k=3, s=2, p=1):
    B, H, W, C, RH, RW, SH, SW, PH, PW = _pool_geom(x, k, s, p)
    P, Q = (H + 2 * PH - RH) // SH + 1, (W + 2 * PW - RW) // SW + 1
    y = torch.empty(_like_spatial(x, B, P, Q, C), dtype=BF16, device=x.device)
    idx = torch.empty(y.shape, dtype=torch.uint8, device=x.device)
    N.call('mpr_bn_relu_maxpool_fwd', x, st.scale if st is not None else None,
           st.shift if st is not None else None, y, idx, B, H, W, C, RH, RW, SH, SW, PH, PW)
    return y, idx


def maxpool_bwd(dy, idx, x_shape, k=3, s=2, p=1):
    dx = torch.empty(x_shape, dtype=BF16, device=dy.device)
    B, H, W, C, RH, RW, SH, SW, PH, PW = _pool_geom(dx, k, s, p)
    N.call('mpr_maxpool_bwd', dy, idx, dx, B, H, W, C, RH, RW, SH, SW, PH, PW)
    return dx


def pool_bn_bwd(dpooled, idx, x, gamma, st, k=3, s=2, p=1, beta=None):
    """Stem backward, fused: maxpool-backward gather + ReLU mask + BatchNorm backward.  -> dx, dgamma, dbeta."""
    B, H, W, C, RH, RW, SH, SW, PH, PW = _pool_geom(x, k, s, p)
    dev = x.device
    rows = x.numel() // C
    parts = torch.empty(N.query('mpr_pool_bn_bwd_rows', B, H, W, C), 2, C, dtype=F32, device=dev)
    geo = (B, H, W, C, RH, RW, SH, SW, PH, PW)
    N.call('mpr_pool_bn_bwd', 0, dpooled, idx, x, st.scale, st.shift, st.mean, st.invstd, None, parts, None, *geo)
    dgamma, dbeta, acc, ret = _bn_grad_targets(gamma, beta, C, dev)
    coef = torch.empty(3, C, dtype=F32, device=dev)
    _bwd_finalize(parts, rows, gamma, st, dgamma, dbeta, acc, coef, C, dev)
    dx = torch.empty_like(x)
    N.call('mpr_pool_bn_bwd', 1, dpooled, idx, x, st.scale, st.shift, None, None, coef, None, dx, *geo)
    return dx, (dgamma if ret else None), (dbeta if ret else None)


def global_pool_fwd(x, mode):
    B, C = x.shape[0], x.shape[-1]
    L = x.numel() // (B * C)
    y = torch.empty(B, C, dtype=F32, device=x.device)
    if mode == 'avg':
        N.call('mpr_global_avgpool_fwd', x, y, B, L, C)
        return y, None
    idx = torch.empty(B, C, dtype=torch.int32, device=x.device)
    N.call('mpr_global_maxpool_fwd', x, y, idx, B, L, C)
    return y, idx


def global_pool_bwd(dy, idx, x_shape, mode):
    dx = torch.empty(x_shape, dtype=BF16, device=dy.device)
    B, C = x_shape[0], x_shape[-1]
    L = dx.numel() // (B * C)
    if mode == 'avg':
        N.call('mpr_global_avgpool_bwd', dy, dx, B, L, C)
    else:
        N.call('mpr_global_maxpool_bwd', dy, idx, dx, B, L, C)
    return dx


# ------------------------------------------------------------------------------------------ fp32 GEMM & friends
def gemm(a, b, trans_a=False, trans_b=False, bias=None, out=None, alpha=1.0, beta=0.0):
    """op(a) @ op(b) for 2-D (or batched 3-D, same batch) fp32 row-major tensors."""
    batched = a.dim() == 3
    if batched:
        nb = a.shape[0]
        a2, b2 = a.shape[1:], b.shape[1:]
    else:
        nb, a2, b2 = 1, a.shape, b.shape
    M, K = (a2[1], a2[0]) if trans_a else (a2[0], a2[1])
    Kb, Nn = (b2[1], b2[0]) if trans_b else (b2[0], b2[1])
    assert K == Kb, f'gemm: inner dimensions differ ({K} vs {Kb})'
    if not (a.is_contiguous() and b.is_contiguous() and (out is None or out.is_contiguous())):
        # (N.ptr accepts any dense permutation -- conv filters are addressed through strides -- but the leading
        # dimensions passed below are those of row-major operands)
        raise N.NativeLibraryError('gemm: operands must be row-major contiguous (use trans_a / trans_b for transposes)')
    if out is None:
        out = torch.empty((nb, M, Nn) if batched else (M, Nn), dtype=F32, device=a.device)
    N.call('mpr_gemm_f32', a, b, out, bias, M, Nn, K, a2[1], b2[1], Nn, int(trans_a), int(trans_b), float(alpha),
           float(beta), nb, a2[0] * a2[1], b2[0] * b2[1], M * Nn)
    return out


def linear_wgrad(dy, x, weight):
    """dW = dy^T x of an nn.Linear (exact fp32).  With optimizer-owned gradient memory it is ACCUMULATED there on the
    weight-gradient side stream -- a leaf of the backward graph, off the dependent chain (the projections sit in the
    forward / backward junction, where nothing else runs) -- and None is returned; otherwise the gradient tensor."""
    tgt = grad_target(weight)
    Nout, K = weight.shape
    if tgt is None or not tgt.is_contiguous():
        return gemm(dy, x, trans_a=True)
    args = (dy, x, tgt, None, Nout, K, dy.shape[0], dy.shape[1], x.shape[1], K, 1, 0, 1.0, 1.0, 1, 0, 0, 0)
    if ASYNC_WGRAD:
        cur, side = _wgrad_stream(dy.device.index)
        side.wait_stream(cur)
        N.call('mpr_gemm_f32', *args, stream_handle=side.cuda_stream)
        _note_arena_stream(dy.device.index, side)
        if KEEPALIVE_WGRAD_OPERANDS and _join_queued[0]:
            _wgrad_keepalive.append((x, dy))
        else:
            x.record_stream(side)
            dy.record_stream(side)
    else:
        N.call('mpr_gemm_f32', *args)
    return None


def accumulate_off_chain(param, g):
    """Gradient of a small parameter computed inside a backward Function (the losses' logit_scale / bias): with
    optimizer-owned gradient memory it is added there on the weight-gradient side stream and None is returned (autograd's
    AccumulateGrad would put a 5-us add kernel + its dispatch into the dependent chain of the junction, where the chip is
    idle and every microsecond counts); otherwise `g` comes back for autograd."""
    tgt = grad_target(param)
    if tgt is None or g is None:
        return g
    if ASYNC_WGRAD:
        cur, side = _wgrad_stream(g.device.index)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            tgt.add_(g.reshape(tgt.shape))
        _note_arena_stream(g.device.index, side)
        g.record_stream(side)
    else:
        tgt.add_(g.reshape(tgt.shape))
    return None


_root_grads = {}


def backward(loss):
    """loss.backward() with a cached root gradient of one (torch fills a fresh ones_like(loss) per call: a 5-us kernel in
    front of the first backward kernel)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    one = _root_grads.get(key)
    if one is None:
        one = _root_grads[key] = torch.ones_like(loss)
    torch.autograd.backward(loss, one)


def tail_fwd(feat, meta, denom, p_drop, seed):
    B, Fd = feat.shape
    Mm = 0 if meta is None else meta.shape[1]
    out = torch.empty(B, Fd + Mm, dtype=F32, device=feat.device)
    mask = torch.empty(B, Fd + Mm, dtype=torch.uint8, device=feat.device) if p_drop > 0 else None
    N.call('mpr_tail_fwd', feat, meta, 1.0 / float(denom), float(p_drop), int(seed) & 0xffffffff, out, mask, B, Fd, Mm)
    return out, mask


def tail_bwd(dout, mask, p_drop, Fd):
    B, Wd = dout.shape
    dfeat = torch.empty(B, Fd, dtype=F32, device=dout.device)
    N.call('mpr_tail_bwd', dout, mask, float(p_drop), dfeat, B, Fd, Wd - Fd)
    return dfeat


def softmax_ce(logits, labels=None, want_grad=False):
    rows, C = logits.shape
    dev = logits.device
    argmax = torch.empty(rows, dtype=torch.int64, device=dev)
    if labels is None:
        N.call('mpr_softmax_ce', logits, None, None, None, argmax, None, rows, C)
        return None, argmax, None
    row_loss = torch.empty(rows, dtype=F32, device=dev)
    loss = torch.empty((), dtype=F32, device=dev)
    dlogits = torch.empty_like(logits) if want_grad else None
    N.call('mpr_softmax_ce', logits, labels, row_loss, loss, argmax, dlogits, rows, C)
    return loss, argmax, dlogits


def scale_by_scalar(x, s):
    y = torch.empty_like(x)
    N.call('mpr_scale_by_scalar', x, s, y, x.numel())
    return y


# ------------------------------------------------------------------------------------------ optimiser
def _mark_touched(p):
    p._mpr_touched = True
    for obs in grad_ready_observers:
        obs(p)


def _dense(t):
    """Every element of the storage span used exactly once (any permutation of a contiguous tensor)."""
    if t.numel() <= 1:
        return True
    dims = sorted(((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1))
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True


class FusedSGD:
    """torch.optim.SGD semantics (src/model.py:147-148) as ONE multi-tensor launch per step.

    Gradients and momentum live in two flat fp32 buffers owned by the optimizer: ``p.grad`` is a view of
    ``flat_grad`` with the parameter's own strides (so a channels-last conv weight receives the weight-gradient
    kernel's native [K][R][S][C] result in place), ``zero_grad`` is one memset, the fused backward Functions
    accumulate straight into it (``ops.grad_target``), data-parallel training all-reduces ``flat_grad`` as it is,
    and the pointer table of the update kernel is built once.  A gradient assigned from outside
    (``p.grad = t``) is copied in; a parameter whose gradient is None or was never produced since ``zero_grad``
    is skipped, as torch does."""

    def __init__(self, params, lr, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False):
        self.params = [p for p in params if p.requires_grad]
        self.lr, self.momentum, self.dampening = float(lr), float(momentum), float(dampening)
        self.weight_decay, self.nesterov = float(weight_decay), bool(nesterov)
        self.bufs = {}
        self.steps = 0
        self.flat_grad = self.flat_buf = None
        self._views = None
        self._arena_key = None
        self._table_full = None
        self._seen = set()        # indices (in self.params) of parameters that have taken a step (momentum buffer initialised)

    # ------------------------------------------------------------------ flat buffers
    def _install(self):
        ps = self.params
        key = tuple(p.data_ptr() for p in ps)
        if key == self._arena_key:
            return
        dev = ps[0].device
        offs, total = [], 0
        for p in ps:
            if p.dtype != F32 or not _dense(p) or p.device != dev:
                raise N.NativeLibraryError('FusedSGD needs dense fp32 parameters on one device')
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4          # 16-byte aligned slots
        old_bufs = self.bufs
        self.flat_grad = torch.zeros(total, dtype=F32, device=dev)
        self.flat_buf = torch.zeros(total, dtype=F32, device=dev)
        self._views, self.bufs, rows = [], {}, []
        for p, o in zip(ps, offs):
            v = self.flat_grad[o:o + p.numel()].as_strided(p.shape, p.stride())
            b = self.flat_buf[o:o + p.numel()].as_strided(p.shape, p.stride())
            if id(p) in old_bufs and old_bufs[id(p)] is not None:
                b.copy_(old_bufs[id(p)])
            if not hasattr(p, '_mpr_hooked'):
                p.register_post_accumulate_grad_hook(_mark_touched)
                p._mpr_hooked = True
            p._mpr_grad = v
            if p.grad is None:
                p.grad, p._mpr_touched = v, False
            self._views.append(v)
            self.bufs[id(p)] = b
            rows.append([p.data_ptr(), v.data_ptr(), b.data_ptr(), p.numel()])
        self._rows = rows
        self.offsets = dict((id(p), (o, p.numel())) for p, o in zip(ps, offs))      # parameter -> (offset, numel) in flat_grad
        self._table_full = torch.tensor(rows, dtype=torch.int64).to(dev)
        self._max_numel = max(p.numel() for p in ps)
        self._arena_key = key

    def zero_grad(self, set_to_none=True):
        """One memset of the flat gradient buffer; ``p.grad`` stays a (zero) view of it."""
        if not self.params:
            return
        self._install()
        self.flat_grad.zero_()
        reset_slice_arena()
        for p, v in zip(self.params, self._views):
            if p.grad is not v:
                p.grad = v
            p._mpr_touched = False

    @torch.no_grad()
    def step(self):
        if not self.params:
            return
        self._install()
        join_gradient_streams()      # (also done by the end-of-backward callback; idempotent)
        live = []
        for i, (p, v) in enumerate(zip(self.params, self._views)):
            g = p.grad
            if g is None:
                continue
            if g is not v:                      # assigned from outside: bring it into the flat buffer
                v.copy_(g)
                p.grad = v
                live.append(i)
            elif getattr(p, '_mpr_touched', False):
                live.append(i)
        if not live:
            return
        new = [i for i in live if i not in self._seen]
        if new and len(new) != len(live) and self.momentum:
            # torch would give the newcomers a "first step" (buf = g) of their own
            raise N.NativeLibraryError('FusedSGD: parameter set changed after the first step')
        first = 1 if new else 0
        if len(live) == len(self.params):
            table = self._table_full
            nmax = self._max_numel
        else:
            table = torch.tensor([self._rows[i] for i in live], dtype=torch.int64).to(self.flat_grad.device)
            nmax = max(self._rows[i][3] for i in live)
        N.call('mpr_sgd_multi', table, len(live), nmax, self.lr, self.momentum, self.dampening, self.weight_decay,
               int(self.nesterov), first)
        self._seen.update(live)
        self.steps += 1
        # parameters were mutated through raw pointers: bump their version counters, then refresh every cached
        # bf16 filter panel in one launch (ops.packed_weights would otherwise repack filter by filter)
        FusedSGD._bump_versions([self.params[i] for i in live])
        pack_registry.repack_all()

    @staticmethod
    def _bump_versions(params):
        for p in params:
            torch.autograd.graph.increment_version(p)

    def state_dict(self):
        return {'steps': self.steps, 'momentum_buffers': [self.bufs.get(id(p)) for p in self.params],
                'seen': [i in self._seen for i in range(len(self.params))],
                'hyper': dict(lr=self.lr, momentum=self.momentum, dampening=self.dampening,
                              weight_decay=self.weight_decay, nesterov=self.nesterov)}

    def load_state_dict(self, state):
        """Resume: momentum buffers, step count and hyper-parameters of a state_dict() taken from an optimizer over the
        same parameter list (torch.optim.SGD.load_state_dict semantics)."""
        bufs = state['momentum_buffers']
        if len(bufs) != len(self.params):
            raise ValueError(f'FusedSGD.load_state_dict: {len(bufs)} momentum buffers for {len(self.params)} parameters')
        self._install()
        seen = state.get('seen', [b is not None for b in bufs])
        with torch.no_grad():
            for i, (p, b, sn) in enumerate(zip(self.params, bufs, seen)):
                if b is not None:
                    if tuple(b.shape) != tuple(p.shape):
                        raise ValueError('FusedSGD.load_state_dict: momentum buffer shape mismatch')
                    self.bufs[id(p)].copy_(b.to(p.device))
                if sn:
                    self._seen.add(i)
        self.steps = int(state.get('steps', 0))
        for k, v in state.get('hyper', {}).items():
            setattr(self, k, type(getattr(self, k))(v))
