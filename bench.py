"""Headline benchmark: samples/s (image+profile pairs) of one full train_multi optimisation step
(forward + backward + SGD) on BASELINE config C3 -- ResNet-18 (1-channel 224x224) + ProfileCNN[2,2,2,2]
+ CLIP loss, D = 512, batch 512 per GPU, synthetic on-device data, random-init weights.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `roofline` is measured live with hipEvent pairs around every launch of
the dominant kernel family (the shifted-window conv kernel, forward + data-gradient instantiations) inside
the timed region, `roofline.families` splits the other MFMA kernels the same way; `cpu_baseline` times the
CPU oracle (plain-torch fp32 restatement of the same step) on the SAME batch-512 workload, a few steps.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CARD = os.path.join(ROOT, 'model_cards', 'resnet18_cnn_2_512_clip.yaml')
MFMA_BF16_PEAK_TFLOPS = 2500.0        # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
C3_STEP_TFLOP = 5.36                    # algorithmic work of one batch-512 C3 step
PROF_EVERY = 10                         # per-launch HIP events (roofline) on every 10th timed step: they cost ~0.4 ms on a step


def synthetic_batch(B, T, device, seed, transformer=False):
    """image ~ clamp(N(0.6136, 0.0938), 0, 1)*2-1, profile ~ U[-1, 1]
    (transformer profile encoder: + prepended zero CLS row, time = arange(T + 1), all-False padding mask)."""
    g = torch.Generator(device=device).manual_seed(seed)
    image = (torch.randn(B, 1, T, T, generator=g, device=device) * 0.0938 + 0.6136).clamp_(0, 1) * 2 - 1
    profile = torch.rand(B, T, 6, generator=g, device=device) * 2 - 1
    image_shape = torch.randint(32, 401, (B, 2), generator=g, device=device)
    profile_len = torch.randint(8, 1025, (B, 1), generator=g, device=device)
    batch = {'image': image, 'profile': profile, 'image_shape': image_shape, 'profile_len': profile_len}
    if transformer:
        batch['profile'] = torch.cat((torch.zeros(B, 1, 6, device=device), profile), 1)
        batch['time'] = torch.arange(T + 1, device=device).repeat(B, 1)
        batch['padding_mask'] = torch.zeros(B, T + 1, dtype=torch.bool, device=device)
    return batch


def isolated_conv_rate(B, dev):
    """The dominant kernel with nothing else on the GPU: forward + data gradient of the four 3x3 body convolutions
    of ResNet-18 at the bench batch (event-timed on the current stream, 5 launches each after 2 warm-up)."""
    from multimodal_plankton_recognition_amd import ops
    tot_ms, tot_flop = 0.0, 0.0
    for H, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
        g = ops.ConvGeom((C, C, 3, 3), 1, 1)
        w = torch.randn(C, C, 3, 3, device=dev) * 0.05
        wf, wd = ops.packed_weights(w, g)
        x = torch.randn(B, H, H, C, device=dev).to(torch.bfloat16)
        for fn in (lambda: ops.conv_fwd(x, wf, g, True), lambda: ops.conv_dgrad(x, wd, g, x.shape)):
            for _ in range(2):
                fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5):
                fn()
            b.record()
            torch.cuda.synchronize()
            tot_ms += a.elapsed_time(b) / 5
            tot_flop += 2.0 * B * H * H * C * C * 9
    rate = tot_flop / (tot_ms * 1e-3) / 1e12
    return {'achieved': round(rate, 2), 'frac': round(rate / MFMA_BF16_PEAK_TFLOPS, 4), 'unit': 'TFLOP/s',
            'avg_launch_us': round(tot_ms * 1e3 / 8, 2)}


PMC_FILE = os.path.join(ROOT, 'profiles', 'r03_pmc_traffic.json')
if not os.path.exists(PMC_FILE):
    PMC_FILE = os.path.join(ROOT, 'profiles', 'r02_pmc_traffic.json')


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the committed PMC summary (scripts/prof_pmc.sh + scripts/pmc_summary.py
    run this same command under rocprofv3 --pmc; a process cannot read its own counters), or None."""
    if not os.path.exists(PMC_FILE):
        return None
    fam = json.load(open(PMC_FILE)).get('families', {}).get(family)
    return round(fam['hbm_bytes_per_launch']) if fam and fam.get('launches_per_step') else None


def host_cores():
    """CPU share of this process: the cgroup quota when there is one (a 1-GPU box gets 16 of the node's cores)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(card, T, threads, B):
    """Oracle (CPU restatement, torch fp32) timed on the SAME workload as the GPU line: batch B (512), dropout on,
    1 warm-up + up to 3 timed steps (bounded to ~30 s), median."""
    from oracle import model as OM
    from multimodal_plankton_recognition_amd.model import MultiModel
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                       card['coordination_args'], card['optim_args'])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    cfg = {k: card[k] for k in ('image_encoder_args', 'profile_encoder_args', 'coordination_args', 'optim_args')}
    batch = synthetic_batch(B, T, 'cpu', 1234)
    batch['buckets'] = 1
    bufs = {}
    t0 = time.time()
    OM.train_step(sd, batch, cfg, bufs, apply_dropout=True)                       # warm-up
    warm = time.time() - t0
    times = []
    t_end = time.time() + max(30.0 - warm, 0.0)
    while len(times) < 3 and (time.time() < t_end or not times):
        t0 = time.time()
        OM.train_step(sd, batch, cfg, bufs, apply_dropout=True)
        times.append(time.time() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {'value': round(B / med, 2), 'unit': 'samples/s', 'cores': threads, 'kind': 'port', 'cpu': cpu_model(),
            'sample': f'oracle (torch fp32 CPU restatement) full optimisation step at batch {B}, dropout '
                      f"{card['image_encoder_args'].get('dropout', 0.1)}, 1 warm-up + {len(times)} timed steps, median "
                      f'{med:.2f} s/step'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=None, help='per-GPU batch (default: the card: 512)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--card', default=CARD, help='model card (default: BASELINE C3; model_cards/vit_base_transformer_clip.yaml '
                                                 'with --batch 128 is the per-GPU share of C5)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the HIP path has no CPU fallback)'
    local_rank %= torch.cuda.device_count()          # (rehearsals put several ranks on one GPU: MPR_DIST_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from multimodal_plankton_recognition_amd import _native as N, ops
    from multimodal_plankton_recognition_amd.model import MultiModel
    from multimodal_plankton_recognition_amd import distributed as D

    card = yaml.safe_load(open(args.card))
    c3 = os.path.abspath(args.card) == os.path.abspath(CARD)
    from multimodal_plankton_recognition_amd import transformer as TF
    TF.set_precision((card.get('trainer_args') or {}).get('precision'))      # transformer stacks only (16-mixed -> bf16 MFMA)
    is_tf = 'num_head' in card['profile_encoder_args']
    T = card['target_size']
    B = args.batch or card['bs']
    if world > 1:
        D.init(dev)

    torch.manual_seed(0)                                      # identical init on every rank
    model = MultiModel(card['dim_embedding'], card['image_encoder_args'], card['profile_encoder_args'],
                       card['coordination_args'], card['optim_args']).to(dev).train()
    opt = model.configure_optimizers()
    stepper = D.DataParallelStep(model, opt, world) if world > 1 else None
    batch = synthetic_batch(B, T, dev, 1234 + rank, transformer=is_tf)
    batch['buckets'] = card['buckets']

    def one_step():
        if stepper is not None:
            return stepper.step(batch)
        opt.zero_grad()
        loss = model.training_step(batch, 0)
        ops.backward(loss)              # (what Trainer.fit does: loss.backward() with a cached root gradient)
        opt.step()
        return loss

    for _ in range(args.warmup):
        one_step()
    model.train_loss.clear()

    lib = N.lib()
    lib.mpr_prof_reset()
    lib.mpr_prof_enable(1)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # step boundaries on the step's stream
    if world > 1:
        D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        # the per-launch HIP events of the roofline measurement cost ~5 us of stream bubble each (~0.4 ms per step):
        # they are recorded on every PROF_EVERY-th step of the timed region only (two of the default twenty)
        lib.mpr_prof_enable(1 if i % PROF_EVERY == 0 else 0)
        loss = one_step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        D.barrier()
    elapsed = time.perf_counter() - t0
    lib.mpr_prof_enable(0)
    if world > 1:
        elapsed = D.max_over_ranks(elapsed)
    loss_val = float(loss.detach())
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    unprofiled = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps) if i % PROF_EVERY)
    median_unprofiled = unprofiled[len(unprofiled) // 2] if unprofiled else median_ms

    def collect(kind):
        ms, work, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int()
        lib.mpr_prof_collect(kind, ctypes.byref(ms), ctypes.byref(work), ctypes.byref(n))
        return ms.value, work.value, n.value

    if rank == 0:
        prof_steps = len(range(0, args.steps, PROF_EVERY))
        ms_per_step = elapsed / args.steps * 1e3

        def family(kinds, name, pmc_key, algo_kinds=None):
            ms = work = 0.0
            n = 0
            for k in kinds:
                a, b, c = collect(k)
                ms, work, n = ms + a, work + b, n + c
            by = ctypes.c_double()
            algo = 0.0
            for k in kinds:
                lib.mpr_prof_collect_bytes(k, ctypes.byref(by))
                algo += by.value
            ach = work / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            return {'kernel': name, 'achieved': round(ach, 2), 'unit': 'TFLOP/s', 'frac': round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    'launches': n, 'avg_launch_us': round(ms * 1e3 / max(n, 1), 2),
                    'share_of_step_time': round(ms / max(prof_steps, 1) / ms_per_step, 3),
                    'algorithmic_bytes_per_launch': round(algo / max(n, 1)),
                    'traffic': pmc_traffic(pmc_key) if c3 else None}

        fam = {'win_fwd': family([6], 'conv_win_kernel / conv_win_l1_kernel (64 channels: filter in registers), forward (3x3 stride 1)', 'win_fwd'),
               'win_dgrad': family([7], 'conv_win_kernel, data gradient (+ fused skip add, ReLU mask, BatchNorm-backward sums)', 'win_dgrad'),
               'dma_fwd_dgrad': family([0, 1], 'conv_igemm_dma_kernel (stride-2 3x3 and 1x1 convolutions forward + data gradient, layer4 3x3 forward)', 'dma_fwd_dgrad'),
               'wgrad_win': family([8], 'conv_wgrad_win_kernel (3x3 stride 1 weight gradients)', 'wgrad_win'),
               'wgrad_dma': family([2], 'conv_wgrad_dma_kernel (other weight gradients)', 'wgrad_dma')}
        dom = family([6, 7], 'conv_win_kernel / conv_win_l1_kernel (shifted-window implicit-GEMM conv: the forward passes of 10 and the data gradients '
                             "of 13 of ResNet-18's 20 convolutions, about three quarters of its forward + data-gradient FLOPs)", 'win_fwd_dgrad')
        if dom['launches'] == 0:
            # cards without 3x3 / stride-1 convolutions (transformer encoders: every linear is a one-tap implicit GEMM)
            dom = family([0, 1], 'conv_igemm_dma_kernel (LDS-DMA implicit GEMM: the linears of the transformer encoders, forward + '
                                 'data gradient)', 'dma_fwd_dgrad')
        out = {
            'metric': 'samples/sec (image+profile pairs) for train_multi', 'value': round(B * world * args.steps / elapsed, 1),
            'unit': 'samples/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16', 'data': 'synthetic',
            'ms_per_step_median': round(median_ms, 3), 'ms_per_step_median_unprofiled': round(median_unprofiled, 3),
            'config': {'workload': ('C3: model_cards/resnet18_cnn_2_512_clip.yaml -- ResNet-18 (1x224x224) + ProfileCNN[2,2,2,2] '
                                    '+ CLIP, D=512, full step (fwd+bwd+SGD nesterov), dropout 0.1') if c3 else
                                   (f"{os.path.relpath(args.card, ROOT)} -- {card['image_encoder_args']['name']} + "
                                    f"{'ProfileTransformer' if is_tf else 'profile encoder'} + "
                                    f"{card['coordination_args']['method']}, transformer precision {TF._PRECISION[0]}, full step"),
                       'per_gpu_batch': B, 'global_batch': B * world,
                       'parallelism': f'dp{world}' if world > 1 else 'single',
                       'loss': round(loss_val, 5)},
            'roofline': {'bound': 'mfma', 'kernel': dom['kernel'], 'achieved': dom['achieved'], 'peak': MFMA_BF16_PEAK_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': dom['frac'], 'traffic': dom['traffic'],
                         'traffic_unit': 'HBM bytes per launch (rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE, '
                                         f'profiles/{os.path.basename(PMC_FILE)})',
                         'algorithmic_bytes_per_launch': dom['algorithmic_bytes_per_launch'],
                         'launches': dom['launches'], 'avg_launch_us': dom['avg_launch_us'],
                         'share_of_step_time': dom['share_of_step_time'],
                         'families': fam,
                         # so that the headline cannot look better than the step: the largest consumer of step time by launch
                         # duration (the window weight gradients, on their side stream) and the whole step against the same peak
                         'largest_time_consumer': {k: fam['wgrad_win'][k] for k in ('kernel', 'achieved', 'frac', 'launches',
                                                                                    'avg_launch_us', 'share_of_step_time')},
                         'whole_step': ({'achieved': round(C3_STEP_TFLOP * B / 512 / ms_per_step * 1e3, 2), 'unit': 'TFLOP/s',
                                         'frac': round(C3_STEP_TFLOP * B / 512 / ms_per_step * 1e3 / MFMA_BF16_PEAK_TFLOPS, 4),
                                         'work': f'{C3_STEP_TFLOP} TFLOP per batch-512 step (SURVEY 8d: 10.47 GFLOP per sample)'}
                                        if c3 else None),
                         'alone': isolated_conv_rate(B, dev) if c3 else None,
                         'note': 'achieved / avg_launch_us: hipEvent-timed on the launch stream INSIDE every 10th timed step, i.e. '
                                 'while the profile branch and the weight-gradient kernels run beside it on other streams; '
                                 '"alone" is the same kernel on the four ResNet-18 body shapes with the GPU to itself; '
                                 'ms_per_step is wall clock over all steps, ms_per_step_median the median of per-step HIP '
                                 'event intervals (ms_per_step_median_unprofiled: steps without per-launch events)'},
        }
        if world == 1 and not args.no_cpu_baseline and c3:
            # host share of a 1-GPU box is 16 cores (the node reports all of them): never oversubscribe
            threads = host_cores()
            out['cpu_baseline'] = cpu_baseline(card, T, threads, B)
        print(json.dumps(out), flush=True)
    if world > 1:
        D.shutdown()


if __name__ == '__main__':
    main()
