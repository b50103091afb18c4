"""CPU: the C-ABI library builds/loads and exports every symbol include/mpr_hip.h (the boundary) and
include/mpr_hip_debug.h (tuning knobs / timing hooks) declare; calling a kernel wrapper with CPU tensors fails loudly
(there is no CPU fallback)."""
import os
import re

import pytest
import torch

from multimodal_plankton_recognition_amd import _native


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_native.LIB_PATH), 'run __graft_entry__.build() first'
    names = _native.exported_symbols()
    decl = {}
    for path in (_native.HEADER_PATH, _native.DEBUG_HEADER_PATH):
        text = open(path).read()
        decl[path] = set(re.findall(r'\b(mpr_\w+)\s*\(', re.sub(r'/\*.*?\*/', ' ', text, flags=re.S)))
    boundary, debug = decl[_native.HEADER_PATH], decl[_native.DEBUG_HEADER_PATH]
    assert boundary and debug and not (boundary & debug)
    assert boundary | debug == set(names)
    # nothing that makes results wrong, and no kernel-variant switch, is part of the drop-in boundary
    assert not any(n.startswith('mpr_conv_debug_') or 'variant' in n for n in boundary)
    assert _native.query('mpr_abi_version') == 1
    assert _native.lib().mpr_target_arch() == b'gfx950'


def test_size_queries_match_documented_tiling():
    rows = lambda *a: _native.query('mpr_conv_fwd_stat_rows', *a)
    # default: every tile adds into one of 4 slice rows (no pre-reduction launch before the BatchNorm that follows)
    assert rows(512, 56, 56, 64, 64, 3, 3, 1, 1, 1, 1) == 4
    old_slices = _native.query('mpr_conv_set_stat_slices', 0)      # one row per tile: the documented tilings
    try:
        _check_tile_rows(rows)
    finally:
        _native.query('mpr_conv_set_stat_slices', old_slices)
    assert _native.query('mpr_loss_workspace_floats') >= 1024


def _check_tile_rows(rows):
    # 3x3 / stride 1 / pad 1: shifted-window kernel, 256 positions of the padded raster (H+1) x (W+1) per tile
    # (N <= 64: the persistent form, one partial row per workgroup, at most 2 workgroups per CU)
    assert rows(512, 56, 56, 64, 64, 3, 3, 1, 1, 1, 1) == min(512, -(-512 * 57 * 57 // 256))
    assert rows(8, 56, 56, 64, 64, 3, 3, 1, 1, 1, 1) == -(-8 * 57 * 57 // 256)
    assert rows(512, 28, 28, 128, 128, 3, 3, 1, 1, 1, 1) == -(-512 * 29 * 29 // 256)
    old = _native.query('mpr_conv_set_window', 0)
    try:        # plain LDS-DMA implicit GEMM: 256-row tiles at N = 64, 128-row tiles above
        assert rows(512, 56, 56, 64, 64, 3, 3, 1, 1, 1, 1) == 512 * 56 * 56 // 256
        assert rows(512, 28, 28, 128, 128, 3, 3, 1, 1, 1, 1) == 512 * 28 * 28 // 128
    finally:
        _native.query('mpr_conv_set_window', old)
    assert rows(512, 28, 28, 128, 64, 3, 3, 2, 2, 1, 1) == 512 * 28 * 28 // 128          # stride 2: never the window kernel
    assert rows(4, 28, 28, 128, 128, 3, 3, 1, 1, 1, 1) == 4 * 28 * 28 // 128 + 1          # small problem: register-staged kernel


def test_no_cpu_fallback():
    from multimodal_plankton_recognition_amd import ops
    with pytest.raises(_native.NativeLibraryError):
        ops.gemm(torch.zeros(4, 4), torch.zeros(4, 4))
    from multimodal_plankton_recognition_amd.coordination import CLIPLoss
    with pytest.raises(_native.NativeLibraryError):
        CLIPLoss()(torch.randn(4, 8), torch.randn(4, 8), 1)


def test_modules_keep_reference_state_dict_keys(golden):
    from multimodal_plankton_recognition_amd.profile_encoder import ProfileCNN
    from multimodal_plankton_recognition_amd.model import MultiModel
    g = golden('profile_cnn_b8_2222')
    m = ProfileCNN(dim_in=6, blocks=[2, 2, 2, 2], base_channels=8)
    assert sorted(m.state_dict()) == sorted(k[3:] for k in g if k.startswith('sd.'))
    mm = MultiModel(dim_embed=32, image_encoder_args=dict(name='resnet18'),
                    profile_encoder_args=dict(dim_in=6, blocks=[1, 1, 1, 1], base_channels=8),
                    coordination_args=dict(method='siglipplus'), optim_args=dict(lr=1e-3))
    keys = set(mm.state_dict())
    for k in ('image_encoder.backbone.conv1.weight', 'image_encoder.backbone.layer2.0.downsample.1.running_var',
              'image_projection.weight', 'profile_encoder.layer4.0.downsample.0.weight', 'profile_projection.weight',
              'loss.siglip.logit_scale', 'loss.siglip.bias'):
        assert k in keys, k
    assert mm.image_projection.weight.shape == (32, 514) and mm.profile_projection.weight.shape == (32, 65)
    with pytest.raises(Exception, match='Coordination loss not found'):
        MultiModel(dim_embed=32, image_encoder_args=dict(name='resnet18'),
                   profile_encoder_args=dict(dim_in=6, blocks=[1, 1, 1, 1], base_channels=8),
                   coordination_args=dict(method='nope'), optim_args=dict(lr=1e-3))
