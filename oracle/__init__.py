"""CPU oracle for the train_multi hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it, and only as the checker / timed CPU
baseline.  The shipped package ``multimodal_plankton_recognition_amd`` never
imports this package and raises when its HIP library is missing.

What it is: a plain-torch fp32 *functional* restatement (state_dict in,
tensors out) of the reference algorithm for the path named in
BASELINE.json:north_star.  Every function cites the reference file:line it
follows (paths relative to /root/reference).

Pinning status:
  * coordination losses, ProfileCNN, ProfileTransformer, ProfileLSTM and the
    bias-free projection heads are PINNED: tests/golden/*.npz were generated
    by importing the reference's own ``src/coordination.py`` and
    ``src/profile_encoder.py`` in the build container
    (tests/golden/make_golden.py, run with ``python3 -B``) and
    tests/test_oracle_golden.py checks this restatement against them.
  * the image backbone (ResNet-18 / ViT) lives in un-vendored, un-pinned
    ``timm`` (call site src/image_encoder.py:16,24) which is absent here, and
    the reference holds no test or golden vector at that boundary:
    image-branch parity is UNPINNED -- the restatement follows the published
    torchvision/timm ResNet-18 BasicBlock topology and is checked only against
    torch.nn building blocks.
"""
