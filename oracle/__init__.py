"""CPU oracle for the sliding-window 3D-U-Net inference hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the timed CPU baseline.  The
product path (``fast-nnunet_amd/``) never imports this package and fails loudly
when the HIP extension is missing.

What it restates (reference = /root/reference, 77even/Fast-nnUNet):

* ``oracle.sliding_window`` - tile starts, Gaussian importance map, padding,
  slicer order, fp16 accumulate / normalise, TTA mirroring, fold ensembling
  (distillation/nnunetv2/inference/sliding_window_prediction.py:10-54,
  distillation/nnunetv2/inference/predict_from_raw_data.py:470-680).
* ``oracle.unet`` - PlainConvUNet / ResidualEncoderUNet forward built from
  torch's own CPU primitives (conv3d / instance_norm / leaky_relu /
  conv_transpose3d), state-dict compatible with the key schema the reference's
  call sites imply (nnUNetDistillationTrainer.py:74-274).
* ``oracle.topology`` - plans.json -> architecture description, including the
  distilled-student rule ``max(f // r, 8)`` (nnUNetDistillationTrainer.py:678).

Parity pinning
--------------
* sliding-window part: PINNED.  ``tests/golden/make_golden.py`` imports the
  reference's own ``nnUNetPredictor`` (with import shims for the third-party
  packages that are absent in this image) and dumps golden vectors that
  ``tests/test_oracle_golden.py`` replays against this restatement.
* network part: "parity unpinned" beyond torch's primitives.  The network
  bodies live in ``dynamic_network_architectures`` (un-vendored, un-pinned
  dependency, distillation/setup.py:7-10; not installed here), and the
  reference holds no golden logits for it.  The restatement follows the
  constructor arguments at the reference's call sites and the published
  module structure of that package.
"""
