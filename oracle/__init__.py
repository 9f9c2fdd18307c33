"""CPU oracle for the csl-gan DP discriminator step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``csl_gan_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and there only as the checker / timed baseline.

The oracle is a plain PyTorch-CPU fp32 (optionally fp64) restatement of the
reference path named by BASELINE.json ``north_star``.  Each function cites the
reference ``file:line`` it follows (paths relative to /root/reference).

Pinning status (see DESIGN.md "Oracle"):
  * gradient penalty, ``Discriminator.aux_loss``, ``Logger`` — pinned against the
    reference's own importable modules (tests/golden/make_golden.py runs them in
    the build container and commits the vectors).
  * model stacks (DCResNet / MNIST vanilla; init order, G forward, D forward, G.loss,
    train_G gradients) — pinned against the reference's own classes, executed from their
    source in the build container (tests/golden/make_golden.py: reference_model_classes);
    vectors in tests/golden/model_*.npz and upsample_conv.npz, checked by tests/test_oracle_golden.py.
  * per-sample gradients / clip / noise / immediate sensitivity — the arithmetic
    lives in the un-pinned third-party fork ``git+git://github.com/twosixlabs/opacus``
    which is not in the container: PARITY UNPINNED at that boundary.  The oracle
    follows the mathematical definition (micro-batch-of-one autograd) and the
    semantics written down in SURVEY.md §8 (a7)-(a13).
"""
