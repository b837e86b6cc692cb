"""CPU oracle for the RDFC-GAN train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``rdfc_gan_amd/`` (the product) may
import this package.  The only legitimate users are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``, and
there only as the checker / the thing timed as a CPU baseline -- never as a
fallback for a missing HIP extension.

What is here
------------
``nets``     plain ``torch.nn`` restatement of G_A2B (RDFGenerator), G_B2A
             (ResnetGenerator) and PatchGAN with the reference's state_dict keys.
``mdconv``   modulated deformable convolution fwd/bwd: a plain-C restatement of
             the reference's CUDA kernels (``mdconv_oracle.c``) plus a pure
             torch formulation used to cross-check it.
``nlspn``    NLSPN (offset/affinity/TGASS/propagation) on top of ``mdconv``.
``losses``   GANLoss / L1 / L2 weighted-sum reductions.
``step``     the RDFCGAN.optimize_parameters schedule on the hot-path nets.

Pinning status (see DESIGN.md "Oracle"):
* nets / losses / step: pinned against the imported reference (run in the build
  container by ``tests/golden/make_golden.py``; outputs committed as fixtures).
* mdconv: the reference's DCN extension is CUDA-only (its CPU entry points are
  ``AT_ERROR`` stubs), so no reference output can be produced here.  It is pinned
  by the reference's own property tests (``deformconv/test.py``: zero-offset ==
  conv2d, identity kernel, im2col-step invariance, finite-difference gradcheck)
  and by an independent autograd formulation.  "parity unpinned" for absolute
  NLSPN values beyond those properties.
"""
