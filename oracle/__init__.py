"""CPU oracle for the DVAE + GRBM training path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker.  The product package
(``image-generation_amd/``) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):

* ``oracle.nets``   – encoder / decoder restatement.  PINNED against the
  reference's own importable modules (/root/reference/src/encoder.py:18-49,
  /root/reference/src/decoder.py:18-62) through the fixtures written by
  ``tests/golden/make_golden.py``.
* ``oracle.common`` – heaviside latent_to_discrete, greedy_get_subgraph,
  get_graph_mapping, train_grbm schedule, push_to_deque: PINNED against
  /root/reference/src/utils/common.py and friends through fixtures.
* ``oracle.plugin`` – DiscreteVariationalAutoencoder, GraphRestrictedBoltzmannMachine,
  GaussianKernel, maximum_mean_discrepancy_loss.  These live in the un-vendored
  PyPI package ``dwave-pytorch-plugin~=0.3`` (/root/reference/requirements.txt:4)
  which is absent from this image and from /root/reference: **parity unpinned**.
  The restatement follows the published API and README maths; every
  assumption is a named switch.
* ``oracle.gibbs`` / ``oracle/gibbs_ref.c`` – the block-Gibbs sampler that
  stands in for the QPU draw.  The reference has no sampler
  (/root/reference/src/utils/common.py:123-138 calls a D-Wave QPU), so this
  restatement IS the definition; it is pinned by Philox known-answer vectors and
  exact-enumeration statistics, and the HIP kernel must match it bit for bit.
"""
