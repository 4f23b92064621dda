"""CPU oracle for the POPE hot path (TEST INFRASTRUCTURE ONLY).

This package is a CPU (torch fp32 / numpy) restatement of the reference
algorithm for the hot path: DINOv2 ViT-S/14 forward, the dense dual-softmax /
mutual-nearest-neighbour matcher and the streaming top-3 proposal vote.

It is the *checker*, never the product:
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it;
  * nothing under ``pope_amd/`` imports it, and the product path raises when
    the HIP library is missing instead of falling back to this code.

Pinning: the restatement is validated against the reference's own Python code
imported from ``/root/reference`` in the build container
(``oracle/gen_golden.py``), with seeded synthetic weights in the reference's
state-dict layout; the resulting vectors are committed under ``tests/golden``.
The reference has no test-suite/golden vectors of its own.
"""
