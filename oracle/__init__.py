"""CPU oracle for the EgoNet hot path (TEST INFRASTRUCTURE ONLY).

This package is a CPU restatement of the reference algorithm for the path
HRNet heat-map backbone -> key-point decode -> FC lifter -> pose solve.  It is
the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under
``egonet_amd/`` (the product) imports it, and the product raises when its HIP
library is missing instead of routing here.

Pinning: the reference ships no tests, golden vectors or checkpoints
(SURVEY.md section 4), so the oracle is pinned against outputs of the
reference itself, produced in the build container by importing
``/root/reference`` (``tests/golden/make_golden.py``) and committed as
fixtures under ``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every
oracle function against those fixtures.
"""
