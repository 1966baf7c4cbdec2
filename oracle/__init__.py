"""CPU oracle for the LDMAE hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain-PyTorch / numpy restatement of the reference algorithm
(isno0907/ldmae) for the LightningDiT flow-matching train step and the VMAE
masked-token encoder.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  Nothing under ``ldmae_amd/`` imports it, and the product path has no
CPU fallback -- it raises when the HIP library is missing.

Parity pin: the reference ships no tests and no golden vectors,
so the oracle is pinned by golden vectors generated in the build container by
importing the reference itself from /root/reference with third-party shims
(``tests/golden/make_golden.py``; the vectors are committed under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks them).

Every function cites the reference file:line (relative to /root/reference) it
restates.
"""
