"""CPU oracle for the pix2pixHD audio-SR hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (numpy fp64 for the MDCT family, plain
torch-CPU fp32 functional ops for the conv networks) of the reference
algorithms on the north-star path.  It exists so that the HIP kernels can be
checked against something that runs anywhere.

Rules (enforced by tests/test_no_oracle_in_product.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import anything from here;
  * nothing under ``pix2pixhdaudiosr_amd/`` may import it -- the product path
    fails loudly when the HIP library is missing instead of falling back.

Pinning: every function here is checked against golden vectors under
``tests/golden/`` that were produced by importing the reference itself
(``tools/gen_golden.py``, run in the build container where ``/root/reference``
is mounted) and against the reference's own known-answer values (SURVEY.md
section 4: DCT/dB KATs, parameter counts, round-trip MSE).
"""
