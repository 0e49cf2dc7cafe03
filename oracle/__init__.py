"""CPU oracle for the MI355X FlashInfer hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product path (flashinfer-ai_amd/) never does -- it fails loudly without the HIP library instead.
"""
