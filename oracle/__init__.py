"""CPU oracle for the ssd_amd hot path.  TEST INFRASTRUCTURE ONLY -- see oracle/ops.py header."""
