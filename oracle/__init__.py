"""CPU oracle for the render-and-compare path.  TEST INFRASTRUCTURE ONLY (see oracle.py)."""
