"""CPU oracle for the TUCH contact path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  The product (tuch_amd/) never does.
"""
