"""CPU oracle for the PICCOLO sampling-loss hot path — TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
piccolo_amd/ (the product) never does.
"""
