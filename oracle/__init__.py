"""CPU oracle for the signature-PDE solver -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package (as the checker / timed CPU baseline). The product package
``sigkernel_amd`` never does.
"""
from .oracle import *  # noqa
