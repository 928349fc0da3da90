#!/usr/bin/env python3
"""Store rate of target-block candidates by allocation depth: K scattered blocks of 4.4 GB allocated one after the other and
all held -- does the rate depend on how far into the device's memory a block lies?   python tools/depth_probe.py [K]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mate_amd import _native  # noqa: E402
K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N, R, row = 4096, 256, 4192
held, rates = [], []
for k in range(K):
    b = _native.ScatteredBlock(0, N * R * row)
    rates.append(round(b.store_rate(N, row)))
    held.append(b)
print('GB/s by allocation order (4.4 GB each):', rates)
