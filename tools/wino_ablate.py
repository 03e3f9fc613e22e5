#!/usr/bin/env python3
"""DIAGNOSTIC: where a tile block's time goes in the eight-wave Winograd kernel: time against the number of chunks per block
(Cin) at fixed tile count, with parts of the epilogue switched off (wrong results, valid timing)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_generation_amd import _lib, dev

def timeit(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

_lib.set_option("enc_wino", 1)
B, side = 4096, 8
L, M = side.bit_length() - 1, B * side * side
for Cout in (128, 128, 64):  # (the first pass warms the clocks up: its rows read high)
    for abl in (0, 2, 4, 6):
        row = []
        for Cin in (16, 32, 64, 128, 256):
            x = torch.randn(M, Cin, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") / 30
            u = None
            with _lib.option_scope(wino_waves=8 + 16 * abl):
                t = timeit(lambda: dev.conv_wino(x, w, 0, M, Cin, Cout, L, stats=True))
            row.append(t)
        tasks = (M // 4 // 64) * (Cout // 64) / 256  # block-tasks per CU
        tc = (row[4] - row[3]) / (128 / 8) / tasks; te = row[3] / tasks - 16 * tc
        print(f"Cout {Cout} abl {abl}: " + " ".join(f"{t:7.1f}" for t in row) + f" us for Cin 16..256   per chunk {tc:.2f} us, per block-task beside its chunks {te:.2f} us")
