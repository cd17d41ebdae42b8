#!/usr/bin/env python3
"""Micro-benchmark of the conv GEMM kernel variants on the bench's shapes (CIFAR, B=64).
Times nfk_conv through the C ABI with HIP events on the launch stream; interleaved rounds in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nf_distillation_amd import capi  # noqa: E402
from nf_distillation_amd.capi import check, ptr  # noqa: E402

lib = capi.lib()
dev = "cuda"


def run(shape, tune, iters=30):
    name, B, H, W, Cin, taps, N, epi = shape
    M = B * H * W
    K = taps * Cin
    Kp = lib.nfk_padded_k(K)
    x = torch.randn(M, Cin, device=dev)
    w = torch.randn(N, Cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1, device=dev) * 0.05
    wt = torch.zeros(N * Kp, device=dev)
    check(lib.nfk_repack(ptr(w), N, Cin, taps, ptr(wt), None, capi.stream()))
    S = lib.nfk_conv_splitk(M, N, K) if epi == 0 else 1
    out = torch.empty(S * M * N, device=dev)
    e0, e1 = torch.randn(N, device=dev), torch.rand(N, device=dev) + 0.5
    lib.nfk_tune(capi.TUNE_F32_TILES, tune)
    s = capi.stream()
    for _ in range(3):
        check(lib.nfk_conv(ptr(x), Cin, B, H, W, Cin, taps, ptr(wt), N, ptr(out), N, epi, ptr(e0), ptr(e1), s))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        check(lib.nfk_conv(ptr(x), Cin, B, H, W, Cin, taps, ptr(wt), N, ptr(out), N, epi, ptr(e0), ptr(e1), s))
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / iters
    lib.nfk_tune(capi.TUNE_F32_TILES, 0)
    return us, 2.0 * M * N * K / (us * 1e-6) / 1e12


SHAPES = [
    ("conv2 l1 1x1 512->512 M=16384", 64, 16, 16, 512, 1, 512, 1),
    ("conv1 l1 3x3 6->512   M=16384", 64, 16, 16, 6, 9, 512, 1),
    ("conv2 l2 1x1 512->512 M=4096 ", 64, 8, 8, 512, 1, 512, 1),
    ("conv1 l2 3x3 12->512  M=4096 ", 64, 8, 8, 12, 9, 512, 1),
    ("conv2 l3 1x1 512->512 M=1024 ", 64, 4, 4, 512, 1, 512, 1),
    ("conv3 l1 3x3 512->12  M=16384", 64, 16, 16, 512, 9, 12, 0),
    ("conv3 l2 3x3 512->24  M=4096 ", 64, 8, 8, 512, 9, 24, 0),
    ("conv3 l3 3x3 512->48  M=1024 ", 64, 4, 4, 512, 9, 48, 0),
]
TUNES = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4".split(","))]

if __name__ == "__main__":
    for shp in SHAPES:
        res = []
        for rnd in range(2):
            for t in TUNES:
                us, tf = run(shp, t)
                if rnd == 1:
                    res.append(f"t{t}: {us:7.1f}us {tf:6.1f}TF")
        print(shp[0], " | ".join(res), flush=True)
