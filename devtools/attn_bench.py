#!/usr/bin/env python3
"""fp32 attention of Segmenter ViT-S/16 at 512x512 (B=8, 6 heads x 64, 1025 tokens): libsea_hip M7 (fp32 MFMA flash
attention) vs ATen scaled_dot_product_attention, forward and forward+backward, interleaved rounds, random data."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from semseg import _native as N  # noqa: E402

N.lib()
for B, H, T in ((8, 6, 1025), (8, 6, 1175)):
    D, scale = 64, 64 ** -0.5
    qkv = torch.randn(B, T, 3, H, D, device="cuda")
    gout = torch.randn(B, T, H * D, device="cuda")
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    flop_f = 4.0 * B * H * T * T * D
    flop_b = 10.0 * B * H * T * T * D   # the 5 products a flash backward needs (M7 runs 7: S recomputed twice)

    def hip_f():
        return N.attention_qkv(qkv, scale)

    o, lse = hip_f()

    def hip_b():
        return N.attention_qkv_backward(qkv, o, lse, gout, scale)

    qa = qkv.detach().clone().requires_grad_(True)

    def aten_f():
        qq, kk, vv = qa.permute(2, 0, 3, 1, 4)
        return F.scaled_dot_product_attention(qq, kk, vv, scale=scale)

    ya = aten_f()

    def aten_b():
        return torch.autograd.grad(ya, [qa], grad_outputs=gout.view(B, T, H, D).permute(0, 2, 1, 3), retain_graph=True)

    fns = {"M7 fwd": (hip_f, flop_f), "ATen sdpa fwd": (aten_f, flop_f), "M7 bwd": (hip_b, flop_b), "ATen sdpa bwd": (aten_b, flop_b)}
    for f, _ in fns.values():
        f()
    torch.cuda.synchronize()
    ts = {k: [] for k in fns}
    for _ in range(9):
        for kname, (f, _) in fns.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                f()
            b.record()
            torch.cuda.synchronize()
            ts[kname].append(a.elapsed_time(b) / 4)
    for kname, (f, fl) in fns.items():
        ms = sorted(ts[kname])[4]
        print(f"B={B} H={H} T={T}  {kname:16s} {ms * 1e3:8.1f} us   {fl / ms / 1e9:7.1f} TFLOP/s (5-product flop count for bwd)", flush=True)
