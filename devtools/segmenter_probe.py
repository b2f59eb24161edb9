#!/usr/bin/env python3
"""Diagnostic: where one forward + input-gradient backward of Segmenter ViT-S/16 (BASELINE configs[2]) spends
its time (torch profiler, grouped by op and input shape)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402

import bench  # noqa: E402

torch.backends.cudnn.benchmark = True


def main():
    torch.manual_seed(0)
    model = bench.make_model("vit_small_patch16_224", 151).eval().cuda()
    for p in model.parameters():
        p.requires_grad_(False)
    x = torch.rand(8, 3, 512, 512, device="cuda")
    with torch.no_grad():
        low, size = model.forward_lowres(x)
    dl = torch.randn_like(low)

    def step():
        xi = x.clone().requires_grad_(True)
        y, _ = model.forward_lowres(xi)
        (g,) = torch.autograd.grad(y, xi, dl)
        return g

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"forward_lowres + input gradient: {(time.perf_counter() - t0) * 100:.2f} ms; low-res logits {tuple(low.shape)}")
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=40,
                                                              max_name_column_width=48, max_shapes_column_width=70))


if __name__ == "__main__":
    main()
