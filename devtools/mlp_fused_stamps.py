#!/usr/bin/env python3
"""Where a loop iteration of the fused MLP kernel (C = 96) spends its cycles: the diagnostic build with s_memtime stamps.
    python devtools/mlp_fused_stamps.py [fwd|bwd]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch  # noqa: E402
from semseg import _native as N  # noqa: E402

bwd = len(sys.argv) > 1 and sys.argv[1] == "bwd"
C, M = 96, 131072
H = 4 * C
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(M, C, generator=g, device="cuda")
w1 = torch.randn(H, C, generator=g, device="cuda") * 0.05
b1 = torch.randn(H, generator=g, device="cuda") * 0.1
w2 = torch.randn(C, H, generator=g, device="cuda") * 0.03
b2 = torch.randn(C, generator=g, device="cuda") * 0.1
res = torch.randn(M, C, generator=g, device="cuda")
gy = torch.randn(M, C, generator=g, device="cuda") * 1e-3
word = lambda v: torch.tensor([float(v)], dtype=torch.float32, device="cuda").view(torch.int32)  # noqa: E731
a1, a2 = word(x.abs().max().item() * 1.7), word((x.abs().max() * w1.abs().sum(1) + b1.abs()).max().item())
mul = (w2.abs().sum(0).max() * 1.13).float().reshape(1)
P1, P2 = N.gemm_split_pack(w1, terms=22), N.gemm_split_pack(w2, terms=22)
P2t, P1t = N.gemm_split_pack(w2, trans=True, terms=22), N.gemm_split_pack(w1, trans=True, terms=22)
y = torch.empty(M, C, device="cuda")
WPB = 4   # waves per block of the C = 96 kernels
nw = (M + 32 * WPB - 1) // (32 * WPB) * WPB
dbg = torch.zeros(nw, 8, dtype=torch.int64, device="cuda")
p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
for _ in range(30):   # sustained load: the clock the loop really runs at
    N._check(N.lib().sea_mlp_fused_stamps(int(bwd), p(gy), C, p(x), C, p(P1.data), p(b1), p(P2t.data if bwd else P2.data), p(P1t.data),
                                          p(b2), p(res), p(y), M, C, p(a1), p(a2), p(mul), p(dbg), N._stream()), "stamps")
torch.cuda.synchronize()
d = dbg.double().cpu()
names = ["(epilogue, whole kernel: see below)", "interleaved matrix + element-wise work" + (" + first products" if bwd else ""), "wait for the DMA", "barrier"]
tot = d[:, 4].mean().item()
print(f"{'backward' if bwd else 'forward'} C={C} M={M}: {nw} waves, 13 loop iterations, loop total {tot:.0f} ticks per wave (s_memtime = shader cycles)")
for k, n in list(enumerate(names))[1:]:
    print(f"   {n:40s} {d[:, k].mean().item() / 13:8.0f} ticks per iteration   ({100 * d[:, k].mean().item() / tot:4.1f} %)   "
          f"min wave {d[:, k].min().item() / 13:7.0f}  max wave {d[:, k].max().item() / 13:7.0f}")
print(f"   prologue {d[:, 7].mean().item():.0f} ticks per wave (min {d[:, 7].min().item():.0f}, max {d[:, 7].max().item():.0f}); epilogue "
      f"{d[:, 0].mean().item():.0f} (min {d[:, 0].min().item():.0f}, max {d[:, 0].max().item():.0f}); loop {tot:.0f}")
blk = d.view(-1, WPB, 8)
first_round = blk[:256 * 8 // WPB]
print(f"   first 256 blocks: prologue {first_round[:, :, 7].mean().item():.0f}, later blocks: {blk[256 * 8 // WPB:, :, 7].mean().item():.0f}; "
      f"epilogue first {first_round[:, :, 0].mean().item():.0f}, later {blk[256 * 8 // WPB:, :, 0].mean().item():.0f}")
start = d[:, 5]
print(f"   start stamps span {(start.max() - start.min()).item():.0f} ticks over the grid; barrier per wave of a block: "
      + " ".join(f"{blk[:, w, 3].mean().item() / 13:.0f}" for w in range(WPB)))
