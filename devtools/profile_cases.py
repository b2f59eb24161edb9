#!/usr/bin/env python3
"""Launch every attack-side kernel a few times, COLD (ring of buffers > 1.5 GB where the working set is smaller than
the 256 MiB Infinity Cache), for rocprofv3:

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cases -- python3 devtools/profile_cases.py
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 .../profile_cases.py     (and WRITE_SIZE, SQ counters)

Each case prints `CASE <tag> <kernel-name substring> <algorithmic bytes> <moved bytes>` so that
tools/summarize_profile.py --cases can price the kernel rows.  B=8, 512x512 throughout.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd"), os.path.join(ROOT, "robust-segmentation_amd", "tools")]
import torch  # noqa: E402

from semseg import _native as N  # noqa: E402
from k2_lab import make_set  # noqa: E402

N.lib()
B, H, W = 8, 512, 512
HW = H * W
REPS = int(os.environ.get("SEA_PROFILE_REPS", "6"))


def case(tag, kname, alg, moved, group=0, groups=1):
    """kname: regular expression on the demangled kernel name; group/groups: the case's launches are the group-th of
    `groups` equal runs of the matching launches in time order (cases that share one kernel instantiation)"""
    print(f"CASE {tag} | {kname} | {alg} | {moved} | {group} | {groups}", flush=True)


# ---- K2 ------------------------------------------------------------------------------------------------------------
for C in (21, 151):
    for dtype in (torch.float32, torch.bfloat16):
        s = 2 if dtype == torch.bfloat16 else 4
        for grad in (True, False):
            set_bytes = B * C * HW * s * (2 if grad else 1)
            nsets = max(2, -(-1600 * 2 ** 20 // set_bytes))
            sets = []
            for i in range(nsets):
                lg, y8 = make_set(C, dtype, i)
                sets.append((lg, y8, torch.empty_like(lg) if grad else None))
            w = torch.rand(C, device="cuda")
            pred = torch.empty(B, H, W, dtype=torch.uint8, device="cuda")
            ws = N.loss_workspace(B, HW, "cuda")
            for r in range(max(1, REPS // nsets) + 1):
                for lg, y8, dl in sets:
                    N.loss_fwd_bwd(lg, y8, w, 1, 3, 1.0 / HW, grad, pred=pred, workspace=ws, dlogits=dl, defer=True)
            torch.cuda.synchronize()
            dn = "float" if dtype == torch.float32 else "__hip_bfloat16"
            if grad and C == 151:
                kn = f"loss_nchw_split<{dn}, 151,"
            elif grad or C <= 32:
                kn = f"loss_nchw_reg<{dn}, {C}, \\d+, {'true' if grad else 'false'},"
            else:
                kn = f"loss_nchw_fwd<{dn},"
            case(f"K2 C={C} {str(dtype)[6:]} {'+grad' if grad else 'no-grad'} cold", kn,
                 B * HW * ((2 if grad else 1) * C * s + 16), B * HW * ((2 if grad else 1) * C * s + 2))
            del sets

# ---- K2u (low-res logits: the working set is small by construction; the kernel is not HBM bound) ----------------------
for gi, (C, hl, lab) in enumerate(((21, 128, "x4"), (151, 128, "x4"), (151, 32, "x16"))):
    low = torch.randn(B, C, hl, hl, device="cuda") * 3
    y8 = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear").max(1)[1].to(torch.uint8)
    w = torch.rand(C, device="cuda")
    dlow = torch.empty_like(low)
    pred = torch.empty(B, H, W, dtype=torch.uint8, device="cuda")
    for grad in (True, False):
        for r in range(REPS):
            N.loss_fwd_bwd_upsampled(low, y8, w, 1, 3, 1.0 / HW, grad, pred=pred, dlow=dlow if grad else None)
        torch.cuda.synchronize()
    # round 6: the power-of-two kernel loss_upsampled_pow2_kernel<S, lane slots of the class vector, GRAD> (+ a small combine
    # kernel for the cells' top / bottom partial rows, priced with the gather it completes: listed on its own line)
    S, NS = H // hl, -(-C // 64)
    # (called through the C ABI every class count takes the power-of-two kernel; the ATTACKER fuses only from 96 classes on)
    kt, kf = f"loss_upsampled_pow2_kernel<{S}, {NS}, true>", f"loss_upsampled_pow2_kernel<{S}, {NS}, false>"
    case(f"K2u C={C} {lab} +grad", kt, B * HW * (2 * C * 4 + 16), B * (2 * C * hl * hl * 4 + 2 * HW))
    case(f"K2u C={C} {lab} no-grad", kf, B * HW * (C * 4 + 16), B * (C * hl * hl * 4 + 2 * HW))

# ---- K1 / K5 / K6 / K4 (25 MB tensors: ring of 16 sets = 2 GB) ----------------------------------------------------------------
g = torch.Generator(device="cuda").manual_seed(1)
ring = [[torch.rand(B, 3, H, W, generator=g, device="cuda") for _ in range(5)] for _ in range(16)]
step = torch.full((B,), 16 / 255, device="cuda")
fl = torch.ones(3, B, dtype=torch.uint8, device="cuda")
fl[2] = 0
n = B * 3 * HW * 4
for r in range(2):
    for x, xa, xo, gr, out in ring:
        N.apgd_linf_step(x, xa, xo, gr, step, 8 / 255, 0.75, out=out)
    for x, xa, xo, gr, out in ring:
        N.pgd_linf_step(x, xa, gr, 1e-2, 4 / 255, delta_out=out, x_in_out=xo)
    for x, xa, xo, gr, out in ring:
        N.linf_project(xa, x, 8 / 255, out=out)
    for x, xa, xo, gr, out in ring:
        N.select_copy(fl, xa, gr, x, xo, out)
torch.cuda.synchronize()
case("K1 apgd_linf_step (4 in, 1 out) cold", "apgd_linf_step_v", 5 * n, 5 * n)
case("K6 pgd_linf_step (3 in, 2 out) cold", "[^a]pgd_linf_step", 5 * n, 5 * n)
case("K5 linf_project (2 in, 1 out) cold", "ew2_v4<1>", 3 * n, 3 * n)
case("K4 select_copy (2 in, 3 out) cold", "select_copy_v", 5 * n, 5 * n)
del ring

# ---- copy / read probes (1 GiB) ------------------------------------------------------------------------------------------
src = torch.empty(256 * 2 ** 20, dtype=torch.float32, device="cuda").normal_()
dst = torch.empty_like(src)
sink = torch.zeros(4096, device="cuda")
st = torch.cuda.current_stream().cuda_stream
L = N.lib()
for r in range(REPS):
    L.sea_probe_stream_copy(src.data_ptr(), dst.data_ptr(), src.numel() * 4, 1, st)
    L.sea_probe_stream_read(src.data_ptr(), sink.data_ptr(), src.numel() * 4, st)
torch.cuda.synchronize()
case("probe copy 1 GiB", "stream_copy_kernel", 2 * src.numel() * 4, 2 * src.numel() * 4)
case("probe read 1 GiB", "stream_read_kernel", src.numel() * 4, src.numel() * 4)
print("profile cases done")
