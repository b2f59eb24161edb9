import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
import torch
import bench
from semseg import attacker as A
from semseg.utils.utils import ADE_WTS
dev = torch.device("cuda", 0)
torch.backends.cudnn.benchmark = True
model, x, y = bench.build_case(0, 8, 151, "ConvNeXt-S_CVST", dev)
w = torch.tensor(ADE_WTS, device=dev)
for n_iter, tag in ((26, "n_iter 26"), (40, "n_iter 40"), (301, "n_iter 301")):
    run = A.ApgdRun(model, x, y, 8 / 255, n_iter, "mask-ce-bal", "ce-avg", True, 151, w, x.clone())
    run.start()
    for i in range(5):
        run.step(i)
    torch.cuda.synchronize()
    ts = []
    t0 = time.perf_counter()
    for i in range(5, 25):
        t = time.perf_counter()
        run.step(i)
        ts.append((time.perf_counter() - t) * 1e3)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(tag, "enqueue %.2f ms/step wall %.2f ms/step" % (t_enq * 50, (time.perf_counter() - t0) * 50), "per-step host ms:", " ".join("%.0f" % v for v in ts), flush=True)
