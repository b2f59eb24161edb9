#!/usr/bin/env python3
"""What stays allocated on the device after tools.infer returns?  (lists live CUDA tensors by size)"""
import gc
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "robust-segmentation_amd")
sys.path[:0] = [ROOT, PKG]
os.chdir(PKG)
import torch  # noqa: E402
import yaml  # noqa: E402

from tools import infer  # noqa: E402

td = tempfile.mkdtemp()
cfg = yaml.safe_load(open(os.path.join(PKG, "configs", "pascalvoc_convnext.yaml")))
cfg["SAVE_DIR"] = td + "/"
yaml.safe_dump(cfg, open(td + "/cfg.yaml", "w"))
torch.cuda.synchronize()
base = torch.cuda.memory_allocated()
for rep in range(int(os.environ.get("REPS", "2"))):
  infer.main(["--cfg", td + "/cfg.yaml", "--eps", "8", "--n_iter", "40", "--synthetic", "8", "--image_size", "512", "--batch_size", "8",
            "--deterministic", "--attack", "mask-ce-bal"])
  gc.collect()
  torch.cuda.empty_cache()
  print("after evaluation", rep, "held MiB", (torch.cuda.memory_allocated() - base) / 2 ** 20, flush=True)
print("held MiB", (torch.cuda.memory_allocated() - base) / 2 ** 20, "reserved MiB", torch.cuda.memory_reserved() / 2 ** 20)
seen = {}
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            st = o.untyped_storage()
            seen[st.data_ptr()] = (st.nbytes(), tuple(o.shape), o.dtype, [type(r).__name__ for r in gc.get_referrers(o)][:6])
    except Exception:
        pass
for ptr, (nb, shape, dt, refs) in sorted(seen.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{nb / 2 ** 20:9.1f} MiB  {shape} {dt}  referrers {refs}")
