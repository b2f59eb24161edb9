#!/usr/bin/env python3
"""Golden trajectories of the reference's L2 branch of apgd_train (build container only):

    python oracle/gen_l2_goldens.py

Runs nmndeep/Robust-Segmentation's own ``semseg.attacker.apgd_train(norm="L2")`` (semseg/attacker.py:412-436 for the step,
323-339 and 528-551 for the schedule it shares with L-inf) on the tiny seeded models of oracle/tiny_models.py and writes
tests/golden/g14_apgd_l2_*.npz: inputs, start point and the four returned tensors.  Only data is written.  The reference
draws no random start for L2 (attacker.py:291-294 handles L-inf only), so every run starts from a supplied ``x_init``."""
import contextlib
import io
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SEA_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(HERE, "shims"), REF, ROOT]

import numpy as np  # noqa: E402
import torch  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def npz(name, **arrs):
    np.savez_compressed(os.path.join(OUT, name + ".npz"),
                        **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print("wrote", name)


def main():
    os.chdir(REF)
    torch.set_num_threads(2)
    import semseg.attacker as A
    from autoattack.other_utils import Logger
    from semseg.utils.utils import VOC_WTS
    from oracle.tiny_models import PointwiseNet, TinyConvNet, make_labels
    logger = Logger(None)
    for netname, Net in (("conv", TinyConvNet), ("pw", PointwiseNet)):
        for C in (5, 21):
            net = Net(C, seed=C)
            g = torch.Generator().manual_seed(1400 + C)
            x = torch.rand(3, 3, 16, 16, generator=g)
            y = make_labels(net, x, ignore_frac=0.05, flip_frac=0.1, seed=C)
            w = torch.tensor(VOC_WTS) if C == 21 else torch.rand(C, generator=g)
            eps = 0.75
            d = torch.randn(x.shape, generator=g)
            x_init = (x + 0.5 * eps * d / d.flatten(1).norm(dim=1).view(-1, 1, 1, 1)).clamp(0, 1)
            for loss in ("mask-ce-avg", "mask-ce-bal", "js-avg"):
                for n_iter in (10, 25):
                    with contextlib.redirect_stdout(io.StringIO()):
                        xb, acc, lb, xba = A.apgd_train(net, x, y, "L2", eps, n_iter=n_iter, use_rs=False, loss=loss,
                                                        track_loss="ce-avg", logger=logger, x_init=x_init, num_classes=C,
                                                        weights=w, early_stop=True)
                    npz(f"g14_apgd_l2_{netname}_C{C}_{loss}_{n_iter}", x=x, y=y, w=w, x_init=x_init, eps=np.float64(eps),
                        x_best=xb, acc=acc, loss_best=lb, x_best_adv=xba)


if __name__ == "__main__":
    main()
