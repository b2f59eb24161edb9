"""Batch-composition probe: does image 0 get bitwise the same logits / input gradient whatever shares its batch?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "robust-segmentation_amd")]
from semseg.models import UperNetForSemanticSegmentation
from semseg import attacker as A
torch.manual_seed(0)
m = UperNetForSemanticSegmentation("ConvNeXt-T_CVST", 21, None).cuda().eval()
x = torch.rand(4, 3, 512, 512, device="cuda")
def run(idx):
    xb = x[idx].contiguous()
    xi, lg = A._forward_logits(m, xb, True)
    g = torch.randn(lg.shape[1:], device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).expand_as(lg).contiguous()
    (gx,) = torch.autograd.grad(lg, [xi], grad_outputs=g)
    return lg.detach()[0].clone(), gx[0].clone()
ref = run([0, 1])
for idx in ([0, 1], [0, 2], [0, 3], [0], [0, 1, 2]):
    l, g = run(idx)
    print(idx, "logits equal", torch.equal(l, ref[0]), "grad equal", torch.equal(g, ref[1]), (l - ref[0]).abs().max().item(), (g - ref[1]).abs().max().item())
# leaf-level: first module whose output for image 0 depends on the batch
def leaves(idx):
    acts = []
    hs = [mm.register_forward_hook(lambda mod, i, o, a=acts: a.append((mod.__class__.__name__, o.detach()[0].clone())) if torch.is_tensor(o) else None)
          for mm in m.modules() if len(list(mm.children())) == 0]
    with torch.no_grad(), A._FrozenParameters(m):
        m(x[idx].contiguous())
    for h in hs: h.remove()
    return acts
a, b = leaves([0, 1]), leaves([0, 2])
for i, ((n1, t1), (n2, t2)) in enumerate(zip(a, b)):
    if t1.shape == t2.shape and not torch.equal(t1, t2):
        print("first batch-dependent leaf:", i, n1, tuple(t1.shape), (t1 - t2).abs().max().item()); break
else:
    print("all leaves batch-independent (no-grad forward)")
