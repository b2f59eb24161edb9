import sys; sys.path[:0]=['/root/repo','/root/repo/robust-segmentation_amd']
import torch
from semseg import _native as N
N.lib()
for C in (96,192):
    M=1000
    g=torch.Generator(device='cuda').manual_seed(C)
    x=torch.randn(M,C,generator=g,device='cuda')*3+1
    lw=torch.rand(C,generator=g,device='cuda')+0.5; lb=torch.randn(C,generator=g,device='cuda')*0.1
    yn,mean,rstd=N.layernorm(x,lw,lb,1e-6)
    y2=torch.empty_like(x); m2=torch.empty(M,device='cuda'); r2=torch.empty(M,device='cuda')
    N._check(N.lib().sea_probe_ln_rows(x.data_ptr(),lw.data_ptr(),lb.data_ptr(),1e-6,M,C,y2.data_ptr(),m2.data_ptr(),r2.data_ptr(),N._stream()),'probe')
    torch.cuda.synchronize()
    print(C,'mean equal',torch.equal(mean,m2),(mean-m2).abs().max().item(),'rstd equal',torch.equal(rstd,r2),(rstd-r2).abs().max().item(),'yn equal',torch.equal(yn,y2),(yn-y2).abs().max().item(), 'rows with mean diff', int((mean!=m2).sum()), 'rstd diff', int((rstd!=r2).sum()))
    # check sum order hypothesis on CPU double-free: compute leaf sums
