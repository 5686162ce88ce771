import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch, torch.nn.functional as F
from conftest import rel_err
from deeplip_amd import autograd as ag
torch.manual_seed(0)
for (M, C) in [(832, 1500), (832, 1472), (832, 1536), (64, 1500), (832, 508)]:
    x = (torch.randn(M, C) * 1.5 + 0.3).requires_grad_(); gamma = (torch.rand(C) + 0.5).requires_grad_(); beta = (torch.randn(C) * 0.2).requires_grad_()
    dy = torch.randn(M, C)
    ref = F.leaky_relu(F.batch_norm(x.double(), None, None, gamma.double(), beta.double(), training=True, eps=1e-5), 0.2)
    ref.backward(dy.double())
    xg = x.detach().cuda().requires_grad_(); gg = gamma.detach().cuda().requires_grad_(); bg = beta.detach().cuda().requires_grad_()
    y = ag.BNRowsActFn.apply(xg, gg, bg, torch.zeros(C).cuda(), torch.ones(C).cuda(), 0.1, 1e-5, 0.2, False)
    y.backward(dy.cuda()); torch.cuda.synchronize()
    eb = np.abs(bg.grad.cpu().numpy() - beta.grad.numpy())
    print(M, C, "y", rel_err(y.detach().cpu().numpy(), ref.detach().numpy()), "dx", rel_err(xg.grad.cpu().numpy(), x.grad.numpy()),
          "dgamma", rel_err(gg.grad.cpu().numpy(), gamma.grad.numpy()), "dbeta", rel_err(bg.grad.cpu().numpy(), beta.grad.numpy()), "worst col", int(eb.argmax()))
