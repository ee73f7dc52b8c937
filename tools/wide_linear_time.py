"""Time of the wide first RoI Linear (512 x 20 736 x 256): own fp32-MFMA kernels against the library products (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, time
from glenet_amd import dense_path as dp
dev=torch.device('cuda')
x=torch.randn(512,20736,device=dev); w=torch.randn(256,20736,device=dev)/144; gy=torch.randn(512,256,device=dev)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n*1e3
print('own fwd %.1f us  dgrad %.1f  wgrad %.1f' % (t(lambda: dp.wide_linear_forward(x,w)), t(lambda: dp.wide_linear_input_grad(gy,w)), t(lambda: dp._SplitKLinearFn.weight_grad(x,gy,w))))
dp.OWN_WIDE_LINEAR=False
class C: 
    def save_for_backward(self,*a): pass
print('lib fwd %.1f us  dgrad %.1f  wgrad %.1f' % (t(lambda: dp._SplitKLinearFn.forward(C(),x,w)), t(lambda: gy@w), t(lambda: dp._SplitKLinearFn.weight_grad(x,gy,w))))
