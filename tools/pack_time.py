import torch, sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from transmf_ad_amd import ops
dev='cuda:0'
for cin,cout in ((32,32),(32,64),(64,64),(64,128),(128,256)):
    w=torch.randn((cout,cin,3,3,3),device=dev)
    for _ in range(3): ops.pack_weights_wino(w,True,True)
    torch.cuda.synchronize(); e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.pack_weights_wino(w,True,True)
    e1.record(); e1.synchronize()
    print(cin,cout,'pack us', e0.elapsed_time(e1)/20*1e3)
