import sys; sys.path.insert(0,'/root/repo')
import torch
from video_watermarking_forgery_detection_amd import ops
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)*1e3/n
x=torch.rand(16,3,256,256,device='cuda')
for kind,name in ((ops.BICUBIC,'bicubic'),(ops.BILINEAR,'bilinear')):
    for (H,W),out in (((256,256),(179,179)),((179,179),(256,256)),((256,256),(192,192))):
        xin=torch.rand(16,3,H,W,device='cuda'); y=ops.resample_fwd(xin,(0,H,0,W),out,kind,clamp01=True); gy=torch.randn_like(y)
        for yc in (None,y):
            a=t(lambda: ops.resample_bwd(gy,yc,(H,W),(0,H,0,W),kind,separable=True)); b=t(lambda: ops.resample_bwd(gy,yc,(H,W),(0,H,0,W),kind,separable=False))
            print(f"{name} {H}x{W}->{out[0]}x{out[1]} clampmask={yc is not None}: separable {a:.1f} us (incl. the workspace allocation), gather {b:.1f} us")
