import torch, time
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev='cuda'
print([n for n in dir(torch.ops.aten) if 'miopen' in n][:20])
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e6
for (ci,co,k,H,W,s) in [(64,64,3,112,200,1),(64,256,1,112,200,1),(256,64,1,112,200,1),(128,128,3,56,100,1),(512,128,1,56,100,1),(256,256,3,28,50,1),(1024,256,1,28,50,1)]:
    x=torch.randn(6,ci,H,W,device=dev).half().contiguous(memory_format=torch.channels_last)
    w=(torch.randn(co,ci,k,k,device=dev)*0.05).half().contiguous(memory_format=torch.channels_last)
    b=torch.randn(co,device=dev).half()
    z=torch.randn(6,co,H,W,device=dev).half().contiguous(memory_format=torch.channels_last)
    p=k//2
    t0=bench(lambda: F.conv2d(x,w,None,1,p))
    t1=bench(lambda: torch.relu_(F.conv2d(x,w,b,1,p)))
    try:
        t2=bench(lambda: torch.ops.aten.miopen_convolution_relu(x,w,b,[1,1],[p,p],[1,1],1))
        y1=torch.relu_(F.conv2d(x,w,b,1,p)); y2=torch.ops.aten.miopen_convolution_relu(x,w,b,[1,1],[p,p],[1,1],1)
        e2=(y1.float()-y2.float()).abs().max().item()
    except Exception as e:
        t2=-1; e2=str(e)[:80]
    try:
        t3=bench(lambda: torch.ops.aten.miopen_convolution_add_relu(x,w,z,1.0,b,[1,1],[p,p],[1,1],1))
        t3b=bench(lambda: torch.relu_(F.conv2d(x,w,b,1,p).add_(z)))
    except Exception as e:
        t3=-1; t3b=str(e)[:80]
    print(f'{ci}->{co} k{k} {H}x{W}: conv only {t0:.1f} us, conv+bias+relu {t1:.1f}, fused {t2:.1f} (diff {e2}); +add: separate {t3b}, fused {t3}')
