"""Times the implicit-GEMM 3x3 convolution (di_conv3x3_fwd) against the library convolution on the hot path's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for (n, Cin, H, W, Cout) in [(6, 256, 112, 200, 128), (1, 512, 180, 180, 128), (1, 128, 180, 180, 128), (1, 128, 180, 180, 10)]:
    g = torch.Generator(device='cuda').manual_seed(0)
    x = torch.randn(n, Cin, H, W, device='cuda', generator=g).half().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(Cin, Cout, 3, padding=1).cuda().half()
    packed = ops.pack_conv3x3(conv.weight, conv.bias)
    with torch.no_grad():
        t_mine = timeit(lambda: ops.conv3x3(x, *packed))
        t_v1 = timeit(lambda: ops.conv3x3(x, *packed, use_staged=False))
        t_lib = timeit(lambda: conv(x))
    gf = 2.0 * n * H * W * Cout * Cin * 9 / 1e9
    print(f'conv3x3 {n}x{Cin}x{H}x{W} -> {Cout}: HIP {t_mine:8.1f} us ({gf / t_mine * 1e3:7.1f} TF)   from-L2 kernel {t_v1:8.1f} us   library {t_lib:8.1f} us ({gf / t_lib * 1e3:7.1f} TF)')
