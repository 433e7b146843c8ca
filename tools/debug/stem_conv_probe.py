"""The ResNet stem (7x7, stride 2, 3 -> 64) through MIOpen with the input padded to 4 / 8 channels (channels-last fp16)."""
import time, torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
def bench(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6
for ci in (3, 4, 8):
    x = torch.randn(6, ci, 448, 800, device='cuda').half().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, ci, 7, 7, device='cuda') * 0.05).half().contiguous(memory_format=torch.channels_last)
    print(ci, 'channels:', round(bench(lambda: F.conv2d(x, w, None, 2, 3)), 1), 'us')
x = torch.randn(6, 64, 224, 400, device='cuda').half().contiguous(memory_format=torch.channels_last)
print('max_pool 3x3 s2:', round(bench(lambda: F.max_pool2d(x, 3, 2, 1)), 1), 'us')
