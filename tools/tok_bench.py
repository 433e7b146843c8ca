"""GPU micro-benchmark of the token kernels: every case is captured 20x in a hipGraph and replayed (no host overhead in
the numbers).  Usage: python tools/tok_bench.py"""
import math, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepinteraction_amd import ops

DEV = 'cuda'
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(DEV)
SP = lambda w: ops.split_hi_lo(w)
PK = lambda w: ops.pack_linear(w)
B, Q = 1, 200
M = B * Q


_big = torch.randn(8192, 8192, device=DEV, dtype=torch.float16)


def timeit(name, fn, reps=20, iters=40):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    for _ in range(12):                      # clocks up: ~100 ms of GEMMs, then the measured replays back to back
        _big @ _big
    for _ in range(5):
        gr.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        gr.replay()
    e.record(); torch.cuda.synchronize()
    print(f'{name:60s} {s.elapsed_time(e) * 1e3 / (reps * iters):8.2f} us')


x, y = r(M, 128), torch.empty(M, 128, device=DEV)
qkv = r(M, 384)
qk, vt = qkv[:, :256].contiguous(), r(B, 128, 208)
ln = (r(128), r(128))
W = {k: PK(r(n, kk) / math.sqrt(kk)) for k, (n, kk) in dict(a=(128, 128), q=(384, 128), f1=(512, 128), f2=(128, 512), h=(384, 256)).items()}
bias = {k: r(n) for k, n in dict(a=128, q=384, f1=512, f2=128, h=384).items()}
timeit('empty program (load, store)', lambda: ops.TokenProgram().load(0, x).store(0, y).run(B, Q))
timeit('load, linear 128->128, store', lambda: ops.TokenProgram().load(0, x).linear(0, 1, W['a'], bias['a']).store(1, y).run(B, Q))
timeit('load, linear 128->384, store', lambda: ops.TokenProgram().load(0, x).linear(0, 1, W['q'], bias['q']).store(1, y).run(B, Q))
timeit('load, linear 128->512 gelu, linear 512->128, store',
       lambda: ops.TokenProgram().load(0, x).linear(0, 1, W['f1'], bias['f1'], act=2).linear(1, 2, W['f2'], bias['f2']).store(2, y).run(B, Q))
timeit('load, 4x linear 128->128, store', lambda: ops.TokenProgram().load(0, x).linear(0, 1, W['a'], bias['a']).linear(1, 0, W['a'], bias['a'])
       .linear(0, 1, W['a'], bias['a']).linear(1, 0, W['a'], bias['a']).store(0, y).run(B, Q))
timeit('attn, store', lambda: ops.TokenProgram().attn(0, qk, vt, 0.25).store(0, y).run(B, Q))
timeit('load, 4x rowop LN, store', lambda: ops.TokenProgram().load(0, x).rowop(0, 0, ln=ln).rowop(0, 0, ln=ln).rowop(0, 0, ln=ln).rowop(0, 0, ln=ln).store(0, y).run(B, Q))
from deepinteraction_amd import decoder_fused
xw = ops.split_rows(r(M, 128))
wd, bd = decoder_fused._dyn_layout(r(32768, 128) / 11, r(32768))
timeit('wide 128->32768', lambda: ops.token_wide(xw, wd, bd))
roi = ops.split_rows(r(M * 49, 128)).view(M, 49, 256)
params = ops.token_wide(xw, wd, bd)
n1, n2 = (r(128), r(128)), (r(128), r(128))
timeit('dynconv', lambda: ops.dynconv(roi, params, n1, n2))
f2 = ops.dynconv(roi, params, n1, n2)
wo = ops.pack_ksteps(r(128, 6272) / 80)
timeit('splitk 6272->128', lambda: ops.token_splitk(f2, wo))

def stamped(name, prog):
    st = torch.zeros(32, dtype=torch.int64, device=DEV)
    for _ in range(3):
        prog().run(B, Q, stamps=st)
    torch.cuda.synchronize()
    t = st.cpu().tolist()
    n = len(prog().steps)
    kinds = {1: 'load', 2: 'parts', 3: 'attn', 4: 'combine', 5: 'linear', 6: 'rowop', 7: 'store', 8: 'heads'}
    print(name, 'cycles per step:', [(kinds[s.kind], t[i + 1] - t[i]) for i, s in enumerate(prog().steps) if t[i + 1]], 'total', t[n] - t[0])


stamped('4x linear', lambda: ops.TokenProgram().load(0, x).linear(0, 1, W['a'], bias['a']).linear(1, 0, W['a'], bias['a'])
        .linear(0, 1, W['a'], bias['a']).linear(1, 0, W['a'], bias['a']).store(0, y))
stamped('ffn', lambda: ops.TokenProgram().load(0, x).linear(0, 1, W['f1'], bias['f1'], act=2).linear(1, 2, W['f2'], bias['f2']).store(2, y))
stamped('attn', lambda: ops.TokenProgram().attn(0, qk, vt, 0.25).store(0, y))
stamped('rowops', lambda: ops.TokenProgram().load(0, x).rowop(0, 0, ln=ln).rowop(0, 0, ln=ln).store(0, y))

nol = lambda n: (lambda: [ops.TokenProgram().load(0, x)] and None)
def rep(n, **kw):
    p = ops.TokenProgram().load(0, x)
    for _ in range(n):
        p.rowop(0, 0, **kw)
    return p.store(0, y)
timeit('load, 8x rowop (no LN: no global access), store', lambda: rep(8).run(B, Q))
timeit('load, 8x rowop LN, store', lambda: rep(8, ln=ln).run(B, Q))
stamped('8 rowop noLN', lambda: rep(8))
stamped('8 rowop LN', lambda: rep(8, ln=ln))
