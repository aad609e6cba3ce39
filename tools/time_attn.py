import sys, torch
sys.path.insert(0, '/root/repo')
from semi_tts_amd import ops
dev = torch.device('cuda:0')
B, L, A, E, F, K = 32, 43, 256, 512, 32, 31
g = torch.Generator().manual_seed(0)
r = lambda *s: torch.randn(*s, generator=g).to(dev)
pq, pm, mem = r(B, A), r(B, L, A), r(B, L, E)
w_prev = torch.softmax(r(B, L), -1); w_cum = w_prev * 2
wc, wl, v = r(F, 2, K) * 0.3, r(A, F) * 0.3, r(1, A)
w1, c1, x1 = torch.empty(B, L, device=dev), torch.empty(B, L, device=dev), torch.empty(B, E, device=dev)
s_buf = torch.empty(B, L, A, device=dev)
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print('full step  %.2f us' % t(lambda: ops.attn_step(pq, pm, mem, w_prev, w_cum, w1, c1, wc, wl, v, x1)))
print('pre        %.2f us' % t(lambda: ops.attn_pre(pm, w_prev, w_cum, wc, wl, s_buf)))
print('fin        %.2f us' % t(lambda: ops.attn_fin(pq, s_buf, mem, w_cum, v, w1, c1, x1, F, K)))
