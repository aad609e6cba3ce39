import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from semi_tts_amd import ops
dev = torch.device('cuda')
def run(B, H, K):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, K, generator=g); c = torch.randn(B, H, generator=g)
    w = torch.randn(4 * H, K, generator=g) * K ** -0.5
    gates = x @ w.t()
    i, f, gg, o = [gates[:, j * H:(j + 1) * H] for j in range(4)]
    c_ref = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    packed = ops.pack_weight([w.to(dev)], [K], 4 * H, lstm_H=H, ldws=[K])
    xb = ops.tile_rows(x.to(dev))
    h_t16 = torch.zeros(ops.t16_floats(B, H), device=dev)
    c_out = torch.empty(B, H, device=dev)
    ops.lstm_cell_packed(packed, ops.t16_view(xb, K=K), 16 * ops.kb16(K), None, None, c.to(dev), ops.t16_view(h_t16, K=H), c_out, B, H)
    e = (c_out.cpu() - c_ref).abs()
    print(B, H, K, 'err', float(e.max()), 'bad rows', sorted(set(torch.nonzero(e > 1e-4)[:, 0].tolist()))[:40], 'bad cols', len(set(torch.nonzero(e > 1e-4)[:, 1].tolist())))
for B, H, K in [(32, 1024, 256), (32, 1024, 768), (32, 1024, 1024), (32, 1024, 1792), (32, 64, 1792), (16, 1024, 1792), (32, 1024, 2560), (32, 256, 1792), (32, 512, 1792)]:
    run(B, H, K)
