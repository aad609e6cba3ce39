import sys, json, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from helpers import full_tacotron
from semi_tts_amd.runtime import GraphedDecoder
from semi_tts_amd.synthetic import synthetic_batch
dev=torch.device('cuda:0')
B,L,T=(64,171,1065) if 'c5' in sys.argv else (32,43,258)
steps=T//3
m=full_tacotron(dev, seed=1234, prenet_dropout=0.5)
txt,spk,_=synthetic_batch(B,L,T,seed=100)
txt,spk=torch.from_numpy(txt).to(dev),torch.from_numpy(spk).to(dev)
with torch.no_grad(): mem=m.encoder(txt,None).contiguous()
for pre,fin in (((4,2),(4,4),(8,2),(8,4),(4,8),(8,8),(2,4)) if 'c5' in sys.argv else ((2,2),(4,2),(1,2),(2,4),(4,4),(2,1),(2,8))):
    m.decoder.attn_pre_parts, m.decoder.attn_fin_parts = pre, fin
    m.decoder.__dict__.pop('_packed_cache', None)
    gd=GraphedDecoder(m.decoder,B,L,T,dev); gd.memory.copy_(mem); gd.spkr.copy_(spk); gd.capture()
    for _ in range(5): gd(redraw=True)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(10): gd(redraw=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
    print('pre_parts %d fin_parts %d: %.2f us/step' % (pre,fin,dt/steps*1e6))
