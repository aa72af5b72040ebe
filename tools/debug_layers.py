"""Stage-by-stage comparison of the HIP forward against the emulated oracle (diagnostic, GPU box)."""
import sys, torch
sys.path.insert(0, '.')
from tests.test_model_gpu import _pair, _batch
from tests.util import rel_l2
dev = torch.device('cuda:0')
ref, m = _pair('vit_tiny_patch16_224', dev, depth=3)
x, _ = _batch(3, 224, 40, 5)
ref.eval(); m.eval()
taps = {}
with torch.no_grad():
    ref(x, emulate=True, taps=taps)
    m._sync_weights()
    outs, saved = m._forward_impl(x.to(dev), m._draw_drop_scales(3, dev), save=True)
B, N, D = 3, m.num_tokens, m.embed_dim
names = ['x', 'm1', 'r1', 'y1', 'qkv', 'o', 'lse', 'x2', 'm2', 'r2', 'y2', 'h', 'a']
for li, L in enumerate(saved['layers']):
    d = dict(zip(names, L))
    if li == 0:
        print('x0', rel_l2(d['x'].float().reshape(B, N, D), taps['x0']), (d['x'].float().reshape(B,N,D).cpu() != taps['x0']).float().mean().item())
    for k in ['y1', 'qkv', 'o', 'x2', 'y2', 'h', 'a']:
        t = taps[f'{li}.{k}']
        g = d[k].float().reshape(t.shape).cpu()
        print(li, k, 'rel %.2e' % rel_l2(g, t), 'frac differing %.4f' % (g != t).float().mean().item())
