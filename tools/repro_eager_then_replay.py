"""Reproducer kept from a round-2 bug hunt: graph replays of train(), a validate()
at another batch size, eager train() steps on fresh batch tensors, then a
replay of the OLD graphs (argument `plain`: the state before the fix, garbage
penalties in ~40 % of processes) or of re-captured ones (`recapture`).  train()
now drops its graphs whenever an eager step runs (wgan_gp.py)."""
import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import oracle as O
from calciumgan_amd import _lib, nets
from calciumgan_amd.gan.algorithms import get_algorithm
from calciumgan_amd.gan.models import get_models
hp = O.make_hparams(256, 16, 8, m=2)
hp.verbose = 0
gen, dis = get_models(hp, None)
gan = get_algorithm(hp, gen, dis, None)
rng = np.random.RandomState(0)
data = torch.tensor(rng.uniform(0, 1, (64, 256, 16)).astype(np.float32), device='cuda')
def batch(i):
    jj = torch.arange(8 * (i % 8), 8 * (i % 8) + 8, device='cuda')
    return data.index_select(0, jj)
for i in range(10):
    gan.train(batch(i))
v = gan.validate(data[:6])
torch.cuda.synchronize()
gan._use_graph = False
for i in range(5):
    gan.train(batch(i))
torch.cuda.synchronize()
st = gan._get_state(8)
def chk(tag):
    items = dict(dis_w=dis.net.params.data, dis_m=dis.net.params.m, dis_v=dis.net.params.v, gen_w=gen.net.params.data,
                 gen_m=gen.net.params.m, gen_v=gen.net.params.v)
    for i, op in enumerate(dis.net.w_fwd): items['dpack%d' % i] = op.buf.float()
    for i, op in enumerate(gen.net.w_fwd): items['gpack%d' % i] = op.buf.float()
    bad = [k for k, t in items.items() if not bool(torch.isfinite(t).all())]
    big = {k: float(t.abs().max()) for k, t in items.items() if float(t.abs().max()) > 1e3}
    print(tag, 'nonfinite:', bad, 'big:', big)
chk('before replay')
gan._use_graph = True
if len(sys.argv) > 1 and sys.argv[1] == 'recapture':
    st.pop('graph')
o = gan.train(batch(0))
torch.cuda.synchronize()
print('out', float(o[0]), float(o[1]), float(o[2]))
print('gp per critic step', st['gp'].cpu().numpy(), 'loss', st['loss'].cpu().numpy().tolist())
g = st['graph']
print('lr_dev', g['lr_dev'].cpu().numpy(), 'shifts', g['shifts_dev'].cpu().numpy())
print('host steps', gan.dis_optimizer.host_steps, gan.gen_optimizer.host_steps)
chk('after replay')
badrun = not np.isfinite([float(o[1])]).all() or abs(float(o[1])) > 1e3
print('VERDICT', 'BAD' if badrun else 'ok')
for k, v in sorted(nets._TILE_CACHE.items()):
    print('TILE', k, v)
