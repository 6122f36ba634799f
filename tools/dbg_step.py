import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import sac1_oracle as so
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters, Learner
def mk(seed):
    opt=HyperParameters(); opt.seed=seed
    return Learner(opt), so.Config()
l1,cfg=mk(2); l2,_=mk(2)
batch,eps=so.synthetic_batch(cfg,seed=5)
l1.train(batch,eps=eps)
g=l2.compute_gradients(batch,eps=eps); l2.apply_gradients(g)
for which,nm in ((_lib.SAC1_MAIN,'main'),(_lib.SAC1_TARGET,'target'),(_lib.SAC1_ADAM_M,'m'),(_lib.SAC1_ADAM_V,'v'),(_lib.SAC1_GRAD,'grad')):
    a=l1.export(which).cpu().numpy(); b=l2.export(which).cpu().numpy(); off=0
    for name,shape in so.param_specs(cfg):
        n=int(np.prod(shape)); x,y=a[off:off+n],b[off:off+n]; off+=n
        nd=int((x!=y).sum())
        if nd: print(nm,name,'ndiff',nd,'of',n,'maxabs %.3e'%np.abs(x-y).max(), 'first', np.flatnonzero(x!=y)[:5], x[x!=y][:3], y[x!=y][:3])
print('steps', l1.opt_steps(), l2.opt_steps())
