import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import sac1_oracle as so
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters, Learner
opt=HyperParameters(); opt.seed=0
L=Learner(opt); cfg=so.Config(); params=so.init_params(cfg,0)
rs=np.random.RandomState(10)
for k in params:
    if k.endswith("bias"): params[k]=rs.uniform(-0.05,0.05,params[k].shape).astype(np.float32)
L.set_weights(list(params.keys()), list(params.values()))
batch,eps=so.synthetic_batch(cfg,seed=1234)
o64=so.Sac1Oracle(cfg,params,torch.float64); w=o64.step(batch,*eps)
losses,_=L.train(batch,eps=eps,return_outputs=True)
print("losses", losses.cpu().numpy(), [float(w[k]) for k in ("pi_loss","q1_loss","q2_loss")])
g=L.export(_lib.SAC1_GRAD).cpu().numpy(); g64=o64.flat("grads"); off=0
for name,shape in so.param_specs(cfg):
    n=int(np.prod(shape)); a,b=g[off:off+n],g64[off:off+n]
    print("%-26s maxerr %.3e  max|ref| %.3e  %s"%(name,np.abs(a-b).max(),np.abs(b).max(), "BAD" if np.abs(a-b).max()>2e-4*np.abs(b).max() else ""))
    if np.abs(a-b).max()>2e-4*np.abs(b).max(): print("   gpu", a[:6], "ref", b[:6])
    off+=n
