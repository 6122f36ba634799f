import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import sac1_oracle as so
from distributed_drl_amd import _lib
from distributed_drl_amd.agent import HyperParameters, Learner
opt=HyperParameters(); opt.seed=5; opt.obs_dim=5; opt.act_dim=3; opt.hidden_sizes=(70,45); opt.batch_size=37
L=Learner(opt); cfg=so.Config(obs_dim=5,act_dim=3,hidden1=70,hidden2=45,batch=37,alpha=opt.alpha,gamma=opt.gamma,lr=opt.lr,polyak=opt.polyak)
params=so.init_params(cfg,5)
L.set_weights(list(params.keys()), list(params.values()))
batch,eps=so.synthetic_batch(cfg,seed=9)
o64=so.Sac1Oracle(cfg,params,torch.float64); w=o64.step(batch,*eps)
losses,_=L.train(batch,eps=eps,return_outputs=True)
print("losses", losses.cpu().numpy(), [float(w[k]) for k in ("pi_loss","q1_loss","q2_loss")])
g=L.export(_lib.SAC1_GRAD).cpu().numpy(); g64=o64.flat("grads"); off=0
print("global tol", 2e-4*np.abs(g64).max())
for name,shape in so.param_specs(cfg):
    n=int(np.prod(shape)); a,b=g[off:off+n],g64[off:off+n]
    print("%-26s maxerr %.3e  max|ref| %.3e"%(name,np.abs(a-b).max(),np.abs(b).max()))
    off+=n
