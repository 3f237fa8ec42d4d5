"""How far do two float32 evaluations of the SAME PPO update drift apart?  (CPU, oracle against itself.)

A PPO update is 160 clipped-surrogate Adam steps.  Perturbing the rewards by 1e-6 relative (a few ulp: what any change of
summation order upstream produces) and replaying the identical update through oracle/sg_oracle.c moves the final policy
weights by up to ~3e-4 absolute (~1 % of the update's length in L2, 17 % of the entries beyond 1e-4 relative), and a
1e-5 perturbation moves them by the same amount: the clip / min / max decisions of rows sitting on a branch boundary flip,
and Adam turns a flipped near-zero gradient into a full lr-sized step.  The three losses the update reports still agree to
<1e-4.  tests/test_gpu_benchpath.py uses these measurements to size its trajectory-level tolerances.
    python tools/trajectory_sensitivity.py [hopper]   (~1 minute of one core; `hopper` = SplitPolicy h100 at the
    HopperCombinedEnv-v1 shapes: 4.3e-02 relative L2, 5.0e-04 worst entry, 39 % of the entries under a 1e-5 perturbation)
"""
import numpy as np, time, sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
rng=np.random.default_rng(0)
SPLIT = len(sys.argv) > 1 and sys.argv[1] == 'hopper'   # HopperCombinedEnv-v1 shapes, SplitPolicy h100
T,N,O,A,H=(128,256,14,7,100) if SPLIT else (128,512,47,12,64)
d=orc.dims(orc.KIND_SPLIT if SPLIT else orc.KIND_MLP,O,A,H,1)
n=orc.policy_num_params(d)
# plausible init: small weights
par0=(rng.standard_normal(n)*0.1).astype(np.float32)
if not SPLIT: par0[-A:]=-0.5
obs=rng.standard_normal((T+1,N,O)).astype(np.float32)
noise=rng.standard_normal((T*N,A)).astype(np.float32)
v,act,lp=orc.policy_act(d,par0,obs[:-1].reshape(-1,O),noise)
actions=act.reshape(T,N,A); logp=lp.reshape(T,N); vp=np.zeros((T+1,N),np.float32); vp[:T]=v.reshape(T,N)
masks=(rng.random((T+1,N))>0.01).astype(np.float32); bad=np.ones((T+1,N),np.float32)
rewards=np.clip(rng.standard_normal((T,N)),-10,10).astype(np.float32)
nv=orc.policy_forward(d,par0,obs[T])[0][:,0]
perms=np.stack([rng.permutation(T*N) for _ in range(10)]).astype(np.int64)
cfg=orc.ppo_cfg(0.2,10,16,0.5,0.0,3e-4,1e-5,0.5,True)
outs=[]
for eps in (0.0, 1e-6, 1e-5):
    rw=(rewards*(1+eps*rng.standard_normal(rewards.shape))).astype(np.float32)
    ret,vp2=orc.compute_returns(rw,vp,masks,bad,nv,1,0.99,0.95,1)
    p=par0.copy(); ad=orc.AdamState(n)
    t=time.time(); l=orc.ppo_update(d,p,ad,cfg,obs,actions,vp2,ret,logp,perms); print(eps,l,time.time()-t,flush=True)
    outs.append(p)
for i in (1,2):
    e=np.abs(outs[i]-outs[0]); tol=5e-5+1e-4*np.abs(outs[0])
    print('perturbation', (1e-6,1e-5)[i-1], 'max abs', e.max(), 'frac out of tol', np.mean(e>tol), 'rel l2 of update', np.linalg.norm(outs[i]-outs[0])/np.linalg.norm(outs[0]-par0), 'max move', np.abs(outs[0]-par0).max())
