import time, numpy as np, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bammmotif2_amd as bm
from bammmotif2_amd import synth
N,L0,W,K=50000,200,20,2
pwm=synth.make_pwm(W); codes,off=synth.make_sequences(N,L0,pwm)
t=time.time(); pk=bm.PackedSeqs.from_codes(codes,off); print("pack",time.time()-t, pk.n_exceptions)
ctx=bm.Context(0); print(ctx.device_name())
ss=bm.SeqSet(ctx,pk); print(ss.info())
vbg=np.full(bm.bg_size(2),0.25,np.float32)
A=synth.alpha_matrix(synth.default_alpha(K),W); v0=synth.bamm_from_pwm((0.7*pwm+0.075).astype(np.float32),K)
em=bm.EM(ctx,ss,K,W,vbg,A,v0,0.3)
em.iterate(3); ctx.sync()
t=time.time(); em.iterate(20); ctx.sync(); dt=time.time()-t
print("20 iters",dt, "it/s",20/dt, "kernel", em.kernel_time())
print(em.trace()[0][:5])
