import sys, time, numpy as np
sys.path.insert(0, ".")
import bammmotif2_amd as bm
from bammmotif2_amd import synth
for N in (125000, 1000000):
  for K in (0,1):
    for W in (6, 8):
        L0=200
        pwm=synth.make_pwm(W,1234); codes,off=synth.make_sequences(N,L0,pwm,1234,plant_frac=0.5)
        pk=bm.PackedSeqs.from_codes(codes,off,False,seed=42); vbg=pk.bg_model(2,np.array([1,10,10],np.float32))
        A=synth.alpha_matrix(synth.default_alpha(K),W); v0=synth.bamm_from_pwm((0.7*pwm+0.075).astype(np.float32),K)
        out=[]
        for kv in ({}, {"grouped":0}):
            ctx=bm.Context(0)
            if kv: ctx.set_tuning(**kv)
            ss=bm.SeqSet(ctx,pk)
            em=bm.EM(ctx,ss,K,W,vbg,A,v0,0.3,max_iterations=100,n_seqs_bound=N); em.iterate(25); ctx.sync()
            t=time.perf_counter(); em.iterate(20); ctx.sync(); out.append((time.perf_counter()-t)/20*1e3)
            em.close(); ss.close(); ctx.close()
        print("N %d K %d W %d: grouped %.3f ms, per-column %.3f ms" % (N,K,W,out[0],out[1]), flush=True)
