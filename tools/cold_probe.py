import sys, os, time
ROOT='/root/repo'
sys.path.insert(0,ROOT); sys.path.insert(0,ROOT+'/tools'); sys.path.insert(0,ROOT+'/tests')
import numpy as np, bench, mia_amd
w=bench.make_workload(1,1_000_000,1)
def T(f,*a):
    t=time.perf_counter(); r=f(*a); return r,(time.perf_counter()-t)*1e3
for rep in range(2):
    t0=time.perf_counter(); hip=mia_amd.MiaHip(0); t_create=(time.perf_counter()-t0)*1e3
    _,t_p=T(hip.set_pssm,w["pssm"])
    _,t_u=T(hip.upload_reads,w["stored"].reshape(-1), w["offsets"], w["rc"], np.ones(w["n"],np.uint8), w["as_"], w["ae"])
    _,t_r=T(hip.realign,w["ref"],True); hip.sync()
    al=hip.alignments()
    _,t_c=T(hip.cull,4000,0.0,0.0,0)
    _,t_t=T(hip.tally)
    c,t_k=T(hip.consensus,1)
    _,t_i1=T(hip.iterate,w["ref"],True)
    _,t_i2=T(hip.iterate,c,True)
    print("rep",rep,"create %.1f set_pssm %.1f upload %.1f | realign %.1f cull %.1f tally %.1f consensus %.1f | iterate(after stepwise) %.1f, again %.1f ms"%(t_create,t_p,t_u,t_r,t_c,t_t,t_k,t_i1,t_i2))
    hip.close()
hip=mia_amd.MiaHip(0); hip.set_pssm(w["pssm"]); hip.upload_reads(w["stored"].reshape(-1), w["offsets"], w["rc"], np.ones(w["n"],np.uint8), w["as_"], w["ae"])
_,t1=T(hip.iterate,w["ref"],True); _,t2=T(hip.iterate,w["ref"],True)
print("fresh context: first iterate %.1f ms, second %.1f ms"%(t1,t2))
