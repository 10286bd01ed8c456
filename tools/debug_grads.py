import sys, numpy as np, torch
sys.path.insert(0, '.')
from ams_amd import spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
from oracle.student_torch import StudentOracle
CI=[0,1,2,10,11,13]
H=int(sys.argv[1]) if len(sys.argv)>1 else 64
W0=Wt.synthetic_weights(S.build_spec(),0)
frames,labels=synth.SyntheticVideo(H,6,CI).clip()
B=4
eng=StudentEngine(CI,H,2*H,max_batch=B,trainable=True); eng.load_variables(W0)
o=StudentOracle(W0,CI,dtype=torch.float64); o32=StudentOracle(W0,CI)
for step in range(3):
    eng.load_variables(o.get_vars()); o32.restore(o.get_vars())
    fr,lb=frames[step:step+B],labels[step:step+B]
    lo,go=o.gradients(fr.astype(np.float32),lb)
    _,g32=o32.gradients(fr.astype(np.float32),lb)
    ls=eng.train_step(fr,lb,1e-3).cpu().numpy()
    g=eng.grads.cpu().numpy().astype(np.float64)
    gn=max(float(v.abs().max()) for v in go.values())
    rows=[]
    for v in eng.spec.trainable:
        want=go[v.name].numpy().reshape(-1); got=g[v.offset:v.offset+v.size]; f32=g32[v.name].numpy().reshape(-1).astype(np.float64)
        fl=max(np.abs(want).max(),1e-3*gn); fl2=max(np.linalg.norm(want),1e-3*gn*np.sqrt(want.size))
        rows.append((np.abs(got-want).max()/fl, np.abs(f32-want).max()/fl, np.linalg.norm(got-want)/fl2, np.linalg.norm(f32-want)/fl2, v.name))
    r=np.array([x[:4] for x in rows])
    print("step",step,"loss",ls[0]/ls[1],lo,"median max-err gpu %.2e f32 %.2e | median l2 gpu %.2e f32 %.2e | worst max gpu %.2e f32 %.2e | worst l2 gpu %.2e f32 %.2e"%(np.median(r[:,0]),np.median(r[:,1]),np.median(r[:,2]),np.median(r[:,3]),r[:,0].max(),r[:,1].max(),r[:,2].max(),r[:,3].max()))
    for x in sorted(rows,key=lambda x:-x[2])[:4]: print("   l2 worst: gpu %.2e f32 %.2e max gpu %.2e f32 %.2e %s"%(x[2],x[3],x[0],x[1],x[4]))
    o.train_step(fr.astype(np.float32),lb,1e-3)
