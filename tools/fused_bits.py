"""Are the fused kernels bit-identical to the layer-by-layer plan?  (first block: yes; expand+depthwise: ~1e-5, different
GEMM chunking.)  usage: python tools/fused_bits.py"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from ams_amd import hip, spec as S, synth, weights as Wt
from ams_amd.engine import StudentEngine
CI=[0,1,2,10,11,13]; H=256
W0=Wt.synthetic_weights(S.build_spec(),0)
frames,labels=synth.SyntheticVideo(H,4,CI,seed=1).clip()
eng=StudentEngine(CI,H,2*H,max_batch=4,trainable=False); eng.load_variables(W0); eng.freeze()
h,w=eng.lowres
low=lambda: eng.logits_lowres.view(-1,h,w,32)[:4,:,:,:19].cpu().numpy().copy()
eng.predict(frames); ref=low()
for name,fn in (("first_block off", lambda: eng.set_fuse_first_block(False)), ("expand_dw off", lambda: eng.set_fuse_expand_dw(0)), ("expand_dw all", lambda: eng.set_fuse_expand_dw(2))):
    fn(); eng.predict(frames); cur=low()
    print(name, "bit-identical" if np.array_equal(cur,ref) else "max rel diff %.2e"%(np.abs(cur-ref).max()/np.abs(ref).max()))
