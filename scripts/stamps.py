"""Reads the per-wavefront work/wait cycle counters of a -DCTC_FUSED4_STAMPS diagnostic build (CTC_AMD_LIB=...)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf_seq2seq_losses_amd import _lib, ops
import bench
B,T,U,V=256,1000,128,256
host,dev=bench.make_inputs(B,T,U,V,0,False,torch.device("cuda:0"))
prep=ops.Prepared(dev["labels"],dev["logits"],dev["label_length"],dev["logit_length"],0,U=U)
n=_lib.workspace_bytes(_lib.WS_LOSS_GRAD,0,B,T,V,U)
ws=torch.zeros(n,dtype=torch.uint8,device="cuda:0")
for _ in range(3): ops.loss_grad(0,_lib.WRT_LOGITS,prep,True,workspace=ws)
torch.cuda.synchronize()
# locate off_dummy: layout mirrors ctc::make_layout
al=lambda x:(x+255)&~255
NL=2; UP=128; ERS=UP+4; SRS=2*UP+8
o=0; o=al(o+B*T*ERS*4); o=al(o+B*(T+1)*SRS*4); o=al(o+B*(T+1)*SRS*4); o=al(o+B*8); off_dummy=o
NH=4; NW=4+2*NH
st=ws[off_dummy:off_dummy+B*NW*32].view(torch.int64).cpu().numpy().reshape(B,NW,4)
names=["main A","main B","recompute A","recompute B"]+[f"helper A{h}" for h in range(NH)]+[f"helper B{h}" for h in range(NH)]
for i,nm in enumerate(names):
    w, wt, w1, wt1 = (st[:, i, k].mean() for k in range(4))
    print(f"{nm:12s}: phase1 work {w1:7.0f} wait {wt1:7.0f} | phase2 work {w - w1:7.0f} wait {wt - wt1:7.0f} | total {w + wt:7.0f} cycles")
