"""Diagnostic (test infrastructure: uses the oracle): one tiny case printed side by side, HIP path vs oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import tf_seq2seq_losses_amd as ctc
from oracle import ctc_oracle as O
np.set_printoptions(precision=4, suppress=True, linewidth=200)
logits = np.zeros((2,2,3), np.float32)
labels = np.array([[1,2],[1,2]], np.int32); ll = np.array([2,1], np.int32); tl = np.array([2,2], np.int32)
lp = O.logit_to_logproba(logits.astype(np.float64)).astype(np.float32)
d = ctc.ClassicCtcLossData(torch.tensor(labels).cuda(), torch.tensor(lp).cuda(), torch.tensor(ll).cuda(), torch.tensor(tl).cuda(), 0)
r = O.ClassicCtcLossData(labels, lp, ll, tl, 0)
print("loss", d.loss.cpu().numpy(), r.loss)
print("alpha gpu\n", np.exp(d.alpha.cpu().numpy()[1])); print("alpha ref\n", np.exp(r.alpha[1]))
print("beta gpu\n", np.exp(d.beta.cpu().numpy()[1])); print("beta ref\n", np.exp(r.beta[1]))
print("grad gpu\n", d.gradient.cpu().numpy()[1]); print("grad ref\n", r.gradient[1])
