"""MI355X-native CTC loss (loss, analytic gradient, analytic Hessian) behind the call signature of
alexeytochin/tf_seq2seq_losses.  HIP kernels for gfx950 in csrc/, C ABI in include/ctc_amd.h."""
from .losses import (  # noqa: F401
    classic_ctc_loss,
    simplified_ctc_loss,
    simple_ctc_loss,
    ctc_loss,
    ctc_loss_from_logproba,
    ClassicCtcLossData,
    SimplifiedCtcLossData,
)
from .ops import check_labels  # noqa: F401

__version__ = "0.1.0"
__all__ = ["classic_ctc_loss", "simplified_ctc_loss", "simple_ctc_loss", "ctc_loss", "ctc_loss_from_logproba",
           "ClassicCtcLossData", "SimplifiedCtcLossData", "check_labels"]
