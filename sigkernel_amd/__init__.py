"""sigkernel_amd -- MI355X-native signature-PDE-kernel engine.

Drop-in for the ``SigKernel.compute_kernel / compute_Gram / compute_mmd`` path of
crispitagorico/sigkernel: Python host code on PyTorch-ROCm calling hand-written HIP kernels
(gfx950) through the C ABI in ``include/sigkernel_amd.h``.
"""
from ._routes import routes
from .static_kernels import LinearKernel, RBFKernel
from .sigkernel import SigKernel, _SigKernel, _SigKernelGram, k_kgrad
from .stats import SigCHSIC, c_alpha, hypothesis_test
from .transforms import AddTime, LeadLag, add_time, lead_lag, transform

__all__ = ["SigKernel", "LinearKernel", "RBFKernel", "_SigKernel", "_SigKernelGram", "hypothesis_test", "SigCHSIC",
           "c_alpha", "transform", "add_time", "lead_lag", "AddTime", "LeadLag", "k_kgrad", "routes"]
__version__ = "0.1.0"
