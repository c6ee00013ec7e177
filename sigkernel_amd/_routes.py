"""Route switches of the host layer: which fused HIP kernels a call may take.  They exist for A/B measurements and for the
tests that compare a fused route with the unfused one; every combination gives the same results to the stated tolerances.

Read from the environment ONCE, when the package is imported -- no `os.environ` look-up on any call path (the C library does
the same with its SK_* knobs, sk_abi.hip):

    SK_NO_FUSED_RBF      RBFKernel: no fused forward / adjoint (static kernel in its own pass, increments in HBM)
    SK_NO_FUSED_MB       no multi-band fused forward (long or wide paths stream precomputed increments)
    SK_NO_FUSED_ADJOINT  gradients by the unfused adjoint (W = dk/dinc in HBM, then the static-kernel chain rule)
    SK_NO_FUSED_DERIV    compute_kernel_and_derivatives_Gram through three precomputed increment arrays
    SK_FUSEDMB_NO_Y32    the multi-band kernels stage fp32 inputs as fp64
    SK_NO_MMD_STREAMS    compute_mmd on ONE stream also while a hipGraph is captured (default: the three Gram matrices of a captured
                         training-sized step -- up to 128 x 128 pairs each -- on three streams = parallel branches of the graph)
    SK_NO_MERGED_LOSS    compute_mmd / compute_scoring_rule / compute_expected_scoring_rule of training-sized batches as the reference's
                         composition of compute_Gram calls (default: one Gram block K(X, [X; Y]), sigkernel._SigKernelLoss)
    SK_NO_ADJOINT_SWAP   gradients for long first paths against short second ones never through the one-band rbf adjoint on (y, x) with the
                         second-argument sums (default: rbf, dim <= 4, fp64 Gram calls whose second paths fit its lanes)
    SK_NO_LOSS_LAUNCH    the merged loss route without the one-launch glue of csrc/sk_loss.hip (staging of [X; Y], K(X, [X; Y]) + the
                         triangle of K(Y, Y) in ONE forward launch, value / weights / gradient fold as single kernels): torch ops instead
    SK_NO_STREAM         memory first: LinearKernel / RBFKernel calls of path dim <= 16, dyadic <= 2 NEVER hold increments in HBM, also
                         on short paths, where the multi-band kernels mostly sweep padding and the (row-tiled) streaming route is the
                         faster default (sk_route_query with SK_ROUTE_NO_STREAM)

A running process flips them through the attributes of `sigkernel_amd.routes` (tests: monkeypatch.setattr), or calls
`routes.reload()` after changing the environment."""
import os

_ENV = {"no_fused_rbf": "SK_NO_FUSED_RBF", "no_fused_mb": "SK_NO_FUSED_MB", "no_fused_adjoint": "SK_NO_FUSED_ADJOINT",
        "no_fused_deriv": "SK_NO_FUSED_DERIV", "no_stream": "SK_NO_STREAM", "no_mmd_streams": "SK_NO_MMD_STREAMS",
        "no_merged_loss": "SK_NO_MERGED_LOSS", "no_loss_launch": "SK_NO_LOSS_LAUNCH", "no_adjoint_swap": "SK_NO_ADJOINT_SWAP"}


class Routes:
    __slots__ = tuple(_ENV)

    def __init__(self):
        self.reload()

    def reload(self):
        for attr, name in _ENV.items():
            setattr(self, attr, bool(os.environ.get(name)))

    def __repr__(self):
        return "Routes(" + ", ".join("%s=%s" % (a, getattr(self, a)) for a in _ENV) + ")"


routes = Routes()
