"""Oracle-backed stand-in for sigkernel_amd._lib.HipBackend -- TESTS ONLY.

Lets the `-m "not gpu"` suite exercise the host logic (autograd wiring, tiling, MMD formula,
row sharding over gloo) on CPU tensors.  The product never imports this.
"""
import numpy as np
import torch

from oracle import oracle as O


class OracleBackend:
    name = "oracle-fake"

    def increments(self, G):
        return torch.from_numpy(O.increments(G.detach().double().numpy())).to(G.dtype)

    def static_increments(self, kind, param, X, Y, gram):
        import sigkernel_amd
        k = sigkernel_amd.LinearKernel(param) if kind == 0 else sigkernel_amd.RBFKernel(param)
        if kind == 0 and gram:
            k = sigkernel_amd.LinearKernel()
        G = k.Gram_matrix(X, Y) if gram else k.batch_kernel(X, Y)
        return self.increments(G)

    def static_adjoint(self, kind, param, X, Y, W, scale, gram):
        import sigkernel_amd
        k = sigkernel_amd.LinearKernel(param) if kind == 0 else sigkernel_amd.RBFKernel(param)
        if kind == 0 and gram:
            k = sigkernel_amd.LinearKernel()
        Xg = X.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            G = k.Gram_matrix(Xg, Y) if gram else k.batch_kernel(Xg, Y)
        dG = self.increments_adjoint(W, scale)
        (g,) = torch.autograd.grad(G, Xg, dG)
        return g

    def static_adjoint2(self, kind, param, X, Y, W, scale, b0=0):
        """dL/dY[b0:] of the Gram pairs (a, b): autograd through the static kernel w.r.t. its second argument."""
        import sigkernel_amd
        k = sigkernel_amd.LinearKernel() if kind == 0 else sigkernel_amd.RBFKernel(param)
        Yg = Y.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            G = k.Gram_matrix(X, Yg)
        dG = self.increments_adjoint(W, scale)
        (g,) = torch.autograd.grad(G, Yg, dG)
        return g[b0:]

    def increments_adjoint(self, W, scale=None):
        dG = torch.from_numpy(O.increments_adjoint(W.detach().double().numpy())).to(W.dtype)
        if scale is not None:
            dG = dG * scale.reshape(scale.shape + (1, 1))
        return dG

    def solve_fwd(self, inc_c, dyadic, naive=False, flags=0, want_grid=False, want_edges=False):
        a = inc_c.detach().double().numpy()
        if want_grid or want_edges:
            out, grid = O.solve_coarse(a, dyadic, naive, want_grid=True)
            edges = np.concatenate([grid[..., -1, :], grid[..., :, -1]], axis=-1)
            return (torch.from_numpy(out).to(inc_c.dtype), torch.from_numpy(grid).to(inc_c.dtype) if want_grid else None,
                    torch.from_numpy(edges) if want_edges else None)
        return torch.from_numpy(O.solve_coarse(a, dyadic, naive)).to(inc_c.dtype)

    def solve_fwd_keep_edges(self, inc_c, dyadic, naive=False):
        return self.solve_fwd(inc_c, dyadic, naive), torch.zeros(1)      # a token: the fake adjoint needs no edges

    def solve_adj(self, inc_c, dyadic, naive=False, flags=0, edges=None):
        out, W = O.adjoint_coarse(inc_c.detach().double().numpy(), dyadic, naive)
        return torch.from_numpy(out).to(inc_c.dtype), torch.from_numpy(W).to(inc_c.dtype)

    def deriv_increments(self, G0, G1, G2, eps):
        d1 = -(1. / eps) * G0
        d2 = (1. / eps) * G1
        dd1 = -(1. / eps) * d1
        dd2 = -(2. / eps) * d2
        dd3 = (1. / eps ** 2) * G2
        inc = self.increments(G0)
        inc_d = self.increments(d1) + self.increments(d2)
        inc_dd = self.increments(dd1) + self.increments(dd2) + self.increments(dd3)
        return torch.stack([inc, inc_d, inc_dd])

    def solve_deriv(self, inc3, dyadic, flags=0):
        a = inc3.detach().double().numpy()
        return tuple(torch.from_numpy(v).to(inc3.dtype) for v in O.solve_deriv_coarse(a[0], a[1], a[2], dyadic))

    def static_deriv_increments(self, kind, param, X0, X1, X2, Y, eps):
        import sigkernel_amd
        k = sigkernel_amd.LinearKernel() if kind == 0 else sigkernel_amd.RBFKernel(param)
        return self.deriv_increments(k.Gram_matrix(X0, Y), k.Gram_matrix(X1, Y), k.Gram_matrix(X2, Y), eps)
