"""Oracle for the force-closure QP: qpth 0.0.18 batched PDIPM, forward + KKT-implicit backward.

TEST INFRASTRUCTURE.  qpth is a pip dependency of the reference (``graspqp/pyproject.toml:34``,
``qpth==0.0.18``) that is not in the reference tree and not installable here, so this file restates
its published algorithm (Amos & Kolter, OptNet, arXiv:1703.00443 sec. 3 + the solver structure of
``qpth/qp.py`` / ``qpth/solvers/pdipm/batch.py``) as it is driven from the reference call site
``metrics/solver/qp_solver.py:8,101-126``::

    QPFunction(verbose=False, maxIter=12, eps=5e-2)(Q, p, G, h, A=empty, b=empty)
    min 1/2 z'Qz + p'z  s.t.  Gz <= h          (notImprovedLim=3, no equalities)

Two forward implementations are kept and tested against each other:
  * ``pdipm_forward``          -- qpth's own linear algebra: pre-factor Q, R = G Q^-1 G',
                                  S = R + diag(1/d) (nineq x nineq), solve_kkt by block elimination.
  * ``pdipm_forward_box``      -- the same iteration specialised to G = [I; -I] with the KKT system
                                  reduced to one nz x nz SPD solve (Q + diag(d_u + d_l)); this is the
                                  form the HIP kernel uses.
PARITY UNPINNED except for the reference's known-answer test (tests/metrics/test_solver.py:5-27).

Documented deviation (both implementations, and the HIP kernel): qpth's ``get_step`` replaces the
ratios of non-blocking entries by ``max(1, a.max())`` where ``a.max()`` runs over the WHOLE batch
tensor.  Its only effect is on rows without any blocking entry, whose step becomes
min(1, 0.999*max(1, global max)); we use the row-independent value obtained when the global max
is >= 1.001001 (always the case in practice: some entry of some row has dv -> -0), i.e. step 1.
"""

import torch

INF = float("inf")


# --------------------------------------------------------------------------------------------------
# qpth-form linear algebra
# --------------------------------------------------------------------------------------------------
def _solve(M, r):
    return torch.linalg.solve(M, r.unsqueeze(-1)).squeeze(-1)


def _solve_kkt(Q, G, S, d, rx, rs, rz):
    """qpth solve_kkt (LU_PARTIAL): returns dx, ds, dz."""
    invQ_rx = _solve(Q, rx)
    h = (G @ invQ_rx.unsqueeze(-1)).squeeze(-1) + rs / d - rz
    w = -_solve(S, h)
    g1 = -rx - (G.transpose(1, 2) @ w.unsqueeze(-1)).squeeze(-1)
    g2 = -rs - w
    dx = _solve(Q, g1)
    ds = g2 / d
    dz = w
    return dx, ds, dz


def _get_step(v, dv):
    """Row-wise max step keeping v + a dv >= 0 (see module docstring for the global-max waiver)."""
    a = -v / dv
    a = torch.where(dv > 0, torch.full_like(a, INF), a)
    return a.min(1)[0]  # NaN propagates (torch.min), as in qpth: a NaN row stays NaN and never becomes best


def pdipm_forward(Q, p, G, h, eps=5e-2, maxIter=12, notImprovedLim=3, history=None):
    """Batched PDIPM in qpth's block form.  Q (B,nz,nz), p (B,nz), G (B,m,nz) or (m,nz), h (B,m).

    Returns (x, lam, slack, n_iter) with qpth's batch-global stopping and per-row best iterate.
    """
    B, nz = p.shape
    if G.dim() == 2:
        G = G.unsqueeze(0).expand(B, -1, -1)
    m = G.shape[1]
    dt = Q.dtype
    R = G @ torch.linalg.solve(Q, G.transpose(1, 2))

    def factor(d):
        return R + torch.diag_embed(1.0 / d)

    d = torch.ones(B, m, dtype=dt)
    S = factor(d)
    x, s, z = _solve_kkt(Q, G, S, d, p, torch.zeros(B, m, dtype=dt), -h)
    Mn = s.min(1)[0]
    s = torch.where((Mn < 0)[:, None], s - Mn[:, None] + 1, s)
    Mn = z.min(1)[0]
    z = torch.where((Mn < 0)[:, None], z - Mn[:, None] + 1, z)

    best = None
    nNotImproved = 0
    it = 0
    for it in range(maxIter):
        rx = (G.transpose(1, 2) @ z.unsqueeze(-1)).squeeze(-1) + (Q @ x.unsqueeze(-1)).squeeze(-1) + p
        rs = z
        rz = (G @ x.unsqueeze(-1)).squeeze(-1) + s - h
        mu = torch.abs((s * z).sum(1) / m)
        resids = torch.linalg.norm(rz, dim=1) + torch.linalg.norm(rx, dim=1) + m * mu
        d = z / s
        S = factor(d)
        if best is None:
            best = {"resids": resids.clone(), "x": x.clone(), "z": z.clone(), "s": s.clone()}
            nNotImproved = 0
        else:
            I = resids < best["resids"]  # False for NaN: a NaN iterate never becomes best
            nNotImproved = 0 if bool(I.any()) else nNotImproved + 1
            best["resids"] = torch.where(I, resids, best["resids"])
            for k, v in (("x", x), ("z", z), ("s", s)):
                best[k] = torch.where(I[:, None], v, best[k])
        if history is not None:
            history.append({"resids": resids.clone(), "mu": mu.clone(), "x": x.clone(), "z": z.clone(), "s": s.clone()})
        if nNotImproved == notImprovedLim or bool(best["resids"].max() < eps) or bool(mu.min() > 1e32):
            return best["x"], best["z"], best["s"], it + 1

        dx_a, ds_a, dz_a = _solve_kkt(Q, G, S, d, rx, rs, rz)
        alpha = torch.minimum(torch.minimum(_get_step(z, dz_a), _get_step(s, ds_a)), torch.ones(B, dtype=dt))
        t1 = s + alpha[:, None] * ds_a
        t2 = z + alpha[:, None] * dz_a
        sig = ((t1 * t2).sum(1) / (s * z).sum(1)) ** 3
        rs2 = ((-mu * sig)[:, None] + ds_a * dz_a) / s
        zx = torch.zeros(B, nz, dtype=dt)
        zm = torch.zeros(B, m, dtype=dt)
        dx_c, ds_c, dz_c = _solve_kkt(Q, G, S, d, zx, rs2, zm)
        dx, ds, dz = dx_a + dx_c, ds_a + ds_c, dz_a + dz_c
        alpha = torch.minimum(0.999 * torch.minimum(_get_step(z, dz), _get_step(s, ds)), torch.ones(B, dtype=dt))
        x = x + alpha[:, None] * dx
        s = s + alpha[:, None] * ds
        z = z + alpha[:, None] * dz
    return best["x"], best["z"], best["s"], it + 1


# --------------------------------------------------------------------------------------------------
# box-constrained reduced form (what the HIP kernel implements)
# --------------------------------------------------------------------------------------------------
def _solve_kkt_box(Q, d, rx, rs, rz):
    """G = [I; -I]: (Q + diag(d_u + d_l)) dx = -rx - G'(d*rz - rs); dz = d*(G dx + rz) - rs; ds = (-rs - dz)/d."""
    nz = Q.shape[-1]
    t = d * rz - rs
    rhs = -rx - (t[:, :nz] - t[:, nz:])
    M = Q + torch.diag_embed(d[:, :nz] + d[:, nz:])
    dx = _solve(M, rhs)
    Gdx = torch.cat([dx, -dx], dim=1)
    dz = d * (Gdx + rz) - rs
    ds = (-rs - dz) / d
    return dx, ds, dz


def pdipm_forward_box(Q, p, lower, upper, eps=5e-2, maxIter=12, notImprovedLim=3, history=None):
    """Same iteration as ``pdipm_forward`` for  lower <= z <= upper  (h = [upper; -lower])."""
    B, nz = p.shape
    m = 2 * nz
    dt = Q.dtype
    h = torch.cat([upper, -lower], dim=1)

    def Gx(v):
        return torch.cat([v, -v], dim=1)

    def Gt(v):
        return v[:, :nz] - v[:, nz:]

    d = torch.ones(B, m, dtype=dt)
    x, s, z = _solve_kkt_box(Q, d, p, torch.zeros(B, m, dtype=dt), -h)
    Mn = s.min(1)[0]
    s = torch.where((Mn < 0)[:, None], s - Mn[:, None] + 1, s)
    Mn = z.min(1)[0]
    z = torch.where((Mn < 0)[:, None], z - Mn[:, None] + 1, z)
    best = None
    nNotImproved = 0
    it = 0
    for it in range(maxIter):
        rx = Gt(z) + (Q @ x.unsqueeze(-1)).squeeze(-1) + p
        rs = z
        rz = Gx(x) + s - h
        mu = torch.abs((s * z).sum(1) / m)
        resids = torch.linalg.norm(rz, dim=1) + torch.linalg.norm(rx, dim=1) + m * mu
        d = z / s
        if best is None:
            best = {"resids": resids.clone(), "x": x.clone(), "z": z.clone(), "s": s.clone()}
            nNotImproved = 0
        else:
            I = resids < best["resids"]
            nNotImproved = 0 if bool(I.any()) else nNotImproved + 1
            best["resids"] = torch.where(I, resids, best["resids"])
            for k, v in (("x", x), ("z", z), ("s", s)):
                best[k] = torch.where(I[:, None], v, best[k])
        if history is not None:
            history.append({"resids": resids.clone(), "mu": mu.clone(), "x": x.clone(), "z": z.clone(), "s": s.clone()})
        if nNotImproved == notImprovedLim or bool(best["resids"].max() < eps) or bool(mu.min() > 1e32):
            return best["x"], best["z"], best["s"], it + 1
        dx_a, ds_a, dz_a = _solve_kkt_box(Q, d, rx, rs, rz)
        alpha = torch.minimum(torch.minimum(_get_step(z, dz_a), _get_step(s, ds_a)), torch.ones(B, dtype=dt))
        t1 = s + alpha[:, None] * ds_a
        t2 = z + alpha[:, None] * dz_a
        sig = ((t1 * t2).sum(1) / (s * z).sum(1)) ** 3
        rs2 = ((-mu * sig)[:, None] + ds_a * dz_a) / s
        dx_c, ds_c, dz_c = _solve_kkt_box(Q, d, torch.zeros(B, nz, dtype=dt), rs2, torch.zeros(B, m, dtype=dt))
        dx, ds, dz = dx_a + dx_c, ds_a + ds_c, dz_a + dz_c
        alpha = torch.minimum(0.999 * torch.minimum(_get_step(z, dz), _get_step(s, ds)), torch.ones(B, dtype=dt))
        x = x + alpha[:, None] * dx
        s = s + alpha[:, None] * ds
        z = z + alpha[:, None] * dz
    return best["x"], best["z"], best["s"], it + 1


def stop_rule(resid, mu, eps=5e-2, notImprovedLim=3):
    """qpth's batch-global stopping rule replayed on the (B, maxIter) tables of per-iteration residuals / mu
    (the loop head of ``pdipm_forward`` above: best-tracking, nNotImproved, the three stop conditions).

    -> (k_star, best_iter): k_star = index of the iteration at which the batch stops (n_iter = k_star + 1; maxIter - 1 if
    no condition fires), best_iter (B,) = each row's best iteration among 0..k_star (first strict minimum; an iterate
    with a NaN residual never replaces the best)."""
    resid = torch.as_tensor(resid)
    mu = torch.as_tensor(mu)
    B, T = resid.shape
    best = resid[:, 0].clone()
    bi = torch.zeros(B, dtype=torch.long)
    n_not = 0
    k_star = T - 1
    for it in range(T):
        if it > 0:
            I = resid[:, it] < best
            n_not = 0 if bool(I.any()) else n_not + 1
            best = torch.where(I, resid[:, it], best)
            bi = torch.where(I, torch.full_like(bi, it), bi)
        if n_not == notImprovedLim or bool(best.max() < eps) or bool(mu[:, it].min() > 1e32):
            k_star = it
            break
    return k_star, bi


# --------------------------------------------------------------------------------------------------
# autograd function with qpth's implicit backward
# --------------------------------------------------------------------------------------------------
LAST = {"n_iter": None}  # iteration count of the most recent forward (tests compare it with the HIP kernels')


class _QPFunctionFn(torch.autograd.Function):
    """x = argmin QP;  backward = one KKT solve at the returned (x, lam, slack)  (qpth/qp.py backward).

    d = clamp(lam,1e-8)/clamp(slack,1e-8); (dx,_,dlam) = solve_kkt(d, dl_dx, 0, 0);
    grad_Q = 1/2 (dx x' + x dx'); grad_p = dx; grad_h = -dlam (G is a constant here).
    """

    @staticmethod
    def forward(ctx, Q, p, G, h, eps, maxIter, box_form):
        Qd, pd, hd = Q.detach(), p.detach(), h.detach()
        if box_form:
            nz = pd.shape[1]
            x, lam, slack, nit = pdipm_forward_box(Qd, pd, -hd[:, nz:], hd[:, :nz], eps, maxIter)
        else:
            x, lam, slack, nit = pdipm_forward(Qd, pd, G.detach(), hd, eps, maxIter)
        ctx.save_for_backward(x, lam, slack, Qd, G.detach())
        ctx.n_iter = nit
        ctx.box_form = bool(box_form)
        LAST["n_iter"] = nit
        return x

    @staticmethod
    def backward(ctx, dl_dx):
        x, lam, slack, Q, G = ctx.saved_tensors
        B, nz = x.shape
        if G.dim() == 2:
            G = G.unsqueeze(0).expand(B, -1, -1)
        m = G.shape[1]
        d = torch.clamp(lam, min=1e-8) / torch.clamp(slack, min=1e-8)
        if ctx.box_form:  # the same KKT solve in the reduced nz x nz form (large batches: no (B, 2nz, 2nz) tensors)
            z0 = torch.zeros(B, 2 * nz, dtype=Q.dtype)
            dx, _, dlam = _solve_kkt_box(Q, d, dl_dx, z0, z0)
            dQ = 0.5 * (dx.unsqueeze(2) * x.unsqueeze(1) + x.unsqueeze(2) * dx.unsqueeze(1))
            return dQ, dx, None, -dlam, None, None, None
        R = G @ torch.linalg.solve(Q, G.transpose(1, 2))
        S = R + torch.diag_embed(1.0 / d)
        z0 = torch.zeros(B, m, dtype=Q.dtype)
        dx, _, dlam = _solve_kkt(Q, G, S, d, dl_dx, z0, z0)
        dQ = 0.5 * (dx.unsqueeze(2) * x.unsqueeze(1) + x.unsqueeze(2) * dx.unsqueeze(1))
        return dQ, dx, None, -dlam, None, None, None


def QPFunction(eps=5e-2, maxIter=12, box_form=False):
    """Callable with the ``qpth.qp.QPFunction(...)(Q, p, G, h, A, b)`` calling convention."""

    def f(Q, p, G, h, A=None, b=None):
        return _QPFunctionFn.apply(Q, p, G, h, eps, maxIter, box_form)

    return f


def lsq_box_qp(A, b, min_bound, max_bound, eps=5e-2, maxIter=12, box_form=False):
    """Restates ``SQPLsqSolver.solve`` (reference qp_solver.py:60-134) for A (B,m,nz), b (B,m).

    Q = A'A + 1e-4 I, p = -A'b, G = [I;-I], h = [max; -min];  value = 1/2 |b - A x|^2.
    Returns (value (B,), x (B,nz)), differentiable wrt A and b.
    """
    B, mdim, nz = A.shape
    dt = A.dtype
    Q = A.transpose(1, 2) @ A + torch.eye(nz, dtype=dt).unsqueeze(0) * 1e-4
    p = (-A.transpose(1, 2) @ b.unsqueeze(-1)).squeeze(-1)
    G = torch.cat([torch.eye(nz, dtype=dt), -torch.eye(nz, dtype=dt)], dim=0)
    u = torch.ones(B, nz, dtype=dt) * max_bound
    l = torch.ones(B, nz, dtype=dt) * min_bound
    h = torch.cat([u, -l], dim=-1)
    x = QPFunction(eps, maxIter, box_form)(Q, p, G, h)
    value = 0.5 * ((b - (A @ x.unsqueeze(-1)).squeeze(-1)) ** 2).sum(-1)
    return value, x
