"""CPU prototype of the smoothed-aggregation hierarchy of csrc/sagg.hip (development aid: explores cycle
shapes and iteration counts with numpy / scipy before a variant is written as kernels; not part of the
product or the test suite).

    python tools/sa_proto.py N [variant ...]

Mirrors the device setup: hashed-priority distance-2 MIS aggregation (6 rounds), P = (I - w D^-1 A) P_tent
truncated to 4 entries per row (rest lumped), Galerkin products, damped-Jacobi smoothing, tail levels
(<= 1024 rows) with their own V(3,3), dense coarsest solve, flexible CG (Polak-Ribiere).
"""
import sys, time
import numpy as np
import scipy.sparse as sp

OMEGA = 0.85
OMEGA_P = 0.70  # level 0 (csrc/sagg.hip; 2/3 until the end of round 5); the device uses 0.85 on the coarse levels:
# pass omega_p per level to build_P to mirror that (see tools/experiments/README.md)
PW = 4
MIS_ROUNDS = 6
TAIL_MAX_N = 1024
COARSEST = 64


def grid_matrix(N):
    idx = np.arange(N * N).reshape(N, N)
    a = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    b = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    n = N * N
    g = np.ones(a.size)
    A = sp.coo_matrix((np.concatenate([g, g, -g, -g]), (np.concatenate([a, b, a, b]), np.concatenate([a, b, b, a]))),
                      shape=(n, n)).tocsr()
    A = A[: n - 1, : n - 1].tocsr()  # last node is ground
    rhs = np.zeros(n - 1)
    rhs[0] = 1.0
    return A, rhs


def hash30(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x & np.uint32(0x3fffffff)


def nbr_max(A, v):
    """max of v over the closed neighbourhood (pattern of A, which holds the diagonal)"""
    g = v[A.indices]
    out = np.maximum.reduceat(g, A.indptr[:-1])
    return np.maximum(out, v)


def aggregate(A, rounds=MIS_ROUNDS, dist=2):
    n = A.shape[0]
    T = (np.uint32(1) << np.uint32(30)) | hash30(np.arange(n))
    for _ in range(rounds):
        m = T.copy()
        for _d in range(dist):
            m = nbr_max(A, m)
        und = (T >> 30) == 1
        root = und & (m == T)
        out = und & ~root & ((m >> 30) == 3)
        T = np.where(root, T | (np.uint32(2) << np.uint32(30)), T)
        T = np.where(out, np.uint32(0), T)
    state = T >> 30
    isroot = state == 3
    near_root = nbr_max(A, np.where(isroot, np.uint32(1), np.uint32(0))) > 0
    flag = isroot | ((state == 1) & ~near_root)
    ident = np.cumsum(flag) - 1
    nc = int(flag.sum())
    # roots and their neighbours (root of highest priority)
    key = np.where(flag, (T | (np.uint32(3) << np.uint32(30))).astype(np.int64) * (1 << 31) + ident, -1)
    g = key[A.indices]
    best = np.maximum.reduceat(g, A.indptr[:-1])
    agg1 = np.where(flag, ident, np.where(best >= 0, best & ((1 << 31) - 1), -1)).astype(np.int64)
    # the rest joins the aggregate of its strongest assigned neighbour (repeated for distance-3 / -4 sets)
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    agg = agg1
    for _ in range(max(1, dist - 1) + 2):
        if (agg >= 0).all():
            break
        w = np.abs(A.data) * (A.indices != rows) * (agg[A.indices] >= 0)
        order = np.lexsort((-w, rows))
        first = order[A.indptr[:-1]]
        far = np.where(w[first] > 0, agg[A.indices[first]], -1)
        agg = np.where(agg >= 0, agg, far)
    assert (agg >= 0).all()
    return agg, nc


def build_P(A, agg, nc, pw=PW, omega_p=OMEGA_P):
    n = A.shape[0]
    Pt = sp.csr_matrix((np.ones(n), (np.arange(n), agg)), shape=(n, nc))
    d = A.diagonal()
    S = sp.identity(n, format="csr") - omega_p * sp.diags(1.0 / d) @ A
    P = (S @ Pt).tocsr()
    if pw:
        # keep own aggregate + the first (pw - 1) others in column-appearance order; lump the rest into own
        P.sort_indices()
        rows = np.repeat(np.arange(n), np.diff(P.indptr))
        own = P.indices == agg[rows]
        # rank of the non-own entries within the row (the device takes them in A-slot order: close enough)
        nonown_rank = np.cumsum(~own) - np.repeat(np.cumsum(~own)[P.indptr[:-1]] - (~own)[P.indptr[:-1]], np.diff(P.indptr))
        keep = own | (nonown_rank < pw - 1)
        lump = np.bincount(rows[~keep], weights=P.data[~keep], minlength=n)
        data = P.data.copy()
        data[own] += lump[rows[own]]
        P = sp.csr_matrix((data[keep], (rows[keep], P.indices[keep])), shape=(n, nc))
    return P


class Level:
    pass


def setup(A, trace=True, pw=PW, dist=2, rounds=MIS_ROUNDS, dist_by_level=None, smooth_twice_from=None):
    levels = []
    while True:
        L = Level()
        L.A = A.tocsr()
        L.n = A.shape[0]
        L.dinv = 1.0 / A.diagonal()
        levels.append(L)
        if L.n <= COARSEST and len(levels) > 1:
            L.inv = np.linalg.inv(A.toarray())
            break
        lv = len(levels) - 1
        d_here = dist_by_level[min(lv, len(dist_by_level) - 1)] if dist_by_level else dist
        agg, nc = aggregate(L.A, rounds=rounds + 2 * (d_here - 2), dist=d_here)
        L.P = build_P(L.A, agg, nc, pw=pw if d_here == 2 else 0)
        if smooth_twice_from is not None and lv >= smooth_twice_from and d_here > 2:
            d = L.A.diagonal()
            S = sp.identity(L.n, format="csr") - OMEGA_P * sp.diags(1.0 / d) @ L.A
            L.P = (S @ L.P).tocsr()
        L.R = L.P.T.tocsr()
        A = (L.R @ L.A @ L.P).tocsr()
        A.eliminate_zeros()
    if trace:
        print("levels:", " ".join(f"{L.n}/{L.A.nnz / L.n:.1f}" for L in levels))
    return levels


def est_rho(L, iters=15):
    rng = np.random.default_rng(0)
    v = rng.standard_normal(L.n)
    for _ in range(iters):
        v = L.dinv * (L.A @ v)
        lam = np.linalg.norm(v)
        v /= lam
    return lam


def smooth(L, b, x, nu, omega=OMEGA, kind="jacobi"):
    if kind == "jacobi":
        for _ in range(nu):
            x = x + omega * L.dinv * (b - L.A @ x)
        return x
    if kind == "cheb":  # Chebyshev on [rho/alpha, rho] of D^-1 A, degree nu
        rho = L.rho * 1.05
        lo = rho / L.cheb_ratio
        theta, delta = 0.5 * (rho + lo), 0.5 * (rho - lo)
        sigma = theta / delta
        rk = 1.0 / sigma
        r = L.dinv * (b - L.A @ x)
        dk = r / theta
        for k in range(nu):
            x = x + dk
            if k + 1 == nu:
                break
            r = L.dinv * (b - L.A @ x)
            rk1 = 1.0 / (2 * sigma - rk)
            dk = rk1 * rk * dk + 2 * rk1 / delta * r
            rk = rk1
        return x
    raise ValueError(kind)


def cycle(levels, l, b, cfg):
    L = levels[l]
    if l == len(levels) - 1:
        return L.inv @ b
    if cfg.get("exact") == l:
        if not hasattr(L, "lu"):
            import scipy.sparse.linalg as spla
            L.lu = spla.splu(L.A.tocsc())
        return L.lu.solve(b)
    in_tail = L.n <= TAIL_MAX_N
    nu = cfg["tail_nu"] if in_tail else cfg["nu"][min(l, len(cfg["nu"]) - 1)]
    kind = "jacobi" if in_tail else cfg.get("smoother", "jacobi")
    om = cfg.get("omega", OMEGA)
    x = smooth(L, b, np.zeros_like(b), nu, om, kind)
    r = b - L.A @ x
    rc = L.R @ r
    C = levels[l + 1]
    k = cfg["klevels"]
    use_k = (l + 1) in cfg["kset"] if "kset" in cfg else l < k
    if use_k and (C.n > TAIL_MAX_N or cfg.get("ktail")) and l + 1 != len(levels) - 1:
        c1 = cycle(levels, l + 1, rc, cfg)
        v1 = C.A @ c1
        rho1, alpha1 = c1 @ v1, c1 @ rc
        r2 = rc - (alpha1 / rho1) * v1
        cfg.setdefault("_kstat", [0, 0])
        cfg["_kstat"][0] += 1
        if cfg.get("kskip") and np.linalg.norm(r2) <= cfg["kskip"] * np.linalg.norm(rc):
            cfg["_kstat"][1] += 1
            x = x + L.P @ ((alpha1 / rho1) * c1)
            return smooth(L, b, x, nu, om, kind)
        c2 = cycle(levels, l + 1, r2, cfg)
        v2 = C.A @ c2
        gamma, beta, alpha2 = c2 @ v1, c2 @ v2, c2 @ r2
        rho2 = beta - gamma * gamma / rho1
        e = (alpha1 / rho1 - gamma * alpha2 / (rho1 * rho2)) * c1 + (alpha2 / rho2) * c2
    else:
        e = cycle(levels, l + 1, rc, cfg)
        for _ in range(cfg.get("gamma", 1) - 1 if (l + 1 < cfg.get("wlevels", 0) + 1) else 0):  # W-cycle
            e = e + cycle(levels, l + 1, rc - C.A @ e, cfg)
    x = x + L.P @ e
    return smooth(L, b, x, nu, om, kind)


def additive_cycle(levels, b, cfg):
    """Additive (BPX-like) cycle: every level smooths the restricted residual independently (symmetric: nu Jacobi
    sweeps from zero, which is a polynomial in D^-1 A applied to the residual), the corrections are summed up the
    hierarchy.  One restriction chain down, one prolongation chain up, all smoothing in parallel."""
    rs = [b]
    for l in range(len(levels) - 1):
        rs.append(levels[l].R @ rs[-1])
    es = []
    for l, L in enumerate(levels):
        if l == len(levels) - 1:
            es.append(L.inv @ rs[l])
        else:
            nu = cfg["tail_nu"] if L.n <= TAIL_MAX_N else cfg["nu"][min(l, len(cfg["nu"]) - 1)]
            es.append(smooth(L, rs[l], np.zeros_like(rs[l]), cfg.get("add_nu", 2 * nu), cfg.get("omega", OMEGA), "jacobi"))
    e = es[-1]
    for l in range(len(levels) - 2, -1, -1):
        e = es[l] + levels[l].P @ e
    return e


def fcg(levels, b, cfg, tol=1e-13, maxit=300):
    A = levels[0].A
    x = np.zeros_like(b)
    r = b.copy()
    bb = b @ b
    p = None
    Ap = None
    rz_old = alpha = 0.0
    for it in range(maxit):
        rr = r @ r
        if rr <= tol * tol * bb:
            return it, np.sqrt(rr / bb)
        z = additive_cycle(levels, r, cfg) if cfg.get("additive") else cycle(levels, 0, r, cfg)
        rz = z @ r
        if it == 0:
            p = z.copy()
        else:
            beta = -alpha * (z @ Ap) / rz_old
            p = z + beta * p
        Ap = A @ p
        alpha = rz / (p @ Ap)
        x += alpha * p
        r -= alpha * Ap
        rz_old = rz
    return maxit, np.sqrt((r @ r) / bb)


VARIANTS = {
    "base": dict(nu=[1, 1, 2], tail_nu=3, klevels=1),
    "v": dict(nu=[1, 1, 2], tail_nu=3, klevels=0),
    "v222": dict(nu=[2, 2, 2], tail_nu=3, klevels=0),
    "k2": dict(nu=[1, 1, 2], tail_nu=3, klevels=2),
    "v122": dict(nu=[1, 2, 2], tail_nu=3, klevels=0),
    "v133": dict(nu=[1, 3, 3], tail_nu=3, klevels=0),
    "w1": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, gamma=2, wlevels=1),
    "w1_122": dict(nu=[1, 2, 2], tail_nu=3, klevels=0, gamma=2, wlevels=1),
    "cheb2": dict(nu=[2, 2, 2], tail_nu=3, klevels=0, smoother="cheb"),
    "cheb3": dict(nu=[3, 3, 3], tail_nu=3, klevels=0, smoother="cheb"),
    "cheb133": dict(nu=[1, 3, 3], tail_nu=3, klevels=0, smoother="cheb"),
    "exact1": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, exact=1),
    "exact2": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, exact=2),
    "exact3": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, exact=3),
    "exact1_2": dict(nu=[2, 1, 2], tail_nu=3, klevels=0, exact=1),
    "k212": dict(nu=[2, 1, 2], tail_nu=3, klevels=1),
    "k222": dict(nu=[2, 2, 2], tail_nu=3, klevels=1),
    "k312": dict(nu=[3, 1, 2], tail_nu=3, klevels=1),
    "k211": dict(nu=[2, 1, 1], tail_nu=3, klevels=1),
    "k111": dict(nu=[1, 1, 1], tail_nu=3, klevels=1),
    "kcheb212": dict(nu=[2, 1, 2], tail_nu=3, klevels=1, smoother="cheb"),
    "k233": dict(nu=[2, 3, 3], tail_nu=3, klevels=1),
    "k2_222": dict(nu=[2, 2, 2], tail_nu=3, klevels=2),
    "k2_212": dict(nu=[2, 1, 2], tail_nu=3, klevels=2),
    "k2_112": dict(nu=[1, 1, 2], tail_nu=3, klevels=2),
    "k3_212": dict(nu=[2, 1, 2], tail_nu=3, klevels=3),
    "kskip25": dict(nu=[1, 1, 2], tail_nu=3, klevels=1, kskip=0.25),
    "kskip35": dict(nu=[1, 1, 2], tail_nu=3, klevels=1, kskip=0.35),
    "kskip50": dict(nu=[1, 1, 2], tail_nu=3, klevels=1, kskip=0.5),
    "add2": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, additive=True, add_nu=2),
    "add1": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, additive=True, add_nu=1),
    "add3": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, additive=True, add_nu=3),
    "kat2": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, kset={2}),
    "kat2_122": dict(nu=[1, 2, 2], tail_nu=3, klevels=0, kset={2}),
    "kat2_212": dict(nu=[2, 1, 2], tail_nu=3, klevels=0, kset={2}),
    "kat2_113": dict(nu=[1, 1, 3], tail_nu=3, klevels=0, kset={2}),
    "kat23": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, kset={2, 3}, ktail=True),
    "kat3": dict(nu=[1, 1, 2], tail_nu=3, klevels=0, kset={3}, ktail=True),
    "kcheb2": dict(nu=[2, 2, 2], tail_nu=3, klevels=1, smoother="cheb"),
}


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    names = sys.argv[2:] or ["base", "v"]
    A, b = grid_matrix(N)
    t0 = time.time()
    import os
    dbl = [int(v) for v in os.environ["SA_DIST"].split(",")] if os.environ.get("SA_DIST") else None
    levels = setup(A, dist_by_level=dbl, smooth_twice_from=1 if os.environ.get("SA_TWICE") else None)
    for L in levels[:-1]:
        L.rho = est_rho(L)
        L.cheb_ratio = 4.0
    print("rho(D^-1 A) per level:", " ".join(f"{L.rho:.3f}" for L in levels[:-1]), f"  setup {time.time() - t0:.1f} s")
    for nm in names:
        cfg = VARIANTS[nm]
        t0 = time.time()
        it, res = fcg(levels, b, cfg)
        print(f"{nm:10s} iterations {it:3d}  relres {res:.1e}  ({time.time() - t0:.1f} s)  K-cycle second steps skipped {cfg.get('_kstat', [0, 0])[1]} of {cfg.get('_kstat', [0, 0])[0]}")


if __name__ == "__main__":
    main()
