"""numpy reference for tests of the reduced-system band solver (csrc/band_kernels2.h, band_sub.h): random block-banded SPD
systems in the band storage of the C ABI, the dense solution, and the intermediates of the substructured elimination
(same algebra as scripts/lab/substructure_proto.py)."""
import numpy as np


def random_band_system(nrows_per_comp, b, dc, seed=0):
    """-> band (N, b+1, dc, dc) [block d of row i = (i, i-d)], dense A (n, n), comp_ptr, rhs (2, n)"""
    rng = np.random.default_rng(seed)
    N = int(sum(nrows_per_comp)); n = N * dc
    A = np.zeros((n, n)); comp_ptr = [0]
    for rows in nrows_per_comp:
        c0 = comp_ptr[-1]
        for i in range(c0, c0 + rows):
            for j in range(max(c0, i - b), i + 1):
                B = rng.normal(size=(dc, dc))
                if i == j:
                    B = (B + B.T) / 2
                A[i*dc:(i+1)*dc, j*dc:(j+1)*dc] = B
                A[j*dc:(j+1)*dc, i*dc:(i+1)*dc] = B.T
        comp_ptr.append(c0 + rows)
    A += np.eye(n) * (2.5 * (2 * b + 1) * dc ** 0.5)
    band = np.zeros((N, b + 1, dc, dc))
    for i in range(N):
        for d in range(0, min(b, i) + 1):
            band[i, d] = A[i*dc:(i+1)*dc, (i-d)*dc:(i-d+1)*dc]
    rhs = rng.normal(size=(2, n))
    return band, A, np.array(comp_ptr, np.int32), rhs


def segment_table(comp_ptr, b, P):
    """The partition csrc/band_sub.h:sub_build makes for a forced P."""
    segs, seps = [], []
    for c in range(len(comp_ptr) - 1):
        c0, rows = int(comp_ptr[c]), int(comp_ptr[c + 1] - comp_ptr[c])
        p = P
        while p > 1 and (rows - (p - 1) * b) // p < b + 1:
            p -= 1
        m_total = rows - (p - 1) * b; pos = c0
        for i in range(p):
            m = m_total // p + (1 if i < m_total % p else 0)
            segs.append((pos, pos + m, pos + m + b if i + 1 < p else pos + m, i > 0)); pos += m
            if i + 1 < p:
                seps.append((pos, len(segs))); pos += b
    return segs, seps


def substructure_intermediates(A, rhs, segs, seps, b, dc):
    """Z (Q, n), D (nsep, Q, Q) lower, T (nsep, 2, Q) as the device computes them."""
    n = A.shape[0]; Q = b * dc
    Wm = A.copy(); Y = rhs.T.copy()
    blk = lambda M, i, j: M[i*dc:(i+1)*dc, j*dc:(j+1)*dc]
    G = {}
    for (r0, r1, re, _) in segs:
        for j in range(r0, r1):
            Ljj = np.linalg.cholesky(blk(Wm, j, j)); G[j] = np.linalg.inv(Ljj)
            Y[j*dc:(j+1)*dc] = G[j] @ Y[j*dc:(j+1)*dc]
            nb = min(b, re - 1 - j)
            for k in range(1, nb + 1):
                blk(Wm, j + k, j)[:] = blk(Wm, j + k, j) @ G[j].T
            for i in range(1, nb + 1):
                for k in range(1, i + 1):
                    blk(Wm, j + i, j + k)[:] -= blk(Wm, j + i, j) @ blk(Wm, j + k, j).T
                Y[(j+i)*dc:(j+i+1)*dc] -= blk(Wm, j + i, j) @ Y[j*dc:(j+1)*dc]
    Z = np.zeros((Q, n))
    for (r0, r1, re, has_left) in segs:
        if not has_left:
            continue
        p0 = r0 - b
        C = np.zeros(((re - r0) * dc, Q))
        for k in range(r0, min(r0 + b, r1)):
            for c in range(max(p0, k - b), r0):
                C[(k-r0)*dc:(k-r0+1)*dc, (c-p0)*dc:(c-p0+1)*dc] = blk(Wm, k, c)
        L = np.zeros(((re - r0) * dc, (re - r0) * dc))
        for k in range(r0, re):
            for j in range(max(r0, k - b), k):
                if j < r1:
                    L[(k-r0)*dc:(k-r0+1)*dc, (j-r0)*dc:(j-r0+1)*dc] = blk(Wm, k, j)
            L[(k-r0)*dc:(k-r0+1)*dc, (k-r0)*dc:(k-r0+1)*dc] = np.linalg.inv(G[k]) if k < r1 else np.eye(dc)
        Zs = np.linalg.solve(L, C)
        Z[:, r0*dc:re*dc] = Zs.T
    D = np.zeros((len(seps), Q, Q)); T = np.zeros((len(seps), 2, Q))
    for s, (p0, rs) in enumerate(seps):
        r0, r1, re, _ = segs[rs]
        Dj = np.tril(Wm[p0*dc:(p0+b)*dc, p0*dc:(p0+b)*dc]); Dj = Dj + np.tril(Dj, -1).T
        Zs = Z[:, r0*dc:r1*dc]
        D[s] = np.tril(Dj - Zs @ Zs.T)
        T[s] = (Y[p0*dc:(p0+b)*dc] - Zs @ Y[r0*dc:r1*dc]).T
    return Z, D, T
