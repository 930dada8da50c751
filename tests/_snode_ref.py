"""numpy restatement of csrc/snode.h's elimination, driven by the plan tables ssfm_snode_plan_probe returns (host only): the same halves, steps, block tables, exchange
and redundant solve of [M, T] -- so the CPU suite checks the planner and the algorithm; the GPU suite (test_snode_gpu.py) then checks the kernel against the dense solve."""
import numpy as np

Q = 30


def gather(plan, blocks, dc, tab, nrows, ncols, ncb):
    out = np.zeros((nrows, ncols))
    if tab < 0:
        return out
    t = plan["tab"]
    for i in range(nrows):
        a, u = divmod(i, dc)
        for j in range(ncols):
            b, v = divmod(j, dc)
            e = t[tab + a * ncb + b]
            if e == -1:
                continue
            if e == -2:
                out[i, j] = 1.0 if u == v else 0.0
            else:
                blk = blocks[e & 0x3fffffff]
                out[i, j] = blk[v, u] if (e & 0x40000000) else blk[u, v]
    return out


def gather_rhs(plan, rhs2, dc, node, nrows, load):
    out = np.zeros((nrows, 2))
    if not load:
        return out
    cams = plan["node_cam"][node]
    for i in range(nrows):
        a, u = divmod(i, dc)
        if cams[a] >= 0:
            out[i] = rhs2[:, cams[a] * dc + u]
    return out


def solve(plan, blocks, rhs2, dc):
    """-> Y (2, Nc * dc) in camera order"""
    S, CAPT = plan["S"], plan["CAPT"]
    Nc = int(plan["node_cam"].max()) + 1
    Y = np.zeros((2, Nc * dc))
    state = []
    for hr in plan["half_rec"]:
        ns, s0, partner, owner, tnode, QT = (int(x) for x in hr[:6])
        QT = QT if tnode >= 0 else 0
        sr = plan["step_rec"][s0:s0 + ns]
        D = gather(plan, blocks, dc, hr[9], Q, Q, S)
        E = gather(plan, blocks, dc, hr[10], QT, Q, S) if QT else np.zeros((0, Q))
        ATT = gather(plan, blocks, dc, hr[11], QT, QT, CAPT) if QT else np.zeros((0, 0))
        g = [gather_rhs(plan, rhs2, dc, hr[13], Q, ns > 0 or owner)] + [gather_rhs(plan, rhs2, dc, sr[k][1], Q, sr[k][5] != 0) for k in range(ns)]
        gT = gather_rhs(plan, rhs2, dc, tnode, QT, owner) if QT else np.zeros((0, 2))
        fac = []
        for k in range(ns):
            Cm = gather(plan, blocks, dc, sr[k][2], Q, Q, S)
            L = np.linalg.cholesky(np.tril(D) + np.tril(D, -1).T)                   # the kernel reads the lower triangle only
            Lsd = np.linalg.solve(L, Cm.T).T
            LTP = np.linalg.solve(L, E.T).T
            y = np.linalg.solve(L, g[k])
            g[k + 1] = g[k + 1] - Lsd @ y
            gT = gT - LTP @ y
            Dn = gather(plan, blocks, dc, sr[k][3], Q, Q, S) - Lsd @ Lsd.T
            En = (gather(plan, blocks, dc, sr[k][4], QT, Q, S) if QT else np.zeros((0, Q))) - LTP @ Lsd.T
            ATT = ATT - LTP @ LTP.T
            fac.append((L, Lsd, LTP, y, int(sr[k][0])))
            D, E = Dn, En
        state.append(dict(hr=hr, DM=D, EM=E, ATT=ATT, gM=g[ns], gT=gT, fac=fac, QT=QT))
    for hi, st in enumerate(state):
        hr = st["hr"]; partner = int(hr[2]); owner = int(hr[3]); QT = st["QT"]
        DM, EM, ATT, gM, gT = st["DM"], st["EM"], st["ATT"], st["gM"], st["gT"]
        if partner >= 0:
            o = state[partner]
            DM, EM, ATT, gM, gT = DM + o["DM"], EM + o["EM"], ATT + o["ATT"], gM + o["gM"], gT + o["gT"]
        LM = np.linalg.cholesky(np.tril(DM) + np.tril(DM, -1).T)
        LTM = np.linalg.solve(LM, EM.T).T
        yM = np.linalg.solve(LM, gM); gT = gT - LTM @ yM
        if QT:
            TT = ATT - LTM @ LTM.T
            LT = np.linalg.cholesky(np.tril(TT) + np.tril(TT, -1).T)
            xT = np.linalg.solve(LT.T, np.linalg.solve(LT, gT))
        else:
            xT = np.zeros((0, 2))
        xM = np.linalg.solve(LM.T, yM - LTM.T @ xT)

        def put(node, x, nrows):
            cams = plan["node_cam"][node]
            for i in range(nrows):
                a, u = divmod(i, dc)
                if cams[a] >= 0:
                    Y[:, cams[a] * dc + u] = x[i]
        if owner:
            put(int(hr[12]), xM, Q)
            if QT:
                put(int(hr[4]), xT, QT)
        xN = xM
        for L, Lsd, LTP, y, node in reversed(st["fac"]):
            x = np.linalg.solve(L.T, y - Lsd.T @ xN - LTP.T @ xT)
            put(node, x, Q); xN = x
    return Y
