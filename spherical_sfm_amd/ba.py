"""Python host side of the bundle-adjustment path: thin objects over the C ABI (include/ssfm.h).

`Context` = one GPU + one HIP stream (+ optional RCCL communicator); `BundleAdjuster` keeps a flattened
problem resident in HBM (ssfm_ba_create / reset / run / download).  `optimize()` is the one-call form that
the C++ `sphericalsfm::SfM::Optimize` shim uses (ssfm_ba_solve).
"""
import ctypes as C
import numpy as np
from . import _lib
from ._lib import BAProblemC, BAOptionsC, BASummaryC, c_double_p, c_i32_p, c_u8_p, c_i64_p

TERMINATION = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE", 3: "NOTHING_TO_DO"}


def default_options(**kw):
    o = BAOptionsC()
    _lib.lib().ssfm_ba_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class Context:
    def __init__(self, device=-1, stream=None):
        self._p = C.c_void_p()
        L = _lib.lib()
        rc = L.ssfm_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._p))
        if rc != 0:
            raise _lib.SsfmError(f"ssfm_ctx_create failed ({rc}): {L.ssfm_last_error(None).decode()}")
        self.nranks, self.rank = 1, 0
        self._adjusters = []          # weak references to the live BundleAdjusters of this context: closed with it (handles die before their context)

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.lib().ssfm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, uid: bytes, nranks: int, rank: int):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid)
        _lib.check(_lib.lib().ssfm_comm_init(self._p, buf, nranks, rank), self._p)
        self.nranks, self.rank = nranks, rank

    def comm_init_host(self, nranks: int, rank: int, allreduce):
        """Bring-your-own collective: allreduce(array (n,) float64 view, op) must reduce IN PLACE over all ranks
        (op 0 = sum, 1 = max).  Used with gloo/MPI, and by the tests to run two ranks on one GPU."""
        def _hook(user, buf, n, op):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(int(n),)), int(op))
                return 0
            except Exception:            # noqa: BLE001 - reported through the C status
                import traceback; traceback.print_exc()
                return 1
        self._hook = _lib.HOST_ALLREDUCE_FN(_hook)            # keep alive
        _lib.check(_lib.lib().ssfm_comm_init_host(self._p, nranks, rank, self._hook, None), self._p)

    def close(self):
        if self._p:
            for ref in self._adjusters:
                adj = ref()
                if adj is not None:
                    adj.close()
            self._adjusters = []
            _lib.lib().ssfm_ctx_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _ProblemBuffers:
    """Owns contiguous numpy copies of a BAProblem-like object and the C struct pointing at them."""
    def __init__(self, prob):
        self.cams = np.ascontiguousarray(prob.cameras, np.float64).copy()
        self.pts = np.ascontiguousarray(prob.points, np.float64).copy()
        self.focal = np.array([prob.focal], np.float64)
        self.xy = np.ascontiguousarray(prob.obs_xy, np.float64)
        self.oc = np.ascontiguousarray(prob.obs_cam, np.int32)
        self.op = np.ascontiguousarray(prob.obs_pt, np.int32)
        self.rf = np.ascontiguousarray(prob.rot_fixed, np.uint8)
        self.tf = np.ascontiguousarray(prob.trans_fixed, np.uint8)
        self.pf = np.ascontiguousarray(prob.pt_fixed, np.uint8)
        p = lambda a, t: a.ctypes.data_as(t)
        self.c = BAProblemC(len(self.cams), len(self.pts), len(self.oc), p(self.cams, c_double_p), p(self.pts, c_double_p),
                            p(self.focal, c_double_p), p(self.xy, c_double_p), p(self.oc, c_i32_p), p(self.op, c_i32_p),
                            p(self.rf, c_u8_p), p(self.tf, c_u8_p), p(self.pf, c_u8_p), 1 if prob.focal_fixed else 0)


def plan(prob, nranks=1, rank=0):
    """Host-only (no GPU): flatten + shard.  Returns (info dict, point ids of this rank, obs_used mask, camera order)."""
    b = _ProblemBuffers(prob)
    info = _lib.BAPlanInfoC()
    ids = np.zeros(len(b.pts), np.int32); used = np.zeros(len(b.oc), np.uint8); pos = np.zeros(len(b.cams), np.int32)
    _lib.check(_lib.lib().ssfm_ba_plan(C.byref(b.c), nranks, rank, C.byref(info), ids.ctypes.data_as(c_i32_p),
                                       used.ctypes.data_as(c_u8_p), pos.ctypes.data_as(c_i32_p)))
    d = info.as_dict()
    return d, ids[:d["num_points_used"]].copy(), used, pos


def optimize(ctx, prob, options=None, **kw):
    """One call: flatten + upload + device LM + scatter back.  Returns (cameras, points, focal, summary)."""
    b = _ProblemBuffers(prob)
    o = options or default_options(**kw)
    s = BASummaryC()
    _lib.check(_lib.lib().ssfm_ba_solve(ctx._p, C.byref(b.c), C.byref(o), C.byref(s)), ctx._p)
    return b.cams, b.pts, float(b.focal[0]), s.as_dict()


def band_solve_probe(ctx, dc, comp_ptr, band, Y, dump=False):
    """ssfm_band_solve_probe: band (N, b+1, dc, dc) lower block band, Y (2, N*dc) -> (X (2, N*dc), info dict[, Z, D, T])."""
    band = np.ascontiguousarray(band, np.float64); N, W = band.shape[0], band.shape[1]; b = W - 1
    Yc = np.ascontiguousarray(Y, np.float64).copy(); cp = np.ascontiguousarray(comp_ptr, np.int32)
    info = np.zeros(3, np.int32); Q = b * dc
    Z = np.zeros((Q, N * dc)) if dump else None
    D = np.zeros((max(N // max(b, 1), 1), Q, Q)) if dump else None
    T = np.zeros((max(N // max(b, 1), 1), 2, Q)) if dump else None
    ptr = lambda a: a.ctypes.data_as(c_double_p) if a is not None else None
    _lib.check(_lib.lib().ssfm_band_solve_probe(ctx._p, dc, N, b, len(cp) - 1, cp.ctypes.data_as(c_i32_p), ptr(band), ptr(Yc),
                                                info.ctypes.data_as(c_i32_p), ptr(Z), ptr(D), ptr(T)), ctx._p)
    d = dict(segments=int(info[0]), separators=int(info[1]), failed=int(info[2]))
    return (Yc, d, Z, D, T) if dump else (Yc, d)


def snode_solve_probe(ctx, dc, row_ptr, col_idx, S_val, rhs2, nr=2):
    """ssfm_snode_solve_probe: block-CSR S (blocks (nnz, dc, dc)), rhs2 (2, Nc*dc) -> (Y (2, Nc*dc), info dict); info['applies'] is False when the plan does not apply."""
    rp = np.ascontiguousarray(row_ptr, np.int32); ci = np.ascontiguousarray(col_idx, np.int32); Nc = len(rp) - 1
    Sv = np.ascontiguousarray(S_val, np.float64); R = np.ascontiguousarray(rhs2, np.float64)
    Y = np.zeros((2, Nc * dc)); info = np.zeros(4, np.int32)
    _lib.check(_lib.lib().ssfm_snode_solve_probe(ctx._p, dc, Nc, rp.ctypes.data_as(c_i32_p), ci.ctypes.data_as(c_i32_p), Sv.ctypes.data_as(c_double_p),
                                                 R.ctypes.data_as(c_double_p), nr, Y.ctypes.data_as(c_double_p), info.ctypes.data_as(c_i32_p)), ctx._p)
    return Y, dict(applies=bool(info[0]), workgroups=int(info[1]), t_rows=int(info[2]), failed=int(info[3]))


def snode_plan_probe(dc, row_ptr, col_idx, num_cus=256):
    """ssfm_snode_plan_probe (host only): None when the plan does not apply, else dict(nhalf, qtm, S, CAPT, half_rec (nhalf, 16), step_rec (steps, 8), node_cam (nodes, CAPT), tab)."""
    rp = np.ascontiguousarray(row_ptr, np.int32); ci = np.ascontiguousarray(col_idx, np.int32); Nc = len(rp) - 1
    sizes = np.zeros(8, np.int32); p32 = lambda a: a.ctypes.data_as(c_i32_p)
    L = _lib.lib()
    tl = np.zeros(1, np.int32)
    assert L.ssfm_snode_plan_probe(dc, Nc, p32(rp), p32(ci), num_cus, p32(sizes), None, None, None, None, p32(tl)) == 0
    if not sizes[0]:
        return None
    hr = np.zeros(sizes[5], np.int32); sr = np.zeros(max(sizes[6], 1), np.int32); nc = np.zeros(sizes[7], np.int32); tab = np.zeros(max(int(tl[0]), 1), np.int32); tl[0] = len(tab)
    assert L.ssfm_snode_plan_probe(dc, Nc, p32(rp), p32(ci), num_cus, p32(sizes), p32(hr), p32(sr), p32(nc), p32(tab), p32(tl)) == 0
    return dict(nhalf=int(sizes[1]), qtm=int(sizes[2]), S=int(sizes[3]), CAPT=int(sizes[4]), half_rec=hr.reshape(-1, 16), step_rec=sr[:sizes[6]].reshape(-1, 8),
                node_cam=nc.reshape(-1, int(sizes[4])), tab=tab)


RETRI_MODE_TRACE, RETRI_MODE_ENUMERATE = 0, 1


def retriangulate(ctx, prob, mode=None):
    """SfM::Retriangulate (reference src/sfm.cpp:156-192) on the GPU -> (points (Np,3), num_inliers (Np,)).
    mode: None = the library default (trace replay), RETRI_MODE_TRACE or RETRI_MODE_ENUMERATE (ssfm.h: ssfm_retriangulate_mode)."""
    b = _ProblemBuffers(prob)
    nin = np.zeros(len(b.pts), np.int32)
    if mode is None:
        _lib.check(_lib.lib().ssfm_retriangulate(ctx._p, C.byref(b.c), nin.ctypes.data_as(c_i32_p)), ctx._p)
    else:
        _lib.check(_lib.lib().ssfm_retriangulate_mode(ctx._p, C.byref(b.c), int(mode), nin.ctypes.data_as(c_i32_p), None, None), ctx._p)
    return b.pts, nin


def retriangulate_ex(ctx, prob):
    """The same with its trace -> (points, num_inliers, iterations (Np,), local-optimisation runs (Np,), inlier flags (M,) per observation)."""
    b = _ProblemBuffers(prob)
    nin = np.zeros(len(b.pts), np.int32); st = np.zeros(2 * max(len(b.pts), 1), np.uint32); fl = np.zeros(max(len(b.oc), 1), np.uint8)
    _lib.check(_lib.lib().ssfm_retriangulate_ex(ctx._p, C.byref(b.c), nin.ctypes.data_as(c_i32_p), st.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                fl.ctypes.data_as(_lib.c_u8_p)), ctx._p)
    st = st[:2 * len(b.pts)].reshape(-1, 2)
    return b.pts, nin, st[:, 0].copy(), st[:, 1].copy(), fl[:len(b.oc)].astype(bool)


def tri_probe(ctx, prob, what, task_pt, lists, X_in=None):
    """TriangulationEstimator's pieces on the device (ssfm.h: ssfm_tri_probe) -> (tasks, 4)"""
    b = _ProblemBuffers(prob)
    task_pt = np.ascontiguousarray(task_pt, np.int32); T = len(task_pt)
    ptr = np.zeros(T + 1, np.int32)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    flat = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32).reshape(-1) for l in lists]) if ptr[-1] else np.zeros(1, np.int32), np.int32)
    X = np.ascontiguousarray(X_in if X_in is not None else np.zeros((T, 3)), np.float64).reshape(-1)
    out = np.zeros(4 * T)
    _lib.check(_lib.lib().ssfm_tri_probe(ctx._p, C.byref(b.c), what, T, task_pt.ctypes.data_as(c_i32_p), ptr.ctypes.data_as(c_i32_p), flat.ctypes.data_as(c_i32_p),
                                         X.ctypes.data_as(c_double_p), out.ctypes.data_as(c_double_p)), ctx._p)
    return out.reshape(T, 4)


class BundleAdjuster:
    def __init__(self, ctx, prob, options=None, **kw):
        self.ctx = ctx
        self.buf = _ProblemBuffers(prob)
        self.opt = options or default_options(**kw)
        self._h = C.c_void_p()
        rc = _lib.lib().ssfm_ba_create(ctx._p, C.byref(self.buf.c), C.byref(self.opt), C.byref(self._h))
        if rc != 0:                    # the library has already torn the half-built handle down (*out is null on failure)
            self._h = C.c_void_p()
            _lib.check(rc, ctx._p)
        import weakref
        ctx._adjusters = [r for r in ctx._adjusters if r() is not None]      # drop the references of adjusters that are gone (the list only ever grew)
        ctx._adjusters.append(weakref.ref(self))

    def reset(self):
        _lib.check(_lib.lib().ssfm_ba_reset(self._h), self.ctx._p)

    def run(self):
        s = BASummaryC()
        _lib.check(_lib.lib().ssfm_ba_run(self._h, C.byref(s)), self.ctx._p)
        return s.as_dict()

    def download(self):
        _lib.check(_lib.lib().ssfm_ba_download(self._h, C.byref(self.buf.c)), self.ctx._p)
        return self.buf.cams, self.buf.pts, float(self.buf.focal[0])

    def evaluate(self):
        M = len(self.buf.oc)
        cost = C.c_double(0); res = np.zeros((M, 2)); jac = np.zeros((M, 2, 10))
        _lib.check(_lib.lib().ssfm_ba_evaluate(self._h, C.byref(cost), res.ctypes.data_as(c_double_p), jac.ctypes.data_as(c_double_p)), self.ctx._p)
        return cost.value, res, jac

    def set_profiling(self, on=True):
        _lib.check(_lib.lib().ssfm_ba_set_profiling(self._h, 1 if on else 0), self.ctx._p)

    def kernel_times(self):
        n = 32
        names = ((C.c_char * 32) * n)(); launches = (C.c_int64 * n)(); ms = (C.c_double * n)()
        k = _lib.lib().ssfm_ba_kernel_times(self._h, n, C.cast(names, C.c_void_p), C.cast(launches, c_i64_p), C.cast(ms, c_double_p))
        return {names[i].value.decode(): {"launches": int(launches[i]), "total_ms": float(ms[i])} for i in range(k)}

    def close(self):
        if self._h:
            _lib.lib().ssfm_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
