"""Host-side mirror of the reference's solver classes over the C ABI.

``CFS_FANUC(obs, sys_info, ROBOT).optimizer()`` and ``PSGCFS_FANUC(obs, sys_info, ROBOT).optimizer()``
keep the reference's names, argument meaning and result fields (Lib/CFS_FANUC.m:40-79,
Lib/PSGCFS_FANUC.m:43-82, Lib/EVAL.m): ``u, x_, Ainq, binq, iter_O, total_iter, eval.cost_all,
eval.e_cost_all, eval.e_u_all, eval.cost_new``.  Every numerical step happens in libcfs_hip.so on
the GPU; this module only packs arguments.  ``CFSBatch`` is the batched form (the build's added
outer dimension: stochastic seeds, RRT node pairs, start/goal/obstacle variations).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib
from .robotproperty2 import to_c_robot

try:
    import torch
except Exception:  # pragma: no cover
    torch = None


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    if a is None:
        return None
    if torch is not None and isinstance(a, torch.Tensor):
        return C.c_void_p(a.data_ptr())
    return a.ctypes.data_as(C.c_void_p)


def obs_to_array(obs):
    """obs cell -> (nobs, 6) rows [obs{j}.l(:,1); obs{j}.l(:,2)]; mesh obstacles (obs{j}.mesh) get a row of zeros."""
    return _f64(np.stack([np.zeros(6) if "mesh" in o else
                          np.concatenate([np.asarray(o["l"], float)[:, 0], np.asarray(o["l"], float)[:, 1]]) for o in obs]))


def obs_meshes(obs):
    """The Mesh objects of an obs cell; mesh obstacles must come after the line obstacles (cfs_problem_set_meshes)."""
    flags = ["mesh" in o for o in obs]
    if any(flags) and flags != sorted(flags):
        raise ValueError("mesh obstacles must follow the line-segment obstacles in the obs cell")
    return [o["mesh"] for o in obs if "mesh" in o]


class CFSBatch:
    """A problem family (robot, horizon, cost matrix, limits, obstacle count and margins) on one GPU,
    solving batches of problems that differ in start/goal (x_init, xR1, ff, caug), obstacles and noise."""

    def __init__(self, sys_info, nobs, margin, mode="CFS", max_batch=1, device=None, check_dynamics=True, use_weights="auto"):
        """use_weights: True -> cfs_problem_create_from_weights(sys_info.weights) (the library assembles QQ, Qaug and alpha
        itself: neither crosses the boundary); False -> cfs_problem_create(sys_info.QQ, ...); "auto" -> the weights path, but
        only if the QQ the library assembles from sys_info.weights IS sys_info.QQ (to 1e-12 of its largest entry; checked
        with cfs_problem_family) and alpha agrees; a sys_info whose QQ / Baug were edited after build_sys_info is solved
        through the dense path with the dynamics check, exactly as given."""
        s = sys_info
        self.mode = mode
        self.H, self.nj = int(s.H), int(s.njoint)
        self.ns, self.nn, self.nx = 2 * self.nj, self.H * self.nj, self.H * 2 * self.nj
        self.nobs, self.K = int(nobs), int(s.MAX_O_ITER)
        self.max_batch = int(max_batch)
        self.robot = s.robot
        self.rows = self.nobs * self.H * (1 + 2 * self.nj)
        lib = _lib.lib()
        if device is not None:
            _lib.check(lib.cfs_set_device(int(device)))
        d = _lib.cfs_problem_desc()
        d.robot = to_c_robot(s.robot)
        d.mode = _lib.MODE[mode]
        d.H, d.njoint, d.nobs = self.H, self.nj, self.nobs
        wts = getattr(s, "weights", None)
        self.from_weights = bool(use_weights) and wts is not None if use_weights == "auto" else bool(use_weights)
        if use_weights == "auto" and self.from_weights and not self._double_integrator(s):
            self.from_weights = False                    # edited Aaug / Baug: the dense path validates them (CFS_ERR_DYNAMICS)
        if self.from_weights and wts is None:
            raise ValueError("use_weights=True needs sys_info.weights")
        keep = [None if self.from_weights else np.asfortranarray(s.QQ, dtype=np.float64), _f64(s.lim),
                _f64(np.asarray(margin, float).reshape(-1))]
        d.QQ, d.lim, d.margin = _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2])
        if keep[2].size != self.nobs:
            raise ValueError("margin must have one entry per obstacle")
        check_dynamics_asked = check_dynamics
        if self.from_weights:
            check_dynamics = False                       # Aaug / Baug are implied (and built) by the library
        if check_dynamics and getattr(s, "Aaug", None) is not None:
            keep.append(np.asfortranarray(s.Aaug, dtype=np.float64))
            d.Aaug = _ptr(keep[-1])
        if check_dynamics and getattr(s, "Baug", None) is not None:
            keep.append(np.asfortranarray(s.Baug, dtype=np.float64))
            d.Baug = _ptr(keep[-1])
        if mode == "CFS":
            keep.append(_f64(s.MAX_input))
            d.MAX_input = _ptr(keep[-1])
        d.epsilon_O, d.MAX_O_ITER, d.alpha = float(s.epsilon_O), self.K, float(getattr(s, "alpha", 0.0))
        d.max_batch = self.max_batch
        h = C.c_void_p()
        self._lib = lib
        if self.from_weights:
            w = _lib.cfs_cost_weights()
            keep += [np.asfortranarray(wts["Qp"], dtype=np.float64), np.asfortranarray(wts["Qv"], dtype=np.float64),
                     np.asfortranarray(wts["Rblk"], dtype=np.float64)]
            w.Qp, w.Qv, w.Rblk = _ptr(keep[-3]), _ptr(keep[-2]), _ptr(keep[-1])
            w.q_cross, w.w_stage, w.w_terminal, w.cR = float(wts["q_cross"]), float(wts["w_stage"]), float(wts["w_terminal"]), float(wts["cR"])
            _lib.check(lib.cfs_problem_create_from_weights(C.byref(d), C.byref(w), C.byref(h)))
            self._h = h
            if use_weights == "auto" and not self._weights_match(s):
                # the caller's matrices are not the ones its weights assemble: solve what was given, through the dense path
                self.close()
                CFSBatch.__init__(self, sys_info, nobs, margin, mode=mode, max_batch=max_batch, device=device,
                                  check_dynamics=check_dynamics_asked, use_weights=False)
                return
        else:
            _lib.check(lib.cfs_problem_create(C.byref(d), C.byref(h)))
            self._h = h

    def _weights_match(self, s):
        """does the QQ (and alpha) the library assembled from sys_info.weights equal what sys_info carries?"""
        QQ = getattr(s, "QQ", None)
        if QQ is None:
            return True
        QQ = np.asarray(QQ, float)
        got, alpha = self.family()
        if QQ.shape != got.shape or not np.abs(got - QQ).max() <= 1e-12 * np.abs(QQ).max():
            return False
        a = float(getattr(s, "alpha", 0.0))
        return not (self.mode == "PSGCFS" and a != 0.0 and not abs(alpha - a) <= 1e-9 * abs(a))

    def _double_integrator(self, s):
        """is sys_info.Baug (when present) the double integrator the structured products assume (robotproperty2.m:136-139)?"""
        Bm = getattr(s, "Baug", None)
        if Bm is None:
            return True
        dt, H, nj = float(s.robot.delta_t), self.H, self.nj
        Bm = np.asarray(Bm, float)
        if Bm.shape != (H * 2 * nj, H * nj):
            return False
        Bm = Bm.reshape(H, 2 * nj, H, nj)
        i, k = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
        want = np.zeros_like(Bm)
        for c in range(nj):
            want[:, c, :, c] = np.where(k <= i, ((i - k) + 0.5) * dt * dt, 0.0)
            want[:, nj + c, :, c] = np.where(k <= i, dt, 0.0)
        return bool(np.abs(Bm - want).max() <= 1e-12 * (1.0 + np.abs(want).max()))

    def set_launch_order(self, order="auto"):
        """Workgroup w of the next solves handles problem order[w] (cfs_set_launch_order): "auto" (default: most violated
        initial trajectories first), "identity", or a permutation of 0..B-1.  Results do not depend on it; a launch is as
        long as its longest problem plus that problem's wait for a compute unit."""
        if isinstance(order, str):
            _lib.check(self._lib.cfs_set_launch_order(self._h, None, {"auto": 0, "identity": -1}[order]))
            return
        o = np.ascontiguousarray(order, dtype=np.int32)
        _lib.check(self._lib.cfs_set_launch_order(self._h, _ptr(o), int(o.size)))

    def family(self):
        """(QQ, alpha) the handle was built with (cfs_problem_family)."""
        QQ = np.zeros((self.nn, self.nn), order="F")
        a = C.c_double(0.0)
        _lib.check(self._lib.cfs_problem_family(self._h, _ptr(QQ), C.byref(a)))
        return QQ, a.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.cfs_problem_destroy(self._h)
            self._h = None

    __del__ = close

    def set_meshes(self, meshes):
        """The last len(meshes) of the nobs obstacles are these mesh.Mesh objects from now on (cfs_problem_set_meshes)."""
        self._meshes = list(meshes)                      # keep them alive as long as the handle uses them
        arr = (C.c_void_p * max(len(self._meshes), 1))(*[m._h for m in self._meshes])
        _lib.check(self._lib.cfs_problem_set_meshes(self._h, len(self._meshes), arr))

    # ---- whole solve ------------------------------------------------------------------------------
    def solve(self, x_init, xR1, ff, caug, obs, noise=None):
        """Host arrays in, host arrays out (cfs_solve_batch)."""
        x_init, xR1, ff, caug, obs = _f64(x_init), _f64(xR1), _f64(ff), _f64(caug).reshape(-1), _f64(obs)
        B = x_init.shape[0]
        assert x_init.shape == (B, self.nx) and xR1.shape == (B, self.ns) and ff.shape == (B, self.nn)
        assert caug.shape == (B,) and obs.shape == (B, self.nobs, 6)
        i = _lib.cfs_batch_in()
        i.B = B
        i.x_init, i.xR1, i.ff, i.caug, i.obs = _ptr(x_init), _ptr(xR1), _ptr(ff), _ptr(caug), _ptr(obs)
        if noise is not None:
            noise = _f64(noise)
            assert noise.ndim == 3 and noise.shape[0] == B and noise.shape[2] == self.nn
            i.noise, i.noise_rows = _ptr(noise), noise.shape[1]
        r = SimpleNamespace(u=np.zeros((B, self.nn)), x_=np.zeros((B, self.nx)), cost_all=np.zeros((B, self.K)),
                            e_cost_all=np.zeros((B, self.K)), e_u_all=np.zeros((B, self.K)),
                            iter_O=np.zeros(B, np.int32), total_iter=np.zeros(B, np.int32), status=np.zeros(B, np.int32))
        o = _lib.cfs_batch_out()
        o.u, o.x_, o.cost_all, o.e_cost_all, o.e_u_all = _ptr(r.u), _ptr(r.x_), _ptr(r.cost_all), _ptr(r.e_cost_all), _ptr(r.e_u_all)
        o.iter_O, o.total_iter, o.status = _ptr(r.iter_O), _ptr(r.total_iter), _ptr(r.status)
        _lib.check(self._lib.cfs_solve_batch(self._h, C.byref(i), C.byref(o)))
        return r

    def alloc_outputs(self, B, device):
        """Device-resident output buffers (torch CUDA tensors) for solve_device."""
        z = lambda *shape, dt=torch.float64: torch.zeros(*shape, dtype=dt, device=device)  # noqa: E731
        return SimpleNamespace(u=z(B, self.nn), x_=z(B, self.nx), cost_all=z(B, self.K), e_cost_all=z(B, self.K),
                               e_u_all=z(B, self.K), iter_O=z(B, dt=torch.int32), total_iter=z(B, dt=torch.int32),
                               status=z(B, dt=torch.int32))

    def solve_device(self, x_init, xR1, ff, caug, obs, noise=None, out=None, stream=None):
        """torch CUDA tensors in/out; enqueues on `stream` (default: torch's current stream) and
        returns without synchronising (cfs_solve_batch_device)."""
        B = x_init.shape[0]
        for t in (x_init, xR1, ff, caug, obs) + ((noise,) if noise is not None else ()):
            assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
        if out is None:
            out = self.alloc_outputs(B, x_init.device)
        i = _lib.cfs_batch_in()
        i.B = B
        i.x_init, i.xR1, i.ff, i.caug, i.obs = _ptr(x_init), _ptr(xR1), _ptr(ff), _ptr(caug), _ptr(obs)
        if noise is not None:
            i.noise, i.noise_rows = _ptr(noise), noise.shape[1]
        o = _lib.cfs_batch_out()
        o.u, o.x_, o.cost_all, o.e_cost_all, o.e_u_all = _ptr(out.u), _ptr(out.x_), _ptr(out.cost_all), _ptr(out.e_cost_all), _ptr(out.e_u_all)
        o.iter_O, o.total_iter, o.status = _ptr(out.iter_O), _ptr(out.total_iter), _ptr(out.status)
        if stream is None:
            stream = torch.cuda.current_stream(x_init.device).cuda_stream
        _lib.check(self._lib.cfs_solve_batch_device(self._h, C.byref(i), C.byref(o), C.c_void_p(stream)))
        return out

    # ---- CHOMP (row f4) ------------------------------------------------------------------------------------
    def chomp(self, x_init, xR1, ff, caug, obs, u0, D, epsilon):
        """CHOMP_FANUC.optimizer() for B problems (cfs_chomp_batch); host arrays in and out."""
        x_init, xR1, ff, caug, obs, u0 = _f64(x_init), _f64(xR1), _f64(ff), _f64(caug).reshape(-1), _f64(obs), _f64(u0)
        D, epsilon = _f64(np.asarray(D, float).reshape(-1)), _f64(np.asarray(epsilon, float).reshape(-1))
        B = x_init.shape[0]
        assert x_init.shape == (B, self.nx) and u0.shape == (B, self.nn) and obs.shape == (B, self.nobs, 6)
        assert D.size == self.nobs and epsilon.size == self.nobs
        i = _lib.cfs_batch_in()
        i.B = B
        i.x_init, i.xR1, i.ff, i.caug, i.obs = _ptr(x_init), _ptr(xR1), _ptr(ff), _ptr(caug), _ptr(obs)
        r = SimpleNamespace(u=np.zeros((B, self.nn)), x_=np.zeros((B, self.nx)), cost_all=np.zeros((B, self.K)),
                            e_cost_all=np.zeros((B, self.K)), e_u_all=np.zeros((B, self.K)),
                            iter_O=np.zeros(B, np.int32), total_iter=np.zeros(B, np.int32), status=np.zeros(B, np.int32))
        o = _lib.cfs_batch_out()
        o.u, o.x_, o.cost_all, o.e_cost_all, o.e_u_all = _ptr(r.u), _ptr(r.x_), _ptr(r.cost_all), _ptr(r.e_cost_all), _ptr(r.e_u_all)
        o.iter_O, o.total_iter, o.status = _ptr(r.iter_O), _ptr(r.total_iter), _ptr(r.status)
        _lib.check(self._lib.cfs_chomp_batch(self._h, C.byref(i), _ptr(u0), _ptr(D), _ptr(epsilon), C.byref(o)))
        return r

    # ---- per-problem setup on the device (row f2) ------------------------------------------------------
    def set_state_cost(self, Qaug):
        """The drivers' state-cost matrix Qaug (main_FANUC.m:79-84), once per handle."""
        q = np.asfortranarray(Qaug, dtype=np.float64)
        assert q.shape == (self.nx, self.nx)
        _lib.check(self._lib.cfs_set_state_cost(self._h, _ptr(q)))

    def build_terms_device(self, x0, xg, stream=None):
        """(x_init, xR1, ff, caug) as CUDA tensors for B (start, goal) pairs given as CUDA tensors (B, njoint)."""
        B = x0.shape[0]
        assert x0.is_cuda and xg.is_cuda and x0.dtype == torch.float64 and x0.is_contiguous() and xg.is_contiguous()
        z = lambda *sh: torch.empty(*sh, dtype=torch.float64, device=x0.device)  # noqa: E731
        x_init, xR1, ff, caug = z(B, self.nx), z(B, self.ns), z(B, self.nn), z(B)
        if stream is None:
            stream = torch.cuda.current_stream(x0.device).cuda_stream
        _lib.check(self._lib.cfs_build_terms_device(self._h, B, _ptr(x0), _ptr(xg), _ptr(x_init), _ptr(xR1), _ptr(ff), _ptr(caug),
                                                    C.c_void_p(stream)))
        return x_init, xR1, ff, caug

    def build_terms_from_routes_device(self, routes, stream=None):
        """(x_init, xR1, ff, caug) for B RRT routes given as a CUDA tensor (B, nwp, njoint): cubic zero-velocity
        resampling to H+1 samples + cost terms, on the device (RRTstar_CFS.m:94-110, 159-163)."""
        B, nwp = routes.shape[0], routes.shape[1]
        assert routes.is_cuda and routes.dtype == torch.float64 and routes.is_contiguous() and routes.shape[2] == self.nj
        z = lambda *sh: torch.empty(*sh, dtype=torch.float64, device=routes.device)  # noqa: E731
        x_init, xR1, ff, caug = z(B, self.nx), z(B, self.ns), z(B, self.nn), z(B)
        if stream is None:
            stream = torch.cuda.current_stream(routes.device).cuda_stream
        _lib.check(self._lib.cfs_build_terms_from_routes_device(self._h, B, _ptr(routes), nwp, _ptr(x_init), _ptr(xR1), _ptr(ff),
                                                                _ptr(caug), C.c_void_p(stream)))
        return x_init, xR1, ff, caug

    def build_terms_from_ragged_routes_device(self, routes, nwp, stream=None):
        """The same for routes of different lengths as cfs_rrt_grow_device leaves them: routes (B, nwp_stride, njoint) CUDA
        float64, nwp (B,) CUDA int32 rows used per route (cfs_build_terms_from_ragged_routes_device)."""
        B, stride = routes.shape[0], routes.shape[1]
        assert routes.is_cuda and routes.dtype == torch.float64 and routes.is_contiguous() and routes.shape[2] == self.nj
        assert nwp.is_cuda and nwp.dtype == torch.int32 and nwp.is_contiguous() and nwp.shape == (B,)
        z = lambda *sh: torch.empty(*sh, dtype=torch.float64, device=routes.device)  # noqa: E731
        x_init, xR1, ff, caug = z(B, self.nx), z(B, self.ns), z(B, self.nn), z(B)
        if stream is None:
            stream = torch.cuda.current_stream(routes.device).cuda_stream
        _lib.check(self._lib.cfs_build_terms_from_ragged_routes_device(self._h, B, _ptr(routes), stride, _ptr(nwp), _ptr(x_init), _ptr(xR1),
                                                                       _ptr(ff), _ptr(caug), C.c_void_p(stream)))
        return x_init, xR1, ff, caug

    # ---- measurement ------------------------------------------------------------------------------
    def profile(self, on=True):
        _lib.check(self._lib.cfs_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        """(ms in the fused solve kernel, ms in the MFMA batched product, solves) since the last read;
        HIP events on the stream the kernels were launched on."""
        a, b, n = C.c_double(0), C.c_double(0), C.c_int(0)
        _lib.check(self._lib.cfs_profile_read(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    # ---- EVAL (Lib/EVAL.m:51-53, 75-78) ------------------------------------------------------------------
    def cost_b(self, ff, caug, want_u=False):
        """Cost_b = get_Cost_b() for B problems: cost of the unconstrained minimiser -H^{-1} ff (cfs_cost_b)."""
        ff, caug = _f64(np.atleast_2d(ff)), _f64(caug).reshape(-1)
        B = ff.shape[0]
        assert ff.shape == (B, self.nn) and caug.shape == (B,)
        cost = np.zeros(B)
        ub = np.zeros((B, self.nn)) if want_u else None
        _lib.check(self._lib.cfs_cost_b(self._h, B, _ptr(ff), _ptr(caug), _ptr(cost), _ptr(ub)))
        return (cost, ub) if want_u else cost

    def get_cost(self, u, ff, caug):
        """get_cost(u) = 0.5 u'QQ u + ff'u + caug for B given u (cfs_get_cost)."""
        u, ff, caug = _f64(np.atleast_2d(u)), _f64(np.atleast_2d(ff)), _f64(caug).reshape(-1)
        B = u.shape[0]
        assert u.shape == (B, self.nn) and ff.shape == (B, self.nn) and caug.shape == (B,)
        cost = np.zeros(B)
        _lib.check(self._lib.cfs_get_cost(self._h, B, _ptr(u), _ptr(ff), _ptr(caug), _ptr(cost)))
        return cost

    # ---- developer / test switches (cfs_debug_*, per handle) ---------------------------------------------
    def debug_options(self, warm_max=0, polish_tol=0.0, **flags):
        """cfs_debug_set_options: flags from _lib.DBG (gather_rollouts, no_refine, no_warm_start, no_certificate, no_prune,
        no_auto_order, tier_w1); no flags = the defaults."""
        mask = 0
        for k, v in flags.items():
            if v:
                mask |= _lib.DBG[k]
        _lib.check(self._lib.cfs_debug_set_options(self._h, mask, int(warm_max), float(polish_tol)))

    def stamps(self, B=None):
        """B given: enable the cycle stamps for the next solves of <= B problems (0: off); B None: read them, (n, 12) uint64."""
        if B is not None:
            _lib.check(self._lib.cfs_debug_stamps(self._h, int(B), None))
            self._stamps_B = int(B)
            return None
        out = np.zeros((self._stamps_B, 12), np.uint64)
        _lib.check(self._lib.cfs_debug_stamps(self._h, self._stamps_B, _ptr(out)))
        return out

    def trace(self, b=None, cap=0):
        """b given: trace the active-set steps of problem b (cap records; cap 0: off); b None: read, (n, 8) float64."""
        if b is not None:
            _lib.check(self._lib.cfs_debug_trace_begin(self._h, int(b), int(cap)))
            self._trace_cap = int(cap)
            return None
        buf = np.zeros((self._trace_cap + 1) * 8)
        _lib.check(self._lib.cfs_debug_trace_read(self._h, _ptr(buf)))
        return buf[8:8 + 8 * int(buf[0])].reshape(-1, 8)

    def log_u(self, on=True):
        """log u after every outer iteration of the next solves (either solver; cfs_debug_log_u)."""
        _lib.check(self._lib.cfs_debug_log_u(self._h, 1 if on else 0))

    def read_u_log(self, B):
        out = np.zeros((B, self.K, self.nn))
        _lib.check(self._lib.cfs_debug_read_u_log(self._h, int(B), _ptr(out)))
        return out

    # ---- pieces -----------------------------------------------------------------------------------
    def linearize(self, x_, obs):
        x_, obs = _f64(x_), _f64(obs)
        B = x_.shape[0]
        dist = np.zeros((B, self.nobs, self.H))
        lid = np.zeros((B, self.nobs, self.H), np.int32)
        grad = np.zeros((B, self.nobs, self.H, self.nj))
        _lib.check(self._lib.cfs_linearize(self._h, B, _ptr(x_), _ptr(obs), _ptr(dist), _ptr(lid), _ptr(grad)))
        return dist, lid, grad

    def get_con(self, x_, u, xR1, obs):
        x_, u, xR1, obs = _f64(x_), _f64(u), _f64(xR1), _f64(obs)
        B = x_.shape[0]
        A = np.zeros((B, self.nn, self.rows))  # per problem rows x nn column-major == (nn, rows) C-order
        b = np.zeros((B, self.rows))
        _lib.check(self._lib.cfs_get_con(self._h, B, _ptr(x_), _ptr(u), _ptr(xR1), _ptr(obs), _ptr(A), _ptr(b)))
        return A.transpose(0, 2, 1), b

    def qp(self, lin, u_lin, xR1, dist, grad, want_lambda=True):
        lin, u_lin, xR1, dist, grad = _f64(lin), _f64(u_lin), _f64(xR1), _f64(dist), _f64(grad)
        B = lin.shape[0]
        u = np.zeros((B, self.nn))
        lam = np.zeros((B, self.nobs * self.H + 4 * self.nn)) if want_lambda else None
        it, st = np.zeros(B, np.int32), np.zeros(B, np.int32)
        _lib.check(self._lib.cfs_qp(self._h, B, _ptr(lin), _ptr(u_lin), _ptr(xR1), _ptr(dist), _ptr(grad), _ptr(u),
                                    _ptr(lam), _ptr(it), _ptr(st)))
        return u, lam, it, st


def dist_arm(robot, theta, obs_l, want_pos=False):
    """[d, linkid] = dist_arm_*(theta, base, obs_l, robot) for N configurations x nobs obstacle axes
    (theta: (N, nj); obs_l: (nobs, 6))."""
    theta, obs_l = _f64(np.atleast_2d(theta)), _f64(np.atleast_2d(obs_l))
    N, nj = theta.shape
    nobs = obs_l.shape[0]
    d = np.zeros((N, nobs))
    lid = np.zeros((N, nobs), np.int32)
    pos = np.zeros((N, nj, 2, 3)) if want_pos else None
    rb = to_c_robot(robot)
    _lib.check(_lib.lib().cfs_dist_arm(C.byref(rb), nj, N, _ptr(theta), nobs, _ptr(obs_l), _ptr(d), _ptr(lid), _ptr(pos)))
    return (d, lid, pos) if want_pos else (d, lid)


class EVAL:
    """Lib/EVAL.m: ``eval = EVAL(sys_info)``, ``Cost_b = eval.get_Cost_b()`` (main_FANUC.m:131-132), ``get_cost(u)``,
    ``store_result(u)``, ``stop_outer(iter_O)`` with the reference's field names (:9-37).  The two cost functions run in
    libcfs_hip.so (cfs_cost_b, cfs_get_cost: QQ*u on the matrix cores); the stop test and the history appends are the
    reference's host-side bookkeeping.  Inside ``optimizer()`` all of this happens in the fused kernel; the methods are for
    callers that use EVAL on its own, as main_FANUC.m does for the baseline cost."""

    def __init__(self, sys_info, device=None):
        self.sys_info = sys_info
        self.epsilon_O, self.MAX_O_ITER = sys_info.epsilon_O, sys_info.MAX_O_ITER     # EVAL.m:43-44
        self.x_ = np.asarray(sys_info.x_, float).reshape(-1).copy()                   # :46
        self.x_old = np.ones_like(self.x_)                                            # :47
        self.u_old = None
        self.Cost_b = 0.0
        self.total_iter = 0
        self.cost_old, self.cost_new = 100000.0, 0.0                                  # :29-30
        self.cost_all, self.e_cost_all, self.e_u_all = np.zeros(0), np.zeros(0), np.zeros(0)
        self._device, self._batch = device, None

    def _family(self):
        if self._batch is None:      # the family handle (QQ and its inverse on the device); obstacles play no role in the costs
            self._batch = CFSBatch(self.sys_info, 1, [0.0], mode="CFS", max_batch=1, device=self._device)
        return self._batch

    def get_cost(self, u):
        """cost = 0.5*u'*Qaug*u + paug'*u + caug with the drivers' Qaug = QQ, paug = ff (EVAL.m:51-53, main_FANUC.m:110-112)."""
        s = self.sys_info
        return float(self._family().get_cost(np.asarray(u, float).reshape(1, -1), _f64(s.ff).reshape(1, -1), np.array([s.caug], float))[0])

    def get_Cost_b(self):
        """cost of the unconstrained QP's minimiser (EVAL.m:75-78)."""
        s = self.sys_info
        self.Cost_b = float(self._family().cost_b(_f64(s.ff).reshape(1, -1), np.array([s.caug], float))[0])
        return self.Cost_b

    def store_result(self, u):
        """EVAL.m:55-59."""
        u = np.asarray(u, float).reshape(-1)
        u_old = np.zeros_like(u) if self.u_old is None else np.asarray(self.u_old, float).reshape(-1)
        self.cost_all = np.append(self.cost_all, self.cost_new)
        self.e_cost_all = np.append(self.e_cost_all, abs(self.cost_old - self.cost_new))
        self.e_u_all = np.append(self.e_u_all, float(np.linalg.norm(u_old - u)))
        return self

    def stop_outer(self, iter_O):
        """EVAL.m:61-73: stop when ||x_ - x_old|| < epsilon_O or iter_O > MAX_O_ITER."""
        stop = False
        if float(np.linalg.norm(np.asarray(self.x_, float).reshape(-1) - np.asarray(self.x_old, float).reshape(-1))) < self.epsilon_O:
            print(f"Converged at step{iter_O}")
            stop = True
        if iter_O > self.MAX_O_ITER:
            print("MAX_ITER")
            stop = True
        return stop


class _SolverBase:
    MODE = "CFS"
    MARGIN_KEY = "epsilon"

    def __init__(self, obs, sys_info, ROBOT="M16iB", device=None):
        self.obs, self.sys_info, self.ROBOT = obs, sys_info, ROBOT
        if getattr(sys_info.robot, "name", ROBOT) != ROBOT:
            raise ValueError(f"sys_info.robot is {sys_info.robot.name!r} but ROBOT={ROBOT!r}")
        self.nn = sys_info.H * sys_info.nu
        self.x_ = np.asarray(sys_info.x_, float).reshape(-1).copy()
        self.u = np.zeros(self.nn)
        self.Ainq = self.binq = None
        self.eval = EVAL(sys_info)
        self.iter_O, self.total_iter, self.status = 1, 0, None
        self._batch = CFSBatch(sys_info, len(obs), [o[self.MARGIN_KEY] for o in obs], mode=self.MODE, max_batch=1,
                               device=device)
        meshes = obs_meshes(obs)
        if meshes:
            self._batch.set_meshes(meshes)

    def _args(self):
        s = self.sys_info
        xR1 = np.asarray(s.xR, float).reshape(s.nstate, -1)[:, 0]
        return xR1[None], _f64(s.ff).reshape(1, -1), np.array([s.caug], float), obs_to_array(self.obs)[None]

    def get_con(self):
        """self.Ainq / self.binq at the current (x_, u): dense, reference row order."""
        xR1, _, _, obs = self._args()
        A, b = self._batch.get_con(self.x_[None], self.u[None], xR1, obs)
        self.Ainq, self.binq = A[0], b[0]
        return self

    def optimizer(self, noise=None):
        xR1, ff, caug, obs = self._args()
        nz = None if noise is None else _f64(noise)[None]
        r = self._batch.solve(np.asarray(self.sys_info.x_, float).reshape(1, -1), xR1, ff, caug, obs, noise=nz)
        n = int(r.iter_O[0]) - 1
        self.u, self.x_ = r.u[0], r.x_[0]
        self.iter_O, self.total_iter, self.status = int(r.iter_O[0]), int(r.total_iter[0]), int(r.status[0])
        self.eval.cost_all, self.eval.e_cost_all, self.eval.e_u_all = r.cost_all[0, :n], r.e_cost_all[0, :n], r.e_u_all[0, :n]
        self.eval.cost_new = float(r.cost_all[0, n - 1]) if n > 0 else float(self.sys_info.caug)
        self.eval.x_ = self.x_
        if self.status == 0:
            print(f"Converged at step{self.iter_O}")  # EVAL.m:66
        elif self.status == 1:
            print("MAX_ITER")  # EVAL.m:70
        return self


class CFS_FANUC(_SolverBase):
    """Lib/CFS_FANUC.m -- margin obs{j}.epsilon, QP with QQ/ff and +-MAX_input bounds."""
    MODE, MARGIN_KEY = "CFS", "epsilon"


class PSGCFS_FANUC(_SolverBase):
    """Lib/PSGCFS_FANUC.m -- margin obs{j}.D, noisy gradient step + projection.  The normrnd draws
    (PSGCFS_FANUC.m:109) are passed in explicitly: optimizer(noise=(rows, nn) array of N(0, 0.1^2))."""
    MODE, MARGIN_KEY = "PSGCFS", "D"


class CHOMP_FANUC:
    """Lib/CHOMP_FANUC.m -- ``CHOMP_FANUC(obs_, sys_info, uref, ROBOT).optimizer()``.  ``obs_`` is the reference's cell:
    ``obs_[0] = dict(num_obs=n)`` followed by the n obstacles (``l``, ``D``, ``epsilon``) (M16iB/CHOMP.m:26-29)."""

    def __init__(self, obs, sys_info, uu, ROBOT="M16iB", device=None):
        self.obs, self.sys_info, self.ROBOT = obs, sys_info, ROBOT
        if getattr(sys_info.robot, "name", ROBOT) != ROBOT:
            raise ValueError(f"sys_info.robot is {sys_info.robot.name!r} but ROBOT={ROBOT!r}")
        n = int(obs[0]["num_obs"])
        self._obstacles = list(obs[1:1 + n])
        self.nn = sys_info.H * sys_info.nu
        self.x_ = np.asarray(sys_info.x_, float).reshape(-1).copy()
        self.u = np.asarray(uu, float).reshape(-1).copy()
        self.eval = EVAL(sys_info)
        self.iter_O, self.total_iter = 1, 0
        self._batch = CFSBatch(sys_info, n, [o["epsilon"] for o in self._obstacles], mode="CFS", max_batch=1, device=device)

    def optimizer(self):
        s = self.sys_info
        xR1 = np.asarray(s.xR, float).reshape(s.nstate, -1)[:, 0]
        r = self._batch.chomp(np.asarray(s.x_, float).reshape(1, -1), xR1[None], _f64(s.ff).reshape(1, -1), np.array([s.caug], float),
                              obs_to_array(self._obstacles)[None], self.u[None], [o["D"] for o in self._obstacles],
                              [o["epsilon"] for o in self._obstacles])
        n = int(r.iter_O[0]) - 1
        self.u, self.x_, self.iter_O = r.u[0], r.x_[0], int(r.iter_O[0])
        self.eval.cost_all, self.eval.e_cost_all, self.eval.e_u_all = r.cost_all[0, :n], r.e_cost_all[0, :n], r.e_u_all[0, :n]
        if n > 0:
            self.eval.cost_new = float(r.cost_all[0, n - 1])
        print("MAX_ITER")  # EVAL.m:70 (the loop never converges by distance: eval.x_ is not refreshed)
        return self
