"""Registration front-ends on the HIP solver.

FrustumRegistration-compatible call            evaluation/frustum_reg/src/registration.cpp:190-206
restart driver (solve_P_random_perturb)        evaluation/registration_lsq.py:142-186
get_initial_guess                              evaluation/registration_lsq.py:196-220
device-resident pipeline (labels -> pose)      replaces the .npy hand-off + process fan-out of
                                               evaluation/visualize_and_save_data.py:174-186 -> registration_lsq.py:284-343

The reference draws the restart list from unseeded ``random`` inside the loop
(registration_lsq.py:163-164); here the list is an explicit input (or drawn from a seeded numpy
Generator) so that two implementations can be compared on identical hypotheses.
"""
import math

import numpy as np
import torch

from . import ops

__version__ = "deepi2p_amd-1"


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("deepi2p_amd.registration needs a HIP device (there is no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def solvePGivenK_batched(points, labels, K, init_y_angles, init_Ts, H, W, t_xyz_lower_bound, t_xyz_upper_bound,
                         max_iter, is_2d, return_all=False):
    """R hypotheses of ONE frame in one launch.  numpy in / numpy out.
    -> (P[R,4,4], cost[R], best) ; with return_all also (params[R,np], iters[R])."""
    dev = _dev()
    pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float64), device=dev).unsqueeze(0)
    lab = torch.as_tensor(np.ascontiguousarray(labels).astype(np.int32), device=dev).unsqueeze(0)
    Kt = torch.as_tensor(np.ascontiguousarray(K, dtype=np.float64), device=dev).reshape(1, 3, 3)
    ys = torch.as_tensor(np.ascontiguousarray(init_y_angles, dtype=np.float64), device=dev).reshape(1, -1)
    R = ys.shape[1]
    Ts = torch.as_tensor(np.ascontiguousarray(init_Ts, dtype=np.float64), device=dev).reshape(1, R, 3)
    params, cost, iters = ops.solve_batched(pts, lab, Kt, ys, Ts, H, W, t_xyz_lower_bound, t_xyz_upper_bound,
                                            max_iter, is_2d)
    # every hypothesis' P: treat the R hypotheses as R "frames" of one candidate each
    npar = params.shape[2]
    _, P, _ = ops.select_best(params.view(R, 1, npar), cost.view(R, 1), is_2d)
    best, _, _ = ops.select_best(params, cost, is_2d)
    out = (P.cpu().numpy(), cost.view(-1).cpu().numpy(), int(best.item()))
    if return_all:
        out = out + (params.view(R, npar).cpu().numpy(), iters.view(-1).cpu().numpy())
    return out


def solvePGivenK(points, labels, K, init_y_angle, init_T, H, W, t_xyz_lower_bound, t_xyz_upper_bound, max_iter,
                 is_debug, is_2d):
    """Drop-in for FrustumRegistration.solvePGivenK: -> (P 4x4, final_cost, residuals[3*N_in+N_out])."""
    dev = _dev()
    pts = torch.as_tensor(np.ascontiguousarray(points, dtype=np.float64), device=dev).unsqueeze(0)
    lab = torch.as_tensor(np.ascontiguousarray(labels).astype(np.int32), device=dev).unsqueeze(0)
    Kt = torch.as_tensor(np.ascontiguousarray(K, dtype=np.float64), device=dev).reshape(1, 3, 3)
    ys = torch.tensor([[float(init_y_angle)]], dtype=torch.float64, device=dev)
    Ts = torch.as_tensor(np.ascontiguousarray(init_T, dtype=np.float64), device=dev).reshape(1, 1, 3)
    params, cost, iters = ops.solve_batched(pts, lab, Kt, ys, Ts, H, W, t_xyz_lower_bound, t_xyz_upper_bound,
                                            max_iter, is_2d)
    _, P, _ = ops.select_best(params, cost, is_2d)
    res, counts, fcost = ops.solver_residuals(pts, lab, Kt, params.view(1, -1), H, W, is_2d)
    n = int(counts.item())
    if is_debug:
        print("deepi2p_amd solvePGivenK: iterations=%d final_cost=%.6g" % (int(iters.item()), float(fcost.item())))
    return P[0].cpu().numpy(), float(fcost.item()), res[0, :n].cpu().numpy()


def get_P_diff(P_pred_np, P_gt_np):
    """evaluation/registration_lsq.py:87-95: RTE = |t| of P_pred^-1 P_gt, RRE = sum |euler_xzy| in degrees."""
    from scipy.spatial.transform import Rotation
    P_diff = np.dot(np.linalg.inv(P_pred_np), P_gt_np)
    t_diff = np.linalg.norm(P_diff[0:3, 3])
    angles = Rotation.from_matrix(P_diff[0:3, 0:3]).as_euler("xzy", degrees=True)
    return t_diff, float(np.sum(np.abs(angles)))


def wrap_in_pi(x):
    x = math.fmod(x + math.pi, math.pi * 2)
    if x < 0:
        x += math.pi * 2
    return x - math.pi


def get_initial_guess(pc_np, coarse_predictions_np):
    """-> (P_init, init_y_angle, pc_front, labels_front); the reduction runs on the device."""
    dev = _dev()
    pts = torch.as_tensor(np.ascontiguousarray(pc_np, dtype=np.float64), device=dev).unsqueeze(0)
    lab = torch.as_tensor(np.ascontiguousarray(coarse_predictions_np).astype(np.int32), device=dev).unsqueeze(0)
    yaw0, lab_out, has = ops.initial_guess(pts, lab)
    y = float(yaw0.item())
    front = (lab_out[0] >= 0).cpu().numpy()
    c, s = math.cos(y), math.sin(y)
    P_init = np.identity(4)
    P_init[0:3, 0:3] = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    return P_init, y, pc_np[:, front], coarse_predictions_np[front]


def draw_restarts(rng, R, ry_sigma, init_t_amplitude, F=None):
    """Seeded stand-in for registration_lsq.py:163-164: (ry_noise, t_init) ; yaw0 is added on device."""
    shape = (R,) if F is None else (F, R)
    ry = rng.normal(0.0, ry_sigma, size=shape)
    Ts = np.zeros(shape + (3,))
    Ts[..., 2] = rng.uniform(-init_t_amplitude, init_t_amplitude, size=shape)
    return ry, Ts


def restart_list(init_y_angle, ry_sigma, init_t_amplitude, iteration_num, thread_num=None, rng=None):
    """The restart list of the reference's driver (registration_lsq.py:147-164), as arrays: ry = init_y_angle + gauss(0, ry_sigma),
    t = (0, 0, uniform(-amp, amp)), drawn in that order restart by restart.
    * ``thread_num`` given: the reference's wave arithmetic decides how many restarts there are -- ceil(n / threads) waves, the last one
      with n - threads * floor(n / threads) processes (:150-156), i.e. ``n % threads == 0`` silently DROPS the last wave (64 restarts on
      8 threads run 56); ``None`` runs all ``iteration_num``.
    * ``rng``: an object with ``gauss`` / ``uniform`` (Python's ``random`` module or a ``random.Random``) is consumed exactly as the
      reference consumes ``random`` -- the list is then the reference's, draw for draw (tests/golden/lsq_restart_golden.npz); a numpy
      Generator (or None) draws normal / uniform vectors instead."""
    n = int(iteration_num)
    if thread_num is not None:
        batch_num = math.ceil(n / thread_num)
        last = n - thread_num * math.floor(n / thread_num)
        n = max(batch_num - 1, 0) * thread_num + (last if batch_num > 0 else 0)
    if rng is not None and hasattr(rng, "gauss"):
        ys, Ts = np.zeros(n), np.zeros((n, 3))
        for i in range(n):
            ys[i] = init_y_angle + rng.gauss(0, ry_sigma)
            Ts[i, 2] = rng.uniform(-init_t_amplitude, init_t_amplitude)
        return ys, Ts
    rng = rng if rng is not None else np.random.default_rng()
    noise, Ts = draw_restarts(rng, n, ry_sigma, init_t_amplitude)
    return init_y_angle + noise, Ts


def solve_P_random_perturb(pc_np, coarse_predictions_np, K_np, H, W, init_t_amplitude, init_y_angle, ry_sigma,
                           t_lowerbound, t_upperbound, iteration_num, is_2d, thread_num=None, rng=None,
                           restarts=None, max_iter=500):
    """Reference signature (registration_lsq.py:142-145).  All restarts run in ONE launch; ``thread_num`` only enters the number of
    restarts (see restart_list: the reference's wave arithmetic, including the dropped last wave).  max_iter = 500 is what the reference
    hard-codes (:176).  -> (P, cost, residuals) of the minimum-cost restart, the FIRST one on ties (the reference keeps a result only
    when it is strictly smaller, :137); (None, 1e20, None) when no restart runs, like the reference's untouched dictionary."""
    if restarts is None:
        ys, Ts = restart_list(init_y_angle, ry_sigma, init_t_amplitude, iteration_num, thread_num, rng)
    else:
        ys, Ts = restarts
    if len(ys) == 0:
        return None, 1e20, None
    P, cost, best, params, _ = solvePGivenK_batched(pc_np, coarse_predictions_np, K_np, ys, Ts, H, W, t_lowerbound,
                                                    t_upperbound, max_iter, is_2d, return_all=True)
    dev = _dev()
    pts = torch.as_tensor(np.ascontiguousarray(pc_np, dtype=np.float64), device=dev).unsqueeze(0)
    lab = torch.as_tensor(np.ascontiguousarray(coarse_predictions_np).astype(np.int32), device=dev).unsqueeze(0)
    Kt = torch.as_tensor(np.ascontiguousarray(K_np, dtype=np.float64), device=dev).reshape(1, 3, 3)
    res, counts, _ = ops.solver_residuals(pts, lab, Kt, torch.as_tensor(params[best:best + 1], device=dev), H, W, is_2d)
    return P[best], float(cost[best]), res[0, :int(counts.item())].cpu().numpy()


class RegistrationPipeline:
    """Device-resident labels -> pose for a batch of frames: initial guess, front filter, R restarts per
    frame, argmin.  Nothing leaves HBM between the classifier and the pose."""

    def __init__(self, H, W, R=60, ry_sigma=10 * math.pi / 180, init_t_amplitude=10.0,
                 t_lowerbound=(-5, -0.1, -10), t_upperbound=(5, 0.1, 10), max_iter=500, is_2d=True, seed=0):
        self.H, self.W, self.R = H, W, R
        self.ry_sigma, self.amp = ry_sigma, init_t_amplitude
        self.lb, self.ub = list(t_lowerbound), list(t_upperbound)
        self.max_iter, self.is_2d = max_iter, is_2d
        self.rng = np.random.default_rng(seed)

    def draw(self, F, device):
        noise, Ts = draw_restarts(self.rng, self.R, self.ry_sigma, self.amp, F=F)
        return (torch.as_tensor(noise, dtype=torch.float64, device=device),
                torch.as_tensor(Ts, dtype=torch.float64, device=device))

    def draw_on_device(self, F, device, seed):
        """The restart list drawn on the device (Philox counter-based: a function of (seed, frame, restart) only)."""
        noise = torch.empty((F, self.R), dtype=torch.float64, device=device)
        Ts = torch.empty((F, self.R, 3), dtype=torch.float64, device=device)
        ops.call("di2p_draw_restarts", int(seed), F, self.R, float(self.ry_sigma), float(self.amp), ops.ptr(noise), ops.ptr(Ts),
                 ops.stream())
        return noise, Ts

    def __call__(self, pc_f32, labels_i32, K_f64, restarts):
        """pc f32[F,3,N], labels i32[F,N], K f64[F,3,3], restarts = (ry_noise f64[F,R], t_init f64[F,R,3])
        -> dict(P f64[F,4,4], cost f64[F], best i32[F], yaw0, costs f64[F,R], iters i32[F,R])."""
        F, _, N = pc_f32.shape
        pts64 = torch.empty((F, 3, N), dtype=torch.float64, device=pc_f32.device)
        ops.call("di2p_f32_to_f64", ops.ptr(pc_f32), ops.ptr(pts64), F * 3 * N, ops.stream())
        yaw0, lab_front, has_inside = ops.initial_guess(pts64, labels_i32)
        noise, Ts = restarts
        sweeps = torch.empty(noise.shape, dtype=torch.int32, device=pc_f32.device)
        params, cost, iters = ops.solve_batched(pc_f32, lab_front, K_f64, noise, Ts, self.H, self.W, self.lb, self.ub,
                                                self.max_iter, self.is_2d, yaw0=yaw0, sweeps=sweeps)
        best, P, bc = ops.select_best(params, cost, self.is_2d, has_inside=has_inside)
        return dict(P=P, cost=bc, best=best, yaw0=yaw0, costs=cost, iters=iters, sweeps=sweeps, params=params, labels_front=lab_front)
