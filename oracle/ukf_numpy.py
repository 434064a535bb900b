"""NumPy restatement of the filterpy pieces the reference env calls.

TEST INFRASTRUCTURE ONLY (see oracle/ssa_oracle.c header): imported by tests/
and by tests/golden/gen_golden.py, never by the product package.

filterpy is a third-party dependency of the reference (requirements.txt:14,
unpinned; reference era => filterpy 1.4.5) that is absent from
/root/reference and from this image, so its arithmetic is restated here from
the published algorithm (Van der Merwe scaled sigma points + Julier/Uhlmann
unscented transform) with filterpy's conventions, driven by *callbacks* exactly
like filterpy is -- the golden generator plugs in the reference's own
fx / hx / mean_z / residual_z / msqrt callables.  Parity of this restatement is
"unpinned" beyond the reference's tests.py Test 6/7 thresholds.

Reference call sites (file:line under /root/reference):
  envs/ssa_tasker_simple_2.py:110      Q_discrete_white_noise
  envs/ssa_tasker_simple_2.py:211-218  UKF / MerweScaledSigmaPoints construction
  envs/ssa_tasker_simple_2.py:275      UKF.predict()
  envs/ssa_tasker_simple_2.py:301-304  UKF.update(z, **hx_kwargs); .y .S .sigmas_h
"""
import numpy as np


def Q_discrete_white_noise(dim, dt=1., var=1., block_size=1, order_by_dim=True):
    """filterpy.common.Q_discrete_white_noise for dim == 2:
    q = [[dt^4/4, dt^3/2], [dt^3/2, dt^2]]; order_by_dim=False -> kron(q, I)."""
    if dim != 2:
        raise ValueError("only dim == 2 is used by the reference")
    Q = np.array([[.25 * dt ** 4, .5 * dt ** 3],
                  [.5 * dt ** 3, dt ** 2]])
    if order_by_dim:
        out = np.zeros((dim * block_size, dim * block_size))
        for b in range(block_size):
            out[b * dim:(b + 1) * dim, b * dim:(b + 1) * dim] = Q
        return out * var
    N = dim * block_size
    D = np.zeros((N, N))
    for i, x in enumerate(Q.ravel()):
        f = np.eye(block_size) * x
        ix, iy = (i // dim) * block_size, (i % dim) * block_size
        D[ix:ix + block_size, iy:iy + block_size] = f
    return D * var


class MerweScaledSigmaPoints:
    """SURVEY 8a U1."""

    def __init__(self, n, alpha, beta, kappa, sqrt_method=None, subtract=None):
        self.n = n
        self.alpha = alpha
        self.beta = beta
        self.kappa = kappa
        self.sqrt = sqrt_method
        self.subtract = np.subtract if subtract is None else subtract
        self._compute_weights()

    def num_sigmas(self):
        return 2 * self.n + 1

    def sigma_points(self, x, P):
        n = self.n
        x = np.asarray(x, dtype=float)
        P = np.atleast_2d(P)
        lambda_ = self.alpha ** 2 * (n + self.kappa) - n
        U = self.sqrt((lambda_ + n) * P)
        sigmas = np.zeros((2 * n + 1, n))
        sigmas[0] = x
        for k in range(n):
            sigmas[k + 1] = self.subtract(x, -U[k])
            sigmas[n + k + 1] = self.subtract(x, U[k])
        return sigmas

    def _compute_weights(self):
        n = self.n
        lambda_ = self.alpha ** 2 * (n + self.kappa) - n
        c = .5 / (n + lambda_)
        self.Wc = np.full(2 * n + 1, c)
        self.Wm = np.full(2 * n + 1, c)
        self.Wc[0] = lambda_ / (n + lambda_) + (1 - self.alpha ** 2 + self.beta)
        self.Wm[0] = lambda_ / (n + lambda_)
        self.scale = lambda_ + n


def unscented_transform(sigmas, Wm, Wc, noise_cov=None, mean_fn=None, residual_fn=None):
    kmax, n = sigmas.shape
    if mean_fn is None:
        x = np.dot(Wm, sigmas)
    else:
        x = mean_fn(sigmas, Wm)
    if residual_fn is np.subtract or residual_fn is None:
        y = sigmas - x[np.newaxis, :]
        P = np.dot(y.T, np.dot(np.diag(Wc), y))
    else:
        P = np.zeros((n, n))
        for k in range(kmax):
            y = residual_fn(sigmas[k], x)
            P += Wc[k] * np.outer(y, y)
    if noise_cov is not None:
        P += noise_cov
    return x, P


class UnscentedKalmanFilter:
    """SURVEY 8a U3 / U5.  `resample_after_predict` selects between the two
    published variants of predict(): False keeps the propagated sigma points
    for update(), True redraws them from the prior."""

    def __init__(self, dim_x, dim_z, dt, hx, fx, points, sqrt_fn=None, x_mean_fn=None,
                 z_mean_fn=None, residual_x=None, residual_z=None, resample_after_predict=False):
        self.x = np.zeros(dim_x)
        self.P = np.eye(dim_x)
        self.Q = np.eye(dim_x)
        self.R = np.eye(dim_z)
        self._dim_x, self._dim_z = dim_x, dim_z
        self.points_fn = points
        self._dt = dt
        self.hx, self.fx = hx, fx
        self.x_mean, self.z_mean = x_mean_fn, z_mean_fn
        self.Wm, self.Wc = points.Wm, points.Wc
        self.residual_x = np.subtract if residual_x is None else residual_x
        self.residual_z = np.subtract if residual_z is None else residual_z
        self.sigmas_f = np.zeros((points.num_sigmas(), dim_x))
        self.sigmas_h = np.zeros((points.num_sigmas(), dim_z))
        self.resample_after_predict = resample_after_predict
        self.inv = np.linalg.inv
        self.y = np.zeros(dim_z)
        self.S = np.zeros((dim_z, dim_z))

    def predict(self, dt=None, **fx_args):
        if dt is None:
            dt = self._dt
        sigmas = self.points_fn.sigma_points(self.x, self.P)
        for i, s in enumerate(sigmas):
            self.sigmas_f[i] = self.fx(s, dt, **fx_args)
        self.x, self.P = unscented_transform(self.sigmas_f, self.Wm, self.Wc, self.Q,
                                             self.x_mean, self.residual_x)
        if self.resample_after_predict:
            self.sigmas_f = self.points_fn.sigma_points(self.x, self.P)
        self.x_prior = np.copy(self.x)
        self.P_prior = np.copy(self.P)

    def update(self, z, R=None, **hx_args):
        if R is None:
            R = self.R
        elif np.isscalar(R):
            R = np.eye(self._dim_z) * R
        sigmas_h = []
        for s in self.sigmas_f:
            sigmas_h.append(self.hx(s, **hx_args))
        self.sigmas_h = np.atleast_2d(sigmas_h)
        zp, self.S = unscented_transform(self.sigmas_h, self.Wm, self.Wc, R, self.z_mean, self.residual_z)
        self.SI = self.inv(self.S)
        Pxz = np.zeros((self.sigmas_f.shape[1], self.sigmas_h.shape[1]))
        for i in range(self.sigmas_f.shape[0]):
            dx = self.residual_x(self.sigmas_f[i], self.x)
            dz = self.residual_z(self.sigmas_h[i], zp)
            Pxz += self.Wc[i] * np.outer(dx, dz)
        self.K = np.dot(Pxz, self.SI)
        self.y = self.residual_z(z, zp)
        self.x = self.x + np.dot(self.K, self.y)
        self.P = self.P - np.dot(self.K, np.dot(self.S, self.K.T))
        self.x_post = self.x.copy()
        self.P_post = self.P.copy()
