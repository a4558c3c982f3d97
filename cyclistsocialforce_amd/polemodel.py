"""The Balancing Rider control-behaviour model as far as the stepping engine needs it: the desired closed-loop poles of the
bicycle-rider system as a function of speed (parameters.py:1352-1411; controlbehavior.py: PoleModel.get_component_mean_function,
:1583-1650).

The reference keeps a "pole model" per rider-bicycle model: a Gaussian mixture over (speed, pole features) fitted behind a
preprocessing pipeline (log-shift of the real parts, Yeo-Johnson power transform, standard scaling), stored as YAML
(data/balancingriderparams/*.yaml).  With `stochastic_control_behavior=False` (the default) a rider's poles are the MEAN of
one mixture component conditioned on the current speed, and the reference does not use that mean directly: it evaluates it at
250 speeds in [1.5, 5.5] m/s, fits a straight line per pole feature (sklearn LinearRegression) and predicts from the line -
at any speed.  `component_mean_functions` restates exactly that chain and returns the lines, [components, 5 features,
(intercept, slope)]; the engine evaluates them per tick (include/csf.h: csf_params::br_pole_fun).

The lines of the two model files the reference ships for the Balancing Rider are tabulated below (MEAN_FUNCTIONS), computed by
this module from those files in the build container and pinned there against the reference's own PoleModel
(tests/golden/make_golden_balancingrider.py -> balancingrider.npz: polefun_BR0, polefun_BR1; tests/test_host_api.py).  Any
other model file: pass its path.
"""
import os

import numpy as np

FEATURES = ("v_mean", "p0_real", "p1_real", "p1_imag", "p2_real", "p2_imag")      # the feature set "ImRe5GivenV"

# component mean functions of the reference's model files: [component][feature p0_real, p1_real, p1_imag, p2_real, p2_imag] ->
# (intercept, slope); pole features = intercept + slope * v
MEAN_FUNCTIONS = {}      # (filled in below the functions that computed it)


def _yeo_johnson(x, lam):
    x = np.asarray(x, dtype=float)
    out = np.empty_like(x)
    pos = x >= 0
    if abs(lam) > 1e-12:
        out[pos] = ((x[pos] + 1.0) ** lam - 1.0) / lam
    else:
        out[pos] = np.log1p(x[pos])
    if abs(lam - 2.0) > 1e-12:
        out[~pos] = -(((-x[~pos] + 1.0) ** (2.0 - lam)) - 1.0) / (2.0 - lam)
    else:
        out[~pos] = -np.log1p(-x[~pos])
    return out


def _yeo_johnson_inverse(y, lam):
    y = np.asarray(y, dtype=float)
    out = np.empty_like(y)
    pos = y >= 0
    if abs(lam) > 1e-12:
        out[pos] = (y[pos] * lam + 1.0) ** (1.0 / lam) - 1.0
    else:
        out[pos] = np.expm1(y[pos])
    if abs(lam - 2.0) > 1e-12:
        out[~pos] = 1.0 - (-(2.0 - lam) * y[~pos] + 1.0) ** (1.0 / (2.0 - lam))
    else:
        out[~pos] = -np.expm1(-y[~pos])
    return out


def component_mean_functions(model, speeds=None):
    """model: the dictionary of a pole-model YAML file (or its path).  Returns [components, 5, 2]: (intercept, slope) of the
    line the reference fits to each component's conditional mean over `speeds` (default: its 250 speeds in [1.5, 5.5])."""
    if isinstance(model, (str, os.PathLike)):
        import yaml

        with open(model) as fh:
            model = yaml.safe_load(fh)
    if model["presets"]["feature_set"] != "ImRe5GivenV" or list(model["presets"]["features"]) != list(FEATURES):
        raise NotImplementedError("only the Balancing Rider feature set ImRe5GivenV (speed + one real pole + two complex pairs)")
    gm, pp = model["gmm_data"], model["preprocessing_pipeline"]
    mu = np.array(gm["means"], dtype=float)                            # [K, 6] in the transformed space
    cov = np.array(gm["covariances"], dtype=float)                     # [K, 6, 6]
    if gm["covariance_type"] != "full" or pp.get("power_transform") != "yeo-johnson" or not pp.get("normalize", False):
        raise NotImplementedError("pole models as the reference writes them: full covariances, Yeo-Johnson, standardised")
    lam = np.array(pp["power_transform_params"]["lambdas"], dtype=float)
    mean = np.array(pp["standard_scaler_params"]["mean"], dtype=float)
    scale = np.array(pp["standard_scaler_params"]["scale"], dtype=float)
    logf, a, sign = [], np.zeros(0), np.zeros(0)
    if pp.get("log_transform", False):
        lt = pp["log_transform_params"]
        logf = [int(i) for i in lt["log_transform_features"]]
        a = np.array(lt["a"], dtype=float).reshape(-1)
        sign = np.array(lt["sign"], dtype=float).reshape(-1)
    v = np.linspace(1.5, 5.5, 250) if speeds is None else np.asarray(speeds, dtype=float)
    # the speed through the pipeline (controlbehavior.py:1498-1501: the log-shift does not touch the speed column)
    vt = (_yeo_johnson(v, lam[0]) - mean[0]) / scale[0]
    out = np.zeros((mu.shape[0], 5, 2))
    for k in range(mu.shape[0]):
        # conditional mean of the component given the transformed speed (controlbehavior.py:485-515)
        cond = mu[k, 1:, None] + (cov[k, 1:, 0] / cov[k, 0, 0])[:, None] * (vt[None, :] - mu[k, 0])      # [5, n]
        # ... back through the pipeline: scaler, Yeo-Johnson, log-shift (controlbehavior.py:961-985)
        raw = np.empty_like(cond)
        for f in range(5):
            raw[f] = _yeo_johnson_inverse(cond[f] * scale[f + 1] + mean[f + 1], lam[f + 1])
        for j, f in enumerate(logf):
            raw[f - 1] = (np.exp(raw[f - 1]) + a[j]) * sign[j]
        # the straight line over speed (controlbehavior.py:1625-1631: LinearRegression().fit(speeds, means))
        A = np.c_[np.ones_like(v), v]
        for f in range(5):
            out[k, f] = np.linalg.lstsq(A, raw[f], rcond=None)[0]
    return out


class PoleSampler:
    """`PoleModel.sample_poles(n_samples=1, X_given=v)` of the reference (controlbehavior.py:1414-1469 -> :1337-1412 -> :536-570 ->
    :478-533 -> sklearn's GaussianMixture.sample): ONE draw of the five poles from the model's mixture conditioned on the speed.

    The chain, statement for statement: the speed through the preprocessing pipeline (Yeo-Johnson, scaler; the log-shift leaves
    the speed column alone); every component conditioned on it (mean, covariance, weight x its marginal density at the speed, the
    weights normalised); sklearn's sample(1): `multinomial(1, weights)` picks the component, `multivariate_normal(mean, cov, k)`
    draws from it - both on `rng`; back through the pipeline (scaler, Yeo-Johnson inverse, log-shift inverse); a draw that the
    inverse transform cannot represent is drawn again; (p0_real, p1_real, p1_imag, p2_real, p2_imag) -> [p0, p1, conj p1, p2, conj p2].

    rng: the reference's mixtures carry random_state=None, i.e. they draw from NumPy's GLOBAL generator; `rng=None` does the same
    (np.random.multinomial / multivariate_normal), so `np.random.seed(s)` in front of the same sequence of calls gives the
    reference's numbers (tests/test_host_api.py, against tests/golden/balancingrider_stochastic.npz).  Any object with those two
    methods (numpy.random.RandomState, numpy.random.Generator) may be passed for a stream of one's own.

    One deviation: a draw with a pole in the right half plane is drawn again here; the reference means to (ensure_stable) but its
    loop assigns a tuple into the pole array and raises (controlbehavior.py:1462-1463)."""

    def __init__(self, model):
        from . import polemodel_data

        if isinstance(model, (str, os.PathLike)) and os.path.basename(str(model)) in polemodel_data.MIXTURES and not os.path.exists(str(model)):
            model = polemodel_data.MIXTURES[os.path.basename(str(model))]
        elif isinstance(model, (str, os.PathLike)):
            import yaml

            with open(model) as fh:
                y = yaml.safe_load(fh)
            gm, pp = y["gmm_data"], y["preprocessing_pipeline"]
            lt = pp.get("log_transform_params", {}) if pp.get("log_transform") else {}
            model = {"features": list(y["presets"]["features"]), "means": gm["means"], "covariances": gm["covariances"], "weights": gm["weights"],
                     "lambdas": pp["power_transform_params"]["lambdas"], "scaler_mean": pp["standard_scaler_params"]["mean"],
                     "scaler_scale": pp["standard_scaler_params"]["scale"], "log_features": [int(i) for i in lt.get("log_transform_features", [])],
                     "log_a": np.array(lt.get("a", [])).reshape(-1).tolist(), "log_sign": np.array(lt.get("sign", [])).reshape(-1).tolist()}
        if list(model["features"]) != list(FEATURES):
            raise NotImplementedError("only the Balancing Rider feature set ImRe5GivenV (speed + one real pole + two complex pairs)")
        self.mu = np.array(model["means"], dtype=float)
        self.cov = np.array(model["covariances"], dtype=float)
        self.pi = np.array(model["weights"], dtype=float).ravel()
        self.lam = np.array(model["lambdas"], dtype=float)
        self.mean = np.array(model["scaler_mean"], dtype=float)
        self.scale = np.array(model["scaler_scale"], dtype=float)
        self.logf = [int(i) for i in model["log_features"]]
        self.log_a = np.array(model["log_a"], dtype=float)
        self.log_sign = np.array(model["log_sign"], dtype=float)

    def conditional(self, v):
        """(weights, means [K, 5], covariances [K, 5, 5]) of the mixture conditioned on the speed (controlbehavior.py:478-533)"""
        vt = (_yeo_johnson(np.array([float(v)]), self.lam[0])[0] - self.mean[0]) / self.scale[0]
        K = self.mu.shape[0]
        w, mc, cc = np.zeros(K), np.zeros((K, 5)), np.zeros((K, 5, 5))
        for k in range(K):
            var = self.cov[k, 0, 0]
            c = self.cov[k, 1:, 0]
            mc[k] = self.mu[k, 1:] + c / var * (vt - self.mu[k, 0])
            cc[k] = self.cov[k, 1:, 1:] - np.outer(c, c) / var
            w[k] = self.pi[k] * np.exp(-0.5 * (vt - self.mu[k, 0]) ** 2 / var) / np.sqrt(2 * np.pi * var)
        w = w / w.sum()
        if np.any(w == 0.0):                                           # (:525-528)
            w[w == 0.0] = np.finfo(float).eps * K
            w = w / w.sum()
        return w, mc, cc

    def _back(self, x):
        """a drawn feature vector back through the pipeline (controlbehavior.py:961-985)"""
        raw = np.empty(5)
        with np.errstate(all="ignore"):
            for f in range(5):
                raw[f] = _yeo_johnson_inverse(np.array([x[f] * self.scale[f + 1] + self.mean[f + 1]]), self.lam[f + 1])[0]
            for j, f in enumerate(self.logf):
                raw[f - 1] = (np.exp(raw[f - 1]) + self.log_a[j]) * self.log_sign[j]
        return raw

    def sample(self, v, rng=None):
        rng = np.random if rng is None else rng
        w, mc, cc = self.conditional(v)

        def draw():                                                    # sklearn.mixture.GaussianMixture.sample(1), covariance_type "full"
            counts = rng.multinomial(1, w)
            x = np.vstack([rng.multivariate_normal(mc[k], cc[k], int(counts[k])) for k in range(len(w))])
            return x[0]

        for _ in range(1000):
            raw = self._back(draw())
            for _ in range(101):                                       # (:1377-1394: outside the inverse transform's range: draw again)
                if np.all(np.isfinite(raw)):
                    break
                raw = self._back(draw())
            else:
                raise RuntimeError("Sampling error!")
            poles = [complex(raw[0]), complex(raw[1], raw[2]), complex(raw[1], -raw[2]), complex(raw[3], raw[4]), complex(raw[3], -raw[4])]
            if all(p.real <= 0 for p in poles):                        # ensure_stable (see the class docstring)
                return np.array(poles)
        raise TimeoutError("Couldn't find stable poles after 1000 draws!")


def poles_at(fun, v):
    """the five poles at speed v from one component's lines (parameters.py:1400-1409): p0 real, two conjugate pairs"""
    f = fun[:, 0] + fun[:, 1] * float(v)
    return [complex(f[0]), complex(f[1], f[2]), complex(f[1], -f[2]), complex(f[3], f[4]), complex(f[3], -f[4])]


MEAN_FUNCTIONS.update({
    "BR0_ImRe5GivenV_pole-model-params.yaml": np.array([
        [[7.477367764370247, -7.589580229524328], [-0.6066675056522426, -0.10886032204606376], [1.7881981548329742, 0.04106272039749896],
         [-1.3282378193409805, -0.026664495682316786], [5.327111219864691, 0.0891070929244512]],
    ]),
    "BR1_ImRe5GivenV_pole-model-params.yaml": np.array([
        [[-1.8535775147013693, -1.6314600197495854], [-0.17961038648194527, -0.15971107558265735], [1.173029309963536, 0.14249117463843872],
         [0.6329343335691049, -0.44818335654213626], [2.319802653488204, 0.9166444997604467]],
        [[-0.3523424742877923, -0.5883782873647062], [-0.10273325648260159, -0.2874543508333966], [1.738236648818368, 0.3111051504915781],
         [-0.4709777039447036, -0.7008198641061161], [7.811765217705169, 0.0608601658377002]],
    ]),
})
