"""Shared by the k-ary factor tests: the exact posterior of a synth.kary_model by a dense solve of the joint Gaussian."""
import numpy as np


def dense_posterior(model):
    """(ids, mean, variance) of every non-observed variable: J = sum of prior precisions + sum_f c_f c_f' / q_f, observed variables
    conditioned on their data."""
    meta = model.meta
    used = [int(v) for v in meta["used"]]
    obs = {int(v): float(y) for v, y in zip(model.data_var, model.data_y)}
    free = [v for v in used if v not in obs]
    pos = {v: i for i, v in enumerate(free)}
    n = len(free)
    J, h = np.zeros((n, n)), np.zeros(n)
    for v, m, s in zip(model.prior_var, model.prior_mean, model.prior_variance):
        J[pos[int(v)], pos[int(v)]] += 1.0 / s
        h[pos[int(v)]] += m / s
    cv, cf, ca = (meta["all_coef_var"], meta["all_coef_fac"], meta["all_coef"]) if "all_coef" in meta else (meta["coef_var"], meta["coef_fac"], meta["coef"])
    coef = {(int(v), int(f)): float(a) for v, f, a in zip(cv, cf, ca)}
    for fi, fid in enumerate(meta["kary_ids"]):
        vs = meta["fac_vars"][fi]
        c = {v: (1.0 if v == int(meta["out_var"][fi]) else -coef[(v, int(fid))]) for v in vs}
        b, q = float(meta["b"][fi]), float(meta["q"][fi])
        rhs = b - sum(c[v] * obs[v] for v in vs if v in obs)          # sum over free c_v x_v = rhs + eps
        fv = [v for v in vs if v not in obs]
        for a in fv:
            h[pos[a]] += c[a] * rhs / q
            for bb in fv:
                J[pos[a], pos[bb]] += c[a] * c[bb] / q
    S = np.linalg.inv(J)
    return np.array(free, dtype=np.int64), S @ h, np.diag(S).copy()
