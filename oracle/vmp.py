"""oracle/vmp.py — TEST INFRASTRUCTURE ONLY (never imported by the product).

Array restatement of what ONE `update_marginals!(engine, ids)` call computes on the reference's two variational
state-space test models (test/inference_engine_tests.jl:593-805 "Mean Field", :807-1147 "Structured"):

    ssnoise ~ Gamma, obsnoise ~ Gamma,   x_{i+1} ~ N(x_i, 1/ssnoise),   y_i ~ N(x_i, 1/obsnoise)

Why an array form is faithful: inside one call the scheduler first computes the messages the requested marginals depend
on and only then, in the final round, the marginals and the linked joint marginals (src/inference_engine.jl:576-628).
Messages with weak dependencies read the marginals as they stood BEFORE the call, so a call is a Jacobi step over the
requested variables; the chain messages of the structured model have strong dependencies and come out as the exact
forward/backward recursion.  tests/test_vmp_restatement.py pins this file, call by call, against the C restatement of
the engine (oracle/cortex_ref.c) driven through the transcribed resolvers and rules (tests/vmp_support.py).

Value conventions follow test/runtests.jl:48-93: Normal as (mean, precision), Gamma as (shape, scale)."""
import numpy as np


def _nprod(m1, w1, m2, w2):
    """product(NormalMeanPrecision, NormalMeanPrecision), runtests.jl:78-84 (same operation order)"""
    xi = m1 * w1 + m2 * w2
    w = w1 + w2
    return (1 / w) * xi, w


def _gamma_fold(scales):
    """left fold of product(Gamma, Gamma) over Gamma(1.5, scale_k): runtests.jl:86-88"""
    shape, scale = 1.5, float(scales[0])
    for s in scales[1:]:
        shape, scale = shape + 1.5 - 1, (scale * s) / (scale + s)
    return shape, scale


def _gamma_tree(scales, lo, hi):
    """the default resolver's segment tree over [lo, hi) (src/dependencies.jl:90-173): halves split at lo + len // 2"""
    if hi - lo == 1:
        return 1.5, float(scales[lo])
    mid = lo + (hi - lo) // 2
    a1, s1 = _gamma_tree(scales, lo, mid)
    a2, s2 = _gamma_tree(scales, mid, hi)
    return a1 + a2 - 1, (s1 * s2) / (s1 + s2)


def _gamma_product_default(scales):
    """marginal of a Gamma variable under the DEFAULT variable resolver: all-pairs fold up to degree 5, segment tree above"""
    n = len(scales)
    if n <= 5:
        return _gamma_fold(scales)
    mid = n // 2
    a1, s1 = _gamma_tree(scales, 0, mid)
    a2, s2 = _gamma_tree(scales, mid, n)
    return a1 + a2 - 1, (s1 * s2) / (s1 + s2)


class _Base:
    def __init__(self, y):
        self.y = np.asarray(y, dtype=np.float64)
        n = self.n = len(self.y)
        self.xm, self.xw = np.zeros(n), np.ones(n)          # q(x_i) = N(0, precision 1)      (:722-727)
        self.ss = (1.0, 1.0)                                # q(ssnoise) = Gamma(1, 1)
        self.obs = (1.0, 1.0)

    @staticmethod
    def gmean(g):
        return g[0] * g[1]

    def update(self, which):
        """which: the variable classes named in one update_marginals! call, in request order ("x", "ssnoise",
        "obsnoise"; all x are requested together).  Messages are computed from the state before the call; the final
        round then stores the marginals (and computes the linked joint marginals) in request order."""
        new = {w: getattr(self, "_" + w)() for w in which}
        for w in which:
            getattr(self, "_store_" + w)(new[w])

    def _store_ssnoise(self, v): self.ss = v
    def _store_obsnoise(self, v): self.obs = v


class MeanFieldVMP(_Base):
    """SSMMeanFieldInferenceRequestProcessor + MeanFieldResolver (:599-689): every message reads marginals only."""

    def _x(self):
        n, tau_s, tau_o = self.n, self.gmean(self.ss), self.gmean(self.obs)
        m, w = self.y.copy(), np.full(n, tau_o)              # likelihood_i -> x_i : N(y_i, E obsnoise)
        # fold order = ascending factor id: likelihood_i, transition_{i-1}, transition_i
        mm, ww = _nprod(m[1:], w[1:], self.xm[:-1], np.full(n - 1, tau_s))      # transition_{i-1} -> x_i : N(E x_{i-1}, E ssnoise)
        m[1:], w[1:] = mm, ww
        mm, ww = _nprod(m[:-1], w[:-1], self.xm[1:], np.full(n - 1, tau_s))     # transition_i -> x_i
        m[:-1], w[:-1] = mm, ww
        return m, w

    def _store_x(self, v): self.xm, self.xw = v

    def _ssnoise(self):
        vx = 1 / self.xw
        spread = vx[:-1] + vx[1:] + (self.xm[:-1] - self.xm[1:]) ** 2           # :679-684, deps in ascending id order
        return _gamma_fold(2 / spread)                                           # the mean-field resolver folds all messages

    def _obsnoise(self):
        spread = 1 / self.xw + (self.y - self.xm) ** 2                           # :671-677
        return _gamma_fold(2 / spread)


class StructuredVMP(_Base):
    """SSMStructuredInferenceRequestProcessor + StructuredResolver (:810-1030): belief propagation along the chain with
    the transition precision replaced by its current expectation, joint marginals of neighbouring states linked to the
    x variables, mean-field updates towards the two precisions."""

    def __init__(self, y):
        super().__init__(y)
        self.joint = None          # (mu1, mu2, W11, W12, W22) arrays of the n-1 joint marginals, undefined before update(x)
        self.to_f = None           # the chain messages into the transition factors, as the last update(x) left them
        self.obs_fresh = False     # q(obsnoise) was updated since the states last were

    def update(self, which):
        """One update_marginals! call.  A request of ONE class, or of the two precisions, is the base class's Jacobi step.  A request
        that names the states TOGETHER with precisions (the last call of the reference's experiment, :1113) is evaluated in an order
        that emerges from the lazy readiness flags (src/inference_engine.jl:575-608); pinned against the restated engine
        (tests/test_vmp_restatement.py) it is three calls, class by class — q(ssnoise) (found pending by the first chain message and
        computed on the fly), the states, q(obsnoise) (final round, from the new states) — whenever the states were updated before,
        n - 1 > 5, "ssnoise" comes before "x" in the request and q(obsnoise) was updated since the states last were.  Outside these
        conditions the reference interleaves per variable (some messages read the old expectation, some the new one): no array form."""
        which = list(which)
        if "x" in which and len(which) > 1:
            ok = self.joint is not None and self.n - 1 > 5
            ok = ok and ("ssnoise" not in which or which.index("ssnoise") < which.index("x"))
            ok = ok and ("obsnoise" not in which or self.obs_fresh)
            if not ok:
                raise NotImplementedError("a request of states and precisions together outside the conditions of StructuredVMP.update")
            for w in ("ssnoise", "x", "obsnoise"):
                if w in which:
                    self.update([w])
            return
        super().update(which)
        if "x" in which:
            self.obs_fresh = False
        if "obsnoise" in which:
            self.obs_fresh = True

    def _x(self):
        n, tau_s, tau_o = self.n, self.gmean(self.ss), self.gmean(self.obs)
        lm, lw = self.y, np.full(n, tau_o)                   # likelihood_i -> x_i : N(y_i, E obsnoise)   (:985-989)
        inv_tau = 1 / tau_s
        fm, fw = np.zeros(n), np.zeros(n)                    # transition_{i-1} -> x_i   (forward), i >= 1
        bm, bw = np.zeros(n), np.zeros(n)                    # transition_i -> x_i       (backward), i <= n-2
        to_f_fwd_m, to_f_fwd_w = np.zeros(n), np.zeros(n)    # x_i -> transition_i
        to_f_bwd_m, to_f_bwd_w = np.zeros(n), np.zeros(n)    # x_i -> transition_{i-1}
        for i in range(n - 1):
            if i == 0:
                m, w = lm[0], lw[0]
            else:
                m, w = _nprod(lm[i], lw[i], fm[i], fw[i])    # product of the OTHER messages into x_i, ascending factor id
            to_f_fwd_m[i], to_f_fwd_w[i] = m, w
            fm[i + 1], fw[i + 1] = m, 1 / (1 / w + inv_tau)  # :1004-1010
        for i in range(n - 1, 0, -1):
            if i == n - 1:
                m, w = lm[i], lw[i]
            else:
                m, w = _nprod(lm[i], lw[i], bm[i], bw[i])
            to_f_bwd_m[i], to_f_bwd_w[i] = m, w
            bm[i - 1], bw[i - 1] = m, 1 / (1 / w + inv_tau)
        xm, xw = lm.copy(), lw.copy()                        # marginal: likelihood_i, transition_{i-1}, transition_i
        xm[1:], xw[1:] = _nprod(xm[1:], xw[1:], fm[1:], fw[1:])
        xm[:-1], xw[:-1] = _nprod(xm[:-1], xw[:-1], bm[:-1], bw[:-1])
        return xm, xw, (to_f_fwd_m[:-1].copy(), to_f_fwd_w[:-1].copy(), to_f_bwd_m[1:].copy(), to_f_bwd_w[1:].copy())

    def _joints(self):
        """joint marginal of (x_i, x_{i+1}) at transition_i (:939-967) from deps = [x_i -> f, x_{i+1} -> f, q(ssnoise)],
        q(ssnoise) as it stands at the moment of the computation"""
        m_out, w_out, m_mu, w_mu = self.to_f
        tau_s = self.gmean(self.ss)
        xi_out, xi_mu = w_out * m_out, w_mu * m_mu
        a, b, c = w_out + tau_s, -tau_s, w_mu + tau_s
        det = a * c - b * b
        mu1 = (c * xi_out - b * xi_mu) / det
        mu2 = (a * xi_mu - b * xi_out) / det
        self.joint = (mu1, mu2, a, b * np.ones(self.n - 1), c)

    def _store_x(self, v):
        """final round: the marginals, then the linked joint marginals — with q(ssnoise) as it stands NOW (it may have
        been stored earlier in the same final round)."""
        self.xm, self.xw, self.to_f = v
        self._joints()

    def _ssnoise(self):
        if self.joint is None:     # the joint marginals are not computed yet: the messages are not pending, nothing changes
            return self.ss
        if self.n - 1 > 5:
            # Degree > 5: the marginal hangs off a segment tree, request_inference_for re-arms only the tree nodes, so the
            # scheduler descends into each message (signal.jl:466-490) and finds its joint marginal pending whenever
            # q(ssnoise) changed since the joint was last computed: the joint is refreshed first.  (Refreshing a joint
            # whose inputs did not change reproduces it, so "always" is the same thing.)  With degree <= 5 the messages are
            # direct dependencies, are pending at once, and read the joints as update(x) left them.
            self._joints()
        mu1, mu2, a, b, c = self.joint
        det = a * c - b * b
        v11, v12, v22 = c / det, -b / det, a / det
        spread = v11 - v12 - v12 + v22 + (mu1 - mu2) ** 2    # :1011-1016
        return _gamma_product_default(2 / spread)

    def _obsnoise(self):
        spread = 1 / self.xw + (self.y - self.xm) ** 2       # :990-995
        return _gamma_product_default(2 / spread)


def which_of(ids, x_ids, ssnoise, obsnoise):
    """translate the id list of an update_marginals! call into variable classes in request order"""
    out, xs = [], set(x_ids)
    for i in np.atleast_1d(ids):
        w = "ssnoise" if i == ssnoise else "obsnoise" if i == obsnoise else "x" if int(i) in xs else None
        if w is not None and w not in out:
            out.append(w)
    return out
