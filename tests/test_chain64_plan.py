"""CPU: the work plan of the chain-scan schedule for wide messages (cortex.jl_amd/csrc/cx_chain64_plan.h, the host logic behind
csrc/cx_mv64chain.hip) executed record by record in numpy.

What is checked: the plan — which potentials are composed from which children, which rule is applied to which sources in
which launch — yields, on every path and at every position, the exact forward/backward messages, i.e. what ONE
update_marginals! of the reference leaves on such a graph (src/inference_engine.jl:575-608), pinned by the
exact block-tridiagonal solve (oracle/exact.py).  The executor also enforces the launch semantics the device relies on: jobs of
one launch run in any order and never read what another job of the same launch writes."""
import numpy as np
import pytest

import cortex.jl_amd as cx
from oracle import exact
from tests.hostlogic import Plan64


class Arena:
    def __init__(self, plan, nslots, ptab, btab, d):
        self.d, self.dd = d, d * d
        self.mem = {"zero": np.zeros(plan.msg), "f2v": np.full(nslots * plan.msg, np.nan), "ptab": ptab, "btab": btab,
                    "pot": np.full(max(1, plan.n_pot) * plan.pot, np.nan), "ent": np.full(max(1, plan.n_ent) * plan.msg, np.nan)}
        self.written = {}      # (space, offset) -> job that wrote it in the current launch

    def view(self, handle, n, job=None, write=False):
        sp, off = Plan64.split(handle)
        if sp == "zero":
            assert not write
            return self.mem["zero"][:n]
        key = (sp, off)
        if write:
            assert sp in ("f2v", "pot", "ent")
            self.written[key] = job
        else:
            owner = self.written.get(key)
            assert owner is None or owner == job, f"job {job} reads {key}, which job {owner} of the same launch writes"
        assert off + n <= self.mem[sp].size
        return self.mem[sp][off:off + n]

    def mat(self, handle, job):
        return self.view(handle, self.dd, job).reshape(self.d, self.d)

    def msg(self, handle, job):
        v = self.view(handle, self.d + self.dd, job)
        return v[:self.d], v[self.d:].reshape(self.d, self.d)


def run_plan(plan, ar, rng):
    d, dd = ar.d, ar.dd
    for jobs in plan.compose_launches:
        ar.written = {}
        for j in rng.permutation(len(jobs)):
            out, first, n = (int(x) for x in jobs[j])
            ch = plan.children[first:first + n]
            P, B, C = (ar.mat(ch[0][k], j).copy() for k in (0, 1, 3))
            assert np.array_equal(ar.mat(ch[0][2], j), B.T)
            h, c = ar.view(ch[0][4], d, j).copy(), ar.view(ch[0][5], d, j).copy()
            for r in ch[1:]:
                P2, B2, C2 = (ar.mat(r[k], j) for k in (0, 1, 3))
                assert np.allclose(ar.mat(r[2], j), B2.T, rtol=0, atol=0)
                h2, c2 = ar.view(r[4], d, j), ar.view(r[5], d, j)
                se, sL = np.zeros(d), np.zeros((d, d))
                for s in r[6:9]:
                    e, Lm = ar.msg(s, j)
                    se, sL = se + e, sL + Lm
                Mi = np.linalg.inv(C + sL + P2)
                g = c + se + h2
                P, h = P - B.T @ Mi @ B, h + B.T @ Mi @ g
                C, c = C2 - B2 @ Mi @ B2.T, c2 + B2 @ Mi @ g
                B = B2 @ Mi @ B
            o = ar.view(out, plan.pot, j, write=True)
            o[:] = np.concatenate([P.ravel(), B.ravel(), B.T.ravel(), C.ravel(), h, c])
    for jobs in plan.walk_launches:
        ar.written = {}
        for j in rng.permutation(len(jobs)):
            _, first, n = (int(x) for x in jobs[j])
            for st in plan.steps[first:first + n]:
                e_in, L_in = ar.view(st[6], d, j).copy(), np.zeros((d, d))
                for s in st[0:3]:
                    e, Lm = ar.msg(s, j)
                    e_in, L_in = e_in + e, L_in + Lm
                P, Bt, C = (ar.mat(st[k], j) for k in (3, 4, 5))
                Mi = np.linalg.inv(L_in + P)
                out = ar.view(st[8], d + dd, j, write=True)
                out[:d] = ar.view(st[7], d, j) + Bt.T @ Mi @ e_in
                out[d:] = (C - Bt.T @ Mi @ Bt).ravel()


def chain_inputs(lengths, d, seed, extra_side=False):
    """several independent d-dimensional state-space chains in an abstract slot space: slot p = the likelihood message into
    position p, npos + l / npos + nlinks + l = the forward / backward message of link l; extra_side: a second side slot
    (a prior) at every third position"""
    models = [cx.synth.lgssm_chain(T, d=d, seed=seed + 7 * i) for i, T in enumerate(lengths)]
    A, Q, R = (models[0].meta[k] for k in ("A", "Q", "R"))
    npos, nlinks = sum(lengths), sum(T - 1 for T in lengths)
    nslots = npos + 2 * nlinks + npos
    link_pos, head_f, head_b = [], [], []
    p = 0
    for T in lengths:
        for t in range(T - 1):
            link_pos.append(p + t); head_f.append(t == 0); head_b.append(t == T - 2)
        p += T
    link_pos = np.array(link_pos, dtype=np.int32)
    ls = np.arange(nlinks, dtype=np.int32)
    side = np.full((npos, 3), -1, dtype=np.int32)
    side[:, 0] = np.arange(npos)
    Qi, Ri = np.linalg.inv(Q), np.linalg.inv(R)
    fw = np.concatenate([(A.T @ Qi @ A).ravel(), (Qi @ A).ravel(), Qi.ravel()])
    bw = np.concatenate([Qi.ravel(), (A.T @ Qi).ravel(), (A.T @ Qi @ A).ravel()])
    ptab = np.concatenate([fw, bw])
    btab = np.concatenate([(Qi @ A).T.ravel(), (A.T @ Qi).T.ravel()])
    y = np.concatenate([m.data_y for m in models])
    prior_prec = np.zeros((npos, d, d)); prior_eta = np.zeros((npos, d))
    if extra_side:
        idx = np.arange(0, npos, 3)
        side[idx, 1] = npos + 2 * nlinks + idx
        prior_prec[idx] = 0.5 * np.eye(d); prior_eta[idx] = 0.25
    return dict(models=models, A=A, Q=Q, R=R, npos=npos, nlinks=nlinks, nslots=nslots, link_pos=link_pos, frm=npos + nlinks + ls, to=npos + ls,
                tab_fwd=np.zeros(nlinks, np.int32), tab_bwd=np.ones(nlinks, np.int32), head_f=head_f, head_b=head_b, side=side,
                ptab=ptab, btab=btab, y=y, Ri=Ri, prior_prec=prior_prec, prior_eta=prior_eta)


def check(lengths, d, K0, fan, seed=1, extra_side=False, lanes=1024, root=False):
    g = chain_inputs(lengths, d, seed, extra_side)
    plan = Plan64(d, g["link_pos"], g["frm"], g["to"], g["tab_fwd"], g["tab_bwd"], g["head_f"], g["head_b"], g["side"], K0=K0, fan=fan, lanes=lanes, root=root)
    ar = Arena(plan, g["nslots"], g["ptab"], g["btab"], d)
    f2v = ar.mem["f2v"].reshape(g["nslots"], plan.msg)
    npos, nlinks = g["npos"], g["nlinks"]
    f2v[:npos, :d] = g["y"] @ g["Ri"].T
    f2v[:npos, d:] = g["Ri"].ravel()
    pr = npos + 2 * nlinks
    f2v[pr:pr + npos, :d] = g["prior_eta"]
    f2v[pr:pr + npos, d:] = g["prior_prec"].reshape(npos, -1)
    run_plan(plan, ar, np.random.default_rng(seed))
    plan.arena, plan.inputs = ar, g
    # marginal of every position = its side information + the two chain messages into it
    p = 0
    link = 0
    for m, T in zip(g["models"], lengths):
        Jprior = g["prior_prec"][p:p + T] if extra_side else None
        if extra_side:
            # exact posterior with the extra unary terms: fold them into the observation model by hand (dense solve)
            n = T * d
            J = np.zeros((n, n)); hvec = np.zeros(n)
            Qi = np.linalg.inv(g["Q"]); A = g["A"]
            for t in range(T):
                sl = slice(t * d, (t + 1) * d)
                J[sl, sl] += g["Ri"] + Jprior[t]
                hvec[sl] += g["Ri"] @ m.data_y[t] + g["prior_eta"][p + t]
                if t + 1 < T:
                    s2 = slice((t + 1) * d, (t + 2) * d)
                    J[sl, sl] += A.T @ Qi @ A; J[s2, s2] += Qi; J[s2, sl] -= Qi @ A; J[sl, s2] -= (Qi @ A).T
            S = np.linalg.inv(J)
            em = (S @ hvec).reshape(T, d)
            ecov = np.stack([S[t * d:(t + 1) * d, t * d:(t + 1) * d] for t in range(T)])
        else:
            em, ecov = exact.lgssm_posterior(m.data_y, g["A"], g["Q"], g["R"])
        for t in range(T):
            eta, lam = f2v[p + t, :d].copy(), f2v[p + t, d:].reshape(d, d).copy()
            if extra_side:
                eta += g["prior_eta"][p + t]; lam += g["prior_prec"][p + t]
            if t > 0:
                row = f2v[npos + link + t - 1]; eta += row[:d]; lam += row[d:].reshape(d, d)
            if t < T - 1:
                row = f2v[npos + nlinks + link + t]; eta += row[:d]; lam += row[d:].reshape(d, d)
            cov = np.linalg.inv(lam)
            assert np.allclose(cov, ecov[t], rtol=1e-9, atol=1e-11), (lengths, K0, fan, t)
            assert np.allclose(cov @ eta, em[t], rtol=1e-9, atol=1e-10), (lengths, K0, fan, t)
        p += T; link += T - 1
    return plan


@pytest.mark.parametrize("d", [1, 3])
@pytest.mark.parametrize("T,K0,fan", [(2, 4, 4), (5, 4, 4), (6, 4, 2), (9, 2, 2), (33, 4, 4), (64, 1, 2), (65, 4, 4), (200, 3, 3), (257, 2, 4)])
def test_one_path_every_marginal_exact(T, K0, fan, d):
    plan = check([T], d, K0, fan, seed=T)
    n = T - 1
    assert plan.n_rules >= 2 * n
    if n <= K0:
        assert plan.n_pot == 0 and not any(len(j) for j in plan.compose_launches)      # one block: nothing to compose


@pytest.mark.parametrize("T,K0,fan", [(2, 4, 2), (3, 4, 2), (9, 2, 2), (33, 4, 4), (64, 1, 2), (200, 3, 3)])
def test_root_potential_of_a_time_block(T, K0, fan):
    """Input.root (cx_chain_block_maps for dim 64: a time block of a partitioned chain): every path also gets the ONE potential of its
    two end variables with everything between them summed out (interior side information included, the ends' excluded) — checked
    against the Schur complement of the dense joint precision — and every marginal is still exact."""
    d = 3
    plan = check([T, 5], d, K0, fan, seed=T, root=True)
    g, ar = plan.inputs, plan.arena
    assert len(plan.root_pot) == 2
    p0 = 0
    for i, Tn in enumerate((T, 5)):
        sp, off = Plan64.split(plan.root_pot[i])
        assert sp == "pot"
        rec = ar.mem["pot"][off:off + plan.pot]
        dd = d * d
        P, B, Bt, Cm = (rec[k * dd:(k + 1) * dd].reshape(d, d) for k in range(4))
        h, c = rec[4 * dd:4 * dd + d], rec[4 * dd + d:]
        # dense joint over the path's variables: transitions + the side information of the INTERIOR positions
        A, Qi, Ri = g["A"], np.linalg.inv(g["Q"]), g["Ri"]
        n = Tn * d
        J, hv = np.zeros((n, n)), np.zeros(n)
        for t in range(Tn):
            sl = slice(t * d, (t + 1) * d)
            if 0 < t < Tn - 1:
                J[sl, sl] += Ri
                hv[sl] += Ri @ g["y"][p0 + t]
            if t + 1 < Tn:
                s2 = slice((t + 1) * d, (t + 2) * d)
                J[sl, sl] += A.T @ Qi @ A; J[s2, s2] += Qi; J[s2, sl] -= Qi @ A; J[sl, s2] -= (Qi @ A).T
        ends = np.r_[0:d, n - d:n]
        mid = np.arange(d, n - d)
        if len(mid):
            W = np.linalg.inv(J[np.ix_(mid, mid)])
            Je = J[np.ix_(ends, ends)] - J[np.ix_(ends, mid)] @ W @ J[np.ix_(mid, ends)]
            he = hv[ends] - J[np.ix_(ends, mid)] @ W @ hv[mid]
        else:
            Je, he = J, hv
        # psi = exp(-1/2 xa'P xa - 1/2 xb'C xb + xb'B xa + h'xa + c'xb): the joint precision is [[P, -B'], [-B, C]]
        assert np.allclose(P, Je[:d, :d], rtol=1e-9, atol=1e-11) and np.allclose(Cm, Je[d:, d:], rtol=1e-9, atol=1e-11)
        assert np.allclose(B, -Je[d:, :d], rtol=1e-9, atol=1e-11) and np.array_equal(Bt, B.T)
        assert np.allclose(h, he[:d], rtol=1e-9, atol=1e-11) and np.allclose(c, he[d:], rtol=1e-9, atol=1e-11)
        p0 += Tn


def test_several_paths_of_different_depths_share_the_launches():
    plan = check([1, 2, 40, 1, 7, 150, 3], 2, K0=2, fan=3, seed=5)
    assert plan.levels >= 3
    # the top walk of every composed path sits in the first walk launch, the link walks in the last
    assert len(plan.walk_launches[-1]) == 2 * sum(-(-(T - 1) // 2) for T in (2, 40, 7, 150, 3))


def test_positions_with_a_second_side_input():
    check([30, 11], 2, K0=3, fan=2, seed=9, extra_side=True)


def test_default_block_length_fills_the_lanes():
    g = chain_inputs([5000], 1, 3)
    plan = Plan64(1, g["link_pos"], g["frm"], g["to"], g["tab_fwd"], g["tab_bwd"], g["head_f"], g["head_b"], g["side"], lanes=1024)
    assert plan.K0 == 5 and len(plan.compose_launches[0]) == 1000
    # work count: one composition per link except the first of every block and group; two rules per link + the tree's walks
    assert plan.n_compositions == sum(int(j[:, 2].sum()) - len(j) for j in plan.compose_launches)
    assert plan.n_rules == sum(int(j[:, 2].sum()) for j in plan.walk_launches)


def test_a_position_with_four_inputs_is_refused():
    g = chain_inputs([6], 1, 3)
    g["side"][2] = [2, 2, 2]       # three side slots on an interior position + its entering message
    with pytest.raises(RuntimeError, match="more than three inputs"):
        Plan64(1, g["link_pos"], g["frm"], g["to"], g["tab_fwd"], g["tab_bwd"], g["head_f"], g["head_b"], g["side"], K0=2)


def test_plan_builder_under_address_and_ub_sanitizers():
    """the same builder compiled with -fsanitize=address,undefined (GPU ASan is not available on this pool: the host logic is
    checked on the CPU)"""
    import os
    import subprocess
    import sys

    from cortex.jl_amd import build as B

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # libstdc++ beside libasan: the interceptor of __cxa_throw needs the real one loaded before it (python itself is plain C)
    pre = " ".join(subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libstdc++.so.6"))
    env = dict(os.environ, LD_PRELOAD=pre, CXH_LIB=B.build_hostlogic(asan=True), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1")
    code = ("import sys; sys.path.insert(0, %r); import pytest; sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', %r, '-k', "
            "'several_paths or second_side or refused or (one_path and 257)']))") % (root, os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
