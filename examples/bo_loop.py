"""A complete Bayesian-optimisation loop on the device path, written against the same names a Cornell-MOE user script
uses (cpp_wrappers mirror + GPP stand-in): sample GP hyper-parameters from their posterior (Metropolis walkers on
`evaluate_log_likelihood_at_hyperparameter_list`, all proposals of a step factorised together on the device), build the `GaussianProcessMCMC` ensemble, choose the next q points by multistart optimisation of the
MCMC-averaged q-KG, evaluate, add the points, report the minimiser of the averaged posterior mean.  The flow follows the
reference's examples/main.py (KG branch, :104-260) with its emcee sampler replaced by a 40-line Metropolis sampler (emcee
is not a dependency of this repository) and the per-point Python loops replaced by the batched calls the boundary offers.

    python examples/bo_loop.py [iterations] [q] [num_mcmc]
"""
import sys
import time

import numpy as np

_ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, _ROOT)
sys.path.insert(0, __import__("os").path.join(_ROOT, "tests"))  # the wrapper mirror is test infrastructure (tests/wrappers_mirror.py);
from cornell_moe_amd import GPP  # noqa: E402                     # with the reference installed, its own cpp_wrappers take this place
import wrappers_mirror as cw  # noqa: E402


def branin(x):
    """Branin on [0, 1]^2 (rescaled from [-5, 10] x [0, 15]); minimum 0.397887."""
    a = 15.0 * x[..., 0] - 5.0
    b = 15.0 * x[..., 1]
    return (b - 5.1 / (4 * np.pi ** 2) * a ** 2 + 5.0 / np.pi * a - 6.0) ** 2 + 10.0 * (1 - 1 / (8 * np.pi)) * np.cos(a) + 10.0


def sample_hyperparameters(X, y, num_samples, rng, steps=150, noise=1e-4):
    """Metropolis chains over log(alpha, lengths) under a flat prior on [-4, 4] (the role of emcee in the reference, which
    the reference drives one `compute_log_likelihood` call per walker).  Here 2 x num_samples independent walkers move in
    lockstep and every step is ONE `evaluate_log_likelihood_at_hyperparameter_list` call: the device factorises all
    proposals together (one launch per factorisation step for the whole batch)."""
    dim, n = X.shape[1], X.shape[0]
    W = max(2 * num_samples, 16)
    Xl, yl = list(X.ravel()), list(y)

    def lnprob(H):
        E = np.exp(H)
        flat = np.hstack([E, np.full((H.shape[0], 1), noise)]).ravel()
        lp = np.array(GPP.evaluate_log_likelihood_at_hyperparameter_list(
            list(flat), Xl, yl, dim, n, GPP.LogLikelihoodTypes.log_marginal_likelihood, [1.0, [1.0] * dim], [noise], [], 0,
            H.shape[0], 1, {}))
        lp[np.any(np.abs(H) > 4.0, axis=1)] = -np.inf
        return lp
    H = np.r_[np.log(np.var(y) + 1e-3), np.full(dim, np.log(0.3))] + 0.1 * rng.standard_normal((W, 1 + dim))
    lp = lnprob(H)
    for it in range(steps):
        prop = H + 0.25 * rng.standard_normal(H.shape)
        lpp = lnprob(prop)
        acc = np.log(rng.uniform(size=W)) < lpp - lp
        H[acc], lp[acc] = prop[acc], lpp[acc]
    return np.exp(H[:num_samples])


def main():
    iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    q = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    num_mcmc = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rng = np.random.default_rng(0)
    dim, noise = 2, 1e-4
    X = rng.uniform(size=(8, dim))
    y = branin(X)
    dom = cw.TensorProductDomain([[0.0, 1.0]] * dim)
    rnd = GPP.RandomnessSourceContainer(1)
    rnd.SetExplicitNormalRNGSeed(1)
    rnd.SetExplicitUniformGeneratorSeed(2)
    inner_params = cw.GradientDescentParameters(1, 6, 1, 3, 0.0, 1.0, 0.1, 1e-10)     # examples/main.py:123-130
    outer_params = cw.GradientDescentParameters(40, 20, 2, 4, 0.7, 0.1, 0.5, 1e-8)    # (:132-139, scaled down)
    print("initial best observed value %.4f" % y.min())
    for it in range(iterations):
        t0 = time.perf_counter()
        ys = (y - y.mean()) / y.std()  # the GP sees standardised values (hyper-parameter box [-4, 4] in log space)
        hypers = sample_hyperparameters(X, ys, num_mcmc, rng, noise=noise)
        t1 = time.perf_counter()
        hd = cw.HistoricalData(dim=dim, num_derivatives=0)
        hd.append_sample_points([cw.SamplePoint(X[i], [ys[i]], noise) for i in range(X.shape[0])])
        gpm = cw.GaussianProcessMCMC(hypers, np.full((num_mcmc, 1), noise), hd, [])
        models = gpm.member_models()
        # discretisation of the domain per ensemble member: random points + the member's best posterior-mean point
        ps = cw.PosteriorMeanMCMC(models, 0)
        inner = cw.GradientDescentOptimizer(dom, ps, inner_params)
        cand = rng.uniform(size=(1000, dim))
        discrete_list = []
        for gp in models:
            mu = gp.compute_mean_of_additional_points(cand)            # one batched device call, not 1000
            keep = cand[np.argsort(mu)[:9]]
            x0 = keep[0]
            xs = np.array(GPP.posterior_mean_optimization(gp._gaussian_process, 0, inner.optimizer_parameters,
                                                          [0.0, 1.0] * dim, list(x0), {}))
            discrete_list.append(np.vstack([keep, xs]))
        t2 = time.perf_counter()
        kg = cw.KnowledgeGradientMCMC(gpm, models, 0, inner, discrete_list, q, num_mc_iterations=2 ** 7, randomness=rnd)
        outer = cw.GradientDescentOptimizer(dom, kg, outer_params, 200)
        status = {}
        nxt = cw.multistart_knowledge_gradient_mcmc_optimization(outer, inner, None, discrete_list, q, discrete_list[0].shape[0],
                                                                 randomness=rnd, max_num_threads=1, status=status)
        kg.set_current_point(nxt)
        voi = kg.compute_knowledge_gradient_mcmc()
        t3 = time.perf_counter()
        X = np.vstack([X, nxt])
        y = np.r_[y, branin(nxt)]
        # report: minimiser of the ensemble-averaged posterior mean over candidates + sampled points
        pts = np.vstack([cand, X])
        mean = np.mean([gp.compute_mean_of_additional_points(pts) for gp in models], axis=0)
        report = pts[int(np.argmin(mean))]
        assert len(np.unique(hypers[:, 0])) > 1, "the hyper-parameter chain did not move"
        print("iteration %d: hyper-parameter sampling %.2f s, ensemble + discretisation %.2f s, KG-MCMC optimisation %.2f s "
              "(KG %.4g, found=%s); suggested %s; best observed %.4f; reported point %s f=%.4f" % (
                  it, t1 - t0, t2 - t1, t3 - t2, voi, list(status.values()), np.round(nxt, 3).tolist(), y.min(),
                  np.round(report, 3).tolist(), float(branin(report))), flush=True)
    return y


if __name__ == "__main__":
    main()
