"""N>1 path on CPU: two gloo ranks, each owning a contiguous document shard, exchanging the packed
sufficient statistics once per EM iteration (strutopy_amd.dist) -- must reproduce the single-process
fit.  OracleEngine stands in for the HIP engine (test hook); the exchange / M-step-from-moments logic
is the code under test."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, model_type, iters, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OMP_NUM_THREADS="2")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle_engine import OracleEngine
    from strutopy_amd import dist as sdist
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.stm import STM
    g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    full = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    sdist.init_from_env("gloo")
    comm = sdist.GlooComm()
    lo, hi = sdist.shard_bounds(full.indptr, world)[rank]
    m = STM(documents=full.slice(lo, hi), dictionary=None, content=False, K=int(g["K"]), X=g["X"][lo:hi, 0],
            kappa_interactions=False, max_em_iter=iters, sigma_prior=0, convergence_threshold=1e-12,
            init_type="random", model_type=model_type, comm=comm, engine=OracleEngine(nthreads=2))
    assert m.N_total == full.N
    m.expectation_maximization(saving=False)
    q.put((rank, lo, hi, list(m.last_bounds), m.sigma.copy(), m.beta.copy(), m.mu.copy(), m.eta.copy(),
           getattr(m, "gamma", None)))
    import torch.distributed as tdist
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("case,model_type", [("c1_k10", "STM"), ("toy_ctm", "CTM")])
def test_two_rank_fit_equals_single_process(case, model_type):
    import torch.multiprocessing as mp
    from _oracle_engine import OracleEngine
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.stm import STM
    iters = 2
    g = load_golden(case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, model_type, iters, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    ref = STM(documents=full, dictionary=None, content=False, K=int(g["K"]), X=g["X"][:, 0],
              kappa_interactions=False, max_em_iter=iters, sigma_prior=0, convergence_threshold=1e-12,
              init_type="random", model_type=model_type, engine=OracleEngine())
    ref.expectation_maximization(saving=False)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == full.N
    for rank, lo, hi, bounds, sigma, beta, mu, eta, gamma in res:
        assert np.allclose(bounds, ref.last_bounds, rtol=1e-11)            # the ELBO is the all-reduced sum
        assert np.allclose(sigma, ref.sigma, rtol=1e-8, atol=1e-12)
        assert np.allclose(beta, ref.beta, rtol=1e-8, atol=1e-14)
        assert np.allclose(mu, ref.mu[lo:hi], atol=1e-9) and np.allclose(eta, ref.eta[lo:hi], atol=1e-8)
        if model_type == "STM":
            assert np.allclose(gamma, ref.gamma, rtol=1e-7, atol=1e-10)
    # both ranks finish the (replicated) M-step with identical global parameters
    assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])
    # and the trace still matches the reference's golden trace
    for it in range(iters):
        assert res[0][3][it] == pytest.approx(float(g[f"it{it}_bound"]), rel=1e-8)
