"""N>1 path on CPU: two ranks (torch.distributed/gloo, and the product's own stdlib TCP group), each owning a
contiguous document shard, exchanging the packed sufficient statistics once per EM iteration
(strutopy_amd.dist) -- must reproduce the single-process fit.  OracleEngine stands in for the HIP
engine (test hook); the exchange / M-step-from-moments logic is the code under test."""
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


def _worker(rank, world, port, case, model_type, iters, q, group="gloo", xkind="binary", init="random", outdir=None, mode="ols"):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OMP_NUM_THREADS="2")
    os.environ.pop("STM_RDZV_PORT", None)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _oracle_engine import OracleEngine
    from strutopy_amd import dist as sdist
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.stm import STM
    g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    full = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    if group == "gloo":
        import datetime

        import torch.distributed as tdist
        tdist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        from _gloo_rig import GlooComm
        comm = GlooComm()
    else:   # the product's host group: stdlib sockets, rendezvous file keyed by MASTER_PORT
        comm = sdist.HostComm(sdist.init_from_env(timeout=120))
    lo, hi = sdist.shard_bounds(full.indptr, world)[rank]
    X = _covariate(g, xkind)
    m = STM(documents=full.slice(lo, hi), dictionary=None, content=False, K=int(g["K"]), X=X[lo:hi],
            kappa_interactions=False, max_em_iter=iters, sigma_prior=0, convergence_threshold=1e-12,
            init_type=init, model_type=model_type, mode=mode, comm=comm, engine=OracleEngine(nthreads=1),
            exchange="single" if rank_exchange_single(xkind) else "split")   # (the host reduction is one packed message either way)
    assert m.N_total == full.N
    m.expectation_maximization(saving=outdir is not None, output_dir=outdir)
    d = m.solver_diagnostics()
    q.put((rank, lo, hi, list(m.last_bounds), m.sigma.copy(), m.beta.copy(), m.mu.copy(), m.eta.copy(),
           getattr(m, "gamma", None), d["nit"].copy(), d["nfev"].copy()))
    if outdir is not None:
        np.save(os.path.join(outdir, f"rank{rank}_theta"), m.theta)     # what this rank held when save_model ran
    comm.barrier()
    if group == "gloo":
        tdist.destroy_process_group()
    else:
        comm.group.close()


def rank_exchange_single(xkind):
    return xkind == "sorted3"


def _covariate(g, xkind):
    """binary: the golden's own 0/1 column.  sorted3: a three-level covariate SORTED along the corpus, so each
    contiguous shard sees a different subset of the levels (one shard even sees only {0, 1}, which on its own
    would count as "already 0/1", stm.py:665)."""
    if xkind == "binary":
        return g["X"][:, 0]
    N = len(g["indptr"]) - 1
    lev = np.minimum(np.arange(N) * 3 // max(N - 40, 1), 2)
    if xkind == "pandas_str":   # a string column of a DataFrame: .to_numpy() hands over an object array
        import pandas as pd
        return pd.DataFrame({"field": np.array(["arts", "maths", "stats"], dtype=object)[lev]})
    return lev


@pytest.mark.parametrize("case,model_type,group,xkind", [("c1_k10", "STM", "gloo", "binary"), ("toy_ctm", "CTM", "gloo", "binary"),
                                                         ("c1_k10", "STM", "tcp", "binary"), ("c1_k10", "STM", "tcp", "sorted3"),
                                                         ("c1_k10", "STM", "tcp", "pandas_str")])
def test_two_rank_fit_equals_single_process(case, model_type, group, xkind, tmp_path):
    import pickle

    import torch.multiprocessing as mp
    from _oracle_engine import OracleEngine
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.stm import STM
    iters = 2
    g = load_golden(case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    out2, out1 = str(tmp_path / "two_ranks"), str(tmp_path / "single")
    os.makedirs(out2)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, model_type, iters, q, group, xkind, "random", out2))
             for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    ref = STM(documents=full, dictionary=None, content=False, K=int(g["K"]), X=_covariate(g, xkind),
              kappa_interactions=False, max_em_iter=iters, sigma_prior=0, convergence_threshold=1e-12,
              init_type="random", model_type=model_type, engine=OracleEngine(nthreads=1))
    ref.expectation_maximization(saving=True, output_dir=out1)
    rd = ref.solver_diagnostics()
    searched = np.concatenate([(r[9] != rd["nit"][r[1]:r[2]]) | (r[10] != rd["nfev"][r[1]:r[2]]) for r in res])   # documents whose last search ran differently
    # save_model on the sharded fit (stm.py:1120-1149, called from stm.py:880): rank 0 alone wrote the reference's files, with
    # the N x K arrays of the WHOLE corpus -- exactly the two shards' rows in corpus order -- and the single-process fit's
    # file set / shapes / dtypes
    shard_files = sorted(f for f in os.listdir(out2) if f.startswith("rank"))
    assert shard_files == ["rank0_theta.npy", "rank1_theta.npy"]
    assert sorted(set(os.listdir(out2)) - set(shard_files)) == sorted(os.listdir(out1))
    assert np.array_equal(np.load(os.path.join(out2, "theta_hat.npy")),
                          np.concatenate([np.load(os.path.join(out2, f)) for f in shard_files]))
    assert np.array_equal(np.load(os.path.join(out2, "eta_hat.npy")), np.concatenate([res[0][7], res[1][7]]))
    assert np.array_equal(np.load(os.path.join(out2, "mu_hat.npy")), np.concatenate([res[0][6], res[1][6]]))
    for f in sorted(os.listdir(out1)):
        if f.endswith(".npy"):
            a, b = (np.load(os.path.join(d, f), allow_pickle=True) for d in (out2, out1))
            assert a.shape == b.shape and a.dtype == b.dtype, f
            if f == "X.npy":
                assert np.array_equal(a, b)         # the covariate rows, in corpus order
            elif f in ("eta_hat.npy", "theta_hat.npy"):
                # per document: two shards add beta_ss in another order than one process does (1e-16), and once in a while a
                # document's last line search accepts one step more or less for it (DESIGN.md sections 2 and 7; profiles/HISTORY.md section 9, "noise-level accept /
                # reject") -- such a row must show it in the solver's own counts; every other row holds the tight tolerance
                rows = ~np.all(np.isclose(a, b, rtol=1e-7, atol=1e-8), axis=1)
                assert rows.sum() <= 2 and not np.any(rows & ~searched) and np.allclose(a, b, rtol=0, atol=1e-3), (f, int(rows.sum()))
            else:
                assert np.allclose(a, b, rtol=1e-6, atol=1e-8), f
    with open(os.path.join(out2, "lower_bound.pickle"), "rb") as fh:
        assert pickle.load(fh) == res[0][3]
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == full.N
    for rank, lo, hi, bounds, sigma, beta, mu, eta, gamma, _nit, _nfev in res:
        # the ELBO is the all-reduced sum; the second iteration sees the first one's 1e-16 summation-order differences amplified
        assert np.isclose(bounds[0], ref.last_bounds[0], rtol=1e-12) and np.allclose(bounds, ref.last_bounds, rtol=1e-9)
        assert np.allclose(sigma, ref.sigma, rtol=1e-8, atol=1e-12)
        assert np.allclose(beta, ref.beta, rtol=1e-8, atol=1e-14)
        assert np.allclose(mu, ref.mu[lo:hi], atol=1e-9)
        off = ~np.all(np.isclose(eta, ref.eta[lo:hi], rtol=0, atol=1e-8), axis=1)      # (a noise-level accept / reject, see above)
        assert off.sum() <= 2 and not np.any(off & ~searched[lo:hi]) and np.allclose(eta, ref.eta[lo:hi], atol=1e-3)
        if model_type == "STM":
            assert np.allclose(gamma, ref.gamma, rtol=1e-7, atol=1e-10)
    # both ranks finish the (replicated) M-step with identical global parameters
    assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])
    # and the trace still matches the reference's golden trace
    for it in range(iters if xkind == "binary" else 1):
        assert res[0][3][it] == pytest.approx(float(g[f"it{it}_bound"]), rel=1e-8)


def test_two_rank_lasso_fit_equals_single_process():
    """mode="lasso" (stm.py:677-681) on a document-sharded fit: the coefficients come from coordinate descent on the centred
    Gram matrix (strutopy_amd.stm.lasso_from_moments), i.e. from the all-reduced moments alone -- two ranks, each holding a
    different subset of a three-level covariate's levels, end with the single-process fit, whose host M-step in turn is
    sklearn's Lasso on the full eta (what the reference calls)."""
    import torch.multiprocessing as mp
    from _oracle_engine import OracleEngine
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.stm import STM
    case, iters = "c1_k10", 2
    g = load_golden(case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, "STM", iters, q, "gloo", "sorted3", "random", None, "lasso")) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    out = {}
    for resident in (True, False):     # resident: lasso from the moments; host: sklearn.linear_model.Lasso on eta (stm.py:677-681)
        ref = STM(documents=full, dictionary=None, content=False, K=int(g["K"]), X=_covariate(g, "sorted3"), kappa_interactions=False,
                  max_em_iter=iters, sigma_prior=0, convergence_threshold=1e-12, init_type="random", model_type="STM", mode="lasso",
                  engine=OracleEngine(nthreads=1))
        ref.expectation_maximization(saving=False, resident=resident)
        out[resident] = ref
    ref, host = out[True], out[False]
    assert np.allclose(ref.gamma, host.gamma, rtol=1e-6, atol=1e-9) and np.allclose(ref.sigma, host.sigma, rtol=1e-7, atol=1e-10)
    assert np.allclose(ref.last_bounds, host.last_bounds, rtol=1e-9)
    for rank, lo, hi, bounds, sigma, beta, mu, eta, gamma, _nit, _nfev in res:
        assert gamma.shape == (int(g["K"]) - 1, 3)
        assert np.allclose(bounds, ref.last_bounds, rtol=1e-9)
        assert np.allclose(gamma, ref.gamma, rtol=1e-7, atol=1e-10) and np.allclose(sigma, ref.sigma, rtol=1e-8, atol=1e-12)
        assert np.allclose(beta, ref.beta, rtol=1e-8, atol=1e-14) and np.allclose(mu, ref.mu[lo:hi], atol=1e-9)
    assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])


def test_two_rank_fit_with_spectral_init_equals_single_process():
    """init_type="spectral" (what the reference's canonical caller uses, src/05_train.py:92; stm.py:420-423) on a sharded fit:
    gram (stm.py:122-157) is a sum over documents, so every rank forms it on its own shard, the matrices are summed once,
    and all ranks find the reference's anchors and the single-process beta."""
    import torch.multiprocessing as mp
    from _oracle_engine import OracleEngine
    from strutopy_amd.corpus import PackedCorpus
    from strutopy_amd.spectral import spectral_init
    from strutopy_amd.stm import STM
    case, iters = "c1_k10", 2
    g, gs = load_golden(case), load_golden("spectral_c1")
    assert str(gs["corpus"]) == case
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, "STM", iters, q, "tcp", "binary", "spectral")) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    full = PackedCorpus(g["indptr"], g["indices"], g["counts"], int(g["V"]))
    det = {}
    beta0 = spectral_init(full, int(g["K"]), int(g["V"]), verbose=False, engine=OracleEngine(), details=det)
    assert np.array_equal(det["anchor"].astype(np.int64), gs["anchor"].astype(np.int64))      # the reference's anchors
    assert np.allclose(beta0, gs["beta"], rtol=1e-7, atol=1e-12)
    ref = STM(documents=full, dictionary=None, content=False, K=int(g["K"]), X=_covariate(g, "binary"),
              kappa_interactions=False, max_em_iter=iters, sigma_prior=0, convergence_threshold=1e-12,
              init_type="spectral", model_type="STM", engine=OracleEngine())
    ref.expectation_maximization(saving=False)
    for rank, lo, hi, bounds, sigma, beta, mu, eta, gamma, _nit, _nfev in res:
        # the shards' gram matrices are added in a different order than a single process adds the documents (1e-16), the
        # per-term QPs and two EM iterations amplify that: the first ELBO (a function of the initial beta alone) pins the
        # initialisation, the state after the fit is compared at the amplified level
        assert np.isclose(bounds[0], ref.last_bounds[0], rtol=1e-9) and np.allclose(bounds, ref.last_bounds, rtol=1e-7)
        assert np.allclose(sigma, ref.sigma, rtol=1e-5, atol=1e-8)
        assert np.allclose(beta, ref.beta, rtol=1e-5, atol=1e-11)
        assert np.allclose(mu, ref.mu[lo:hi], atol=1e-6) and np.allclose(eta, ref.eta[lo:hi], atol=1e-6)
    assert np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])   # identical global parameters on both ranks


def _tcp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", STM_RDZV_PORT=str(port))
    sys.path.insert(0, ROOT)
    from strutopy_amd import dist as sdist
    g = sdist.init_from_env(timeout=60)
    out = dict(rank=g.rank, size=g.size)
    out["gather"] = g.allgather(("r", rank))
    out["sum"] = g.allreduce(np.arange(5.0) * (rank + 1))
    out["max"] = g.allreduce(np.array([float(rank), -float(rank)]), op="max")
    out["bcast"] = g.broadcast(b"x" * 128 if rank == 1 else None, src=1)
    g.barrier()
    q.put(out)
    g.barrier()
    g.close()


def test_tcp_group_collectives_three_ranks():
    """strutopy_amd.dist.TcpGroup (stdlib sockets only): what ships the ncclUniqueId and the few host scalars."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tcp_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(3)), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for o in res:
        assert o["size"] == 3 and o["gather"] == [("r", 0), ("r", 1), ("r", 2)]
        assert np.array_equal(o["sum"], np.arange(5.0) * 6) and np.array_equal(o["max"], [2.0, 0.0])
        assert o["bcast"] == b"x" * 128


def test_product_comm_path_is_torch_free():
    """north_star: host code is Python + ctypes, no PyTorch -- importing the package and building the product's
    host group must not import torch (GlooGroup, the CPU test rig, is the only torch user)."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); import strutopy_amd, strutopy_amd.dist as d; "
            "g = d.init_from_env(); c = d.RcclComm(g); assert c.size == 1; "
            "assert 'torch' not in sys.modules, 'torch was imported'" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE")}
    subprocess.run([sys.executable, "-c", code], check=True, env=env)


def _tcp_rank(rank, world, port, secret, q):
    sys.path.insert(0, ROOT)
    from strutopy_amd import dist as sdist
    g = sdist.TcpGroup(rank, world, "127.0.0.1", port=port, timeout=60, secret=secret)
    out = g.allgather((rank, np.arange(3) * rank, {"a": "b" * rank, "n": None}))
    red = g.allreduce(np.full(4, float(rank + 1)))
    q.put((rank, [o[0] for o in out], [o[1].tolist() for o in out], out[1][2], red.tolist()))
    g.close()


def test_tcp_group_survives_garbage_and_rejects_the_wrong_secret():
    """The rendezvous of the host group: nothing is unpickled, a peer must prove the shared secret, and a
    port scan / stale rank / hostile peer on the listening port neither joins nor takes rank 0 down."""
    import multiprocessing as mp
    import struct
    import time
    from strutopy_amd import dist as sdist
    ctx = mp.get_context("spawn")
    port, secret = _free_port(), os.urandom(32)
    q = ctx.Queue()
    p0 = ctx.Process(target=_tcp_rank, args=(0, 2, port, secret, q))
    p0.start()
    time.sleep(1.0)
    for payload in (b"", b"GET / HTTP/1.0\r\n\r\n", os.urandom(64),                       # scan, stray client, noise
                    struct.pack("<Q", 1 << 60) + b"x" * 8,                                     # an absurd frame length
                    sdist._MAGIC + struct.pack("<II", 1, 2) + b"n" * 16 + b"\0" * 32):       # right shape, no secret
        s = socket.create_connection(("127.0.0.1", port), timeout=5)
        s.sendall(payload)
        s.close()
    with pytest.raises(TimeoutError):      # the wrong secret never gets in
        sdist.TcpGroup(1, 2, "127.0.0.1", port=port, timeout=1.5, secret=os.urandom(32))
    p1 = ctx.Process(target=_tcp_rank, args=(1, 2, port, secret, q))
    p1.start()
    got = sorted(q.get(timeout=60) for _ in range(2))
    p0.join(30); p1.join(30)
    assert p0.exitcode == 0 and p1.exitcode == 0
    for r, ranks, arrs, d, red in got:
        assert ranks == [0, 1] and arrs == [[0, 0, 0], [0, 1, 2]] and d == {"a": "b", "n": None} and red == [3.0] * 4


def test_host_group_wire_format_round_trips_without_pickle():
    from strutopy_amd import dist as sdist
    obj = (1, -2.5, True, None, "x", b"\x00\xff", [np.arange(6, dtype=np.int32).reshape(2, 3), np.array(["a", "bc"])],
           {"k": (np.float64(3.0), np.int64(7)), 1: 2, None: "n"})
    back = sdist._decode(sdist._encode(obj))
    assert back[:6] == obj[:6] and np.array_equal(back[6][0], obj[6][0]) and back[6][0].dtype == np.int32
    assert np.array_equal(back[6][1], obj[6][1]) and back[7] == {"k": (3.0, 7), 1: 2, None: "n"}   # keys keep their type
    oa = np.array([["arts", None, 3], ["x", 2.5, True]], dtype=object)      # what pandas gives for string / mixed columns
    ob = sdist._decode(sdist._encode(oa))
    assert ob.dtype == object and ob.shape == oa.shape and ob.tolist() == oa.tolist()
    with pytest.raises(TypeError):
        sdist._encode(object())
    with pytest.raises(TypeError):
        sdist._encode(np.array([object()], dtype=object))         # scalars only inside an object array
    with pytest.raises(TypeError):
        sdist._encode({(1, 2): 3})                                  # no silent str() of keys
    with pytest.raises(ValueError):
        sdist._decode(b"O\x01" + struct_pack_q(1) + b"L" + struct_pack_q(0))   # ... and on the way in
    with pytest.raises(ValueError):
        sdist._decode(b"A\x03\x01|O8" + struct_pack_q(1))          # an object-dtype array header is refused
    with pytest.raises(ValueError):                                # a dict whose key is a list: refused as malformed, not a TypeError
        sdist._decode(b"M" + struct_pack_q(1) + b"L" + struct_pack_q(0) + b"N")
    with pytest.raises(ValueError):                                # an empty object array with a huge sibling dimension
        sdist._decode(b"O\x02" + struct_pack_q(0) + struct_pack_q(1 << 40))
    assert "pickle" not in open(sdist.__file__).read().replace("no pickle", "").replace("unpickled", "")


def struct_pack_q(n):
    import struct
    return struct.pack("<q", n)
