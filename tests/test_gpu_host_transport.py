"""The host-supplied transport of the library's collectives (abcdez_comm_init_host, csrc/abz_comm.hip) in ONE process: the
callbacks play the other ranks, so every offset the library computes -- where this rank's piece is staged, which ranges come
back to the device, how the words of an all-reduce are laid out -- is checked against numpy without a second process.  The
multi-process runs (worlds 2 / 3 / 4 sharing the GPU, bit-identical to the single-process oracle) are in
tests/test_distributed_gloo.py.  Replaces what the threads of `@floop ex` share through memory (src/abcdez_smc.jl:110)."""
import ctypes as C

import numpy as np
import pytest
import torch

import abcdez_amd as A
from abcdez_amd import _lib
from abcdez_amd.engine import HipOps

pytestmark = pytest.mark.gpu


def make_ops():
    spec = A.ModelSpec(A.Normal(0.0, 1.0), A.Normal1D(0.5), seed=3)
    return HipOps(spec)


def host_view(address, nbytes):
    return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(address))


@pytest.mark.parametrize("world,rank", [(1, 0), (2, 0), (2, 1), (3, 1), (4, 3)])
@pytest.mark.parametrize("piece", [1, 64, 4096 + 24])
def test_allgather_stages_own_piece_and_returns_the_others(world, rank, piece):
    ops = make_ops()
    rng = np.random.default_rng(world * 100 + rank * 10 + piece)
    full = rng.integers(0, 256, size=world * piece, dtype=np.uint8)       # what the ranks hold together
    seen = {}

    def allgather(address, piece_bytes):
        v = host_view(address, world * piece_bytes)
        seen["own"] = v[rank * piece_bytes:(rank + 1) * piece_bytes].copy()
        seen["piece"] = piece_bytes
        for r in range(world):
            if r != rank:
                v[r * piece_bytes:(r + 1) * piece_bytes] = full[r * piece_bytes:(r + 1) * piece_bytes]

    ops.comm_init_host(rank, world, allgather)
    assert ops.comm_kind() == 2 and ops.comm_rank() == (rank, world, True)
    dev = torch.full((world * piece + 8,), 0xEE, dtype=torch.uint8, device="cuda")
    dev[rank * piece:(rank + 1) * piece] = torch.from_numpy(full[rank * piece:(rank + 1) * piece]).cuda()
    ops.use_current_stream()
    ops.comm_allgather(dev, piece)
    torch.cuda.synchronize()
    assert seen["piece"] == piece and np.array_equal(seen["own"], full[rank * piece:(rank + 1) * piece])
    got = dev.cpu().numpy()
    assert np.array_equal(got[:world * piece], full) and (got[world * piece:] == 0xEE).all()
    ops.comm_destroy()
    assert ops.comm_kind() == 0 and ops.comm_rank() == (0, 1, False)


CASES = [
    (torch.int64, 0, "sum"), (torch.int64, 0, "min"), (torch.int64, 0, "max"),
    (torch.float64, 1, "sum"), (torch.float64, 1, "min"), (torch.float64, 1, "max"),
    (torch.uint64, 2, "sum"), (torch.uint64, 2, "min"), (torch.uint64, 2, "max"),
]


@pytest.mark.parametrize("with_cb", [False, True])
@pytest.mark.parametrize("tdt,code,op", CASES)
def test_allreduce_by_rank_order_and_by_callback(tdt, code, op, with_cb):
    """without an all-reduce callback the library all-gathers the words and reduces them in rank order (unsigned order for
    uint64: values with the top bit set -- the complemented keys of abcdemc's exchange -- must not compare as negative)"""
    world, rank, n = 4, 2, 5
    ops = make_ops()
    npdt = {torch.int64: np.int64, torch.float64: np.float64, torch.uint64: np.uint64}[tdt]
    rng = np.random.default_rng(7 + code)
    if npdt is np.float64:
        rows = rng.standard_normal((world, n))
    elif npdt is np.int64:
        rows = rng.integers(-2**40, 2**40, size=(world, n), dtype=np.int64)
    else:
        rows = rng.integers(0, 2**64 - 1, size=(world, n), dtype=np.uint64)
        rows[1, 0] = np.uint64(2**64 - 3); rows[0, 0] = np.uint64(5)       # top bit set against a small value
    red = {"sum": lambda a: a[0] + a[1] + a[2] + a[3], "min": lambda a: a.min(axis=0), "max": lambda a: a.max(axis=0)}[op]
    with np.errstate(over="ignore"):
        want = red(rows)
    calls = []

    def allgather(address, piece_bytes):
        calls.append(("ag", piece_bytes))
        v = host_view(address, world * piece_bytes).view(npdt).reshape(world, -1)
        assert np.array_equal(v[rank], rows[rank])
        for r in range(world):
            if r != rank:
                v[r] = rows[r]

    def allreduce(address, nn, dtype, opc):
        calls.append(("ar", nn, dtype, opc))
        v = host_view(address, 8 * nn).view(npdt)
        assert np.array_equal(v, rows[rank])
        v[:] = want

    ops.comm_init_host(rank, world, allgather, allreduce if with_cb else None)
    t = torch.from_numpy(rows[rank].copy()).cuda()
    ops.use_current_stream()
    ops.comm_allreduce(t, op)
    torch.cuda.synchronize()
    assert np.array_equal(t.cpu().numpy(), want)
    assert calls == ([("ar", n, code, {"sum": 0, "min": 1, "max": 2}[op])] if with_cb else [("ag", 8 * n)])


def test_failing_callback_breaks_the_communicator_until_it_is_made_again():
    ops = make_ops()
    lib = ops.lib
    state = {"fail": True}

    def allgather(address, piece_bytes):
        if state["fail"]:
            raise RuntimeError("the transport lost a peer")

    dev = torch.zeros(256, dtype=torch.uint8, device="cuda")
    # misuse
    assert lib.abcdez_comm_init_host(ops.ctx, 0, 2, None, None, None) != 0 and b"required" in lib.abcdez_last_error()
    assert lib.abcdez_comm_allgather(ops.ctx, dev.data_ptr(), 64) != 0 and b"no communicator" in lib.abcdez_last_error()
    cb = _lib.HOST_ALLGATHER_FN(lambda user, buf, n: 0)
    assert lib.abcdez_comm_init_host(ops.ctx, 2, 2, C.cast(cb, C.c_void_p), None, None) != 0 and b"rank < world" in lib.abcdez_last_error()
    ops.comm_init_host(1, 2, allgather)
    assert lib.abcdez_comm_init_host(ops.ctx, 0, 2, C.cast(cb, C.c_void_p), None, None) != 0 and b"already has" in lib.abcdez_last_error()
    # a failing exchange: status, message, and the context refuses further collectives (its peers cannot be told otherwise)
    with pytest.raises(_lib.AbcdezError, match="callback returned 1"):
        ops.comm_allgather(dev, 64)
    state["fail"] = False
    with pytest.raises(_lib.AbcdezError, match="aborted"):
        ops.comm_allgather(dev, 64)
    with pytest.raises(_lib.AbcdezError, match="aborted"):
        ops.comm_allreduce(torch.zeros(2, dtype=torch.int64, device="cuda"), "sum")
    ops.comm_destroy()
    ops.comm_init_host(1, 2, allgather)
    ops.comm_allgather(dev, 64)
    torch.cuda.synchronize()


def test_library_loads_without_rccl_and_opens_it_on_demand():
    """librccl is not a load-time dependency of libabcdez_hip.so: it is opened by the first RCCL call.  A host that
    points ABCDEZ_RCCL_LIB nowhere and hides the default names cannot be emulated in-process once RCCL is loaded, so this checks
    the two observable halves: no DT_NEEDED entry, and the lazy path works on this box."""
    import subprocess

    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in out, out
    ops = make_ops()
    uid = ops.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ops.comm_init(uid, 0, 1)
    assert ops.comm_kind() == 1
    t = torch.arange(4, dtype=torch.int64, device="cuda")
    ops.use_current_stream()
    ops.comm_allreduce(t, "sum")
    torch.cuda.synchronize()
    assert t.tolist() == [0, 1, 2, 3]
    ops.comm_destroy()
