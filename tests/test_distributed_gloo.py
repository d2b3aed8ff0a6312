"""N > 1 path on CPU: gloo jobs of 2, 3, 4 and 8 ranks through the same PopulationEngine the GPU path uses
(packed population: every rank sweeps a chunk of the alive prefix, accept-flag all-gather + replay on the
replicas, distances once per generation; double-buffered storage (abcdemc, legacy abcdesmc): in-place all-gather
of the new rows / logπ / Δ after every sweep; integer counter all-reduce), with the oracle as compute backend.
The RNG is keyed by the global position, so any world size -- powers of two or not -- must reproduce the
single-process run bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(world, outdir, mode="oracle", timeout=900):
    env = dict(os.environ, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py"), str(outdir), mode]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]


@pytest.fixture(scope="module")
def runs(tmp_path_factory):
    out = {}
    for world in (1, 2, 3, 4, 8):
        d = tmp_path_factory.mktemp(f"world{world}")
        run_world(world, d)
        out[world] = d
    return out


@pytest.mark.parametrize("name", ["normal1d", "mvn8", "quad2d", "mvn32", "lv", "further5", "wrapped4", "user_mvn20"])
@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_sharded_run_equals_single_process(runs, name, world):
    ref = np.load(os.path.join(runs[1], f"result_{name}_rank0.npz"))
    for rank in range(world):
        got = np.load(os.path.join(runs[world], f"result_{name}_rank{rank}.npz"))
        assert int(got["world"]) == world
        for k in ("theta", "C", "Wns", "logpi", "eps_hist", "mc_theta", "mc_C"):
            assert np.array_equal(ref[k], got[k], equal_nan=True), (name, world, rank, k)
        if "blobs" in ref.files:        # blob stamps: rebuilt by the replay (abcdesmc) / all-gathered with the rows (abcdemc)
            assert ref["blobs"].shape[0] == ref["C"].shape[0]
            for k in ("blobs", "mc_blobs"):
                assert np.array_equal(ref[k], got[k]), (name, world, rank, k)
        assert float(ref["logZ"]) == float(got["logZ"])
        assert int(ref["nsims"]) == int(got["nsims"]) and int(ref["iters"]) == int(got["iters"])
        assert int(ref["mc_nsims"]) == int(got["mc_nsims"])


def test_uneven_shard_is_rejected(oracle):
    import abcdez_amd as A
    from abcdez_amd.engine import PopulationEngine

    class FakePG:
        pass

    spec = A.ModelSpec(A.Normal(0, 1), A.Normal1D(0.0))
    eng = PopulationEngine(spec, 10, None, ops=oracle.OracleOps(spec))
    assert (eng.lo, eng.hi, eng.world) == (0, 10, 1)


NAMES = ("normal1d", "mvn8", "quad2d", "mvn32", "lv", "further5", "wrapped4", "user_mvn20")


def compare_with_single_process_oracle(ref_dir, hip_dir, world, names=NAMES):
    for name in names:
        ref = np.load(os.path.join(ref_dir, f"result_{name}_rank0.npz"))
        for rank in range(world):
            got = np.load(os.path.join(hip_dir, f"result_{name}_rank{rank}.npz"))
            assert int(got["world"]) == world
            for k in ("theta", "C", "Wns", "logpi", "eps_hist", "mc_theta", "mc_C") + \
                    (("blobs", "mc_blobs") if "blobs" in ref.files else ()):
                assert np.array_equal(ref[k], got[k], equal_nan=True), (name, rank, k)
            assert float(ref["logZ"]) == float(got["logZ"]) and int(ref["nsims"]) == int(got["nsims"])
            assert int(ref["iters"]) == int(got["iters"]) and int(ref["mc_nsims"]) == int(got["mc_nsims"])


@pytest.fixture(scope="module")
def oracle_ref(tmp_path_factory):
    d = tmp_path_factory.mktemp("ref_oracle")
    run_world(1, d, "oracle")
    return d


@pytest.mark.gpu
@pytest.mark.parametrize("world,mode", [(2, "hip"), (4, "hip"), (2, "hip_ar"), (3, "hip")])
def test_library_sharded_entry_points_with_world_gt_1_on_one_gpu(tmp_path_factory, oracle_ref, world, mode):
    """The LIBRARY's multi-rank code with world > 1: 2 / 3 / 4 processes share the single GPU of the test box, each with a context of
    its own, and every exchange is issued by libabcdez_hip.so itself -- abcdez_smc_sweeps_sharded (chunk sweep, flag all-gather at
    flags + rank chunk, replay, device-side test of smc:352, distance all-gather), abcdez_mc_generation_sharded_async (rows at
    ntheta + rank n_local ld, log-priors, distances, stamps, the seven exchange words) and abcdez_comm_allgather for the initial
    population -- over the host transport (abcdez_comm_init_host) with gloo's all-gather underneath ("hip_ar": gloo's all-reduce
    too instead of the library's rank-order reduction).  Every rank must reproduce the single-process CPU-oracle run bit for bit:
    the same assertions as test_sharded_run_equals_single_process.  Replaces `@floop ex` of src/abcdez_smc.jl:110,237 and
    src/abcdez_mc.jl:7,112."""
    hip_dir = tmp_path_factory.mktemp(f"hip_world{world}_{mode}")
    run_world(world, hip_dir, mode, timeout=420)
    compare_with_single_process_oracle(oracle_ref, hip_dir, world)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rccl1", "hip1"])
def test_sharded_code_path_in_a_group_of_one_rank(tmp_path_factory, oracle_ref, mode):
    """The sharded code path (flag all-gather + replay, per-generation distance all-gather, abcdemc's exchange) in a group of ONE
    rank with the collectives forced on: rccl1 -- over the real RCCL backend (all a single-GPU box can run of it; librccl is
    opened lazily by abcdez_comm_unique_id); hip1 -- over the host transport.  Bit-identical to the single-process CPU oracle."""
    hip_dir = tmp_path_factory.mktemp("hip_" + mode)
    run_world(1, hip_dir, mode, timeout=300)
    compare_with_single_process_oracle(oracle_ref, hip_dir, 1)
