"""CPU check of the run-time translation unit of user-supplied simulators: the very text abcdez_ctx_create_user hands to hiprtc
(abcdez_user_translation_unit: a function of the model and the source alone, no device) compiles with hipcc for gfx950 against the
library's own headers -- for the three forms of include/abcdez_hip.h: one thread per row (abz_user_dist), the cooperative form on
rows of 17 .. 64 parameters (abz_user_dist_lanes), the staged form (abz_user_round).  The GPU tests
(tests/test_gpu_user_simulators.py) run what is compiled here.  Replaces dist!(theta, ve) of src/abcdez_smc.jl:137."""
import ctypes as C
import math
import os
import shutil
import subprocess

import pytest

import abcdez_amd as A
from abcdez_amd import _lib

from user_sources import USER_MVN16, USER_SEQ16, USER_SEQ16_ROUNDS, USER_LV, USER_LV_ROUNDS, USER_MVN_LANES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")


def translation_unit(prior, sim):
    spec = A.ModelSpec(prior, sim, seed=1)
    import numpy as np
    data = np.ascontiguousarray(spec.data, dtype=np.float64)
    cm = spec.cstruct(data.ctypes.data if data.size else None)
    lib = _lib.load()
    tu, opts = C.create_string_buffer(1 << 20), C.create_string_buffer(4096)
    source = getattr(sim, "source", None)          # None: a built-in simulator (kernels compiled at run time for wrapper priors)
    keep = spec.ext                                 # (cm.ext points into it)
    n = lib.abcdez_user_translation_unit(C.byref(cm), source.encode() if source is not None else None, tu, len(tu), opts, len(opts))
    assert n > 0, lib.abcdez_last_error()
    return tu.value.decode(), opts.value.decode().split()


CASES = {
    "mvn32_lanes": (A.Factored(*[A.Normal(0, 1)] * 32), A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=(1.0,) * 32),
                    {"-DABZ_USER_L=4", "-DABZ_USER_C=8", "-DABZ_USER_PLAIN=1"}, "smc_swarm_packed_body<ABZ_JIT_SIM, ABZ_USER_L"),
    "mvn20_lanes_padded": (A.Factored(*([A.Normal(0, 1)] * 19 + [A.Gamma(2.0, 1.0)])), A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=(1.0,) * 20),
                           {"-DABZ_USER_L=4", "-DABZ_USER_C=8", "-DABZ_USER_PLAIN=0"}, None),
    "mvn64_lanes": (A.Factored(*[A.Normal(0, 1)] * 64), A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=(1.0,) * 64),
                    {"-DABZ_USER_L=8", "-DABZ_USER_C=8", "-DABZ_USER_PLAIN=1"}, None),
    "mvn200_lanes": (A.Factored(*[A.Normal(0, 1)] * 200), A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=(1.0,) * 200),
                     {"-DABZ_USER_L=8", "-DABZ_USER_C=32", "-DABZ_USER_PLAIN=0"}, None),
    "mvn128_lanes": (A.Factored(*[A.Normal(0, 1)] * 128), A.UserSimulator(USER_MVN_LANES, params=(1.0,), data=(1.0,) * 128),
                     {"-DABZ_USER_L=8", "-DABZ_USER_C=16", "-DABZ_USER_PLAIN=1"}, None),
    "lv_rounds": (A.Factored(*[A.Uniform(0.0, 2.0)] * 4),
                  A.UserSimulator(USER_LV_ROUNDS % {"rounds": 8}, params=(1.0, 0.5, 0.01, 100.0, 0.1), data=(1.0, 0.5) * 16),
                  {"-DABZ_USER_L=1", "-DABZ_USER_C=4", "-DABZ_USER_PLAIN=0"}, "smc_user_rounds_phase2_body"),
    "lv_opaque": (A.Factored(*[A.Uniform(0.0, 2.0)] * 4), A.UserSimulator(USER_LV, params=(1.0, 0.5, 0.01, 100.0, 0.1), data=(1.0, 0.5) * 16),
                  {"-DABZ_USER_L=1", "-DABZ_USER_C=4", "-DABZ_USER_PLAIN=0"}, "smc_split_phase2_body"),
    # 9 to 16 parameters: one lane per particle on rows of 16 doubles, the sweep in two launches, the second one 128 threads wide
    "mvn12_opaque": (A.Factored(*[A.Normal(0, 1)] * 12), A.UserSimulator(USER_MVN16, params=(1.0,), data=(1.0,) * 12),
                     {"-DABZ_USER_L=1", "-DABZ_USER_C=16", "-DABZ_USER_P2_BLOCK=128"}, "smc_split_phase2_body"),
    "seq12_rounds": (A.Factored(*([A.Normal(0, 1)] * 11 + [A.Gamma(2.0, 1.0)])),
                     A.UserSimulator(USER_SEQ16_ROUNDS % {"rounds": 4}, params=(1.0,), data=(1.0,) * 12),
                     {"-DABZ_USER_L=1", "-DABZ_USER_C=16", "-DABZ_USER_PLAIN=0", "-DABZ_USER_P2_BLOCK=128"}, "smc_user_rounds_phase2_body"),
    # BUILT-IN simulators whose model has prior factors of the wrapper families (truncated(...), MixtureModel): the statically compiled
    # sweeps do not carry those log-densities (include/abcdez_spec.h, ABZ_PRIOR_WRAP), so the library compiles this model's sweep,
    # replay and abcdemc kernels at run time with -DABZ_PRIOR_WRAP=1
    "normal1d_wrapped_prior": (A.truncated(A.Gamma(2.0, 1.5), 1.0, 6.0), A.Normal1D(3.0),
                               {"-DABZ_USER_L=1", "-DABZ_USER_C=1", "-DABZ_JIT_SIM=0", "-DABZ_PRIOR_WRAP=1"}, "smc_replay_packed_body"),
    "mvn8_wrapped_prior": (A.Factored(*([A.Normal(0, 1)] * 6 + [A.MixtureModel([A.Normal(-1, 0.5), A.Laplace(1.0, 2.0)], [0.3, 0.7]),
                                                                 A.truncated(A.Cauchy(0.0, 1.0), -2.0, 3.0)])), A.MVNormal((1.0,) * 8),
                           {"-DABZ_USER_L=1", "-DABZ_USER_C=8", "-DABZ_JIT_SIM=1", "-DABZ_PRIOR_WRAP=1"}, None),
    "mvn32_wrapped_prior": (A.Factored(*([A.Normal(0, 1)] * 31 + [A.truncated(A.Gamma(2.0, 1.0), 0.2, 5.0)])), A.MVNormal((1.0,) * 32),
                            {"-DABZ_USER_L=4", "-DABZ_USER_C=8", "-DABZ_JIT_SIM=1", "-DABZ_PRIOR_WRAP=1"}, None),
    "lv_wrapped_prior": (A.Factored(A.truncated(A.Gamma(2.0, 0.5), 0.0, 2.0), A.Uniform(0.0, 2.0), A.Uniform(0.0, 2.0),
                                    A.truncated(A.LogNormal(-1.0, 1.0), 0.0, 2.0)),
                         A.LotkaVolterraRK4((1.0, 0.5) * 16), {"-DABZ_USER_L=1", "-DABZ_USER_C=4", "-DABZ_JIT_SIM=7", "-DABZ_PRIOR_WRAP=1"},
                         "smc_lv_phase2_body"),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_translation_unit_of_every_form_compiles_for_gfx950(name, tmp_path):
    prior, sim, want_opts, want_text = CASES[name]
    tu, opts = translation_unit(prior, sim)
    assert want_opts <= set(opts), opts
    if want_text:
        assert want_text in tu
    if hasattr(sim, "source"):
        assert '#include "abz_user_rounds.h"' in tu and tu.index(sim.source.strip()[:40]) < tu.index('#include "abz_user_rounds.h"')
    else:
        assert "abz_user_init" not in tu                 # the initial population of a built-in simulator stays with the static kernel
    if HIPCC is None:
        pytest.skip("no hipcc here")
    src = tmp_path / "abz_user.hip"
    src.write_text(tu)
    cmd = [HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", *opts,
           "-I", os.path.join(ROOT, "abcdez.jl_amd", "csrc"), "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "u.o")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]


def test_wide_rows_with_blobs_are_refused_by_name():
    lib = _lib.load()
    sim = A.UserSimulator(USER_MVN_LANES + "\n/* abz_user_blob */", params=(1.0,), data=(1.0,) * 32, n_blob=2)
    spec = A.ModelSpec(A.Factored(*[A.Normal(0, 1)] * 32), sim, seed=1)
    import numpy as np
    data = np.ascontiguousarray(spec.data, dtype=np.float64)
    cm = spec.cstruct(data.ctypes.data)
    tu, opts = C.create_string_buffer(1 << 20), C.create_string_buffer(4096)
    assert lib.abcdez_user_translation_unit(C.byref(cm), sim.source.encode(), tu, len(tu), opts, len(opts)) < 0
    assert b"blobs need the whole row in one thread" in lib.abcdez_last_error()
