import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Builds oracle/libnnoracle.so if needed."""
    so = os.path.join(ROOT, "oracle", "libnnoracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    from oracle import cutsel_oracle
    return cutsel_oracle


@pytest.fixture(scope="session")
def golden_boxqp():
    return np.load(os.path.join(GOLDEN, "inst_boxqp.npz"))


@pytest.fixture(scope="session")
def golden_qcqp():
    return np.load(os.path.join(GOLDEN, "inst_qcqp.npz"))


def golden_nn(k):
    return np.load(os.path.join(GOLDEN, "nn_k%d.npz" % k))


def agg_from_arrays(oracle_mod, set_inds, ks, nb_vars, Q_arr):
    sets = [[int(v) for v in set_inds[i, :ks[i]]] for i in range(set_inds.shape[0])]
    return oracle_mod.build_agg_list(sets, nb_vars, list(Q_arr))


BOXQP_TAGS = ["spar020_100_1_d3", "spar020_100_1_d4", "spar040_030_1_d5", "spar030_060_1_d3"]
POINTS = ["mck", "rnd", "psd"]
