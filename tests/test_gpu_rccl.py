"""The N > 1 code path against REAL RCCL on the one-GPU test box: a fresh child
process initialises the `nccl` process group with world size 1, binds the library to torch's stream
and runs the sharded selection -- packed head record, `all_gather_into_tensor` on the library's
stream, replicated merge, rows of the own candidates, fp64 MAX all-reduce -- with
SDPCUT_FORCE_COLLECTIVES=1 so that the collectives are issued although nobody else is there.
What it cannot show is the xGMI latency of an 8-rank gather; the call path, dtypes and stream
ordering are exactly those of the multi-GPU run."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from sdpcutsel_via_nn_amd import _capi, networks, synthetic
from sdpcutsel_via_nn_amd.distributed import DeviceOps, ShardedSelector
from oracle import cutsel_oracle as oracle

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
dist.barrier()
n_local, nv = 200000, 80
wl = synthetic.make_workload(nb_vars=nv, k=3, count=n_local, seed=31)
sc = _capi.Scorer(0)
sc.set_network(3, *networks.load_network(3))
sc.set_instance(nv, wl["Q_arr"])
sc.set_candidates(wl["set_inds"], wl["ks"], global_base=7 * n_local)        # a shard in the middle of the id space
sel = ShardedSelector(DeviceOps(sc, dev), n_local)
assert not sel._solo and sel.n_global == n_local
x = np.full(nv, 0.5); iu = np.triu_indices(nv); rng = np.random.default_rng(2)
points = [wl["vars_values"], np.concatenate([np.where(rng.uniform(size=iu[0].shape[0]) < 0.5, 0.25, 0.5), x])]
calls = 0
for vv in points:
    sc.set_point(vv)
    sc.score(_capi.EIG | _capi.NN)
    E, O = sc.get_scores()
    n_strong = int(((O > 0) & (E < -1e-15)).sum())
    for strat in (1, 2, 4):
        for sel_size in sorted({37, 5000, max(n_strong - 5, 1), n_strong + 40}):
            order, score, new_strat, cnt = oracle.rank_arrays(strat, O, E, sel_size)
            k = min(sel_size, order.shape[0])
            r = sel.select(strat, sel_size)
            assert np.array_equal(r["ids"].cpu().numpy(), order[:k] + 7 * n_local), (strat, sel_size)
            assert np.array_equal(r["scores"].cpu().numpy(), score[:k] + 0.0) and r["new_strat"] == new_strat
            q = sel.select_round(strat, sel_size)
            assert np.array_equal(q["ids"], order[:k] + 7 * n_local), (strat, sel_size)
            assert np.array_equal(q["scores"], score[:k] + 0.0) and q["new_strat"] == new_strat
            assert q["mine"].all()
            lam, coef, rhs, _, ks = sc.cut_rows(order[:k])
            assert np.array_equal(q["lam"], lam) and np.array_equal(q["coef"], coef[:, :9]) and np.array_equal(q["rhs"], rhs)
            if strat == 4:
                assert q["counters"]["strong"] == cnt["strong"] and q["counters"]["violated"] == cnt["violated"]
            calls += 1
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)            # the reduction bench.py ends with
assert float(t.item()) == 1.25
sc.close()
dist.destroy_process_group()
print("rccl world-1 ok:", calls, "selections")
'''


def test_sharded_selection_on_real_rccl_world1(oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29651", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               SDPCUT_FORCE_COLLECTIVES="1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-3000:]
    assert "rccl world-1 ok" in text
