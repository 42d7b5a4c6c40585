"""Two ranks sharing the one GPU of the test box (gloo rendezvous, host-staged all-gather):
the sharded selection must reproduce the single-list ranking head exactly."""
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

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
n_each, nv = 60000, 60
wl = synthetic.make_workload(nb_vars=nv, k=3, count=n_each * world, seed=11)
lo = rank * n_each
sc = _capi.Scorer(0)
sc.set_network(3, *networks.load_network(3))
sc.set_instance(nv, wl["Q_arr"])
sc.set_candidates(wl["set_inds"][lo:lo + n_each], wl["ks"][lo:lo + n_each], global_base=lo)
points = [wl["vars_values"]]
x = np.full(nv, 0.5); iu = np.triu_indices(nv)
rng = np.random.default_rng(2)
points.append(np.concatenate([np.where(rng.uniform(size=iu[0].shape[0]) < 0.5, 0.25, 0.5), x]))   # few strong candidates
sel = ShardedSelector(DeviceOps(sc, dev), n_each)
for vv in points:
    sc.set_point(vv)
    sc.score(_capi.EIG | _capi.NN)
    eig, obj = sc.get_scores()
    parts_e = [torch.empty(n_each, dtype=torch.float64) for _ in range(world)]
    parts_o = [torch.empty(n_each, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(parts_e, torch.from_numpy(eig))
    dist.all_gather(parts_o, torch.from_numpy(obj))
    E, O = torch.cat(parts_e).numpy(), torch.cat(parts_o).numpy()
    n_strong = int(((O > 0) & (E < -1e-15)).sum())
    for strat in (1, 2, 4):
        for sel_size in sorted({37, 5000, max(n_strong - 5, 1), n_strong + 40}):
            r = sel.select(strat, sel_size)
            order, score, new_strat, cnt = oracle.rank_arrays(strat, O, E, sel_size)
            k = min(sel_size, order.shape[0])
            assert np.array_equal(r["ids"].cpu().numpy(), order[:k]), (strat, sel_size, rank)
            assert np.array_equal(r["scores"].cpu().numpy(), score[:k] + 0.0), (strat, sel_size)
            assert r["new_strat"] == new_strat
            if strat == 4:
                assert r["counters"]["strong"] == cnt["strong"] and r["counters"]["violated"] == cnt["violated"]
sc.close()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_ranks_one_gpu_sharded_selection(oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29643", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
