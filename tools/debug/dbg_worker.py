import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, "/root/repo")
from sdpcutsel_via_nn_amd import _capi, networks, synthetic
from sdpcutsel_via_nn_amd.distributed import DeviceOps, ShardedSelector
from oracle import cutsel_oracle as oracle
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
dist.init_process_group("gloo")
n_each, nv = 60000, 60
wl = synthetic.make_workload(nb_vars=nv, k=3, count=n_each * world, seed=11)
lo = rank * n_each
sc = _capi.Scorer(0)
sc.set_network(3, *networks.load_network(3))
sc.set_instance(nv, wl["Q_arr"])
sc.set_candidates(wl["set_inds"][lo:lo + n_each], wl["ks"][lo:lo + n_each], global_base=lo)
sel = ShardedSelector(DeviceOps(sc, dev), n_each)
sc.set_point(wl["vars_values"]); sc.score(3)
eig, obj = sc.get_scores()
pe = [torch.empty(n_each, dtype=torch.float64) for _ in range(world)]; po = [torch.empty(n_each, dtype=torch.float64) for _ in range(world)]
dist.all_gather(pe, torch.from_numpy(eig)); dist.all_gather(po, torch.from_numpy(obj))
E, O = torch.cat(pe).numpy(), torch.cat(po).numpy()
ns = int(((O > 0) & (E < -1e-15)).sum())
sel_size = ns + 40
# local general ranking vs oracle on local data
ids, score, total, new_strat, cnt = sc.rank(4, n_each * world + 1, max_out=sel_size)
o_l, s_l, _, _ = oracle.rank_arrays(4, obj, eig, n_each)
print(rank, "local head equal:", np.array_equal(ids - lo, o_l[:sel_size]), np.array_equal(score, s_l[:sel_size] + 0.0), flush=True)
for st in (1, 2, 4):
    for ss in sorted({37, 5000, ns - 5, ns + 40}):
        r = sel.select(st, ss)
        order, sco, new_strat, cnt = oracle.rank_arrays(st, O, E, ss)
        kk = min(ss, order.shape[0])
        ok = np.array_equal(r["ids"].cpu().numpy(), order[:kk])
        print(rank, "strat", st, "sel", ss, "ok", ok, flush=True)
        if not ok: break
    if not ok: break
sel_size = ss
got = r["ids"].cpu().numpy()
bad = np.nonzero(got != order[:sel_size])[0]
print(rank, "ns", ns, "mismatches", bad.size, "first", bad[:5], flush=True)
if bad.size:
    b = bad[0]
    for p in range(max(0, b - 2), b + 4):
        print(rank, p, "got", got[p], r["scores"][p].item(), O[got[p]], E[got[p]], "| ref", order[p], sco[p], O[order[p]], E[order[p]], flush=True)
ops = sel.ops
s_, i_, q_, tot, c_ = ops.local_head(4, n_each * world + 1, 5000, True)
S = sel._all_gather(s_); I = sel._all_gather(i_); Q = sel._all_gather(q_)
Sn, In, Qn = S.cpu().numpy(), I.cpu().numpy(), Q.cpu().numpy()
for t in (16602, 67757, 98829):
    w = np.nonzero(In == t)[0]
    print(rank, "id", t, "pos", w, [float(Sn[j]).hex() for j in w], [float(Qn[j]).hex() for j in w], flush=True)
ms, mi = ops.merge(S, I, 5000, Q)
mi = mi.cpu().numpy()
print(rank, "merged positions", [int(np.nonzero(mi == t)[0][0]) for t in (16602, 67757, 98829)], flush=True)
ref = np.lexsort((In, -Qn, -Sn))[:5000]
print(rank, "merge equals lexsort:", np.array_equal(mi, In[ref]), flush=True)
sc.close(); dist.destroy_process_group()
