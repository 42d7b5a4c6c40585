import sys, numpy as np
sys.path.insert(0, "/root/repo")
import sdpcutsel_via_nn_amd as pkg
from sdpcutsel_via_nn_amd import harness, _capi
from oracle import cutsel_oracle as oracle
inst = harness.parse_osil("tests/golden/instances/q_20_4_25_1.osil")
n, L = inst["nb_vars"], inst["nb_lifted"]
lp = harness.LinearRelaxation(np.concatenate([inst["Q_arr"], inst["c"]]))
lp.linear_constraints.add(inst["rows"], inst["rhs"], inst["senses"])
lp.linear_constraints.add(*harness.mccormick_rows(n, inst["adj"]))
lp.solve()
vv = np.array(lp.get_values())
print("x =", np.round(vv[L:], 3))
(So, ko), (Sc, kc) = harness.qcqp_covers(inst, 3, _capi.enumerate_cover)
for name, S, ks in (("obj", So, ko), ("cons", Sc, kc)):
    lam = []
    for i in range(ks.shape[0]):
        k = int(ks[i]); s = S[i, :k]
        pos = oracle.triu_positions(s, n)
        lam.append(oracle.get_eigendecomp(k, vv[L:][s], vv[:L][pos], False)[0])
    lam = np.array(lam)
    print(name, "N", lam.size, "violated", int((lam < -1e-15).sum()), "min lam", lam.min())
