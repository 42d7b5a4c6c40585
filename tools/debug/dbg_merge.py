import sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from sdpcutsel_via_nn_amd import _capi
sc = _capi.Scorer(0)
rng = np.random.default_rng(0)
for n in (100, 1000, 2048, 4096, 5000, 8192, 10000, 16384, 20000, 40000):
    for use_sec in (False, True):
        scores = rng.integers(0, 7, n).astype(np.float64)
        sec = rng.integers(0, 3, n).astype(np.float64)
        ids = rng.permutation(10 ** 6)[:n].astype(np.int64)
        ds, dq, di = (torch.from_numpy(a).cuda() for a in (scores, sec, ids))
        os_ = torch.empty(n, dtype=torch.float64, device="cuda"); oi = torch.empty(n, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        sc.merge_topk_device(n, ds.data_ptr(), di.data_ptr(), n, os_.data_ptr(), oi.data_ptr(), dq.data_ptr() if use_sec else None)
        sc.synchronize()
        ref = np.lexsort((ids, -sec, -scores)) if use_sec else np.lexsort((ids, -scores))
        print(n, use_sec, np.array_equal(oi.cpu().numpy(), ids[ref]), flush=True)
