"""Round 4, CPU side (no GPU needed)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_single_gpu_product_imports_and_binds_without_torch():
    """the reference needs numpy + ctypes (cut_select_qp.py:1-14).  With torch masked from the import system the
    package imports, the library loads against the system ROCm and a handle request reaches the device probe (SDPCUT_ENODEVICE
    in the build container, a live handle on a GPU box); torch is never imported on the way."""
    code = r'''
import sys
class _NoTorch(object):
    def find_spec(self, name, path=None, target=None):
        if name == "torch" or name.startswith("torch."):
            raise ImportError("torch is masked for this test")
sys.meta_path.insert(0, _NoTorch())
sys.path.insert(0, %r)
import sdpcutsel_via_nn_amd as pkg
from sdpcutsel_via_nn_amd import _capi, cut_solver, harness, networks, synthetic
lib = pkg.load_library()
assert lib.sdpcut_version() >= 100
try:
    sc = pkg.Scorer(0)
    sc.close()
    print("handle ok")
except pkg.SdpCutError as e:
    assert "(-2)" in str(e) and "no CPU fallback" in str(e), str(e)
    print("no device:", e)
assert "torch" not in sys.modules, "torch was imported"
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    assert b"no device:" in out.stdout or b"handle ok" in out.stdout


def _libm_exp_twin(x, T):
    """Python twin of csrc/libm_exp.h (glibc's exp, the -mfma variant): exact fused multiply-adds through rational arithmetic"""
    import struct
    from fractions import Fraction

    def fma(a, b, c):
        return float(Fraction(a) * Fraction(b) + Fraction(c))

    def bits(v):
        return struct.unpack("<Q", struct.pack("<d", v))[0]

    def dbl(u):
        return struct.unpack("<d", struct.pack("<Q", u & 0xFFFFFFFFFFFFFFFF))[0]
    InvLn2N, Shift = float.fromhex("0x1.71547652b82fep0") * 128.0, float.fromhex("0x1.8p52")
    hi, lo = float.fromhex("-0x1.62e42fefa0000p-8"), float.fromhex("-0x1.cf79abc9e3b3ap-47")
    C2, C3, C4, C5 = (float.fromhex(h) for h in ("0x1.ffffffffffdbdp-2", "0x1.555555555543cp-3", "0x1.55555cf172b91p-5", "0x1.1111167a4d017p-7"))
    kd = InvLn2N * x + Shift
    ki = bits(kd)
    kd -= Shift
    r = fma(kd, lo, fma(kd, hi, x))
    idx = 2 * (ki % 128)
    tail, sbits = dbl(T[idx]), T[idx + 1] + (ki << 45)
    r2 = r * r
    tmp = fma(r2 * r2, fma(r, C5, C4), fma(r2, fma(r, C3, C2), tail + r))
    scale = dbl(sbits)
    return fma(scale, tmp, scale)


def test_libm_exp_port_reproduces_the_host_libm_bit_for_bit():
    """NNs.so's outputs are a function of the host libm's exp.  The port's table (recomputed by build.py) and
    operation sequence give math.exp's bits on the range a tansig of the shipped networks can see (|2 n| <= 62.6) and beyond."""
    import math
    import platform
    import random
    from sdpcutsel_via_nn_amd import build
    # the host assumption of csrc/libm_exp.h: glibc >= 2.28 (ARM's exp) on an x86-64 CPU with FMA (the ifunc's -mfma variant)
    libc, ver = platform.libc_ver()
    try:
        fma_cpu = " fma " in (" " + open("/proc/cpuinfo").read().split("flags", 1)[1].split("\n", 1)[0] + " ")
    except (OSError, IndexError):
        fma_cpu = False
    if not (libc == "glibc" and tuple(int(v) for v in ver.split(".")[:2]) >= (2, 28) and platform.machine() == "x86_64" and fma_cpu):
        pytest.skip("host libm is not glibc >= 2.28 / x86-64 with FMA (%s %s, %s, fma %s): NNs.so itself rounds differently here, "
                    "the bit-identity claim of csrc/libm_exp.h does not apply" % (libc, ver, platform.machine(), fma_cpu))
    T = build.libm_exp_table()
    assert len(T) == 256 and T[0] == 0 and T[1] == 0x3FF0000000000000
    random.seed(5)
    xs = [random.uniform(-63.0, 63.0) for _ in range(6000)] + [random.uniform(-1.0, 1.0) * 10.0 ** random.uniform(-12, 0) for _ in range(2000)]
    xs += [0.0, -0.0, 1.0, -1.0, 62.6, -62.6, 1e-16, 511.9, -511.9]
    bad = [x for x in xs if _libm_exp_twin(x, T) != math.exp(x)]
    assert not bad, bad[:5]


def _proto():
    tools = os.path.join(ROOT, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    import lmin_proto
    return lmin_proto


@pytest.mark.parametrize("k", [2, 3, 4, 5])
def test_lambda_min_solver_numerics_against_lapack(oracle, k):
    """The algorithm of csrc/lmin.h (Householder + block Gershgorin start + Laguerre; numpy twin tools/lmin_proto.py, same operations
    up to fused multiply-adds) against what the reference calls (cut_select_qp.py:796, eigvalsh UPLO="U"): same accuracy class as
    LAPACK itself (<= 3e-15 on matrices of norm 2-4; the parity bound is 2e-13), same classification at -1e-15, no lane left to
    Jacobi on a generic point, <= 8 evaluations."""
    P = _proto()
    from sdpcutsel_via_nn_amd import synthetic
    nv = 60
    wl = synthetic.make_workload(nb_vars=nv, k=k, count=20000, seed=11 + k)
    L = nv * (nv + 1) // 2
    si, vv = wl["set_inds"][:, :k], wl["vars_values"]
    A = P.lifted(vv[L:][si], vv[:L][oracle.triu_positions(si, nv)], k)
    ref = np.linalg.eigvalsh(A, UPLO="U")[:, 0]
    st = {}
    lam, ok = P.lambda_min(A, st)
    assert ok.sum() >= ok.size - 2                     # (a nearly multiple lambda_min is left to Jacobi: none or a stray one here)
    assert np.abs(lam - ref)[ok].max() <= 3e-15
    assert np.array_equal((lam < -1e-15)[ok], (ref < -1e-15)[ok])
    assert st["evals"].max() <= P.K_MAX + 2 and st["evals"].mean() < 4.5


def test_lambda_min_solver_at_structured_vertices(oracle, golden_boxqp):
    """McCormick vertices (x = 0.5, X in {0, 0.5}): reducible tridiagonal forms are iterated block by block, exactly singular ones
    (zero diagonal: the relative deflation criterion never fires) and nearly multiple eigenvalues go to Jacobi -- a few per cent."""
    P = _proto()
    g = golden_boxqp
    for tag in ("spar020_100_1_d3", "spar020_100_1_d4", "spar030_060_1_d3"):
        S, ks, n = g[tag + "_set_inds"], g[tag + "_k"], int(g[tag + "_nb_vars"])
        L = n * (n + 1) // 2
        vv = g[tag + "_mck_vars"]
        for k in np.unique(ks):
            si = S[ks == k][:, :k]
            A = P.lifted(vv[L:][si], vv[:L][oracle.triu_positions(si, n)], int(k))
            ref = np.linalg.eigvalsh(A, UPLO="U")[:, 0]
            st = {}
            lam, ok = P.lambda_min(A, st)
            assert (~ok).mean() <= 0.05, (tag, k, int((~ok).sum()))
            assert np.abs(lam - ref)[ok].max() <= 3e-15
            assert np.array_equal((lam < -1e-15)[ok], (ref < -1e-15)[ok]), (tag, k)
            assert st["split"] > 0


def test_lambda_min_solver_on_exactly_singular_matrices(oracle):
    """Rounds 1 and 3 of a BoxQP run (spar125-075-2, dim 3, recorded trajectory): a per cent of the lifted matrices are exactly
    singular with a two- or threefold zero eigenvalue, and the trailing d_i, e_i of their tridiagonal forms are ALL rounding noise --
    LAPACK's relative deflation rule has nothing to compare with, the absolute one (a coupling below the stopping tolerance,
    csrc/lmin.h SDPCUT_LMIN_ABS_SPLIT) splits them: almost nobody is left to Jacobi, same accuracy, same side of -1e-15 as LAPACK
    for everybody the solver answers.  (Round 2 keeps its Jacobi lanes: LP noise of 1e-15 INSIDE a singular block couples two zero
    eigenvalues at the very level that decides the classification -- not deflated by either rule.)"""
    P = _proto()
    from sdpcutsel_via_nn_amd import _capi, harness
    g = np.load(os.path.join(ROOT, "tests", "golden", "rounds_spar125_075_2_d3_s4.npz"))
    inst = harness.parse_boxqp(os.path.join(ROOT, "tests", "golden", "instances", "spar125-075-2.in"))
    n, L = inst["nb_vars"], inst["nb_lifted"]
    S, ks, N = _capi.enumerate_cover(inst["adj"], 3)
    m = np.nonzero(ks == 3)[0]
    for r in (1, 3):
        vv = g["r%02d_vars" % r]
        A = P.lifted(vv[L:][S[m, :3]], vv[:L][oracle.triu_positions(S[m, :3], n)], 3)
        ref = np.linalg.eigvalsh(A, UPLO="U")[:, 0]
        left = {}
        for abs_split in (0.0, 1.0):
            P.ABS_SPLIT = abs_split
            try:
                lam, ok = P.lambda_min(A)
            finally:
                P.ABS_SPLIT = 1.0
            left[abs_split] = int((~ok).sum())
            assert np.abs(lam - ref)[ok].max() <= 3e-15
            assert np.array_equal((lam < -1e-15)[ok], (ref < -1e-15)[ok])
        assert left[0.0] >= 1000 and left[1.0] <= 20, (r, left)


def test_exact_eigenvalue_tool_on_a_known_matrix():
    """tests/exact_eig.py (exact characteristic polynomial + 80-digit Newton; r4: tools/lmin_truth.py): the noise-floor evidence and
    the per-pair proofs of the trajectory replay rest on it"""
    import exact_eig
    A = np.array([[2.0, 1.0, 0.0], [1.0, 2.0, 1.0], [0.0, 1.0, 2.0]])          # eigenvalues 2 - sqrt 2, 2, 2 + sqrt 2
    t = exact_eig.exact_lambda_min(A, 0.5)
    from decimal import Decimal
    assert abs(t - (Decimal(2) - Decimal(2).sqrt())) < Decimal(10) ** -50
