// The reference's own FFI on this path, exported by the PRODUCT library (SURVEY.md section 8 b):
//
//     nn_library = ctypes.cdll.LoadLibrary('neural_nets/NNs.so')      cut_select_qp.py:297
//     func = nn_library.neural_net_%dD; func.restype = c_double       cut_select_qp.py:299-300
//     func(input_arr)  with input_arr = c_double[d(d+3)/2]            cut_select_qp.py:302, 579-582
//     (same binding in utilities.py:80-89, :157)
//
// NNs.so exports neural_net_{2,3,4,5}D plus the MATLAB-Coder no-ops NNs_initialize / NNs_terminate.
// Here the six symbols sit on a process-wide default handle of the GPU library: one call = a batch
// of one through sdpcut_nn_batch (reference operation order, fp64).  The trained weights are
// compiled in (csrc/_gen/nn_builtin.inc, generated at build time from data/nn_weights.npz), so a
// caller that only knows the reference's binding needs nothing else.  This is the compatibility
// door, not the fast path: the batched entry points of sdpcut.h replace the per-candidate call.
//
// No CPU fallback: without a gfx950 device the functions print the reason once and return NaN
// (the reference's signature has no error channel; a NaN score cannot be mistaken for a result).
#include <cmath>
#include <cstdio>
#include <mutex>

#include "common.h"

#include "_gen/nn_builtin.inc"   // BUILTIN_K[4], BUILTIN_NLAYERS[4], BUILTIN_WIDTHS[4][MAX_LAYERS], BUILTIN_PARAMS_k*, BUILTIN_NPARAMS[4]

static std::mutex g_compat_mu;
static sdpcut_handle g_compat = nullptr;
static bool g_compat_failed = false;

static const double *builtin_params(int i)
{
    switch (i) {
    case 0: return BUILTIN_PARAMS_2;
    case 1: return BUILTIN_PARAMS_3;
    case 2: return BUILTIN_PARAMS_4;
    default: return BUILTIN_PARAMS_5;
    }
}

extern "C" int sdpcut_set_builtin_networks(sdpcut_handle h, int max_k)
{
    if (!h) return SDPCUT_EINVAL;
    SDPCUT_NO_PENDING(h);
    if (max_k < 2 || max_k > SDPCUT_MAX_K) return sdpcut_fail(h, SDPCUT_EINVAL, "max_k must be 2..5");
    for (int i = 0; i < 4 && BUILTIN_K[i] <= max_k; ++i) {
        int rc = sdpcut_set_network(h, BUILTIN_K[i], BUILTIN_NLAYERS[i], BUILTIN_WIDTHS[i], builtin_params(i),
                                    BUILTIN_NPARAMS[i]);
        if (rc) return rc;
    }
    return SDPCUT_OK;
}

// must be called with g_compat_mu held
static sdpcut_handle compat_handle()
{
    if (g_compat || g_compat_failed) return g_compat;
    int dev = 0;
    if (const char *e = std::getenv("SDPCUT_COMPAT_DEVICE")) dev = std::atoi(e);
    sdpcut_handle h = nullptr;
    int rc = sdpcut_create(dev, &h);
    if (rc == SDPCUT_OK) rc = sdpcut_set_builtin_networks(h, SDPCUT_MAX_K);
    if (rc != SDPCUT_OK) {
        std::fprintf(stderr, "libsdpcut_hip: neural_net_kD needs a gfx950 GPU (%s); returning NaN\n",
                     sdpcut_last_error(h));
        if (h) sdpcut_destroy(h);
        g_compat_failed = true;
        return nullptr;
    }
    g_compat = h;
    return h;
}

static double compat_eval(int k, const double *X)
{
    std::lock_guard<std::mutex> lk(g_compat_mu);
    sdpcut_handle h = compat_handle();
    double y = std::nan("");
    if (!h || !X) return y;
    if (sdpcut_nn_batch(h, k, 1, X, &y) != SDPCUT_OK) {
        std::fprintf(stderr, "libsdpcut_hip: neural_net_%dD failed: %s\n", k, sdpcut_last_error(h));
        return std::nan("");
    }
    return y;
}

extern "C" {

double neural_net_2D(const double X[5]) { return compat_eval(2, X); }
double neural_net_3D(const double X[9]) { return compat_eval(3, X); }
double neural_net_4D(const double X[14]) { return compat_eval(4, X); }
double neural_net_5D(const double X[20]) { return compat_eval(5, X); }

// MATLAB Coder's init / terminate are single `ret` instructions in NNs.so; here they bracket the
// lifetime of the default handle (both optional: the first neural_net_kD call initialises lazily).
void NNs_initialize(void)
{
    std::lock_guard<std::mutex> lk(g_compat_mu);
    g_compat_failed = false;      // a later initialise may retry (e.g. after the device became visible)
    (void)compat_handle();
}

void NNs_terminate(void)
{
    std::lock_guard<std::mutex> lk(g_compat_mu);
    if (g_compat) sdpcut_destroy(g_compat);
    g_compat = nullptr;
    g_compat_failed = false;
}

} // extern "C"
