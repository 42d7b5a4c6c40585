// libsdpcut_nns.so -- the reference's own FFI on this path, as a drop-in for neural_nets/NNs.so (include/sdpcut_nns.h):
//
//     nn_library = ctypes.cdll.LoadLibrary('neural_nets/NNs.so')      cut_select_qp.py:297
//     func = nn_library.neural_net_%dD; func.restype = c_double       cut_select_qp.py:299-300
//     func(input_arr)  with input_arr = c_double[d(d+3)/2]            cut_select_qp.py:302, 579-582
//     (same binding in utilities.py:80-89, :157)
//
// NNs.so exports neural_net_{2,3,4,5}D plus the MATLAB-Coder no-ops NNs_initialize / NNs_terminate -- and so does this
// file, and NOTHING else.  (Until round 4 the six names were exported by libsdpcut_hip.so itself: a process that loaded it
// RTLD_GLOBAL next to the reference's real NNs.so got whichever definition resolved first.  Now the GPU library exports
// sdpcut_* only, and this library -- which a maintainer puts in the place of NNs.so -- binds to it privately.)
//
// The six symbols sit on a process-wide default handle of the GPU library: one call = a batch of one through
// sdpcut_nn_batch (reference operation order, fp64, the host libm's exp: bit-identical to NNs.so).  The trained weights
// are compiled into the GPU library (sdpcut_set_builtin_networks).  This is the compatibility door, not the fast path:
// the batched entry points of sdpcut.h replace the per-candidate call.
//
// The GPU library is looked for (1) at $SDPCUT_LIBRARY, (2) next to this file's REAL location (a symlink named NNs.so in the
// reference's neural_nets/ directory works), and opened RTLD_LOCAL.
//
// No CPU fallback: without the GPU library or a gfx950 device the functions print the reason once and return NaN (the
// reference's signature has no error channel; a NaN score cannot be mistaken for a result).
#include <dlfcn.h>
#include <libgen.h>
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/sdpcut_nns.h"

namespace {

typedef struct sdpcut_ctx *sdpcut_handle;
struct Api {
    int (*create)(int, sdpcut_handle *) = nullptr;
    int (*destroy)(sdpcut_handle) = nullptr;
    const char *(*last_error)(sdpcut_handle) = nullptr;
    int (*set_builtin_networks)(sdpcut_handle, int) = nullptr;
    int (*nn_batch)(sdpcut_handle, int, int64_t, const double *, double *) = nullptr;
};

std::mutex g_mu;
Api g_api;
void *g_dl = nullptr;
sdpcut_handle g_handle = nullptr;
bool g_failed = false;

std::string gpu_library_path()
{
    if (const char *e = getenv("SDPCUT_LIBRARY")) return e;
    Dl_info info;
    char real[PATH_MAX];
    if (dladdr((void *)&neural_net_3D, &info) && info.dli_fname && realpath(info.dli_fname, real)) {
        std::string dir = dirname(real);
        return dir + "/libsdpcut_hip.so";
    }
    return "libsdpcut_hip.so";
}

// must be called with g_mu held
bool bind_api()
{
    if (g_dl) return true;
    const std::string path = gpu_library_path();
    void *dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!dl) {
        std::fprintf(stderr, "libsdpcut_nns: cannot open the GPU library %s (%s); neural_net_kD returns NaN -- there is no CPU fallback\n",
                     path.c_str(), dlerror());
        return false;
    }
    Api a;
    a.create = (int (*)(int, sdpcut_handle *))dlsym(dl, "sdpcut_create");
    a.destroy = (int (*)(sdpcut_handle))dlsym(dl, "sdpcut_destroy");
    a.last_error = (const char *(*)(sdpcut_handle))dlsym(dl, "sdpcut_last_error");
    a.set_builtin_networks = (int (*)(sdpcut_handle, int))dlsym(dl, "sdpcut_set_builtin_networks");
    a.nn_batch = (int (*)(sdpcut_handle, int, int64_t, const double *, double *))dlsym(dl, "sdpcut_nn_batch");
    if (!a.create || !a.destroy || !a.last_error || !a.set_builtin_networks || !a.nn_batch) {
        std::fprintf(stderr, "libsdpcut_nns: %s does not export the sdpcut_* entry points this library binds\n", path.c_str());
        dlclose(dl);
        return false;
    }
    g_api = a;
    g_dl = dl;
    return true;
}

// must be called with g_mu held
sdpcut_handle default_handle()
{
    if (g_handle || g_failed) return g_handle;
    if (!bind_api()) {
        g_failed = true;
        return nullptr;
    }
    int dev = 0;
    if (const char *e = getenv("SDPCUT_COMPAT_DEVICE")) dev = atoi(e);
    sdpcut_handle h = nullptr;
    int rc = g_api.create(dev, &h);
    if (rc == 0) rc = g_api.set_builtin_networks(h, 5);
    if (rc != 0) {
        std::fprintf(stderr, "libsdpcut_nns: neural_net_kD needs a gfx950 GPU (%s); returning NaN\n", g_api.last_error(h));
        if (h) g_api.destroy(h);
        g_failed = true;
        return nullptr;
    }
    g_handle = h;
    return h;
}

double eval(int k, const double *X)
{
    std::lock_guard<std::mutex> lk(g_mu);
    sdpcut_handle h = default_handle();
    double y = std::nan("");
    if (!h || !X) return y;
    if (g_api.nn_batch(h, k, 1, X, &y) != 0) {
        std::fprintf(stderr, "libsdpcut_nns: neural_net_%dD failed: %s\n", k, g_api.last_error(h));
        return std::nan("");
    }
    return y;
}

}  // namespace

extern "C" {

double neural_net_2D(const double X[5]) { return eval(2, X); }
double neural_net_3D(const double X[9]) { return eval(3, X); }
double neural_net_4D(const double X[14]) { return eval(4, X); }
double neural_net_5D(const double X[20]) { return eval(5, X); }

// MATLAB Coder's init / terminate are single `ret` instructions in NNs.so; here they bracket the
// lifetime of the default handle (both optional: the first neural_net_kD call initialises lazily).
void NNs_initialize(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_failed = false;      // a later initialise may retry (e.g. after the device became visible)
    (void)default_handle();
}

void NNs_terminate(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_handle) g_api.destroy(g_handle);
    g_handle = nullptr;
    g_failed = false;
}

}  // extern "C"
