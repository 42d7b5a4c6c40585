// The four trained MLPs of the reference, compiled into the library (csrc/_gen/nn_builtin.inc, generated at build time from
// data/nn_weights.npz): sdpcut_set_builtin_networks.  A caller that only knows the reference's binding -- the six symbols of
// NNs.so, exported by the sibling library libsdpcut_nns.so (csrc/nns_compat.cpp, include/sdpcut_nns.h) -- needs nothing else.
#include "common.h"

#include "_gen/nn_builtin.inc"   // BUILTIN_K[4], BUILTIN_NLAYERS[4], BUILTIN_WIDTHS[4][MAX_LAYERS], BUILTIN_PARAMS_k*, BUILTIN_NPARAMS[4]

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
