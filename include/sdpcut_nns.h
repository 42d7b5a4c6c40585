/*
 * sdpcut_nns.h -- the reference's OWN FFI on the cut-scoring path, exported by  libsdpcut_nns.so  (a drop-in for
 * neural_nets/NNs.so; built next to libsdpcut_hip.so, source sdpcutsel_via_nn_amd/csrc/nns_compat.cpp):
 *
 *     nn_library = ctypes.cdll.LoadLibrary(<libsdpcut_nns.so instead of 'neural_nets/NNs.so'>)   cut_select_qp.py:297
 *     func_dim = getattr(nn_library, "neural_net_%dD" % d); func_dim.restype = c_double         :299-300
 *     input_arr = (c_double * (d (d+3) / 2))();  ...  nn(input_arr)                              :302, :579-582
 *     (also utilities.py:80-89, :157)
 *
 * X = [x_rho (d) | Q_slice (d(d+1)/2)], returns the raw network output.  One call = a batch of one (sdpcut_nn_batch of
 * sdpcut.h) on a process-wide default handle of the GPU library (device SDPCUT_COMPAT_DEVICE, default 0) with the
 * built-in networks.  NNs_initialize / NNs_terminate (no-ops in NNs.so) create / destroy that handle; the first
 * neural_net_kD call creates it if needed.  Without the GPU library or a gfx950 device the functions report once on
 * stderr and return NaN -- there is no CPU fallback (SURVEY.md section 8 b lists "CPU twins of each" entry point:
 * deliberately absent).
 *
 * BIT-IDENTICAL to NNs.so for the shipped networks: the reference's summation order without contraction, and exp
 * evaluated as the host libm evaluates it (NNs.so imports exp from libm; glibc >= 2.28 e_exp.c, the x86-64 FMA variant
 * -- csrc/libm_exp.h, which also states what happens outside that libm's main path).  4096 of 4096 recorded outputs per
 * network agree to the last bit (profiles/r04_accuracy.txt).
 *
 * These six names are ALL this library exports; libsdpcut_hip.so exports sdpcut_* only (r5), so loading either of them
 * next to the reference's real NNs.so cannot capture the other's symbols.
 */
#ifndef SDPCUT_NNS_H
#define SDPCUT_NNS_H

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

double neural_net_2D(const double X[5]);
double neural_net_3D(const double X[9]);
double neural_net_4D(const double X[14]);
double neural_net_5D(const double X[20]);
void NNs_initialize(void);
void NNs_terminate(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SDPCUT_NNS_H */
