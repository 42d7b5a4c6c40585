/*
 * ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar CPU restatement of the reference's MATLAB-Coder MLP evaluators
 *   neural_nets/NNs.so : neural_net_{2,3,4,5}D, NNs_initialize, NNs_terminate
 *   (source of truth neural_nets/neural_net_3D.m:47-62 simulation, :69-85 helpers;
 *    2D :43-58, 4D :47-62, 5D :51-69; called from cut_select_qp.py:579-582).
 *
 * Operation order (bit-verified there against NNs.so and pinned
 * again by tests/golden/nn_k*.npz, which were produced by calling the real NNs.so):
 *   xp_i  = ((v_i - xoffset_i) * gain_i) + ymin
 *   hidden: acc = 0; for i ascending: acc += a_i * W[j][i]; acc += b_j;
 *           a'_j = 2 / (exp(acc * -2) + 1) + -1
 *   output: acc = 0; for j ascending: acc += a_j * w_j; acc += b_out;
 *           y = (acc - ymin_out) / gain_out + xoffset_out
 * No FMA contraction: compile with -ffp-contract=off (see oracle/Makefile).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The weights come from oracle/_gen/nn_weights.inc, generated at build time from the
 * committed data fixture sdpcutsel_via_nn_amd/data/nn_weights.npz (oracle/gen_inc.py).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define MAX_WIDTH 64
#define MAX_LAYERS 5

typedef struct {
    int d_in;
    int n_layers;                 /* including the linear output layer */
    int width[MAX_LAYERS];        /* outputs of each layer */
    const double *xoffset, *gain; /* d_in each */
    double ymin;
    const double *W[MAX_LAYERS];  /* row-major [width[l]][fan_in] */
    const double *b[MAX_LAYERS];
    double y_ymin, y_gain, y_xoffset;
} net_t;

#include "_gen/nn_weights.inc" /* defines static const net_t NETS[4] (k = 2..5) */

static double eval_net(const net_t *net, const double *v)
{
    double a[MAX_WIDTH], a2[MAX_WIDTH];
    double *in = a, *out = a2, *t;
    int fan_in = net->d_in;
    for (int i = 0; i < fan_in; ++i)
        in[i] = ((v[i] - net->xoffset[i]) * net->gain[i]) + net->ymin;
    for (int l = 0; l < net->n_layers - 1; ++l) {
        const double *W = net->W[l];
        for (int j = 0; j < net->width[l]; ++j) {
            double acc = 0.0;
            for (int i = 0; i < fan_in; ++i)
                acc += in[i] * W[(size_t)j * fan_in + i];
            acc += net->b[l][j];
            out[j] = 2.0 / (exp(acc * -2.0) + 1.0) + -1.0;
        }
        fan_in = net->width[l];
        t = in; in = out; out = t;
    }
    {
        const int l = net->n_layers - 1;
        const double *w = net->W[l];
        double acc = 0.0;
        for (int j = 0; j < fan_in; ++j)
            acc += in[j] * w[j];
        acc += net->b[l][0];
        return (acc - net->y_ymin) / net->y_gain + net->y_xoffset;
    }
}

/* ---- the six symbols of the reference's FFI (cut_select_qp.py:297-303) ---- */
double neural_net_2D(const double X[5])  { return eval_net(&NETS[0], X); }
double neural_net_3D(const double X[9])  { return eval_net(&NETS[1], X); }
double neural_net_4D(const double X[14]) { return eval_net(&NETS[2], X); }
double neural_net_5D(const double X[20]) { return eval_net(&NETS[3], X); }
void NNs_initialize(void) {}
void NNs_terminate(void) {}

/* ---- batched helpers for the test-suite / cpu_baseline (same arithmetic) ---- */

/* inputs: [N][d_in] row-major; out: [N] raw network output */
int oracle_nn_batch(int k, int64_t N, const double *inputs, double *out)
{
    if (k < 2 || k > 5) return -1;
    const net_t *net = &NETS[k - 2];
    for (int64_t c = 0; c < N; ++c)
        out[c] = eval_net(net, inputs + (size_t)c * net->d_in);
    return 0;
}

/*
 * Optimality score of N k-variable candidates (cut_select_qp.py:569-582, 529-540):
 *   Xarr_inds = packed-triu positions of combinations_with_replacement(set_inds, 2)
 *   max_elem  = k * max|Q_arr[Xarr_inds]| (+1 if 0);  Q_slice = Q_arr[Xarr_inds] / max_elem
 *   S = ((0 + q0*X0) + q1*X1) + ...;  obj = (-S) * max_elem;  obj += nn([x_rho | Q_slice]) * max_elem
 * vars_values = [X packed (L) | x (n)].  nn_raw_out (may be NULL) receives the raw NN output.
 */
int oracle_opt_score_batch(int k, int64_t N, int n, const int32_t *set_inds,
                           const double *vars_values, const double *Q_arr,
                           double *obj_improve, double *nn_raw_out)
{
    if (k < 2 || k > 5) return -1;
    const net_t *net = &NETS[k - 2];
    const int64_t L = (int64_t)n * (n + 1) / 2;
    const double *Xv = vars_values, *xv = vars_values + L;
    for (int64_t c = 0; c < N; ++c) {
        const int32_t *s = set_inds + (size_t)c * k;
        double in[20], q[15], Xs[15];
        int m = 0;
        double amax = 0.0;
        for (int a = 0; a < k; ++a)
            for (int b = a; b < k; ++b) {
                int64_t pos = (int64_t)n * s[a] - (int64_t)s[a] * (s[a] + 1) / 2 + s[b];
                q[m] = Q_arr[pos];
                Xs[m] = Xv[pos];
                if (fabs(q[m]) > amax) amax = fabs(q[m]);
                ++m;
            }
        double max_elem = (double)k * amax;
        if (max_elem == 0.0) max_elem += 1.0;
        double S = 0.0;
        for (int i = 0; i < m; ++i) {
            q[i] = q[i] / max_elem;
            S += q[i] * Xs[i];
        }
        for (int a = 0; a < k; ++a) in[a] = xv[s[a]];
        for (int i = 0; i < m; ++i) in[k + i] = q[i];
        double y = eval_net(net, in);
        double obj = -S * max_elem;
        obj += y * max_elem;
        obj_improve[c] = obj;
        if (nn_raw_out) nn_raw_out[c] = y;
    }
    return 0;
}
