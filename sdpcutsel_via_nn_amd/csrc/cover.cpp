// Semidefinite vertex cover: enumeration of the candidate index sets (SURVEY.md section 8 f,
// row 1).  Host code: it runs once per instance and feeds sdpcut_set_candidates.
//
// Reference: CutSolver._get_sdp_vertex_cover, cut_select_qp.py:399-524 (ch_ext = 0: P^E_dim).
// The nested loops there enumerate, in lexicographic DFS order over increasing vertex ids,
//   * every clique of size dim of the sparsity graph, and
//   * every clique of size 2 <= s < dim that has no extension at all -- neither by a larger
//     vertex (the forward loop) nor by any smaller one (the "look backward" loop),
// i.e. the maximal cliques below the size cap.  A clique that can be extended forward is
// never emitted itself, whatever happens deeper.  The same order falls out of one recursion
// over bitset intersections; the reference's O(n) inner scans become word-wide ANDs.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/sdpcut.h"

namespace {

struct Cover {
    int n, words, dim;
    const uint64_t *adj;      // [n][words], diagonal cleared
    int64_t count, cap;
    int32_t *sets, *ks;       // outputs (may be null: count only)
    int32_t clique[SDPCUT_MAX_K];

    void emit(int size)
    {
        if (sets && count < cap) {
            int32_t *row = sets + count * SDPCUT_MAX_K;
            for (int a = 0; a < SDPCUT_MAX_K; ++a) row[a] = a < size ? clique[a] : -1;
            ks[count] = size;
        }
        ++count;
    }

    // common = intersection of the adjacency rows of clique[0..size)
    void extend(int size, const uint64_t *common)
    {
        if (size == dim) { emit(size); return; }
        const int last = clique[size - 1];
        bool forward = false;
        std::vector<uint64_t> next(words);
        for (int w = (last + 1) >> 6; w < words; ++w) {
            uint64_t bits = common[w];
            if (w == ((last + 1) >> 6)) bits &= ~0ull << ((last + 1) & 63);
            while (bits) {
                const int v = (w << 6) + __builtin_ctzll(bits);
                bits &= bits - 1;
                forward = true;
                for (int u = 0; u < words; ++u) next[u] = common[u] & adj[(size_t)v * words + u];
                clique[size] = v;
                extend(size + 1, next.data());
            }
        }
        if (forward) return;
        // no larger vertex extends it: emitted only if no smaller vertex does either
        // (members of the clique are not in `common`: the diagonal is cleared)
        for (int w = 0; w <= (last >> 6); ++w) {
            uint64_t bits = common[w];
            if (w == (last >> 6)) bits &= (last & 63) ? (~0ull >> (64 - (last & 63))) : 0ull;
            if (bits) return;
        }
        emit(size);
    }
};

} // namespace

extern "C" int sdpcut_enumerate_cover(int32_t nb_vars, const uint8_t *adjacency, int32_t dim, int64_t max_out,
                                      int32_t *set_inds_out, int32_t *ks_out, int64_t *count_out)
{
    if (nb_vars < 2 || !adjacency || dim < 3 || dim > SDPCUT_MAX_K || !count_out || max_out < 0 ||
        (max_out > 0 && (!set_inds_out || !ks_out)))
        return SDPCUT_EINVAL;
    const int n = nb_vars, words = (n + 63) / 64;
    std::vector<uint64_t> adj((size_t)n * words, 0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j)
            if (i != j && (adjacency[(size_t)i * n + j] || adjacency[(size_t)j * n + i]))
                adj[(size_t)i * words + (j >> 6)] |= 1ull << (j & 63);
    Cover c;
    c.n = n; c.words = words; c.dim = dim; c.adj = adj.data();
    c.count = 0; c.cap = max_out;
    c.sets = max_out > 0 ? set_inds_out : nullptr;
    c.ks = ks_out;
    std::vector<uint64_t> common(words);
    for (int i1 = 0; i1 < n; ++i1) {
        const uint64_t *row1 = &adj[(size_t)i1 * words];
        for (int w = (i1 + 1) >> 6; w < words; ++w) {
            uint64_t bits = row1[w];
            if (w == ((i1 + 1) >> 6)) bits &= ~0ull << ((i1 + 1) & 63);
            while (bits) {
                const int i2 = (w << 6) + __builtin_ctzll(bits);
                bits &= bits - 1;
                for (int u = 0; u < words; ++u) common[u] = row1[u] & adj[(size_t)i2 * words + u];
                c.clique[0] = i1;
                c.clique[1] = i2;
                c.extend(2, common.data());
            }
        }
    }
    *count_out = c.count;
    return SDPCUT_OK;
}
