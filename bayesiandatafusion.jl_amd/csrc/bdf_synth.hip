// bdf_synth.hip -- host-only generator of the synthetic sparse relation of configuration C4 (SURVEY 8d "M-C4": rows
// uniform, columns Zipf-like with p(c) ~ 1 / (c + offset), ratings clip(round(3.5 + <u*_row, v*_col> + 0.5 eps), 1, 5)
// from a planted rank-8 model).  The reference's own large-scale benchmark draws its relation with sprand
// (test/benchmark_parallel_latent.jl:8-12); a 10M x 1M relation with 100M observations cannot go through a Python table,
// so the COO triplets are produced here, on all host cores, straight into the caller's arrays.
//
// Counter-based (Philox4x32-10, key = seed): observation k is the same whatever range, rank or thread generates it, so P
// processes can each generate the whole relation (or any part) and agree.
#include "bdf_common.h"
#include <algorithm>
#include <cmath>
#include <thread>

namespace {

constexpr uint32_t P_SYNTH_OBS = 16, P_SYNTH_U = 17, P_SYNTH_V = 18;
constexpr int RANK = 8;

inline void host_normal_pair(const u32x4 &o, double &n0, double &n1)
{
    const double u1 = bdf_u01(o.x, o.y), u2 = bdf_u01(o.z, o.w);
    const double r = std::sqrt(-2.0 * std::log(u1)), t = 6.283185307179586476925286766559 * u2;
    n0 = r * std::cos(t);
    n1 = r * std::sin(t);
}

// planted factor row `i` of stream `purpose`: RANK values 0.5 N(0,1)
inline void planted_row(uint64_t seed, uint32_t purpose, uint64_t i, double *out)
{
    for (int p = 0; p < RANK / 2; p++) {
        double a, b;
        host_normal_pair(bdf_draw(seed, 0, purpose, 0, i, (uint32_t)p), a, b);
        out[2 * p] = 0.5 * a;
        out[2 * p + 1] = 0.5 * b;
    }
}

template <typename F>
void parallel_for(int64_t n, F f)
{
    unsigned hw = std::thread::hardware_concurrency();
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), (int64_t)64, (n + 65535) / 65536}));
    if (nt == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    for (int64_t t = 0; t < nt; t++) th.emplace_back([=]() { f(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}

}  // namespace

extern "C" int bdf_synth_ratings(uint64_t seed, int64_t n_rows, int64_t n_cols, int64_t k_begin, int64_t k_end,
                                 double zipf_offset, double test_fraction, int32_t *rows_out, int32_t *cols_out,
                                 double *vals_out, uint8_t *held_out)
{
    BDF_REQUIRE(n_rows >= 1 && n_cols >= 1 && n_rows < (int64_t)0x7fffffff && n_cols < (int64_t)0x7fffffff, BDF_ERR_ARG,
                "bdf_synth_ratings: dimensions must be in 1..2^31-2");
    BDF_REQUIRE(k_begin >= 0 && k_end >= k_begin, BDF_ERR_ARG, "bdf_synth_ratings: bad observation range");
    BDF_REQUIRE(rows_out && cols_out && vals_out, BDF_ERR_ARG, "bdf_synth_ratings: NULL output");
    BDF_REQUIRE(zipf_offset >= 0.0 && test_fraction >= 0.0 && test_fraction <= 1.0, BDF_ERR_ARG, "bdf_synth_ratings: bad parameter");
    const int64_t n = k_end - k_begin;
    // the planted factors as tables when the range is large enough to pay for them (each row is needed ~n / n_rows times)
    const bool tab_u = n >= n_rows / 4, tab_v = n >= n_cols / 4;
    std::vector<double> U, V;
    if (tab_u) {
        U.resize((size_t)n_rows * RANK);
        parallel_for(n_rows, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) planted_row(seed, P_SYNTH_U, (uint64_t)i, &U[(size_t)i * RANK]); });
    }
    if (tab_v) {
        V.resize((size_t)n_cols * RANK);
        parallel_for(n_cols, [&](int64_t a, int64_t b) { for (int64_t i = a; i < b; i++) planted_row(seed, P_SYNTH_V, (uint64_t)i, &V[(size_t)i * RANK]); });
    }
    const double ratio = zipf_offset > 0.0 ? ((double)n_cols + zipf_offset) / zipf_offset : 0.0;
    parallel_for(n, [&](int64_t a, int64_t b) {
        double ur[RANK], vr[RANK];
        for (int64_t q = a; q < b; q++) {
            const uint64_t k = (uint64_t)(k_begin + q);
            const u32x4 o = bdf_draw(seed, 0, P_SYNTH_OBS, 0, k, 0);
            const double u_row = bdf_u01(o.x, o.y), u_col = bdf_u01(o.z, o.w);
            int64_t r = (int64_t)(u_row * (double)n_rows);
            r = std::min(r, n_rows - 1);
            // inverse CDF of the continuous density ~ 1 / (c + offset) on [0, n_cols); offset 0: uniform
            double cc = zipf_offset > 0.0 ? zipf_offset * (std::pow(ratio, u_col) - 1.0) : u_col * (double)n_cols;
            int64_t c = std::min<int64_t>((int64_t)cc, n_cols - 1);
            c = std::max<int64_t>(c, 0);
            const double *pu = tab_u ? &U[(size_t)r * RANK] : (planted_row(seed, P_SYNTH_U, (uint64_t)r, ur), ur);
            const double *pv = tab_v ? &V[(size_t)c * RANK] : (planted_row(seed, P_SYNTH_V, (uint64_t)c, vr), vr);
            double dot = 0.0;
            for (int d = 0; d < RANK; d++) dot += pu[d] * pv[d];
            double e0, e1;
            const u32x4 o1 = bdf_draw(seed, 0, P_SYNTH_OBS, 0, k, 1);
            host_normal_pair(o1, e0, e1);
            double v = std::nearbyint(3.5 + dot + 0.5 * e0);
            v = std::min(5.0, std::max(1.0, v));
            rows_out[q] = (int32_t)(r + 1);
            cols_out[q] = (int32_t)(c + 1);
            vals_out[q] = v;
            if (held_out) {
                const u32x4 o2 = bdf_draw(seed, 0, P_SYNTH_OBS, 0, k, 2);
                held_out[q] = bdf_u01(o2.x, o2.y) < test_fraction ? 1 : 0;
            }
        }
    });
    return BDF_OK;
}
