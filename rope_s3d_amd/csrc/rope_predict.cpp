// rope_predict / rope_predict_batch: the per-frame stage machine of Predictor.run (robotpose/prediction/predict.py:144-375) on the
// host, written against the public C ABI only — every batch of candidate poses it asks for is rendered and scored on the device;
// what runs here is the reference's control flow, a few dozen doubles of state per frame.
//
// One implementation serves both entry points: the stages below advance B frames in lockstep.  Frames are independent (fresh state
// per frame, predict.py:144-148; predict_dataset.py:43-44 is a plain loop), so at every step of a stage the rows all frames ask for
// travel as ONE device batch, each row scored against its own frame's target (rope_eval_targets) — the reference's ~25 dependent
// batches of 2-26 rows per frame become ~25 batches of B times as many rows per B frames.  rope_predict is the same code with B = 1
// on the context's single target (rope_eval).
//
// The arithmetic is the reference's numpy/Python arithmetic restated step by step, so that the decisions — and with
// them the returned angles — are those of rope_s3d_amd/prediction/predict.py and of the sequential oracle
// (oracle/predictor_ref.py): np.linspace's two formulas, np.mean's left-to-right sums, Python's min/index semantics
// with NaN, np.isclose.  The one piece that is not a transcription is the cubic of scipy's interp1d(kind='cubic'):
// the same not-a-knot spline, solved here by Gaussian elimination on its second derivatives.
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include <string>
#include <vector>

#include "../../include/rope_s3d.h"

void rope_set_error(rope_ctx *c, const std::string &msg);       // rope_abi.hip
void rope_range_push(const char *name);                         // roctx ranges, one per stage (rope_abi.hip; no-ops without the library)
void rope_range_pop();

namespace {

using Vec6 = std::array<double, 6>;
constexpr int HISTORY = 5;                                      // predict.py:30
constexpr double INF = std::numeric_limits<double>::infinity();

struct State {
    Vec6 angles{};                                              // predict.py:148
    Vec6 lr{{0.1, 0.1, 0.1, 0.1, 0.1, 0.1}};                    // predict.py:144, persists across stages
    double history[HISTORY][6] = {};
    double err_history[HISTORY] = {};
    // reference table aliasing (rope_predict_args::lookup_angles_live): the row of the live table `angles` still IS —
    // numpy's view of predict.py:171 — or -1 once a stage has bound the name to another array
    int alias_row = -1;
};

struct Machine {
    rope_ctx *c;
    const rope_predict_args &a;
    const bool batch;                                           // rows go to the resident targets of rope_set_targets
    int64_t evals = 0;
    Machine(rope_ctx *ctx, const rope_predict_args &args, bool many) : c(ctx), a(args), batch(many) {}
    std::vector<Vec6> rows;                                     // the step's rows of all frames ...
    std::vector<int32_t> frame_of;                              // ... and the frame each belongs to
    std::vector<double> err;

    double lim(int j, int side) const { return a.limits[2 * j + side]; }
    void clear() { rows.clear(); frame_of.clear(); }
    int add(int frame, const Vec6 &row) { rows.push_back(row); frame_of.push_back(frame); return (int)rows.size() - 1; }

    // Predictor._error (loss FULL; render_at_pos + _error, predict.py:159-161,475-509) or the TensorSweep score of every row
    int errors(int n_render, int loss = ROPE_LOSS_FULL)
    {
        err.assign(rows.size(), 0.0);
        if (rows.empty()) return ROPE_OK;
        evals += (int64_t)rows.size();
        if (batch) return rope_eval_targets(c, rows[0].data(), frame_of.data(), (int)rows.size(), n_render, loss, nullptr, err.data());
        return rope_eval(c, rows[0].data(), (int)rows.size(), n_render, loss, nullptr, err.data(), nullptr, nullptr, nullptr);
    }
};

// Python's min(list) / list.index(min(list)): the first element nothing is strictly smaller than (a leading NaN stays)
int py_argmin(const double *v, int n)
{
    int k = 0;
    for (int i = 1; i < n; i++)
        if (v[i] < v[k]) k = i;
    return k;
}

// np.argmin: the first NaN if there is one
int np_argmin(const std::vector<double> &v)
{
    for (size_t i = 0; i < v.size(); i++)
        if (std::isnan(v[i])) return (int)i;
    return py_argmin(v.data(), (int)v.size());
}

void push_front(double *hist, int n, double v)
{
    for (int i = n - 1; i > 0; i--) hist[i] = hist[i - 1];
    hist[0] = v;
}

void push_front(double (*hist)[6], const Vec6 &v)
{
    for (int i = HISTORY - 1; i > 0; i--)
        for (int j = 0; j < 6; j++) hist[i][j] = hist[i - 1][j];
    for (int j = 0; j < 6; j++) hist[0][j] = v[j];
}

// np.linspace(start, stop, num)[i] for scalar arguments (numpy/_core/function_base.py): i*step + start, unless the
// step is zero; the last sample is `stop` itself
double linspace_scalar(double start, double stop, int num, int i)
{
    if (i == num - 1) return stop;
    const double div = (double)(num - 1), delta = stop - start, step = delta / div;
    return step == 0.0 ? ((double)i / div) * delta + start : (double)i * step + start;
}

// np.linspace(lo, hi, num) for the six-vectors of a sweep: five of the six steps are zero there, which sends numpy
// down its (i / div) * delta + start formula for every column
Vec6 linspace_rows(const Vec6 &lo, const Vec6 &hi, int num, int i)
{
    if (i == num - 1) return hi;
    Vec6 r;
    const double div = (double)(num - 1);
    bool any_zero = false;
    for (int j = 0; j < 6; j++) any_zero = any_zero || ((hi[j] - lo[j]) / div == 0.0);
    for (int j = 0; j < 6; j++) {
        const double delta = hi[j] - lo[j];
        r[j] = any_zero ? ((double)i / div) * delta + lo[j] : (double)i * (delta / div) + lo[j];
    }
    return r;
}

// interp1d(x, y, kind='cubic')(xq): not-a-knot cubic spline through (x, y), x strictly increasing, n >= 4.
// Unknowns are the second derivatives M at the knots; dense elimination with partial pivoting (n <= a few hundred).
// A NaN among the y makes every output NaN, as interp1d's NaN path does.
void cubic_not_a_knot(const std::vector<double> &x, const std::vector<double> &y, const std::vector<double> &xq, std::vector<double> &out)
{
    const int n = (int)x.size();
    out.assign(xq.size(), std::numeric_limits<double>::quiet_NaN());
    for (int i = 0; i < n; i++)
        if (std::isnan(y[i])) return;
    std::vector<double> h(n - 1);
    for (int i = 0; i < n - 1; i++) {
        h[i] = x[i + 1] - x[i];
        if (!(h[i] > 0.0)) return;
    }
    std::vector<double> A((size_t)n * n, 0.0), M(n, 0.0);
    auto at = [&](int r, int col) -> double & { return A[(size_t)r * n + col]; };
    for (int i = 1; i < n - 1; i++) {
        at(i, i - 1) = h[i - 1];
        at(i, i) = 2.0 * (h[i - 1] + h[i]);
        at(i, i + 1) = h[i];
        M[i] = 6.0 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1]);
    }
    at(0, 0) = h[1]; at(0, 1) = -(h[0] + h[1]); at(0, 2) = h[0];
    at(n - 1, n - 3) = h[n - 2]; at(n - 1, n - 2) = -(h[n - 3] + h[n - 2]); at(n - 1, n - 1) = h[n - 3];
    for (int k = 0; k < n; k++) {
        int p = k;
        for (int r = k + 1; r < n; r++)
            if (std::fabs(at(r, k)) > std::fabs(at(p, k))) p = r;
        if (p != k) {
            for (int col = 0; col < n; col++) std::swap(at(k, col), at(p, col));
            std::swap(M[k], M[p]);
        }
        for (int r = k + 1; r < n; r++) {
            const double f = at(r, k) / at(k, k);
            if (f == 0.0) continue;
            for (int col = k; col < n; col++) at(r, col) -= f * at(k, col);
            M[r] -= f * M[k];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double s = M[k];
        for (int col = k + 1; col < n; col++) s -= at(k, col) * M[col];
        M[k] = s / at(k, k);
    }
    for (size_t q = 0; q < xq.size(); q++) {
        int i = 0;
        while (i < n - 2 && xq[q] >= x[i + 1]) i++;
        const double t0 = xq[q] - x[i], t1 = x[i + 1] - xq[q];
        out[q] = (M[i] * t1 * t1 * t1 + M[i + 1] * t0 * t0 * t0) / (6.0 * h[i]) + (y[i] / h[i] - M[i] * h[i] / 6.0) * t1 +
                 (y[i + 1] / h[i] - M[i + 1] * h[i] / 6.0) * t0;
    }
}

// the engine's argmin (finalize_argmin_kernel): first index of the smallest value, a NaN never beats a number, all NaN: index 0
int engine_argmin(const double *v, int n)
{
    int k = -1;
    for (int i = 0; i < n; i++)
        if (!std::isnan(v[i]) && (k < 0 || v[i] < v[k])) k = i;
    return k < 0 ? 0 : k;
}

// ---- Lookup (predict.py:165-171): argmin of the grid's lookup score, per frame
int stage_lookup(Machine &m, const rope_stage &s, std::vector<State> &sts)
{
    const int B = (int)sts.size(), n = m.a.n_lookup;
    std::vector<int32_t> best((size_t)B, 0);
    int rc;
    if (!m.batch) {
        if (m.a.use_table) rc = rope_lookup_score(m.c, nullptr, &best[0], nullptr);
        else rc = rope_eval(m.c, m.a.lookup_angles, n, s.to_render, ROPE_LOSS_LOOKUP, m.a.lookup_crop, nullptr, nullptr, &best[0], nullptr);
    } else if (m.a.use_table) {
        rc = rope_lookup_score_targets(m.c, best.data(), nullptr, nullptr);
    } else {
        // no stored table: the grid rendered and scored against every frame — frame by frame in one batch each (a grid is
        // thousands of rows: enough to fill the device), so that the shared layers of a grid keep working
        std::vector<int32_t> fo((size_t)n);
        std::vector<double> e((size_t)n);
        rc = ROPE_OK;
        for (int f = 0; f < B && !rc; f++) {
            fo.assign((size_t)n, f);
            rc = rope_eval_targets(m.c, m.a.lookup_angles, fo.data(), n, s.to_render, ROPE_LOSS_LOOKUP, m.a.lookup_crop, e.data());
            best[f] = engine_argmin(e.data(), n);
        }
    }
    if (rc) return rc;
    m.evals += (int64_t)n * B;
    for (int f = 0; f < B; f++) {
        if (best[f] < 0 || best[f] >= n) {
            rope_set_error(m.c, "rope_predict: lookup argmin outside the grid (table built for another grid?)");
            return ROPE_E_ARG;
        }
        const double *table = m.a.lookup_angles_live ? m.a.lookup_angles_live : m.a.lookup_angles;
        for (int j = 0; j < 6; j++) sts[f].angles[j] = table[(size_t)best[f] * 6 + j];
        sts[f].alias_row = m.a.lookup_angles_live ? best[f] : -1;
    }
    return ROPE_OK;
}

// ---- Descent (predict.py:173-230)
int stage_descent(Machine &m, const rope_stage &s, std::vector<State> &sts)
{
    const int B = (int)sts.size();
    for (State &st : sts)
        for (int i = 0; i < 6; i++)
            if (!std::isnan(s.init_rate[i])) st.lr[i] = s.init_rate[i];                 // predict.py:175-177
    const int spec = m.a.speculate < 1 ? 1 : (m.a.speculate > 3 ? 3 : m.a.speculate);
    std::vector<int> joints;
    for (int j = 0; j < 6; j++)
        if ((s.joints >> j) & 1u) joints.push_back(j);
    struct Pairs { int index[3][9][2]; };                                               // rows of a frame's under/over pairs, per level and state
    std::vector<Pairs> pairs((size_t)B);
    std::vector<double> over_err((size_t)B, INF), under_err((size_t)B, INF);
    std::vector<char> active((size_t)B, 1);                                             // frames still inside the stage's loop
    std::vector<Vec6> frontier, next;
    int n_active = B;
    for (int it = 0; it < s.count && n_active > 0; it++) {
        // step sizes of the whole iteration first: a joint's step depends on its own angle and on the history only,
        // and neither changes before that joint's turn (predict.py:184-187)
        for (int f = 0; f < B; f++) {
            if (!active[f]) continue;
            State &st = sts[f];
            for (int idx : joints) {
                const double mean = ((((st.history[0][idx] + st.history[1][idx]) + st.history[2][idx]) + st.history[3][idx]) + st.history[4][idx]) / 5.0;
                if (std::fabs(mean - st.angles[idx]) <= st.lr[idx]) st.lr[idx] *= s.rate_reduction;
                for (int k = 0; k < 6; k++)
                    if (!(st.lr[k] >= m.a.min_ang_inc[k])) st.lr[k] = std::isnan(st.lr[k]) ? st.lr[k] : m.a.min_ang_inc[k];
            }
        }
        // under/over of up to `spec` joints as one batch: the pair of every state the earlier decisions of the group
        // can lead to (+step, -step, stay: 2, 6, 18 rows per frame); the decisions are then read off in the reference's order
        for (size_t g = 0; g < joints.size(); g += (size_t)spec) {
            const int glen = (int)std::min(joints.size() - g, (size_t)spec);
            m.clear();
            for (int f = 0; f < B; f++) {
                if (!active[f]) continue;
                const State &st = sts[f];
                auto &index = pairs[f].index;
                for (auto &lv : index) for (auto &kk : lv) kk[0] = kk[1] = -1;
                frontier.assign(1, st.angles);
                for (int level = 0; level < glen; level++) {
                    const int idx = joints[g + level];
                    next.clear();
                    for (size_t k = 0; k < frontier.size(); k++) {
                        const Vec6 &state = frontier[k];
                        Vec6 under = state;
                        under[idx] -= st.lr[idx];
                        Vec6 over = under;
                        over[idx] += 2 * st.lr[idx];
                        if (m.lim(idx, 0) <= under[idx] && under[idx] <= m.lim(idx, 1)) index[level][k][0] = m.add(f, under);
                        if (m.lim(idx, 0) <= over[idx] && over[idx] <= m.lim(idx, 1)) index[level][k][1] = m.add(f, over);
                        if (level + 1 < glen) {
                            Vec6 up = state, down = state;
                            up[idx] += st.lr[idx];
                            down[idx] -= st.lr[idx];
                            next.push_back(up); next.push_back(down); next.push_back(state);
                        }
                    }
                    frontier.swap(next);
                }
            }
            const int rc = m.errors(s.to_render);
            if (rc) return rc;
            for (int f = 0; f < B; f++) {
                if (!active[f]) continue;
                State &st = sts[f];
                const auto &index = pairs[f].index;
                int k = 0;
                for (int level = 0; level < glen; level++) {
                    const int idx = joints[g + level];
                    under_err[f] = index[level][k][0] >= 0 ? m.err[index[level][k][0]] : INF;
                    over_err[f] = index[level][k][1] >= 0 ? m.err[index[level][k][1]] : INF;
                    if (over_err[f] < under_err[f]) { st.angles[idx] += st.lr[idx]; k = 3 * k; }          // ties and NaN: stay (predict.py:212-215)
                    else if (over_err[f] > under_err[f]) { st.angles[idx] -= st.lr[idx]; k = 3 * k + 1; }
                    else k = 3 * k + 2;
                }
            }
        }
        for (int f = 0; f < B; f++) {
            if (!active[f]) continue;
            State &st = sts[f];
            push_front(st.history, st.angles);
            push_front(st.err_history, HISTORY, under_err[f] < over_err[f] ? under_err[f] : over_err[f]);     // min(over, under) of the LAST joint (predict.py:222)
            const double *eh = st.err_history;
            const double mean_err = ((((eh[0] + eh[1]) + eh[2]) + eh[3]) + eh[4]) / 5.0;
            bool leave = std::fabs(mean_err - eh[0]) / eh[0] < s.early_stop;
            if (!leave) {
                bool settled = true, stuck = true;
                for (int j = 0; j < 6; j++) {
                    double hi = st.history[0][j], lo = st.history[0][j];
                    for (int i = 1; i < HISTORY; i++) { hi = std::fmax(hi, st.history[i][j]); lo = std::fmin(lo, st.history[i][j]); }
                    const double spread = hi - lo, inc = m.a.min_ang_inc[j];
                    const bool close = std::fabs(spread - inc) <= 1e-8 + 1e-5 * std::fabs(inc);      // np.isclose defaults
                    settled = settled && (spread <= inc || close);
                    for (int i = 1; i < 3; i++) stuck = stuck && st.history[i][j] == st.history[0][j];
                }
                leave = settled || stuck;
            }
            if (leave) { active[f] = 0; n_active--; }            // this frame's `break`: no rows from it for the rest of the stage
        }
    }
    // `angles[idx] += rate` edits the array in place (predict.py:212-215): while that array is a row of the live table,
    // the table keeps the steps
    for (State &st : sts)
        if (st.alias_row >= 0)
            for (int j = 0; j < 6; j++) m.a.lookup_angles_live[(size_t)st.alias_row * 6 + j] = st.angles[j];
    return ROPE_OK;
}

// ---- SFlip (predict.py:232-281)
int stage_sflip(Machine &m, const rope_stage &s, std::vector<State> &sts)
{
    const int B = (int)sts.size();
    const double *cam = m.a.camera_pose;
    const double axis = cam[5] * std::fabs(std::cos(cam[3])) + cam[4] * std::fabs(std::sin(cam[3]));     // predict.py:245
    struct Plan { Vec6 temp; bool in_limits, at_limits; int first; };
    std::vector<Plan> plans((size_t)B);
    m.clear();
    for (int f = 0; f < B; f++) {
        Plan &p = plans[f];
        p.temp = sts[f].angles;
        Vec6 &temp = p.temp;
        const double sign = temp[0] > 0.0 ? 1.0 : (temp[0] < 0.0 ? -1.0 : temp[0]);                       // np.sign
        temp[0] = -temp[0] + 2 * axis * sign;
        const bool close_to_limits = 0.15 > std::fabs(m.lim(0, 0) - temp[0]) || 0.15 > std::fabs(m.lim(0, 1) - temp[0]);
        p.in_limits = m.lim(0, 0) <= temp[0] && temp[0] <= m.lim(0, 1);
        p.at_limits = !p.in_limits || close_to_limits;
        // every pose this stage can ask for is known before the first answer: up to three rows per frame
        p.first = m.add(f, sts[f].angles);
        if (p.in_limits) m.add(f, temp);
        if (p.at_limits) {
            Vec6 endpoint = temp;
            endpoint[0] = m.lim(0, 1);
            m.add(f, endpoint);
        }
    }
    const int rc = m.errors(s.to_render);
    if (rc) return rc;
    for (int f = 0; f < B; f++) {
        Plan &p = plans[f];
        State &st = sts[f];
        const double *e = &m.err[p.first];
        const int last = (p.in_limits ? 1 : 0) + (p.at_limits ? 1 : 0);          // this frame's last row
        double base_err = e[0];
        bool aliased = false;                           // `angles = temp` binds both names to one array (predict.py:261)
        if (p.in_limits && e[1] < base_err) { st.angles = p.temp; aliased = true; base_err = e[1]; }
        if (p.at_limits) {
            // predict.py:270-277: both endpoints are written into temp, the comparison sits after the loop, so only the
            // upper limit's error is used — and an accepted flip IS temp, so its S angle becomes the upper limit regardless
            p.temp[0] = m.lim(0, 1);
            if (aliased) st.angles = p.temp;
            if (e[last] < base_err) { st.angles = p.temp; aliased = true; }
        }
        if (aliased) st.alias_row = -1;                 // `angles = temp`: the name leaves the table row (temp was a copy)
    }
    return ROPE_OK;
}

// the sweep range of joint idx about the current angles (predict.py:295-300, 347-353)
void sweep_ends(const Machine &m, const rope_stage &s, const Vec6 &angles, int idx, Vec6 &lo, Vec6 &hi)
{
    lo = angles; hi = angles;
    if (std::isnan(s.range)) { lo[idx] = m.lim(idx, 0); hi[idx] = m.lim(idx, 1); }
    else {
        const double l = lo[idx] - s.range, h = hi[idx] + s.range;
        lo[idx] = m.lim(idx, 0) > l ? m.lim(idx, 0) : l;                    // max(a - range, lower limit)
        hi[idx] = m.lim(idx, 1) < h ? m.lim(idx, 1) : h;                    // min(a + range, upper limit)
    }
}

// ---- InterpolativeSweep (predict.py:283-338)
int stage_isweep(Machine &m, const rope_stage &s, std::vector<State> &sts)
{
    const int B = (int)sts.size(), div = s.count;
    bool have_base = false;                         // base_err is not refreshed between joints (predict.py:288-289)
    std::vector<double> base_err((size_t)B, 0.0);
    std::vector<std::vector<Vec6>> space((size_t)B, std::vector<Vec6>((size_t)div));
    std::vector<Vec6> lo((size_t)B), hi((size_t)B), angs((size_t)B);
    std::vector<std::vector<double>> space_err((size_t)B, std::vector<double>((size_t)div));
    std::vector<double> xs(div), xq((size_t)div * 5), pred;
    for (int idx = 0; idx < 6; idx++) {
        if (!((s.joints >> idx) & 1u)) continue;
        m.clear();
        for (int f = 0; f < B; f++) {
            sweep_ends(m, s, sts[f].angles, idx, lo[f], hi[f]);
            for (int i = 0; i < div; i++) space[f][i] = linspace_rows(lo[f], hi[f], div, i);
            if (!have_base) m.add(f, sts[f].angles);                            // the base pose rides along with the first sweep
            for (int i = 0; i < div; i++) m.add(f, space[f][i]);
        }
        int rc = m.errors(s.to_render);
        if (rc) return rc;
        const int per = div + (have_base ? 0 : 1), off = have_base ? 0 : 1;
        for (int f = 0; f < B; f++) {
            const double *e = &m.err[(size_t)f * per];
            if (!have_base) base_err[f] = e[0];
            for (int i = 0; i < div; i++) { space_err[f][i] = e[off + i]; xs[i] = space[f][i][idx]; }
            for (int i = 0; i < div * 5; i++) xq[i] = linspace_scalar(lo[f][idx], hi[f][idx], div * 5, i);
            cubic_not_a_knot(xs, space_err[f], xq, pred);
            angs[f] = sts[f].angles;
            angs[f][idx] = xq[np_argmin(pred)];
        }
        have_base = true;
        m.clear();
        for (int f = 0; f < B; f++) m.add(f, angs[f]);
        rc = m.errors(s.to_render);
        if (rc) return rc;
        for (int f = 0; f < B; f++) {
            State &st = sts[f];
            const double pred_min_err = m.err[f];
            const int best_sample = py_argmin(space_err[f].data(), div);
            const double errs[3] = {base_err[f], space_err[f][best_sample], pred_min_err};
            const int min_type = py_argmin(errs, 3);                                // ties go to the earlier entry
            if (min_type == 1) { st.angles = space[f][best_sample]; st.alias_row = -1; push_front(st.err_history, HISTORY, space_err[f][best_sample]); }
            else if (min_type == 2) { st.angles = angs[f]; st.alias_row = -1; push_front(st.err_history, HISTORY, pred_min_err); }
            push_front(st.history, st.angles);
        }
    }
    return ROPE_OK;
}

// ---- TensorSweep (predict.py:340-373): the sampled pose with the smallest mean * -std of |sqrt(T) - sqrt(D)| over the whole frame
int stage_tsweep(Machine &m, const rope_stage &s, std::vector<State> &sts)
{
    const int B = (int)sts.size(), div = s.count;
    Vec6 lo, hi;
    for (int idx = 0; idx < 6; idx++) {
        if (!((s.joints >> idx) & 1u)) continue;
        m.clear();
        for (int f = 0; f < B; f++) {
            sweep_ends(m, s, sts[f].angles, idx, lo, hi);
            for (int i = 0; i < div; i++) m.add(f, linspace_rows(lo, hi, div, i));
        }
        const int rc = m.errors(s.to_render, ROPE_LOSS_TSWEEP);
        if (rc) return rc;
        for (int f = 0; f < B; f++) {
            sts[f].angles = m.rows[(size_t)f * div + engine_argmin(&m.err[(size_t)f * div], div)];      // `angles = space[argmin]`: a new array
            sts[f].alias_row = -1;
        }
    }
    return ROPE_OK;
}

int run_stages(rope_ctx *c, const rope_predict_args *a, bool batch, int B, double *angles_out, double *trace_out, int64_t *n_evals)
{
    auto fail = [&](const char *msg) { rope_set_error(c, msg); return (int)ROPE_E_ARG; };
    if (!a || !angles_out) return fail("rope_predict: null pointer");
    if (!a->stages || a->n_stages < 1) return fail("rope_predict: no stages");
    if (!a->limits || !a->camera_pose || !a->min_ang_inc) return fail("rope_predict: limits, camera_pose and min_ang_inc are required");
    if (batch && a->lookup_angles_live) return fail("rope_predict_batch: the reference's table aliasing makes frames depend on their order (lookup_angles_live must be NULL)");
    if (B < 1) return fail("rope_predict_batch: no frames");
    for (int i = 0; i < a->n_stages; i++) {
        const rope_stage &s = a->stages[i];
        switch (s.kind) {
        case ROPE_STAGE_LOOKUP:
            if (!a->lookup_angles || a->n_lookup < 1) return fail("rope_predict: a Lookup stage needs the pose grid");
            if (!a->use_table && !a->lookup_crop) return fail("rope_predict: a Lookup stage without a stored table needs the crop");
            break;
        case ROPE_STAGE_DESCENT:
            if (s.count < 0) return fail("rope_predict: Descent iterations must be >= 0");
            break;
        case ROPE_STAGE_SFLIP:
            break;
        case ROPE_STAGE_ISWEEP:
            if (s.count < 4) return fail("rope_predict: InterpolativeSweep needs at least 4 divisions (cubic interpolation)");
            break;
        case ROPE_STAGE_TSWEEP:
            if (s.count < 1) return fail("rope_predict: TensorSweep needs at least one division");
            break;
        default:
            return fail("rope_predict: unknown stage kind");
        }
        if (s.to_render < 1 || s.to_render > ROPE_MAX_LINKS) return fail("rope_predict: to_render must be 1..6");
    }
    Machine m(c, *a, batch);
    std::vector<State> sts((size_t)B);
    for (int i = 0; i < a->n_stages; i++) {
        const rope_stage &s = a->stages[i];
        int rc = ROPE_OK;
        static const char *const names[] = {"rope:stage:Lookup", "rope:stage:Descent", "rope:stage:SFlip", "rope:stage:InterpolativeSweep", "rope:stage:TensorSweep"};
        rope_range_push(names[s.kind]);
        switch (s.kind) {
        case ROPE_STAGE_LOOKUP: rc = stage_lookup(m, s, sts); break;
        case ROPE_STAGE_DESCENT: rc = stage_descent(m, s, sts); break;
        case ROPE_STAGE_SFLIP: rc = stage_sflip(m, s, sts); break;
        case ROPE_STAGE_ISWEEP: rc = stage_isweep(m, s, sts); break;
        case ROPE_STAGE_TSWEEP: rc = stage_tsweep(m, s, sts); break;
        }
        rope_range_pop();
        if (rc) return rc;
        if (trace_out)
            for (int f = 0; f < B; f++)
                for (int j = 0; j < 6; j++) trace_out[((size_t)f * a->n_stages + i) * 6 + j] = sts[f].angles[j];
    }
    for (int f = 0; f < B; f++)
        for (int j = 0; j < 6; j++) angles_out[(size_t)f * 6 + j] = sts[f].angles[j];
    if (n_evals) *n_evals = m.evals;
    return ROPE_OK;
}

}  // namespace

extern "C" int rope_predict(rope_ctx *c, const rope_predict_args *a, double *angles_out, double *trace_out, int64_t *n_evals)
{
    if (!c) return ROPE_E_ARG;
    return run_stages(c, a, false, 1, angles_out, trace_out, n_evals);
}

extern "C" int rope_predict_batch(rope_ctx *c, const rope_predict_args *a, int n_frames, double *angles_out, double *trace_out, int64_t *n_evals)
{
    if (!c) return ROPE_E_ARG;
    return run_stages(c, a, true, n_frames, angles_out, trace_out, n_evals);
}
