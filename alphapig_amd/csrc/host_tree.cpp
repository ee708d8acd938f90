// libalphapig_host.so -- host side of the batched self-play engine.
//
// Board rules, a struct-of-arrays PUCT tree arena per game, and a scheduler that keeps every
// game's playouts strictly sequential while exposing one pending leaf per game, so G games
// give one coalesced batch of <= G leaves per step.  C ABI in include/alphapig_host.h.
//
// Numerics follow the reference's NumPy-2 behaviour exactly:
//   u     = ((c_puct*P) * sqrt(double(N_parent))) / (1 + n)     c_puct*P rounded to float32
//                                                               when priors are float32
//   score = double(Q) + u, first maximum in child order
//   Q     : INT0 -> PYF (float64 math on +-1.0/0.0) -> F32 (all-float32 math once a net
//           value has passed through; a PYF value is rounded to float32 at that moment)
// Build with -ffp-contract=off: a fused multiply-add would change the roundings above.
#include "alphapig_host.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char *msg) {
    g_err = msg;
    return code;
}

struct Arena {
    std::vector<int32_t> parent, first_child, n;
    std::vector<int16_t> n_child, action;
    std::vector<uint8_t> qk;
    std::vector<double> q, prior;

    int32_t size() const { return (int32_t)parent.size(); }
    void clear() {
        parent.clear(); first_child.clear(); n.clear(); n_child.clear(); action.clear();
        qk.clear(); q.clear(); prior.clear();
    }
    int32_t alloc(int count) {
        int32_t base = size();
        size_t ns = (size_t)base + (size_t)count;
        if (ns > parent.capacity()) {
            size_t cap = std::max(ns, parent.capacity() * 2 + 1024);
            parent.reserve(cap); first_child.reserve(cap); n.reserve(cap); n_child.reserve(cap);
            action.reserve(cap); qk.reserve(cap); q.reserve(cap); prior.reserve(cap);
        }
        parent.resize(ns, -1); first_child.resize(ns, -1); n.resize(ns, 0); n_child.resize(ns, 0);
        action.resize(ns, -1); qk.resize(ns, APZH_Q_INT0); q.resize(ns, 0.0); prior.resize(ns, 0.0);
        return base;
    }
    void reset_root() {
        clear();
        alloc(1);
        prior[0] = 1.0;
    }
    // Address space for a whole search up front (pages are only touched as nodes are created): without it
    // the vectors double at the same node counts in every game, and since the games of a batch run in lock
    // step, all of them reallocate + copy in the SAME apzh_feed call (measured: 5-13 ms spikes against a
    // 0.18 ms median, ~8 % of the self-play step time).
    void reserve_nodes(size_t cap, bool touch) {
        if (cap <= parent.capacity()) return;
        parent.reserve(cap); first_child.reserve(cap); n.reserve(cap); n_child.reserve(cap);
        action.reserve(cap); qk.reserve(cap); q.reserve(cap); prior.reserve(cap);
        if (touch) {                // fault the pages in now (by the thread that will own the game), not inside the search
            alloc((int)cap);
            clear();
        }
    }
    static size_t bytes_per_node() { return 3 * 4 + 2 * 2 + 1 + 2 * 8; }
};

struct Game {
    std::vector<int8_t> cells;      // 0 empty, 1 / 2 owner
    std::vector<int16_t> ply_of;    // ply index at which the cell was filled
    std::vector<int16_t> hist_move;
    std::vector<int8_t> hist_mover;
    int current_player = 1;
    int last_move = -1;
    Arena tree[2];
    int cur = 0;
    int playouts_done = 0;
    bool pending = false;
    int32_t pending_leaf = -1;
    int leaf_player = 1;            // player to move at the pending leaf
    std::vector<int16_t> path;
    int64_t n_net = 0, n_term = 0, peak_nodes = 0;
};

struct MT {
    uint32_t *key;
    int32_t *pos;
    void gen() {
        const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, MA = 0x9908b0dfu;
        uint32_t *mt = key;
        int kk;
        uint32_t y;
        for (kk = 0; kk < 624 - 397; kk++) {
            y = (mt[kk] & UP) | (mt[kk + 1] & LO);
            mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        for (; kk < 623; kk++) {
            y = (mt[kk] & UP) | (mt[kk + 1] & LO);
            mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        }
        y = (mt[623] & UP) | (mt[0] & LO);
        mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? MA : 0u);
        *pos = 0;
    }
    uint32_t next32() {
        if (*pos >= 624) gen();
        uint32_t y = key[(*pos)++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    double next_double() {   // legacy random_sample
        long a = next32() >> 5, b = next32() >> 6;
        return (a * 67108864.0 + b) / 9007199254740992.0;
    }
};

// NumPy's LEGACY distributions (np.random.RandomState: what `np.random.dirichlet` / `np.random.choice` of
// mcts_alphaZero.py:198-201 draw through), restated from their published algorithms: standard exponential by
// inversion, Gaussian by the polar method with the second value cached in the generator state, standard gamma by
// Marsaglia-Tsang (shape >= 1) / the Ahrens-Dieter style rejection for shape < 1.  log / pow / sqrt are the C library's,
// which is what NumPy calls for scalars.
struct LegacyRng {
    MT mt;
    int32_t *has_gauss;
    double *gauss;
    double dbl() { return mt.next_double(); }
    double standard_exponential() { return -std::log(1.0 - dbl()); }
    double gauss_next() {
        if (*has_gauss) {
            const double tmp = *gauss;
            *gauss = 0.0;
            *has_gauss = 0;
            return tmp;
        }
        double f, x1, x2, r2;
        do {
            x1 = 2.0 * dbl() - 1.0;
            x2 = 2.0 * dbl() - 1.0;
            r2 = x1 * x1 + x2 * x2;
        } while (r2 >= 1.0 || r2 == 0.0);
        f = std::sqrt(-2.0 * std::log(r2) / r2);
        *gauss = f * x1;
        *has_gauss = 1;
        return f * x2;
    }
    double standard_gamma(double shape) {
        if (shape == 1.0) return standard_exponential();
        if (shape == 0.0) return 0.0;
        if (shape < 1.0) {
            for (;;) {
                const double U = dbl();
                const double V = standard_exponential();
                if (U <= 1.0 - shape) {
                    const double X = std::pow(U, 1.0 / shape);
                    if (X <= V) return X;
                } else {
                    const double Y = -std::log((1.0 - U) / shape);
                    const double X = std::pow(1.0 - shape + shape * Y, 1.0 / shape);
                    if (X <= (V + Y)) return X;
                }
            }
        }
        const double b = shape - 1.0 / 3.0;
        const double c = 1.0 / std::sqrt(9.0 * b);
        for (;;) {
            double X, V;
            do {
                X = gauss_next();
                V = 1.0 + c * X;
            } while (V <= 0.0);
            V = V * V * V;
            const double U = dbl();
            if (U < 1.0 - 0.0331 * (X * X) * (X * X)) return b * V;
            if (std::log(U) < 0.5 * X * X + b * (1.0 - V + std::log(V))) return b * V;
        }
    }
};

// np.sum of a contiguous float64 vector: NumPy's pairwise summation (blocks of <= 128 elements summed through eight
// running sums, larger inputs halved at multiples of eight).  tests/test_host_sampler.py holds it to np.sum bit for bit.
double np_pairwise_sum(const double *a, long n) {
    if (n < 8) {
        double res = -0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; k++) r[k] = a[k];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; k++) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

}  // namespace

struct apzh_pool {
    apzh_config cfg;
    int hw;
    int nthreads;
    std::vector<Game> games;
    int64_t arena_bytes = 0;        // address space reserved for the tree arenas of all games
    bool arena_touched = false;     // ... and whether it was faulted in at creation
};

namespace {

inline bool line_through(const int8_t *cells, int W, int H, int n, int move, int8_t who) {
    const int h = move / W, w = move % W;
    static const int DH[4] = {0, 1, 1, 1}, DW[4] = {1, 0, 1, -1};
    for (int d = 0; d < 4; d++) {
        int cnt = 1;
        for (int s = 1; s < n; s++) {
            int hh = h + DH[d] * s, ww = w + DW[d] * s;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W || cells[hh * W + ww] != who) break;
            cnt++;
        }
        for (int s = 1; s < n; s++) {
            int hh = h - DH[d] * s, ww = w - DW[d] * s;
            if (hh < 0 || hh >= H || ww < 0 || ww >= W || cells[hh * W + ww] != who) break;
            cnt++;
        }
        if (cnt >= n) return true;
    }
    return false;
}

// game.py:127-158 full scan (ascending cell order; see oracle/board_ref.py on scan order)
inline bool full_scan_winner(const int8_t *cells, int W, int H, int n, int n_stones, int *who) {
    *who = -1;
    if (n_stones < n + 2) return false;
    for (int m = 0; m < W * H; m++) {
        int8_t p = cells[m];
        if (!p) continue;
        int h = m / W, w = m % W;
        static const int DH[4] = {0, 1, 1, 1}, DW[4] = {1, 0, 1, -1};
        for (int d = 0; d < 4; d++) {
            if (DW[d] == 1 && !(w <= W - n)) continue;
            if (DW[d] == -1 && !(w >= n - 1)) continue;
            if (DH[d] == 1 && !(h <= H - n)) continue;
            int step = DH[d] * W + DW[d];
            bool ok = true;
            for (int k = 1; k < n; k++)
                if (cells[m + k * step] != p) { ok = false; break; }
            if (ok) { *who = p; return true; }
        }
    }
    return false;
}

// end-of-game test after `last` was played by `who` on a position that had no winner before
inline void end_after_move(const apzh_pool *P, const int8_t *cells, int n_stones, int last, int8_t who,
                           bool *ended, int *winner) {
    const apzh_config &c = P->cfg;
    if (n_stones >= c.n_in_row + 2 && line_through(cells, c.width, c.height, c.n_in_row, last, who)) {
        *ended = true; *winner = who; return;
    }
    if (n_stones >= P->hw) { *ended = true; *winner = -1; return; }
    *ended = false; *winner = -1;
}

inline void node_update(Arena &t, int32_t i, double v, bool v_f32) {
    t.n[i] += 1;
    if (t.qk[i] == APZH_Q_F32 || v_f32) {
        float q = (float)t.q[i];
        float d = (float)v - q;
        d = 1.0f * d;
        d = d / (float)t.n[i];
        t.q[i] = (double)(q + d);
        t.qk[i] = APZH_Q_F32;
    } else {
        double d = v - t.q[i];
        d = 1.0 * d;
        d = d / (double)t.n[i];
        t.q[i] = t.q[i] + d;
        t.qk[i] = APZH_Q_PYF;
    }
}

inline void backup(Arena &t, int32_t node, double v, bool v_f32) {
    while (node >= 0) {
        node_update(t, node, v, v_f32);
        v = -v;
        node = t.parent[node];
    }
}

inline int32_t select_child(const apzh_pool *P, const Arena &t, int32_t node) {
    const double s = std::sqrt((double)t.n[node]);
    const int32_t b = t.first_child[node], e = b + t.n_child[node];
    const double c = P->cfg.c_puct;
    int32_t best = b;
    double best_v = 0.0;
    if (P->cfg.prior_is_f32) {
        const float cf = (float)c;
        for (int32_t i = b; i < e; i++) {
            double cp = (double)(cf * (float)t.prior[i]);
            double val = t.q[i] + (cp * s) / (double)(1 + t.n[i]);
            if (i == b || val > best_v) { best_v = val; best = i; }
        }
    } else {
        for (int32_t i = b; i < e; i++) {
            double cp = c * t.prior[i];
            double val = t.q[i] + (cp * s) / (double)(1 + t.n[i]);
            if (i == b || val > best_v) { best_v = val; best = i; }
        }
    }
    return best;
}

inline void undo_path(Game &g) {
    for (int16_t m : g.path) g.cells[m] = 0;
    g.path.clear();
}

// descend from the root to a leaf applying moves on g.cells; returns leaf node, sets leaf_player
inline int32_t descend(const apzh_pool *P, Game &g) {
    Arena &t = g.tree[g.cur];
    int32_t node = 0;
    int pl = g.current_player;
    g.path.clear();
    int ply = (int)g.hist_move.size();
    while (t.first_child[node] >= 0) {
        node = select_child(P, t, node);
        int a = t.action[node];
        g.cells[a] = (int8_t)pl;
        g.ply_of[a] = (int16_t)ply++;
        g.path.push_back((int16_t)a);
        pl = 3 - pl;
    }
    g.leaf_player = pl;
    return node;
}

inline void leaf_end(const apzh_pool *P, const Game &g, bool *ended, int *winner) {
    int n_stones = (int)g.hist_move.size() + (int)g.path.size();
    if (g.path.empty()) {
        int who;
        if (full_scan_winner(g.cells.data(), P->cfg.width, P->cfg.height, P->cfg.n_in_row, n_stones, &who)) {
            *ended = true; *winner = who;
        } else if (n_stones >= P->hw) {
            *ended = true; *winner = -1;
        } else {
            *ended = false; *winner = -1;
        }
        return;
    }
    end_after_move(P, g.cells.data(), n_stones, g.path.back(), (int8_t)(3 - g.leaf_player), ended, winner);
}

inline void write_codes(const apzh_pool *P, const Game &g, int player_to_move, int n_stones, uint8_t *codes) {
    const int hw = P->hw;
    for (int m = 0; m < hw; m++) {
        int8_t c = g.cells[m];
        if (!c) { codes[m] = 0; continue; }
        int age = n_stones - 1 - g.ply_of[m];
        if (age > 3) age = 3;
        codes[m] = (uint8_t)((c == player_to_move ? 1 : 5) + age);
    }
    codes[hw] = (uint8_t)((n_stones % 2 == 0) ? 1 : 0);
    int stride = apzh_code_stride(P->cfg.height, P->cfg.width);
    for (int m = hw + 1; m < stride; m++) codes[m] = 0;
}

// run playouts of game g until a net evaluation is needed or the move is complete
inline int advance_one(apzh_pool *P, Game &g, uint8_t *codes) {
    if (g.pending) return APZH_NEED_EVAL;   // idempotent: leaf still waiting for feed()
    Arena &t = g.tree[g.cur];
    while (g.playouts_done < P->cfg.n_playout) {
        int32_t leaf = descend(P, g);
        bool ended; int winner;
        leaf_end(P, g, &ended, &winner);
        if (ended) {
            double lv = (winner == -1) ? 0.0 : (winner == g.leaf_player ? 1.0 : -1.0);
            backup(t, leaf, -lv, false);
            undo_path(g);
            g.n_term++;
            g.playouts_done++;
            continue;
        }
        g.pending = true;
        g.pending_leaf = leaf;
        if (codes) write_codes(P, g, g.leaf_player, (int)g.hist_move.size() + (int)g.path.size(), codes);
        return APZH_NEED_EVAL;
    }
    return APZH_MOVE_READY;
}

inline void finish_pending(apzh_pool *, Game &g, double value, bool v_f32) {
    Arena &t = g.tree[g.cur];
    backup(t, g.pending_leaf, -value, v_f32);
    undo_path(g);
    g.pending = false;
    g.pending_leaf = -1;
    g.n_net++;
    g.playouts_done++;
    if (t.size() > g.peak_nodes) g.peak_nodes = t.size();
}

void reroot(Game &g, int move) {
    Arena &src = g.tree[g.cur];
    int32_t child = -1;
    if (move >= 0 && src.first_child[0] >= 0) {
        int32_t b = src.first_child[0], e = b + src.n_child[0];
        for (int32_t i = b; i < e; i++)
            if (src.action[i] == move) { child = i; break; }
    }
    Arena &dst = g.tree[g.cur ^ 1];
    dst.clear();
    if (child < 0) {
        dst.reset_root();
    } else {
        // breadth-first copy of the subtree; children blocks stay contiguous and ordered
        dst.alloc(1);
        dst.parent[0] = -1; dst.action[0] = src.action[child]; dst.n[0] = src.n[child];
        dst.q[0] = src.q[child]; dst.qk[0] = src.qk[child]; dst.prior[0] = src.prior[child];
        std::vector<std::pair<int32_t, int32_t>> queue;   // (src index, dst index)
        queue.emplace_back(child, 0);
        size_t head = 0;
        while (head < queue.size()) {
            auto [s, d] = queue[head++];
            int nc = src.n_child[s];
            if (src.first_child[s] < 0) continue;
            int32_t sb = src.first_child[s];
            int32_t db = dst.alloc(nc);
            dst.first_child[d] = db;
            dst.n_child[d] = (int16_t)nc;
            for (int k = 0; k < nc; k++) {
                dst.parent[db + k] = d;
                dst.action[db + k] = src.action[sb + k];
                dst.n[db + k] = src.n[sb + k];
                dst.q[db + k] = src.q[sb + k];
                dst.qk[db + k] = src.qk[sb + k];
                dst.prior[db + k] = src.prior[sb + k];
                if (src.first_child[sb + k] >= 0) queue.emplace_back(sb + k, db + k);
            }
        }
    }
    src.clear();
    g.cur ^= 1;
}

inline int board_do_move(apzh_pool *P, Game &g, int move) {
    if (move < 0 || move >= P->hw || g.cells[move] != 0) return APZH_E_ILLEGAL;
    g.cells[move] = (int8_t)g.current_player;
    g.ply_of[move] = (int16_t)g.hist_move.size();
    g.hist_move.push_back((int16_t)move);
    g.hist_mover.push_back((int8_t)g.current_player);
    g.current_player = 3 - g.current_player;
    g.last_move = move;
    return APZH_OK;
}

inline void board_end(const apzh_pool *P, const Game &g, bool *ended, int *winner) {
    int who;
    int n_stones = (int)g.hist_move.size();
    if (full_scan_winner(g.cells.data(), P->cfg.width, P->cfg.height, P->cfg.n_in_row, n_stones, &who)) {
        *ended = true; *winner = who;
    } else if (n_stones >= P->hw) {
        *ended = true; *winner = -1;
    } else {
        *ended = false; *winner = -1;
    }
}

#define CHECK_POOL(p) if (!(p)) return fail(APZH_E_ARG, "null pool")
#define CHECK_GAME(p, g) CHECK_POOL(p); if ((g) < 0 || (g) >= (p)->cfg.n_games) return fail(APZH_E_ARG, "game index out of range")

}  // namespace

extern "C" {

const char *apzh_last_error(void) { return g_err.c_str(); }
int apzh_version(void) { return 1; }

apzh_pool *apzh_create(const apzh_config *cfg) {
    if (!cfg || cfg->width < 1 || cfg->height < 1 || cfg->width * cfg->height > 1024 ||
        cfg->n_games < 1 || cfg->n_in_row < 1 || cfg->n_playout < 0) {
        fail(APZH_E_ARG, "bad config");
        return nullptr;
    }
    if (cfg->width < cfg->n_in_row || cfg->height < cfg->n_in_row) {
        fail(APZH_E_ARG, "board width and height can not be less than n_in_row");
        return nullptr;
    }
    apzh_pool *p = new (std::nothrow) apzh_pool();
    if (!p) { fail(APZH_E_NOMEM, "out of memory"); return nullptr; }
    p->cfg = *cfg;
    p->hw = cfg->width * cfg->height;
    int nt = cfg->n_threads;
#ifdef _OPENMP
    if (nt <= 0) nt = std::min(omp_get_max_threads(), 16);   // a 1-GPU box grants 16 cores
#else
    nt = 1;
#endif
    p->nthreads = nt;
    p->games.resize(cfg->n_games);
    // one search adds at most n_playout * hw nodes to what the re-rooted subtree kept; capped at 2^21 nodes
    // (78 MB of address space per arena) -- larger trees fall back to geometric growth
    const size_t cap = std::min<size_t>((size_t)(cfg->n_playout + 8) * (size_t)(p->hw + 1) * 5 / 4, (size_t)1 << 21);
    // ... and touched up front when the whole pool stays under APZ_HOST_PRETOUCH_GB (default 16 GB; 1024 games of
    // 15x15 at n_playout = 400 are 8.7 GB): first-touch page faults cost more than the tree work itself
    // (feed median 0.35 ms against 0.15 ms once the pages exist) and would otherwise sit in the first
    // two searches of every slot -- most of a short benchmark window.
    // The limit follows the memory this process may really use -- its share of half of min(MemAvailable, cgroup
    // memory.max), at most 96 GB (apzh_pretouch_limit_gb) -- so that the competition-strength slice (BASELINE config 5: n_playout = 1600, 1024 games,
    // 31 GB of arenas) is touched up front as well instead of faulting its pages in inside the first searches.
    // All ranks of a node start together and each reads the same MemAvailable: the share is divided by the number of
    // ranks on this node (LOCAL_WORLD_SIZE, as torchrun and bench.py's own rank spawner export it).
    double limit_gb = 16.0;
    {
        double avail_gb = 0.0;
        if (FILE *f = fopen("/proc/meminfo", "r")) {
            char line[256];
            while (fgets(line, sizeof line, f)) {
                unsigned long long kb;
                if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) avail_gb = (double)kb / 1e6;
            }
            fclose(f);
        }
        if (FILE *f = fopen("/sys/fs/cgroup/memory.max", "r")) {
            unsigned long long b;
            if (fscanf(f, "%llu", &b) == 1 && (avail_gb == 0.0 || (double)b / 1e9 < avail_gb)) avail_gb = (double)b / 1e9;
            fclose(f);
        }
        int local_world = 1;
        if (const char *s = getenv("LOCAL_WORLD_SIZE")) local_world = atoi(s);
        if (avail_gb > 0.0) limit_gb = apzh_pretouch_limit_gb(avail_gb, local_world);
    }
    if (const char *s = getenv("APZ_HOST_PRETOUCH_GB")) limit_gb = atof(s);
    const double total_gb = 2.0 * (double)cfg->n_games * (double)cap * (double)Arena::bytes_per_node() / 1e9;
    const bool touch = total_gb <= limit_gb;
    p->arena_bytes = (int64_t)(2.0 * (double)cfg->n_games * (double)cap * (double)Arena::bytes_per_node());
    p->arena_touched = touch;
    const int ng = cfg->n_games;
#pragma omp parallel for schedule(static) num_threads(nt) if (ng > 8)
    for (int i = 0; i < ng; i++) {
        Game &g = p->games[i];
        g.cells.assign(p->hw, 0);
        g.ply_of.assign(p->hw, 0);
        g.tree[0].reserve_nodes(cap, touch);
        g.tree[1].reserve_nodes(cap, touch);
        g.tree[0].reset_root();
    }
    return p;
}

void apzh_destroy(apzh_pool *p) { delete p; }

int apzh_game_reset(apzh_pool *p, int gi, int start_player) {
    CHECK_GAME(p, gi);
    if (start_player != 0 && start_player != 1) return fail(APZH_E_ARG, "start_player must be 0 or 1");
    Game &g = p->games[gi];
    std::fill(g.cells.begin(), g.cells.end(), 0);
    g.hist_move.clear(); g.hist_mover.clear();
    g.current_player = start_player + 1;
    g.last_move = -1;
    g.tree[0].clear(); g.tree[1].clear();
    g.cur = 0;
    g.tree[0].reset_root();
    g.playouts_done = 0;
    g.pending = false; g.pending_leaf = -1; g.path.clear();
    return APZH_OK;
}

int apzh_game_set_position(apzh_pool *p, int gi, const int16_t *moves, const int8_t *movers, int n,
                           int current_player) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    if (n < 0 || n > p->hw || (current_player != 1 && current_player != 2)) return fail(APZH_E_ARG, "bad position");
    std::fill(g.cells.begin(), g.cells.end(), 0);
    g.hist_move.clear(); g.hist_mover.clear();
    for (int k = 0; k < n; k++) {
        int m = moves[k];
        if (m < 0 || m >= p->hw || g.cells[m] != 0 || (movers[k] != 1 && movers[k] != 2))
            return fail(APZH_E_ILLEGAL, "illegal move in history");
        g.cells[m] = movers[k];
        g.ply_of[m] = (int16_t)k;
        g.hist_move.push_back((int16_t)m);
        g.hist_mover.push_back(movers[k]);
    }
    g.current_player = current_player;
    g.last_move = n ? moves[n - 1] : -1;
    return APZH_OK;
}

int apzh_game_do_move(apzh_pool *p, int gi, int move) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    int rc = board_do_move(p, g, move);
    if (rc) return fail(rc, "illegal move");
    return APZH_OK;
}

int apzh_game_status(apzh_pool *p, int gi, int32_t *out5) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    bool ended; int winner;
    board_end(p, g, &ended, &winner);
    out5[0] = g.current_player; out5[1] = (int32_t)g.hist_move.size();
    out5[2] = ended ? 1 : 0; out5[3] = winner; out5[4] = g.last_move;
    return APZH_OK;
}

int apzh_game_history(apzh_pool *p, int gi, int16_t *moves, int8_t *movers, int cap) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    int n = (int)g.hist_move.size();
    if (cap < n) return fail(APZH_E_ARG, "history buffer too small");
    for (int k = 0; k < n; k++) {
        if (moves) moves[k] = g.hist_move[k];
        if (movers) movers[k] = g.hist_mover[k];
    }
    return n;
}

int apzh_game_has_a_winner(apzh_pool *p, int gi, int32_t *out2) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    int who;
    bool win = full_scan_winner(g.cells.data(), p->cfg.width, p->cfg.height, p->cfg.n_in_row,
                                (int)g.hist_move.size(), &who);
    out2[0] = win ? 1 : 0; out2[1] = who;
    return APZH_OK;
}

int apzh_code_stride(int height, int width) { return ((height * width + 1) + 15) / 16 * 16; }

int apzh_game_codes(apzh_pool *p, int gi, uint8_t *codes) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    write_codes(p, g, g.current_player, (int)g.hist_move.size(), codes);
    return APZH_OK;
}

int apzh_codes_to_planes(const uint8_t *codes, int n, int height, int width, int n_planes, float *planes) {
    if (!codes || !planes || n < 0 || (n_planes != 9 && n_planes != 4)) return fail(APZH_E_ARG, "bad arguments");
    const int hw = height * width, stride = apzh_code_stride(height, width);
    for (int i = 0; i < n; i++) {
        const uint8_t *c = codes + (size_t)i * stride;
        float *out = planes + (size_t)i * n_planes * hw;
        std::memset(out, 0, sizeof(float) * n_planes * hw);
        const float colour = c[hw] ? 1.0f : 0.0f;
        for (int m = 0; m < hw; m++) {
            const int h = m / width, w = m % width;
            const int o = (height - 1 - h) * width + w;          // game.py:94 vertical flip
            const int code = c[m];
            if (n_planes == 9) {
                out[8 * hw + o] = colour;
                if (!code) continue;
                const int opp = code >= 5, age = (code - 1) & 3;
                for (int k = 0; k <= age; k++) out[(6 - 2 * k + opp) * hw + o] = 1.0f;
            } else {
                out[3 * hw + o] = colour;
                if (!code) continue;
                const int opp = code >= 5, age = (code - 1) & 3;
                out[opp * hw + o] = 1.0f;
                if (age == 0) out[2 * hw + o] = 1.0f;
            }
        }
    }
    return APZH_OK;
}

int apzh_advance(apzh_pool *p, const int32_t *games, int n, int32_t *status, uint8_t *codes) {
    CHECK_POOL(p);
    if (!games || !status || n < 0) return fail(APZH_E_ARG, "bad arguments");
    for (int i = 0; i < n; i++)
        if (games[i] < 0 || games[i] >= p->cfg.n_games) return fail(APZH_E_ARG, "game index out of range");
    const int stride = apzh_code_stride(p->cfg.height, p->cfg.width);
#pragma omp parallel for schedule(dynamic, 4) num_threads(p->nthreads) if (n > 8)
    for (int i = 0; i < n; i++) {
        Game &g = p->games[games[i]];
        status[i] = advance_one(p, g, codes ? codes + (size_t)i * stride : nullptr);
    }
    return APZH_OK;
}

static inline void feed_one(apzh_pool *p, Game &g, const float *pr, float value) {
    const int hw = p->hw;
    Arena &t = g.tree[g.cur];
    int cnt = 0;
    for (int m = 0; m < hw; m++) cnt += (g.cells[m] == 0);
    int32_t leaf = g.pending_leaf;
    int32_t base = t.alloc(cnt);
    t.first_child[leaf] = base;
    t.n_child[leaf] = (int16_t)cnt;
    int k = 0;
    for (int m = 0; m < hw; m++) {
        if (g.cells[m]) continue;
        t.parent[base + k] = leaf;
        t.action[base + k] = (int16_t)m;
        t.prior[base + k] = (double)pr[m];
        k++;
    }
    finish_pending(p, g, (double)value, true);
}

int apzh_feed(apzh_pool *p, const int32_t *games, int n, const float *probs, const float *values) {
    CHECK_POOL(p);
    if (!games || !probs || !values || n < 0) return fail(APZH_E_ARG, "bad arguments");
    for (int i = 0; i < n; i++) {
        if (games[i] < 0 || games[i] >= p->cfg.n_games) return fail(APZH_E_ARG, "game index out of range");
        if (!p->games[games[i]].pending) return fail(APZH_E_STATE, "feed without a pending leaf");
    }
#pragma omp parallel for schedule(dynamic, 4) num_threads(p->nthreads) if (n > 8)
    for (int i = 0; i < n; i++) feed_one(p, p->games[games[i]], probs + (size_t)i * p->hw, values[i]);
    return APZH_OK;
}

/* apzh_feed followed by apzh_advance on the same games, one parallel region and one call: the step of a 32-leaf group (BASELINE
 * config 2) is two tree calls + their Python glue otherwise */
int apzh_feed_advance(apzh_pool *p, const int32_t *games, int n, const float *probs, const float *values, int32_t *status,
                      uint8_t *codes) {
    CHECK_POOL(p);
    if (!games || !probs || !values || !status || n < 0) return fail(APZH_E_ARG, "bad arguments");
    std::vector<char> seen((size_t)p->cfg.n_games, 0);
    for (int i = 0; i < n; i++) {
        if (games[i] < 0 || games[i] >= p->cfg.n_games) return fail(APZH_E_ARG, "game index out of range");
        if (!p->games[games[i]].pending) return fail(APZH_E_STATE, "feed without a pending leaf");
        if (seen[games[i]]++) return fail(APZH_E_ARG, "a game twice in one call");
    }
    const int stride = apzh_code_stride(p->cfg.height, p->cfg.width);
#pragma omp parallel for schedule(dynamic, 4) num_threads(p->nthreads) if (n > 8)
    for (int i = 0; i < n; i++) {
        Game &g = p->games[games[i]];
        feed_one(p, g, probs + (size_t)i * p->hw, values[i]);
        status[i] = advance_one(p, g, codes ? codes + (size_t)i * stride : nullptr);
    }
    return APZH_OK;
}

int apzh_feed_sparse(apzh_pool *p, int gi, const int32_t *actions, const double *priors, int n, double value,
                     int value_is_f32) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (!g.pending) return fail(APZH_E_STATE, "feed without a pending leaf");
    if (n < 0 || n > p->hw || (n && (!actions || !priors))) return fail(APZH_E_ARG, "bad arguments");
    Arena &t = g.tree[g.cur];
    int32_t leaf = g.pending_leaf;
    if (n > 0) {
        int32_t base = t.alloc(n);
        t.first_child[leaf] = base;
        t.n_child[leaf] = (int16_t)n;
        for (int k = 0; k < n; k++) {
            t.parent[base + k] = leaf;
            t.action[base + k] = (int16_t)actions[k];
            t.prior[base + k] = priors[k];
        }
    }
    finish_pending(p, g, value, value_is_f32 != 0);
    return APZH_OK;
}

int apzh_pending_path(apzh_pool *p, int gi, int16_t *moves, int cap) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (!g.pending) return fail(APZH_E_STATE, "no pending leaf");
    int n = (int)g.path.size();
    if (cap < n) return fail(APZH_E_ARG, "path buffer too small");
    for (int k = 0; k < n; k++) moves[k] = g.path[k];
    return n;
}

int apzh_playouts_done(apzh_pool *p, int gi) {
    CHECK_GAME(p, gi);
    return p->games[gi].playouts_done;
}

int apzh_set_playouts_done(apzh_pool *p, int gi, int k) {
    CHECK_GAME(p, gi);
    if (k < 0) return fail(APZH_E_ARG, "negative count");
    p->games[gi].playouts_done = k;
    return APZH_OK;
}

int apzh_set_n_playout(apzh_pool *p, int n_playout) {
    CHECK_POOL(p);
    if (n_playout < 0) return fail(APZH_E_ARG, "negative n_playout");
    p->cfg.n_playout = n_playout;
    return APZH_OK;
}

int apzh_node_children(apzh_pool *p, int gi, int node, int32_t *acts, int64_t *visits, double *q, int8_t *qk,
                       double *prior, int32_t *child_ids, int cap, int64_t *node3, double *node_q) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    Arena &t = g.tree[g.cur];
    if (node < 0 || node >= t.size()) return fail(APZH_E_ARG, "node index out of range");
    if (node3) { node3[0] = t.n[node]; node3[1] = t.qk[node]; node3[2] = t.parent[node]; }
    if (node_q) { node_q[0] = t.q[node]; node_q[1] = t.prior[node]; }
    if (t.first_child[node] < 0) return 0;
    int nc = t.n_child[node];
    if (cap < nc) return fail(APZH_E_ARG, "children buffer too small");
    int32_t b = t.first_child[node];
    for (int k = 0; k < nc; k++) {
        if (acts) acts[k] = t.action[b + k];
        if (visits) visits[k] = t.n[b + k];
        if (q) q[k] = t.q[b + k];
        if (qk) qk[k] = (int8_t)t.qk[b + k];
        if (prior) prior[k] = t.prior[b + k];
        if (child_ids) child_ids[k] = b + k;
    }
    return nc;
}

int apzh_set_prior_mode(apzh_pool *p, int prior_is_f32) {
    CHECK_POOL(p);
    p->cfg.prior_is_f32 = prior_is_f32 ? 1 : 0;
    return APZH_OK;
}

int apzh_root_visits_dense(apzh_pool *p, const int32_t *games, int n, int32_t *visits, int32_t *n_children) {
    CHECK_POOL(p);
    if (!games || !visits || n < 0) return fail(APZH_E_ARG, "bad arguments");
    const int hw = p->hw;
    for (int i = 0; i < n; i++) {
        if (games[i] < 0 || games[i] >= p->cfg.n_games) return fail(APZH_E_ARG, "game index out of range");
        Game &g = p->games[games[i]];
        Arena &t = g.tree[g.cur];
        int32_t *row = visits + (size_t)i * hw;
        std::fill(row, row + hw, -1);
        int nc = t.first_child[0] >= 0 ? t.n_child[0] : 0;
        for (int k = 0; k < nc; k++) row[t.action[t.first_child[0] + k]] = t.n[t.first_child[0] + k];
        if (n_children) n_children[i] = nc;
    }
    return APZH_OK;
}

int apzh_update_with_move(apzh_pool *p, int gi, int move) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    reroot(g, move);
    return APZH_OK;
}

int apzh_play_move(apzh_pool *p, int gi, int move, int32_t *out3) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    int rc = board_do_move(p, g, move);
    if (rc) return fail(rc, "illegal move");
    reroot(g, move);
    g.playouts_done = 0;
    if (out3) {
        bool ended; int winner;
        end_after_move(p, g.cells.data(), (int)g.hist_move.size(), move, (int8_t)(3 - g.current_player), &ended,
                       &winner);
        out3[0] = ended ? 1 : 0; out3[1] = winner; out3[2] = (int32_t)g.hist_move.size();
    }
    return APZH_OK;
}

int apzh_play_moves(apzh_pool *p, const int32_t *games, int n, const int32_t *moves, uint8_t *codes_before,
                    int32_t *movers_before, int32_t *out3) {
    if (!p || !games || !moves || !out3 || n < 0) return fail(APZH_E_ARG, "null argument");
    const int stride = apzh_code_stride(p->cfg.height, p->cfg.width);
    for (int i = 0; i < n; i++) {
        if (games[i] < 0 || games[i] >= (int)p->games.size()) return fail(APZH_E_ARG, "game index out of range");
        if (p->games[games[i]].pending) return fail(APZH_E_STATE, "a leaf is pending");
    }
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 8) num_threads(p->nthreads) if (n > 8)
    for (int i = 0; i < n; i++) {
        Game &g = p->games[games[i]];
        if (codes_before) write_codes(p, g, g.current_player, (int)g.hist_move.size(), codes_before + (size_t)i * stride);
        if (movers_before) movers_before[i] = g.current_player;
        if (board_do_move(p, g, moves[i])) {
#pragma omp atomic write
            bad = 1;
            continue;
        }
        reroot(g, moves[i]);
        g.playouts_done = 0;
        bool ended; int winner;
        end_after_move(p, g.cells.data(), (int)g.hist_move.size(), moves[i], (int8_t)(3 - g.current_player), &ended, &winner);
        out3[3 * i] = ended ? 1 : 0; out3[3 * i + 1] = winner; out3[3 * i + 2] = (int32_t)g.hist_move.size();
    }
    if (bad) return fail(APZH_E_ARG, "illegal move");
    return APZH_OK;
}

int apzh_stats(apzh_pool *p, int gi, int64_t *out4) {
    CHECK_GAME(p, gi);
    Game &g = p->games[gi];
    out4[0] = g.n_net; out4[1] = g.n_term; out4[2] = g.tree[g.cur].size(); out4[3] = g.peak_nodes;
    return APZH_OK;
}

int apzh_pool_info(apzh_pool *p, int64_t *out4) {
    if (!p || !out4) return fail(APZH_E_ARG, "null argument");
    int64_t peak = 0, live = 0;
    for (const Game &g : p->games) {
        peak = std::max<int64_t>(peak, g.peak_nodes);
        live += g.tree[g.cur].size();
    }
    out4[0] = p->arena_bytes; out4[1] = p->arena_touched ? 1 : 0; out4[2] = peak; out4[3] = live;
    return APZH_OK;
}

int apzh_pure_get_move(apzh_pool *p, int gi, uint32_t *mt_key624, int32_t *mt_pos, int32_t *acts, int64_t *visits,
                       double *q, int cap, int32_t *n_children) {
    CHECK_GAME(p, gi);
    if (!mt_key624 || !mt_pos) return fail(APZH_E_ARG, "null rng state");
    if (p->cfg.prior_is_f32) return fail(APZH_E_STATE, "pure MCTS needs a pool created with prior_is_f32=0");
    Game &g = p->games[gi];
    if (g.pending) return fail(APZH_E_STATE, "a leaf is pending");
    MT mt{mt_key624, mt_pos};
    Arena &t = g.tree[g.cur];
    t.reset_root();
    const int hw = p->hw;
    std::vector<int16_t> roll;          // rollout moves to undo
    std::vector<int16_t> avail;
    for (int it = 0; it < p->cfg.n_playout; it++) {
        int32_t leaf = descend(p, g);
        bool ended; int winner;
        leaf_end(p, g, &ended, &winner);
        if (!ended) {
            int cnt = 0;
            for (int m = 0; m < hw; m++) cnt += (g.cells[m] == 0);
            int32_t base = t.alloc(cnt);
            t.first_child[leaf] = base;
            t.n_child[leaf] = (int16_t)cnt;
            const double pr = 1.0 / (double)cnt;                       // np.ones(k)/k
            int k = 0;
            for (int m = 0; m < hw; m++) {
                if (g.cells[m]) continue;
                t.parent[base + k] = leaf; t.action[base + k] = (int16_t)m; t.prior[base + k] = pr;
                k++;
            }
        }
        // random rollout (mcts_pure.py:138-157)
        const int player = g.leaf_player;
        int pl = g.leaf_player;
        int n_stones = (int)g.hist_move.size() + (int)g.path.size();
        roll.clear();
        for (int step = 0; step < 1000 && !ended; step++) {
            avail.clear();
            for (int m = 0; m < hw; m++) if (!g.cells[m]) avail.push_back((int16_t)m);
            int best = 0; double bv = -1.0;
            for (size_t k = 0; k < avail.size(); k++) {
                double r = mt.next_double();
                if (r > bv) { bv = r; best = (int)k; }
            }
            int mv = avail[best];
            g.cells[mv] = (int8_t)pl;
            roll.push_back((int16_t)mv);
            n_stones++;
            end_after_move(p, g.cells.data(), n_stones, mv, (int8_t)pl, &ended, &winner);
            pl = 3 - pl;
        }
        double lv = (winner == -1) ? 0.0 : (winner == player ? 1.0 : -1.0);
        for (int16_t m : roll) g.cells[m] = 0;
        backup(t, leaf, -lv, false);
        undo_path(g);
    }
    int nc = t.first_child[0] >= 0 ? t.n_child[0] : 0;
    if (n_children) *n_children = nc;
    if (nc == 0) return fail(APZH_E_STATE, "no legal move");
    if ((acts || visits || q) && cap < nc) return fail(APZH_E_ARG, "children buffer too small");
    int32_t b = t.first_child[0], best = b;
    for (int k = 0; k < nc; k++) {
        if (acts) acts[k] = t.action[b + k];
        if (visits) visits[k] = t.n[b + k];
        if (q) q[k] = t.q[b + k];
        if (t.n[b + k] > t.n[best]) best = b + k;
    }
    int move = t.action[best];
    t.reset_root();                                                     // update_with_move(-1)
    return move;
}

double apzh_pretouch_limit_gb(double avail_gb, int local_world) {
    if (local_world < 1) local_world = 1;
    // half of what is available, split between the ranks of this node, never more than 96 GB per rank.  A single rank keeps
    // a floor of 4 GB (round 3's rule); several ranks get no floor: floors would add up past the half on a small node
    const double share = 0.5 * avail_gb / (double)local_world;
    const double floor_gb = local_world == 1 ? std::min(4.0, avail_gb) : 0.0;
    return std::min(96.0, std::max(floor_gb, share));
}

int apzh_mt_seed(uint32_t seed, uint32_t *key624, int32_t *pos) {
    if (!key624 || !pos) return fail(APZH_E_ARG, "null rng state");
    // np.random.RandomState(seed) / np.random.seed(seed) for an integer seed: Knuth's initialiser, position 624
    key624[0] = seed;
    for (int i = 1; i < 624; i++) key624[i] = 1812433253u * (key624[i - 1] ^ (key624[i - 1] >> 30)) + (uint32_t)i;
    *pos = 624;
    return APZH_OK;
}

double apzh_np_sum(const double *a, int64_t n) { return (a && n > 0) ? np_pairwise_sum(a, (long)n) : 0.0; }

int apzh_root_sample(int g, int hw, const double *e_flat, const int32_t *acts_flat, const int32_t *counts, double alpha,
                     double eps, int with_noise, uint32_t *keys, int32_t *pos, int32_t *has_gauss, double *gauss,
                     double *pi_out, int32_t *moves_out, int n_threads) {
    if (g < 0 || hw <= 0 || !e_flat || !acts_flat || !counts || !keys || !pos || !has_gauss || !gauss || !moves_out)
        return fail(APZH_E_ARG, "null argument");
    std::vector<int64_t> start((size_t)g + 1, 0);
    for (int i = 0; i < g; i++) {
        if (counts[i] <= 0 || counts[i] > hw) return fail(APZH_E_ARG, "a row without children");
        start[i + 1] = start[i] + counts[i];
    }
#ifdef _OPENMP
    const int nt = n_threads > 0 ? n_threads : 1;
#pragma omp parallel num_threads(nt) if (g > 4 && nt > 1)
#endif
    {
        std::vector<double> probs((size_t)hw), cdf((size_t)hw);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
        for (int i = 0; i < g; i++) {
            const int k = counts[i];
            const double *e = e_flat + start[i];
            const int32_t *acts = acts_flat + start[i];
            // probs = e / np.sum(e)                                 (mcts_alphaZero.py:13-16, the tail of softmax)
            const double s = np_pairwise_sum(e, k);
            for (int j = 0; j < k; j++) probs[j] = e[j] / s;
            if (pi_out) {
                double *pi = pi_out + (size_t)i * hw;
                for (int j = 0; j < hw; j++) pi[j] = 0.0;
                for (int j = 0; j < k; j++) pi[acts[j]] = probs[j];
            }
            LegacyRng rng{MT{keys + (size_t)i * 624, pos + i}, has_gauss + i, gauss + i};
            if (with_noise) {
                // noise = dirichlet(alpha * ones(k)): k standard gammas, normalised by the reciprocal of their sum
                double acc = 0.0;
                for (int j = 0; j < k; j++) {
                    cdf[j] = rng.standard_gamma(alpha);
                    acc += cdf[j];
                }
                const double invacc = 1.0 / acc;
                // p = (1 - eps) * probs + eps * noise               (:198-200; no fused multiply-add: -ffp-contract=off)
                const double keep = 1.0 - eps;
                for (int j = 0; j < k; j++) {
                    const double noise = cdf[j] * invacc;
                    const double a = keep * probs[j], b = eps * noise;
                    probs[j] = a + b;
                }
            }
            // choice(acts, p): cdf = cumsum(p); cdf /= cdf[-1]; first index with cdf > one uniform double
            double run = 0.0;
            for (int j = 0; j < k; j++) {
                run = j == 0 ? probs[0] : run + probs[j];
                cdf[j] = run;
            }
            const double last = cdf[k - 1];
            for (int j = 0; j < k; j++) cdf[j] /= last;
            const double u = rng.dbl();
            int lo = 0, hi = k;                                      // searchsorted(side='right')
            while (lo < hi) {
                const int mid = lo + (hi - lo) / 2;
                if (u < cdf[mid]) hi = mid; else lo = mid + 1;
            }
            moves_out[i] = acts[lo < k ? lo : k - 1];
        }
    }
    return APZH_OK;
}

}  // extern "C"
