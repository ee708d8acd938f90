/*
 * alphapig_host.h -- C ABI of libalphapig_host.so: the HOST side of the batched self-play
 * engine (board rules, PUCT tree pool, one-leaf-per-game playout scheduler).
 *
 * Plain C, no torch / numpy types.  Not thread-safe per pool (one driver thread); the
 * library parallelises internally over games.  Every int-returning entry point returns
 * >= 0 on success and a negative APZH_E_* code on failure (text via apzh_last_error()).
 *
 * Reference interfaces replaced (file:line in the reference tree):
 *   Board.init_board / do_move / game_end / has_a_winner / current_state
 *                                  game.py:35-44, :117-125, :160-167, :127-158, :68-94
 *   TreeNode.select / expand / update_recursive / get_value
 *                                  mcts_alphaZero.py:43-49, :34-41, :61-67, :69-80
 *   MCTS._playout / get_move_probs (visit counts) / update_with_move
 *                                  mcts_alphaZero.py:108-139, :141-157, :159-167
 *   mcts_pure.MCTS._playout / _evaluate_rollout / get_move
 *                                  mcts_pure.py:114-136, :138-157, :159-169
 * The reference evaluates one leaf per forward (mcts_alphaZero.py:124); here every game
 * keeps its playouts strictly sequential and exposes at most ONE pending leaf, so a batch
 * of G games yields a batch of <= G leaves per step and each game's tree is bit-identical
 * to the reference's.
 */
#ifndef ALPHAPIG_HOST_H
#define ALPHAPIG_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define APZH_OK 0
#define APZH_E_ARG (-1)       /* bad argument / index out of range        */
#define APZH_E_ILLEGAL (-2)   /* illegal move (cell occupied / off board) */
#define APZH_E_STATE (-3)     /* call not valid in the current state      */
#define APZH_E_NOMEM (-4)

/* advance() status codes */
#define APZH_NEED_EVAL 1      /* a non-terminal leaf is pending: evaluate it, then feed() */
#define APZH_MOVE_READY 2     /* n_playout playouts done: read the root, then play a move  */

/* Q representation states: python int 0 / python float / float32 array */
#define APZH_Q_INT0 0
#define APZH_Q_PYF 1
#define APZH_Q_F32 2

typedef struct apzh_pool apzh_pool;

typedef struct apzh_config {
    int32_t width;         /* board_width  (game.py:25)                                   */
    int32_t height;        /* board_height (game.py:26); square boards as in the reference */
    int32_t n_in_row;      /* game.py:32                                                   */
    int32_t n_games;       /* concurrent game slots G                                      */
    int32_t n_playout;     /* playouts per move (mcts_alphaZero.py:106)                    */
    int32_t prior_is_f32;  /* 1: priors float32, c_puct*P rounded to float32 (net path);
                              0: float64 priors (pure MCTS, mcts_pure.py:24)               */
    int32_t n_threads;     /* <=0: all cores                                               */
    int32_t reserved;
    double c_puct;         /* mcts_alphaZero.py:105                                        */
} apzh_config;

const char *apzh_last_error(void);
int apzh_version(void);

apzh_pool *apzh_create(const apzh_config *cfg);
void apzh_destroy(apzh_pool *p);

/* ---- board of game slot g ------------------------------------------------------------ */
int apzh_game_reset(apzh_pool *p, int g, int start_player);          /* init_board + fresh root */
/* replace the slot's position by an explicit move/mover history (movers are 1 or 2);
 * the tree is left untouched (the reference keeps tree and board decoupled). */
int apzh_game_set_position(apzh_pool *p, int g, const int16_t *moves, const int8_t *movers,
                           int n, int current_player);
int apzh_game_do_move(apzh_pool *p, int g, int move);                 /* Board.do_move only      */
/* out[0]=current_player out[1]=n_moves out[2]=ended out[3]=winner(-1 none/tie) out[4]=last_move */
int apzh_game_status(apzh_pool *p, int g, int32_t *out5);
int apzh_game_history(apzh_pool *p, int g, int16_t *moves, int8_t *movers, int cap);  /* -> n  */
/* has_a_winner() by the reference's full scan: out[0]=win out[1]=winner */
int apzh_game_has_a_winner(apzh_pool *p, int g, int32_t *out2);

/* ---- leaf / position encoding --------------------------------------------------------- */
/* One position = apzh_code_stride(H,W) bytes: byte m (= h*W+w, un-flipped board index) is
 * 0 empty, 1+min(age,3) stone of the player to move, 5+min(age,3) opponent stone
 * (age = plies since it was placed, 0 = last move); byte H*W is the colour-plane value. */
int apzh_code_stride(int height, int width);
int apzh_game_codes(apzh_pool *p, int g, uint8_t *codes);             /* current position        */
/* host expansion codes -> planes[n][n_planes][H][W] float32, n_planes 9 (game.py:68-94) or
 * 4 (game.py:96-115); includes the vertical flip of game.py:94. */
int apzh_codes_to_planes(const uint8_t *codes, int n, int height, int width, int n_planes,
                         float *planes);

/* ---- search --------------------------------------------------------------------------- */
/* For each listed game run playouts until one reaches a non-terminal leaf (status
 * NEED_EVAL, codes row i filled) or n_playout playouts are done (MOVE_READY).  Terminal
 * leaves are backed up in place with +-1.0/0.0 python-float semantics and do not stop the
 * loop.  codes may be NULL when the caller reads leaves another way. */
int apzh_advance(apzh_pool *p, const int32_t *games, int n, int32_t *status, uint8_t *codes);
/* Expand + back up the pending leaf of each listed game from a dense policy row
 * probs[i][H*W] (children = empty cells ascending, priors = probs[i][cell], not
 * renormalised: policy_value_net_mxnet.py:274) and value values[i] (float32 semantics). */
int apzh_feed(apzh_pool *p, const int32_t *games, int n, const float *probs, const float *values);
/* apzh_feed, then apzh_advance, on the same games in one call (no game twice): status / codes as apzh_advance */
int apzh_feed_advance(apzh_pool *p, const int32_t *games, int n, const float *probs, const float *values,
                      int32_t *status, uint8_t *codes);
/* Same for one game with an explicit (action, prior) list in caller order (drop-in
 * policy_value_fn path).  value_is_f32: 1 float32 value, 0 python float. */
int apzh_feed_sparse(apzh_pool *p, int g, const int32_t *actions, const double *priors, int n,
                     double value, int value_is_f32);
/* moves from the root to the pending leaf (for rebuilding a Board at the leaf) -> n */
int apzh_pending_path(apzh_pool *p, int g, int16_t *moves, int cap);
int apzh_playouts_done(apzh_pool *p, int g);
int apzh_set_playouts_done(apzh_pool *p, int g, int k);
int apzh_set_n_playout(apzh_pool *p, int n_playout);

/* node inspection (node 0 = root): children in insertion order -> count; any output pointer
 * may be NULL.  node3[0]=N node3[1]=qkind node3[2]=parent index (-1 root);
 * node_q[0]=Q node_q[1]=prior.  child_ids are node indices valid until the next
 * feed / update_with_move of that game. */
int apzh_node_children(apzh_pool *p, int g, int node, int32_t *acts, int64_t *visits, double *q,
                       int8_t *qk, double *prior, int32_t *child_ids, int cap, int64_t *node3,
                       double *node_q);
/* 1: priors are float32 and c_puct*P is a float32 product; 0: float64 */
int apzh_set_prior_mode(apzh_pool *p, int prior_is_f32);
/* batched: visits[i][H*W] dense (-1 where the root has no child for that cell),
 * n_children[i]; for MOVE_READY games */
int apzh_root_visits_dense(apzh_pool *p, const int32_t *games, int n, int32_t *visits,
                           int32_t *n_children);
/* MCTS.update_with_move: re-root at child `move`, or fresh root if absent / -1 */
int apzh_update_with_move(apzh_pool *p, int g, int move);
/* do_move + update_with_move(move) + playouts_done=0; out3: ended, winner, n_moves */
int apzh_play_move(apzh_pool *p, int g, int move, int32_t *out3);
/* apzh_play_move for n DIFFERENT games in one call (OpenMP over games): game games[i] plays moves[i].  codes_before
 * [n][stride] / movers_before[n] (either may be NULL) receive the position codes and the player to move BEFORE the move --
 * the (state, player) the reference records per ply (game_ai.py:118-122); out3[n][3] as apzh_play_move. */
int apzh_play_moves(apzh_pool *p, const int32_t *games, int n, const int32_t *moves, uint8_t *codes_before,
                    int32_t *movers_before, int32_t *out3);
/* counters: out[0]=net leaf evals out[1]=terminal leaf playouts out[2]=live nodes out[3]=peak nodes */
int apzh_stats(apzh_pool *p, int g, int64_t *out4);
/* pool-wide: out[0]=bytes reserved for the tree arenas, out[1]=1 if they were pre-touched at creation,
 * out[2]=largest tree any game has held (nodes), out[3]=live nodes of all games.  (Build-side addition: the
 * reference allocates one Python object per node, mcts_alphaZero.py:34-41, and has nothing to report.) */
int apzh_pool_info(apzh_pool *p, int64_t *out4);

/* ---- pure-MCTS move (mcts_pure.py) ------------------------------------------------------ */
/* Runs n_playout rollout playouts from a fresh root on game g's position, drawing
 * np.random.rand() values from the MT19937 state (key[624], pos) -- the same legacy stream
 * the reference consumes -- and returns the most visited action.  The state is updated in
 * place.  Optional outputs as apzh_root_children. */
int apzh_pure_get_move(apzh_pool *p, int g, uint32_t *mt_key624, int32_t *mt_pos,
                       int32_t *acts, int64_t *visits, double *q, int cap, int32_t *n_children);

/* ---- root sampling on NumPy's legacy generator (mcts_alphaZero.py:13-16, :152-155, :198-201) ------------------ */
/* np.random.RandomState(seed) for an integer seed: fills key[624] and the position (624). */
int apzh_mt_seed(uint32_t seed, uint32_t *key624, int32_t *pos);
/* np.sum of a contiguous float64 vector (NumPy's pairwise summation), exposed for the tests. */
double apzh_np_sum(const double *a, int64_t n);
/* One move for each of g games, the way MCTSPlayer.get_action draws it.  Row i has counts[i] root children with actions
 * acts_flat[...] (ascending, rows concatenated) and e_flat[...] = exp(x - max x), x = 1/temp * log(visits + 1e-10) -- the
 * elementwise part of the reference's softmax, which the caller evaluates with NumPy itself (its vector exp / log are not
 * the C library's).  Per row: probs = e / np.sum(e); pi_out[i][H*W] (may be NULL) = probs scattered at the actions;
 * with_noise: p = (1 - eps) * probs + eps * RandomState.dirichlet(alpha * ones(k)), else p = probs;
 * move = RandomState.choice(acts, p=p).  Every game draws from its own legacy MT19937 state keys[i][624] / pos[i] /
 * has_gauss[i] / gauss[i] (np.random.get_state() fields), updated in place: bit for bit the stream the reference
 * consumes.  n_threads: OpenMP threads over games. */
int apzh_root_sample(int g, int hw, const double *e_flat, const int32_t *acts_flat, const int32_t *counts, double alpha,
                     double eps, int with_noise, uint32_t *keys, int32_t *pos, int32_t *has_gauss, double *gauss,
                     double *pi_out, int32_t *moves_out, int n_threads);
/* Tree-arena pre-touch limit of one rank, GB: its share (1 / local_world) of half the available memory, at most 96. */
double apzh_pretouch_limit_gb(double avail_gb, int local_world);

#ifdef __cplusplus
}
#endif
#endif
