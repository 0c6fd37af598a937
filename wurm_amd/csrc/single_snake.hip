// single_snake.hip — gfx950 kernels and C-ABI entry points for SingleSnake and SimpleGridworld.
//
// Replaces the reference's op sequences (cited against oscarknagg/wurm):
//   SingleSnake.step      wurm/envs/single_snake.py:197-304   (~60 torch op dispatches + 2-3 host syncs)
//   SingleSnake._observe  wurm/envs/single_snake.py:104-195
//   SingleSnake.reset     wurm/envs/single_snake.py:322-387
//   determine_orientations wurm/utils.py:36-65, food respawn wurm/utils.py:181-232
//   SimpleGridworld.*     wurm/envs/simple_gridworld.py:88-268
// with ONE fused launch per call: one env per wavefront, the env's cells spread over the lanes
// (cell c = lane + 64*k), food/head channels held as per-lane bit sets, the body channel as per-lane ints,
// per-env scalars wave-uniform via ballots and DPP wave reductions (per-lane partials + one reduction, never one
// ballot per k), no host sync, no MFMA (integer/index work, HBM-bound).
// The only LDS use is a one-byte-per-cell class map for the cropped `partial_n` observation on grids > 128 cells and
// for the general (irregular-state) orientation stencil.
// rollout_kernel fuses T step+reset iterations with the env resident in registers; for well-formed start states it
// carries head cell / length / orientation / food cell as scalars (fast_step) instead of re-deriving them.
#include "step_args.hpp"
#include <cstdlib>

namespace wurm {

// ------------------------------------------------------------------------------------------------ state

template <int CPL>
struct Env {
    int body[CPL]; // body channel (SingleSnake only), cell lane + 64k
    u64 food;      // bit k: food at cell lane + 64k
    u64 head;      // bit k: head / agent at cell lane + 64k
};

struct Geo {
    int S, C, lane;
    float rcpS;
    u64 valid;    // bit k: lane + 64k < C
    u64 interior; // bit k: cell is not on the border ring
};

template <int CPL>
__device__ __forceinline__ Geo make_geo(int S)
{
    Geo g;
    g.S = S;
    g.C = S * S;
    g.lane = (int)(threadIdx.x & 63u);
    g.rcpS = 1.0f / (float)S;
    g.valid = 0;
    g.interior = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        if (c < g.C) {
            g.valid |= 1ull << k;
            int y = div_size(c, g.rcpS), x = c - y * S;
            if (y >= 1 && y <= S - 2 && x >= 1 && x <= S - 2) g.interior |= 1ull << k;
        }
    }
    return g;
}

template <int CPL, bool SNAKE>
__device__ __forceinline__ void load_state(const float *__restrict__ envp, const Geo &g, Env<CPL> &e)
{
    float f[CPL], h[CPL], b[CPL];
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        bool v = (g.valid >> k) & 1;
        f[k] = v ? envp[c] : 0.0f;
        h[k] = v ? envp[g.C + c] : 0.0f;
        b[k] = (SNAKE && v) ? envp[2 * g.C + c] : 0.0f;
    }
    e.food = 0;
    e.head = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        if (f[k] > 0.5f) e.food |= 1ull << k;
        if (h[k] > 0.5f) e.head |= 1ull << k;
        e.body[k] = SNAKE ? __float2int_rn(b[k]) : 0;
    }
}

template <int CPL, bool SNAKE>
__device__ __forceinline__ void store_state(float *__restrict__ envp, const Geo &g, const Env<CPL> &e)
{
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        if ((g.valid >> k) & 1) {
            envp[c] = ((e.food >> k) & 1) ? 1.0f : 0.0f;
            envp[g.C + c] = ((e.head >> k) & 1) ? 1.0f : 0.0f;
            if (SNAKE) envp[2 * g.C + c] = (float)e.body[k];
        }
    }
}

constexpr int NO_CELL = 1 << 20;

// lowest cell (row-major) whose bit is set in the per-lane bit set `bits` (bit k <=> cell lane + 64k), or -1.
// One per-lane ctz + one DPP min reduction — no per-k ballots (they cost two SGPRs each and, fully unrolled for
// large grids, drown the kernel in SGPR spills).
__device__ __forceinline__ int first_cell(u64 bits, int lane)
{
    int mine = bits ? lane + 64 * (__ffsll((long long)bits) - 1) : NO_CELL;
    int c = wave_min_i32(mine);
    return c >= NO_CELL ? -1 : c;
}

template <int CPL>
__device__ __forceinline__ int find_head(const Env<CPL> &e)
{
    return first_cell(e.head, (int)(threadIdx.x & 63u));
}

// ------------------------------------------------------------------------------------------------ orientation

// General form of determine_orientations (wurm/utils.py:36-65) for states that are not a well-formed snake
// (e.g. a done env stepped again before reset): neck map in LDS, 4-tap stencil, wave max, first argmax.
template <int CPL>
__device__ __forceinline__ int slow_orientation(const Env<CPL> &e, const Geo &g, int L, signed char *lds)
{
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        if ((g.valid >> k) & 1) {
            int r = e.body[k] - (L - 2);                   // utils.py:51-53 relu(body - (L-2))
            lds[c] = (signed char)(r <= 0 ? 0 : 2 * r - 3); // utils.py:54-55: r=1 -> -1 (neck), r=2 -> +1 (head)
        }
    }
    wave_lds_sync();
    int best0 = -128, best1 = -128, best2 = -128, best3 = -128;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        if ((g.valid >> k) & 1) {
            int y = div_size(c, g.rcpS), x = c - y * g.S;
            int own = lds[c];
            int n0 = y >= 1 ? lds[c - g.S] : 0;       // tap (-1, 0)
            int n1 = x <= g.S - 2 ? lds[c + 1] : 0;   // tap ( 0,+1)
            int n2 = y <= g.S - 2 ? lds[c + g.S] : 0; // tap (+1, 0)
            int n3 = x >= 1 ? lds[c - 1] : 0;         // tap ( 0,-1)
            best0 = max(best0, n0 - own);
            best1 = max(best1, n1 - own);
            best2 = max(best2, n2 - own);
            best3 = max(best3, n3 - own);
        }
    }
    best0 = wave_max_i32(best0);
    best1 = wave_max_i32(best1);
    best2 = wave_max_i32(best2);
    best3 = wave_max_i32(best3);
    int o = 0, bv = best0; // utils.py:63 argmax, first maximum wins
    if (best1 > bv) { bv = best1; o = 1; }
    if (best2 > bv) { bv = best2; o = 2; }
    if (best3 > bv) { bv = best3; o = 3; }
    wave_lds_sync();
    return uniform(o);
}

// determine_orientations (wurm/utils.py:36-65) of the env in registers.  Well-formed snake (exactly one cell == L
// and one == L-1, L >= 2): the filter response is 2 only for the tap pointing from the neck to the head, so the
// orientation follows from the two cells; anything else takes the exact stencil path.
// determine_orientations (wurm/utils.py:36-65) of the env in registers.  Well-formed snake (exactly one cell == L
// and one == L-1, L >= 2): the filter response is 2 only for the tap pointing from the neck to the head, so the
// orientation follows from the two cells; anything else takes the exact stencil path.
// top_two: count of cells equal to L and to L-1 and the lowest such cells (per-lane partials + 3 wave reductions).
template <int CPL>
__device__ __forceinline__ void top_two(const Env<CPL> &e, const Geo &g, int L, int &cntL, int &cntN, int &cellL,
                                        int &cellN)
{
    int packed = 0, cL = NO_CELL, cN = NO_CELL;
#pragma unroll
    for (int k = CPL - 1; k >= 0; --k) {
        const bool v = (g.valid >> k) & 1;
        if (v && e.body[k] == L) { packed += 1; cL = g.lane + 64 * k; }
        if (v && e.body[k] == L - 1) { packed += 1 << 16; cN = g.lane + 64 * k; }
    }
    packed = wave_sum_i32(packed);
    cntL = packed & 0xffff;
    cntN = packed >> 16;
    cellL = wave_min_i32(cL);
    cellN = wave_min_i32(cN);
}

template <int CPL>
__device__ __forceinline__ int orientation_of(const Env<CPL> &e, const Geo &g, int L, signed char *lds)
{
    int cntL, cntN, cellL, cellN;
    top_two<CPL>(e, g, L, cntL, cntN, cellL, cellN);
    if (cntL == 1 && cntN == 1 && L >= 2) {
        int yL = div_size(cellL, g.rcpS), xL = cellL - yL * g.S;
        int yN = div_size(cellN, g.rcpS), xN = cellN - yN * g.S;
        int dy = yL - yN, dx = xL - xN;
        return (dy == 0 && dx == 1) ? 1 : (dy == 1 && dx == 0) ? 2 : (dy == 0 && dx == -1) ? 3 : 0;
    }
    return slow_orientation<CPL>(e, g, L, lds);
}

// ------------------------------------------------------------------------------------------------ food respawn

// _get_food_addition (single_snake.py:306-320, simple_gridworld.py:209-223): +1 food on one uniformly random
// interior cell with nothing on it.  RNG form: the K-th free cell in row-major order, K = mulhi(word, n_free).
template <int CPL, bool SNAKE, bool WRITE>
__device__ __forceinline__ void add_food(Env<CPL> &e, const Geo &g, float *__restrict__ envp, bool use_inject,
                                         int inject_cell, u32 word)
{
    if (use_inject) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            int c = g.lane + 64 * k;
            if (c == inject_cell && ((g.valid >> k) & 1)) {
                e.food |= 1ull << k;
                if (WRITE) envp[c] = 1.0f;
            }
        }
        return;
    }
    const u64 occupied = e.food | e.head;
    u64 fr = 0; // bit k: cell lane + 64k is free
#pragma unroll
    for (int k = 0; k < CPL; ++k)
        if (((g.interior >> k) & 1) && !((occupied >> k) & 1) && (!SNAKE || e.body[k] == 0)) fr |= 1ull << k;
    const int n_free = wave_sum_i32(__popcll(fr));
    if (n_free == 0) return;
    const int K = (int)mulhi_range(word, (u32)n_free);
    // row-major order = k-major, lane-minor: walk the k planes (NOT unrolled: one live ballot at a time)
    int base = 0;
#pragma unroll 1
    for (int k = 0; k < CPL; ++k) {
        const bool b = (fr >> k) & 1;
        const u64 m = ballot(b);
        const int cnt = popc64(m);
        if (K < base + cnt) {
            if (b && base + rank_below(m) == K) {
                e.food |= 1ull << k;
                if (WRITE) envp[g.lane + 64 * k] = 1.0f;
            }
            break;
        }
        base += cnt;
    }
}

// ------------------------------------------------------------------------------------------------ step

struct StepOut {
    long long action; // sanitised action (SingleSnake)
    int headcell;     // head cell after the move, -1 if it left the grid
    float reward;
    int done, selfc, edgec;
    int foodcell;     // small_step only: the food cell after the step (-1: none); -2: the generic path ran
};

// step_core for grids of at most 128 cells whose state is a well-formed snake (exactly one head, on the unique maximum
// L >= 2 of the body channel; exactly one cell L - 1; at most one food cell; no negative values) — the state every
// per-call step of a reset-after-done loop sees.  Same transition, but every wave-level quantity comes from BALLOTS of
// per-lane compares (a v_cmp into an SGPR pair + s_bcnt1 / s_ff1) instead of DPP butterfly reductions: head and food
// cells are the set bits of two masks, L is the body value under the head (one v_readlane), "unique maximum" is
// popc(body == L) == 1 && no lane has body > L, the neck is the set bit of (body == L - 1).  PMC on round 1's
// step_kernel<2> at 65 536 envs: 404 VALU + 237 SALU per env, six compiler-emitted DPP reductions (~20 instructions
// each) among them, and the kernel was issue-bound at 2.1x the time its HBM traffic needs.  Returns false (nothing
// touched) if the state is anything else; step_core then runs its general path.
template <bool WRITE>
__device__ __forceinline__ bool small_step(Env<2> &e, const Geo &g, float *__restrict__ envp, long long a_in, StepOut &out,
                                           u64 seed, u64 call, u64 env_id, bool use_inject, int inject_cell)
{
    const int S = g.S, C = g.C, lane = g.lane;
    const u64 H0 = ballot((e.head & 1) != 0), H1 = ballot((e.head & 2) != 0);
    const u64 F0 = ballot((e.food & 1) != 0), F1 = ballot((e.food & 2) != 0);
    if (popc64(H0) + popc64(H1) != 1 || popc64(F0) + popc64(F1) > 1) return false;
    const int hc = H0 ? first_bit(H0) : 64 + first_bit(H1);
    const int fc = F0 ? first_bit(F0) : (F1 ? 64 + first_bit(F1) : -1);
    const int L = lane_value(hc < 64 ? e.body[0] : e.body[1], hc & 63);           // single_snake.py:210 snake_sizes
    if (L < 2) return false;
    const u64 above = ballot(e.body[0] > L || e.body[1] > L || e.body[0] < 0 || e.body[1] < 0);
    const u64 M0 = ballot(e.body[0] == L), M1 = ballot(e.body[1] == L);
    const u64 N0 = ballot(e.body[0] == L - 1), N1 = ballot(e.body[1] == L - 1);
    if (above != 0 || popc64(M0) + popc64(M1) != 1 || popc64(N0) + popc64(N1) != 1) return false;
    const int neck = N0 ? first_bit(N0) : 64 + first_bit(N1);
    // orientation (wurm/utils.py:36-65) from the two newest cells, as orientation_of
    const int hy = div_size(hc, g.rcpS), hx = hc - hy * S;
    const int yN = div_size(neck, g.rcpS), xN = neck - yN * S;
    const int dy = hy - yN, dx = hx - xN;
    const int o = (dy == 0 && dx == 1) ? 1 : (dy == 1 && dx == 0) ? 2 : (dy == 0 && dx == -1) ? 3 : 0;
    long long a = a_in;
    if ((long long)o == a) a += 2;                                                  // :221-222
    a = a % 4;
    const int ai = (int)(((a % 4) + 4) % 4);
    const int ny = hy - tap_y(ai), nx = hx - tap_x(ai);                            // :225-233
    const bool inside = ny >= 0 && ny < S && nx >= 0 && nx < S;
    const int nh = inside ? ny * S + nx : -1;
    const bool EAT = inside && nh == fc;                                           // :242
    const int under = inside ? lane_value(nh < 64 ? e.body[0] : e.body[1], nh & 63) : 0;
    const bool SELFC = inside && (EAT ? under : max(under - 1, 0)) > 0;            // :252 (after the decay)
    const int grow = L + (EAT ? 1 : 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = lane + 64 * k;
        const int b0 = e.body[k];
        int b = EAT ? b0 : max(b0 - 1, 0);                                         // :246-249
        if (c == nh) b += grow;                                                    // :258-262
        if (WRITE && b != b0) envp[2 * C + c] = (float)b;
        e.body[k] = b;
    }
    if (WRITE && lane == 0) {
        envp[C + hc] = 0.0f;
        if (inside) envp[C + nh] = 1.0f;
        if (EAT) envp[nh] = 0.0f;                                                  // :270-272
    }
    e.head = (inside && lane == (nh & 63)) ? (nh < 64 ? 1ull : 2ull) : 0ull;
    if (EAT) {                                                                     // :277-282
        e.food = 0;
        u32 word = 0;
        if (!use_inject) word = rng_words(seed, call, env_id, RNG_FOOD, 0).w[0];
        add_food<2, true, WRITE>(e, g, envp, use_inject, inject_cell, word);
    }
    const bool EDGEC = !(inside && ny >= 1 && ny <= S - 2 && nx >= 1 && nx <= S - 2); // :290-295
    out.action = a;
    out.headcell = nh;
    out.reward = EAT ? 1.0f : 0.0f;
    out.selfc = SELFC;
    out.edgec = EDGEC;
    out.done = SELFC | EDGEC;
    if (EAT) {
        const u64 G0 = ballot((e.food & 1) != 0), G1 = ballot((e.food & 2) != 0);
        out.foodcell = G0 ? first_bit(G0) : (G1 ? 64 + first_bit(G1) : -1);
    } else {
        out.foodcell = fc;
    }
    return true;
}

// One transition of one env held in registers.  WRITE: changed cells are written through to HBM as they are
// produced (per-call kernels); !WRITE: registers only (rollout kernel).
template <int CPL, bool SNAKE, bool WRITE>
__device__ __forceinline__ void step_core(Env<CPL> &e, const Geo &g, float *__restrict__ envp, long long a_in,
                                          StepOut &out, u64 seed, u64 call, u64 env_id, bool use_inject,
                                          int inject_cell, signed char *lds)
{
    out.foodcell = -2;
    if constexpr (SNAKE && CPL == 2 && WRITE) {
        if (small_step<WRITE>(e, g, envp, a_in, out, seed, call, env_id, use_inject, inject_cell)) return;
        out.foodcell = -2;
    }
    const int S = g.S, C = g.C, lane = g.lane;
    long long a = a_in;
    int L = 0;
    if (SNAKE) {
        int lm = 0;
#pragma unroll
        for (int k = 0; k < CPL; ++k) lm = max(lm, e.body[k]);
        L = uniform(wave_max_i32(lm)); // single_snake.py:210 snake_sizes

        const int o = orientation_of<CPL>(e, g, L, lds); // utils.py:36-65
        if ((long long)o == a) a += 2; // single_snake.py:221-222 (written back in place by the caller)
        a = a % 4;                     // fmod_: sign follows the dividend
    }
    const int ai = (int)(((a % 4) + 4) % 4);

    // head shift (single_snake.py:225-233 / simple_gridworld.py:149-157): by -TAP[a]; off-grid => vanishes
    const int headcell = find_head<CPL>(e);
    int newhead = -1, ny = -1, nx = -1;
    if (headcell >= 0) {
        int hy = div_size(headcell, g.rcpS), hx = headcell - hy * S;
        ny = hy - tap_y(ai);
        nx = hx - tap_x(ai);
        if (ny >= 0 && ny < S && nx >= 0 && nx < S) newhead = ny * S + nx;
    }

    bool eat_l = false;
#pragma unroll
    for (int k = 0; k < CPL; ++k) eat_l |= (lane + 64 * k == newhead) && ((e.food >> k) & 1);
    const bool EAT = ballot(eat_l) != 0; // single_snake.py:242 head_food_overlap

    bool selfc_l = false;
    u64 newbits = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int c = lane + 64 * k;
        const bool is_new = (c == newhead);
        if (SNAKE) {
            const int b0 = e.body[k];
            int b = b0;
            if (!EAT) b = max(b - 1, 0); // :246-249 decay unless food was eaten
            if (is_new) {
                selfc_l |= b > 0;        // :252 self collision (after the decay)
                b += L + (EAT ? 1 : 0);  // :258-262 new head segment
            }
            if (WRITE && b != b0) envp[2 * C + c] = (float)b;
            e.body[k] = b;
        }
        if (is_new) {
            newbits |= 1ull << k;
            if ((e.food >> k) & 1) {     // :270-272 food removal
                e.food &= ~(1ull << k);
                if (WRITE) envp[c] = 0.0f;
            }
        }
        if (WRITE && (((e.head >> k) & 1) != (u64)is_new)) envp[C + c] = is_new ? 1.0f : 0.0f;
    }
    e.head = newbits;
    const bool SELFC = SNAKE && (ballot(selfc_l) != 0);

    if (EAT) { // :277-282
        u32 word = 0;
        if (!use_inject) word = rng_words(seed, call, env_id, RNG_FOOD, 0).w[0];
        add_food<CPL, SNAKE, WRITE>(e, g, envp, use_inject, inject_cell, word);
    }

    // :290-295 edge collision: head not in the interior (on the border ring or gone)
    const bool EDGEC = !(newhead >= 0 && ny >= 1 && ny <= S - 2 && nx >= 1 && nx <= S - 2);

    out.action = a;
    out.headcell = newhead;
    out.reward = EAT ? 1.0f : 0.0f;
    out.selfc = SELFC;
    out.edgec = EDGEC;
    out.done = SELFC | EDGEC;
}

// ------------------------------------------------------------------------------------------------ reset

// _create_envs for one env (single_snake.py:344-387 / simple_gridworld.py:247-268).  inj: SNAKE {seed_y,
// seed_x, direction, food_cell}; GRID {food_cell}.
template <int CPL, bool SNAKE>
__device__ __forceinline__ void reset_core(Env<CPL> &e, const Geo &g, u64 seed, u64 call, u64 env_id,
                                           const int *__restrict__ inj, int start_y, int start_x)
{
    const int S = g.S, lane = g.lane;
    const bool use_inject = inj != nullptr;
    Words w;
    w.w[0] = w.w[1] = w.w[2] = w.w[3] = 0;
    if (!use_inject) w = rng_words(seed, call, env_id, RNG_RESET, 0);
    int hc, sc = -1, tc = -1, foodcell = -1;
    if (SNAKE) {
        int sy, sx, d;
        if (use_inject) {
            sy = inj[0]; sx = inj[1]; d = inj[2]; foodcell = inj[3];
        } else { // randint(4, S-4) twice, randint(4) (:358-359,366)
            sy = 4 + (int)mulhi_range(w.w[0], (u32)(S - 8));
            sx = 4 + (int)mulhi_range(w.w[1], (u32)(S - 8));
            d = (int)(w.w[2] >> 30);
        }
        // conv2d(seed, LENGTH_3_SNAKES[d]) (:372-376): 3 at seed + TAP[d], 2 at the seed, 1 at seed - TAP[d]
        hc = (sy + tap_y(d)) * S + sx + tap_x(d);
        sc = sy * S + sx;
        tc = (sy - tap_y(d)) * S + sx - tap_x(d);
    } else {
        hc = start_y * S + start_x; // simple_gridworld.py:262
        if (use_inject) foodcell = inj[0];
    }
    e.food = 0;
    e.head = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = lane + 64 * k;
        e.body[k] = SNAKE ? (c == hc ? 3 : c == sc ? 2 : c == tc ? 1 : 0) : 0;
        if (c == hc) e.head |= 1ull << k;
    }
    add_food<CPL, SNAKE, false>(e, g, nullptr, use_inject, foodcell, w.w[3]); // :384-385
}

// ------------------------------------------------------------------------------------------------ observations

__device__ __forceinline__ float class_rgb(int cls, int ch, bool snake)
{
    // classes: 0 background, 1 body, 2 head, 3 food, 4 border ring.  single_snake.py:99-128 paints body
    // (0,127,0), head (0,255,0), food (255,0,0) on white, ring black; simple_gridworld.py:84-109 on black.
    switch (cls) {
    case 0: return snake ? 1.0f : 0.0f;
    case 1: return ch == 1 ? 127.0f / 255.0f : 0.0f;
    case 2: return ch == 1 ? 1.0f : 0.0f;
    case 3: return ch == 0 ? 1.0f : 0.0f;
    default: return 0.0f;
    }
}

template <int CPL, bool SNAKE>
__device__ __forceinline__ int cell_class(const Env<CPL> &e, const Geo &g, int k)
{
    if (!((g.interior >> k) & 1)) return 4;
    if ((e.food >> k) & 1) return 3;
    if ((e.head >> k) & 1) return 2;
    if (SNAKE && e.body[k] > 0) return 1;
    return 0;
}

// _observe of one env (single_snake.py:130-195, simple_gridworld.py:111-133) from registers.
// headcell: the env's head cell (-1 = none).
template <int CPL, bool SNAKE>
__device__ __forceinline__ void write_obs(const Env<CPL> &e, const Geo &g, int headcell, float *__restrict__ o,
                                          int mode, int n, signed char *lds)
{
    const int S = g.S, C = g.C, lane = g.lane;
    if (mode == WURM_OBS_DEFAULT) {
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            int c = lane + 64 * k;
            if ((g.valid >> k) & 1) {
                int cls = cell_class<CPL, SNAKE>(e, g, k);
                o[c] = class_rgb(cls, 0, SNAKE);
                o[C + c] = class_rgb(cls, 1, SNAKE);
                o[2 * C + c] = class_rgb(cls, 2, SNAKE);
            }
        }
    } else if (mode == WURM_OBS_PARTIAL) {
        // (2n+1)^2 crop of the zero-padded RGB image around the head, channel-major (single_snake.py:166-193).
        // Each window cell is classified once and written to its three channel planes.
        const int W = 2 * n + 1, W2 = W * W;
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < CPL; ++k)
            if ((g.valid >> k) & 1) lds[lane + 64 * k] = (signed char)cell_class<CPL, SNAKE>(e, g, k);
        wave_lds_sync();
        const int hy = headcell >= 0 ? div_size(headcell, g.rcpS) : 0;
        const int hx = headcell - hy * S;
        const float rcpW = 1.0f / (float)W;
        for (int w = lane; w < W2; w += 64) {
            int wy = div_size(w, rcpW), wx = w - wy * W;
            int y = hy - n + wy, x = hx - n + wx;
            // F.pad zeros (single_snake.py:179); no head: zeros (the reference raises at :191)
            int cls = 4;
            if (headcell >= 0 && y >= 0 && y < S && x >= 0 && x < S) cls = lds[y * S + x];
            o[w] = class_rgb(cls, 0, SNAKE);
            o[W2 + w] = class_rgb(cls, 1, SNAKE);
            o[2 * W2 + w] = class_rgb(cls, 2, SNAKE);
        }
        wave_lds_sync();
    } else if (mode == WURM_OBS_ONE_CHANNEL) { // single_snake.py:142-151
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            int c = lane + 64 * k;
            if ((g.valid >> k) & 1) {
                float v = (e.body[k] > 0 ? 0.5f : 0.0f) + (((e.head >> k) & 1) ? 0.5f : 0.0f) +
                          (((e.food >> k) & 1) ? 1.5f : 0.0f);
                if (!((g.interior >> k) & 1)) v = -1.0f;
                o[c] = v;
            }
        }
    } else if (mode == WURM_OBS_RAW) { // clone of the state
#pragma unroll
        for (int k = 0; k < CPL; ++k) {
            int c = lane + 64 * k;
            if ((g.valid >> k) & 1) {
                o[c] = ((e.food >> k) & 1) ? 1.0f : 0.0f;
                o[C + c] = ((e.head >> k) & 1) ? 1.0f : 0.0f;
                if (SNAKE) o[2 * C + c] = (float)e.body[k];
            }
        }
    } else if (mode == WURM_OBS_POSITIONS) { // argmax of the head and food channels (first maximum; 0 if empty)
        const int fcell = first_cell(e.food, lane);
        int h = headcell < 0 ? 0 : headcell, f = fcell < 0 ? 0 : fcell;
        int hy = div_size(h, g.rcpS), fy = div_size(f, g.rcpS);
        if (lane < 4) o[lane] = (float)(lane == 0 ? hy : lane == 1 ? h - hy * S : lane == 2 ? fy : f - fy * S);
    }
}

// ------------------------------------------------------------------------------------------------ kernels

extern __shared__ __attribute__((aligned(16))) signed char wurm_lds[];

template <int CPL, bool SNAKE>
__global__ __launch_bounds__(256) void step_kernel(StepArgs p)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    if (p.only_flagged && uniform((int)p.done[env]) != (int)GRID_SKIPPED) return; // grid_step_kernel stepped this env
    signed char *lds = wurm_lds + wave * p.lds_per_wave;
    const int NCH = SNAKE ? 3 : 2;
    const Geo g = make_geo<CPL>(p.S);
    float *envp = p.envs + env * NCH * g.C;
    Env<CPL> e;
    load_state<CPL, SNAKE>(envp, g, e);
    const long long a_in = uniform64(load_action(p.actions, p.act_dtype, env));
    const bool inj = p.inject_food != nullptr;
    const int inj_cell = inj ? uniform(p.inject_food[env]) : -1;
    StepOut out;
    step_core<CPL, SNAKE, true>(e, g, envp, a_in, out, p.seed, p.call, (u64)(p.env_offset + env), inj, inj_cell, lds);
    if (g.lane == 0) {
        if (SNAKE) {
            store_action(p.actions, p.act_dtype, env, out.action);
            p.selfc[env] = (uint8_t)out.selfc;
        }
        p.reward[env] = out.reward;
        p.done[env] = (uint8_t)out.done;
        p.edgec[env] = (uint8_t)out.edgec;
        if (p.done_copy) p.done_copy[env] = (uint8_t)out.done;
    }
    if (p.obs_mode != WURM_OBS_NONE)
        write_obs<CPL, SNAKE>(e, g, out.headcell, p.obs + env * p.obs_elems, p.obs_mode, p.obs_n, lds);
}

// One launch for the caller loop `obs, r, d, info = env.step(a); env.reset(d)` (tests/test_single_snake_env.py:24-31,
// experiments/main.py:212-227), in either of two groupings:
//   * deferred reset: envs flagged in p.done_in (the `done` of the PREVIOUS step, whose reset(done) call the host
//     side postponed) are rebuilt first — exactly reset_kernel with call = p.pre_call — then every env is stepped
//     (call = p.call) and observed;
//   * immediate reset (p.post_reset): after the observation of the post-step state (:304) done envs are rebuilt
//     (call = p.call + 1) and stored, as rollout_kernel does per iteration.
// p.obs_after (nullable): what reset(done) returns — the observation of every env after done envs are rebuilt —
// written whether or not the rebuilt state is stored (the deferred reset of the next launch recreates it from the
// same counters).  p.done_copy (nullable): second copy of `done` in a buffer the caller cannot modify.
template <int CPL, bool SNAKE>
__device__ __forceinline__ void fused_step_env(const StepArgs &p, long long env, signed char *lds)
{
    const int NCH = SNAKE ? 3 : 2;
    const Geo g = make_geo<CPL>(p.S);
    float *envp = p.envs + env * NCH * g.C;
    const u64 env_id = (u64)(p.env_offset + env);
    Env<CPL> e;
    const bool pre = p.done_in != nullptr && uniform((int)p.done_in[env]) != 0;
    if (pre) {
        const int *inj = p.inject_pre_reset ? p.inject_pre_reset + env * (SNAKE ? 4 : 1) : nullptr;
        reset_core<CPL, SNAKE>(e, g, p.seed, p.pre_call, env_id, inj, p.start_y, p.start_x);
        store_state<CPL, SNAKE>(envp, g, e); // step_core then writes the cells it changes on top (same wave: in order)
    } else {
        load_state<CPL, SNAKE>(envp, g, e);
    }
    const long long a_in = uniform64(load_action(p.actions, p.act_dtype, env));
    const bool inj = p.inject_food != nullptr;
    const int inj_cell = inj ? uniform(p.inject_food[env]) : -1;
    StepOut out;
    step_core<CPL, SNAKE, true>(e, g, envp, a_in, out, p.seed, p.call, env_id, inj, inj_cell, lds);
    if (g.lane == 0) {
        if (SNAKE) {
            store_action(p.actions, p.act_dtype, env, out.action);
            p.selfc[env] = (uint8_t)out.selfc;
        }
        p.reward[env] = out.reward;
        p.done[env] = (uint8_t)out.done;
        p.edgec[env] = (uint8_t)out.edgec;
        if (p.done_copy) p.done_copy[env] = (uint8_t)out.done;
    }
    if (p.obs_mode != WURM_OBS_NONE)
        write_obs<CPL, SNAKE>(e, g, out.headcell, p.obs + env * p.obs_elems, p.obs_mode, p.obs_n, lds);
    if (!p.post_reset && p.obs_after == nullptr) return;
    int headcell = out.headcell;
    if (out.done) {
        const int *inj_r = p.inject_reset ? p.inject_reset + env * (SNAKE ? 4 : 1) : nullptr;
        reset_core<CPL, SNAKE>(e, g, p.seed, p.call + 1ull, env_id, inj_r, p.start_y, p.start_x);
        if (p.post_reset) store_state<CPL, SNAKE>(envp, g, e);
        headcell = find_head<CPL>(e);
    }
    if (p.obs_after != nullptr && p.obs_mode != WURM_OBS_NONE)
        write_obs<CPL, SNAKE>(e, g, headcell, p.obs_after + env * p.obs_elems, p.obs_mode, p.obs_n, lds);
}

template <int CPL, bool SNAKE>
__global__ __launch_bounds__(256) void fused_step_kernel(StepArgs p)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    if (p.only_flagged && uniform((int)p.done[env]) != (int)GRID_SKIPPED) return; // grid_step_kernel stepped this env
    fused_step_env<CPL, SNAKE>(p, env, wurm_lds + wave * p.lds_per_wave);
}

} // namespace wurm
#include "lane_step.hpp"
namespace wurm {

template <int CPL, bool SNAKE>
__global__ __launch_bounds__(256) void reset_kernel(StepArgs p)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    signed char *lds = wurm_lds + wave * p.lds_per_wave;
    const int NCH = SNAKE ? 3 : 2;
    const Geo g = make_geo<CPL>(p.S);
    float *envp = p.envs + env * NCH * g.C;
    Env<CPL> e;
    const bool d = uniform((int)p.done_in[env]) != 0;
    if (d) {
        const int *inj = p.inject_reset ? p.inject_reset + env * (SNAKE ? 4 : 1) : nullptr;
        reset_core<CPL, SNAKE>(e, g, p.seed, p.call, (u64)(p.env_offset + env), inj, p.start_y, p.start_x);
        store_state<CPL, SNAKE>(envp, g, e);
    } else {
        if (p.obs_mode == WURM_OBS_NONE) return;
        load_state<CPL, SNAKE>(envp, g, e);
    }
    if (p.obs_mode != WURM_OBS_NONE)
        write_obs<CPL, SNAKE>(e, g, find_head<CPL>(e), p.obs + env * p.obs_elems, p.obs_mode, p.obs_n, lds);
}

template <int CPL, bool SNAKE>
__global__ __launch_bounds__(256) void observe_kernel(StepArgs p)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    signed char *lds = wurm_lds + wave * p.lds_per_wave;
    const int NCH = SNAKE ? 3 : 2;
    const Geo g = make_geo<CPL>(p.S);
    Env<CPL> e;
    load_state<CPL, SNAKE>(p.envs + env * NCH * g.C, g, e);
    write_obs<CPL, SNAKE>(e, g, find_head<CPL>(e), p.obs + env * p.obs_elems, p.obs_mode, p.obs_n, lds);
}

// ------------------------------------------------------------------------------------------------ rollout fast path
//
// Inside a rollout nothing but this wave touches the env, so the quantities step_core re-derives from the grid on
// every call — head cell, snake length, orientation, food cell — are known wave-uniform scalars that can simply be
// carried from step to step.  Exactness (same results as step_core on the same state) needs the state to be a
// well-formed snake when the carry starts: at most one head cell, at most one food cell, exactly one body cell == L
// (under the head, if there is a head) and exactly one == L-1, L >= 2.  Then, for an env that is not done:
//   * the new head cell holds L + eat and is the unique maximum, the old head cell holds the unique maximum - 1
//     => next length = L + eat, next orientation = (action + 2) % 4 (head = neck + TAP[o] with the move being -TAP[a]);
//   * a done env (self collision / edge) is rebuilt by the reset that follows every step of a rollout.
// Any other start state runs the generic loop (step_core), which makes no assumption.
struct Fast {
    int hc, hy, hx; // head cell (-1: none) and its row / column
    int L;          // snake length = max body value
    int o;          // orientation
    int food;       // food cell (-1: none)
};

template <int CPL>
__device__ __forceinline__ bool fast_init(const Env<CPL> &e, const Geo &g, Fast &f)
{
    int lm = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) lm = max(lm, e.body[k]);
    const int counts = wave_sum_i32(__popcll(e.head) | (__popcll(e.food) << 16));
    const int nhead = counts & 0xffff, nfood = counts >> 16;
    const int hc = first_cell(e.head, g.lane), fc = first_cell(e.food, g.lane);
    const int L = wave_max_i32(lm);
    int cntL, cntN, cellL, cellN;
    top_two<CPL>(e, g, L, cntL, cntN, cellL, cellN);
    if (nhead > 1 || nfood > 1 || cntL != 1 || cntN != 1 || L < 2 || (hc >= 0 && hc != cellL)) return false;
    int yL = div_size(cellL, g.rcpS), xL = cellL - yL * g.S;
    int yN = div_size(cellN, g.rcpS), xN = cellN - yN * g.S;
    int dy = yL - yN, dx = xL - xN;
    f.o = (dy == 0 && dx == 1) ? 1 : (dy == 1 && dx == 0) ? 2 : (dy == 0 && dx == -1) ? 3 : 0; // as orientation_of
    f.hc = hc;
    f.hy = hc >= 0 ? div_size(hc, g.rcpS) : 0;
    f.hx = hc - f.hy * g.S;
    f.L = L;
    f.food = fc;
    return true;
}

// K-th free interior cell (body == 0; the head cell has body > 0 and the only food was just eaten / the grid was
// just rebuilt) in row-major order — the same choice add_food makes.  Returns the cell or -1.
template <int CPL>
__device__ __forceinline__ int fast_food_cell(const Env<CPL> &e, const Geo &g, u32 word)
{
    u64 fr = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
        if (((g.interior >> k) & 1) && e.body[k] == 0) fr |= 1ull << k;
    const int n_free = wave_sum_i32(__popcll(fr));
    if (n_free == 0) return -1;
    const int K = (int)mulhi_range(word, (u32)n_free);
    int base = 0;
#pragma unroll 1
    for (int k = 0; k < CPL; ++k) {
        const u64 m = ballot((fr >> k) & 1);
        const int cnt = popc64(m);
        if (K < base + cnt) { // the (K - base)-th set bit of m
            const u64 hit = ballot(((m >> g.lane) & 1) && rank_below(m) == K - base);
            return 64 * k + first_bit(hit);
        }
        base += cnt;
    }
    return -1;
}

// step_core with carried scalars (single_snake.py:197-304; same line references as step_core)
// a_small = the action if it is one of 0..3, else -1;  a_mod = action % 4 (C semantics: -3..3).  Both are computed
// once per 64-step tape chunk so that the per-step sanitisation is 32-bit scalar work.
template <int CPL>
__device__ __forceinline__ void fast_step(Env<CPL> &e, const Geo &g, Fast &f, int a_small, int a_mod, StepOut &out,
                                          u64 seed, u64 call, u64 env_id, bool use_inject, int inject_cell)
{
    const int S = g.S;
    const bool rev = f.o == a_small;                                      // :221-222
    const int a_out = rev ? ((f.o + 2) & 3) : a_mod;
    const int ai = a_out & 3;                                             // == ((a_out % 4) + 4) % 4 for -3..3
    int newhead = -1, ny = -1, nx = -1;
    if (f.hc >= 0) {                                                      // :225-233
        ny = f.hy - tap_y(ai);
        nx = f.hx - tap_x(ai);
        if (ny >= 0 && ny < S && nx >= 0 && nx < S) newhead = ny * S + nx;
    }
    const bool inside = newhead >= 0;
    const bool EAT = inside && newhead == f.food;                         // :242
    int sel = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) sel = (newhead >> 6) == k ? e.body[k] : sel;
    const int bnew = inside ? lane_value(sel, newhead & 63) : 0;
    const int bdec = EAT ? bnew : max(bnew - 1, 0);
    const bool SELFC = inside && bdec > 0;                                // :252
    const int grow = f.L + (EAT ? 1 : 0);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int b = e.body[k];
        if (!EAT) b = max(b - 1, 0);                                      // :246-249
        if (g.lane + 64 * k == newhead) b += grow;                        // :258-262
        e.body[k] = b;
    }
    const bool EDGEC = !(inside && ny >= 1 && ny <= S - 2 && nx >= 1 && nx <= S - 2); // :290-295
    f.hc = newhead; f.hy = ny; f.hx = nx;
    f.L = grow;
    f.o = (ai + 2) & 3;
    if (EAT) {                                                            // :270-282
        if (use_inject) f.food = (inject_cell >= 0 && inject_cell < g.C) ? inject_cell : -1;
        else f.food = fast_food_cell<CPL>(e, g, rng_words(seed, call, env_id, RNG_FOOD, 0).w[0]);
    }
    out.action = a_out;
    out.headcell = newhead;
    out.reward = EAT ? 1.0f : 0.0f;
    out.selfc = SELFC;
    out.edgec = EDGEC;
    out.done = SELFC | EDGEC;
}

// reset_core with carried scalars (single_snake.py:344-387)
template <int CPL>
__device__ __forceinline__ void fast_reset(Env<CPL> &e, const Geo &g, Fast &f, u64 seed, u64 call, u64 env_id,
                                           const int *__restrict__ inj)
{
    const int S = g.S;
    Words w;
    w.w[0] = w.w[1] = w.w[2] = w.w[3] = 0;
    int sy, sx, d, fc = -1;
    if (inj) {
        sy = inj[0]; sx = inj[1]; d = inj[2]; fc = inj[3];
        if (fc >= g.C) fc = -1;
    } else {
        w = rng_words(seed, call, env_id, RNG_RESET, 0);
        sy = 4 + (int)mulhi_range(w.w[0], (u32)(S - 8));
        sx = 4 + (int)mulhi_range(w.w[1], (u32)(S - 8));
        d = (int)(w.w[2] >> 30);
    }
    const int hy = sy + tap_y(d), hx = sx + tap_x(d);
    const int hc = hy * S + hx, sc = sy * S + sx, tc = (sy - tap_y(d)) * S + sx - tap_x(d);
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        e.body[k] = c == hc ? 3 : c == sc ? 2 : c == tc ? 1 : 0;
    }
    f.hc = hc; f.hy = hy; f.hx = hx;
    f.L = 3;
    f.o = d;
    f.food = inj ? fc : fast_food_cell<CPL>(e, g, w.w[3]);
}

// food / head bit sets of the Env from the carried scalars (for the generic observation writer and store_state)
template <int CPL>
__device__ __forceinline__ void fast_sync_bits(Env<CPL> &e, const Geo &g, const Fast &f)
{
    e.food = 0;
    e.head = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        if (c == f.food) e.food |= 1ull << k;
        if (c == f.hc) e.head |= 1ull << k;
    }
}

// Per-lane geometry of the partial_n crop, computed once per kernel: lane owns window cells w = lane + 64*i.
constexpr int CROP_NI = 3; // (2n+1)^2 <= 192, i.e. n <= 6
struct Crop {
    int W2;
    int dy[CROP_NI], dx[CROP_NI]; // window cell offset from the head; dy = INT_MIN/2 marks "no such cell"
};

__device__ __forceinline__ Crop make_crop(int lane, int n)
{
    Crop c;
    const int W = 2 * n + 1;
    c.W2 = W * W;
    const float rcpW = 1.0f / (float)W;
#pragma unroll
    for (int i = 0; i < CROP_NI; ++i) {
        int w = lane + 64 * i;
        int wy = div_size(w, rcpW), wx = w - wy * W;
        c.dy[i] = w < c.W2 ? wy - n : -(1 << 20);
        c.dx[i] = wx - n;
    }
    return c;
}

// partial_n crop for grids of at most 128 cells (single_snake.py:166-193): body occupancy as two ballot masks, the
// head and food cells as scalars — no LDS.  A window cell that is off the grid, on the border ring, or seen from
// an env without a head is (0,0,0); otherwise food (1,0,0), head (0,1,0), body (0,127/255,0), background (1,1,1).
template <int CPL>
__device__ __forceinline__ void fast_partial_small(const Env<CPL> &e, const Geo &g, const Fast &f,
                                                   float *__restrict__ o, const Crop &cg, float *lds_copy = nullptr)
{
    static_assert(CPL <= 2, "ballot-mask crop needs <= 128 cells");
    const int S = g.S, W2 = cg.W2;
    const u64 m0 = ballot(e.body[0] > 0), m1 = CPL > 1 ? ballot(e.body[CPL - 1] > 0) : 0;
    const bool has_head = f.hc >= 0;
#pragma unroll
    for (int i = 0; i < CROP_NI; ++i) {
        if (64 * i >= W2) break;
        const int y = f.hy + cg.dy[i], x = f.hx + cg.dx[i];
        if (cg.dy[i] <= -(1 << 19)) continue;
        const bool live = has_head && (unsigned)(y - 1) < (unsigned)(S - 2) && (unsigned)(x - 1) < (unsigned)(S - 2);
        const int cell = y * S + x;
        const bool fd = cell == f.food, hd = cell == f.hc;
        const bool bd = (((cell < 64 ? m0 : m1) >> (cell & 63)) & 1) != 0;
        const float bg = (live && !fd && !hd && !bd) ? 1.0f : 0.0f;
        const float r = (live && fd) ? 1.0f : bg;
        const float gr = (live && !fd) ? (hd ? 1.0f : (bd ? 127.0f / 255.0f : bg)) : 0.0f;
        const int w = g.lane + 64 * i;
        o[w] = r;
        o[W2 + w] = gr;
        o[2 * W2 + w] = bg;
        if (lds_copy) { // the same observation for a consumer inside the kernel (policy_rollout.hpp)
            lds_copy[w] = r;
            lds_copy[W2 + w] = gr;
            lds_copy[2 * W2 + w] = bg;
        }
    }
}

// T fused step+reset iterations with the env resident in registers.  Lane j of the wave buffers the
// per-step scalars of step t0+j; they are flushed every 64 steps.
// OBSK >= 0 fixes the observation mode at compile time and INJ = false compiles the injection plumbing out: the
// flagship configuration (9x9, partial_n / no observation, RNG mode) gets a lean instantiation, everything else the
// fully general one (OBSK = -1, INJ = true).
template <int CPL, bool SNAKE, int OBSK, bool INJ>
__device__ __forceinline__ void rollout_generic(const StepArgs &p, long long env, float *__restrict__ envp, const Geo &g,
                                                Env<CPL> &e, signed char *lds)
{
    const u64 env_id = (u64)(p.env_offset + env);
    const bool inj_f = INJ && p.inject_food != nullptr, inj_r = INJ && p.inject_reset != nullptr;
    const int obs_mode = OBSK >= 0 ? OBSK : p.obs_mode;
    Fast f = {-1, 0, 0, 0, 0, -1};
    bool fast = false;
    if (SNAKE) fast = fast_init<CPL>(e, g, f);
    const bool small_crop = SNAKE && CPL <= 2 && obs_mode == WURM_OBS_PARTIAL && p.obs_n <= 6;
    const Crop cg = make_crop(g.lane, small_crop ? p.obs_n : 0);
    const long long obs_stride = p.N * p.obs_elems;
    float *obs_t = p.obs + env * p.obs_elems; // observation of step t; advanced by obs_stride per step
    u64 call = p.call;                        // step t uses call0 + 2t, its reset call0 + 2t + 1

    for (long long t0 = 0; t0 < p.T; t0 += 64) {
        const int nt = (int)min((long long)64, p.T - t0);
        const long long my_t = t0 + g.lane;
        long long my_a = g.lane < nt ? load_action(p.actions, p.act_dtype, my_t * p.N + env) : 0;
        int my_inj = (inj_f && g.lane < nt) ? p.inject_food[my_t * p.N + env] : -1;
        int my_flags = 0; // bit 0 done, 1 self collision, 2 edge collision, 3 reward
        // Retire the two prefetch loads HERE.  Otherwise the compiler, seeing a register that may still be in flight
        // on loop entry, puts `s_waitcnt vmcnt(0)` in front of the per-step readlane, and every step then also waits
        // for the previous step's observation stores to be acknowledged by HBM (vmcnt counts loads and stores).
        asm volatile("" : "+v"(my_a), "+v"(my_inj));
        const int my_small = (my_a >= 0 && my_a < 4) ? (int)my_a : -1, my_mod = (int)(my_a % 4);
        int my_out = 0; // sanitised action of step t0 + lane (always in -3..3)
        for (int j = 0; j < nt; ++j, obs_t += obs_stride, call += 2) {
            const int inj_cell = INJ ? lane_value(my_inj, j) : -1;
            StepOut out;
            if (SNAKE && fast) {
                fast_step<CPL>(e, g, f, lane_value(my_small, j), lane_value(my_mod, j), out, p.seed, call, env_id, inj_f,
                               inj_cell);
                if (small_crop) {
                    if constexpr (CPL <= 2) fast_partial_small<CPL>(e, g, f, obs_t, cg);
                } else if (obs_mode != WURM_OBS_NONE) {
                    fast_sync_bits<CPL>(e, g, f);
                    write_obs<CPL, SNAKE>(e, g, f.hc, obs_t, obs_mode, p.obs_n, lds);
                }
                if (out.done)
                    fast_reset<CPL>(e, g, f, p.seed, call + 1ull, env_id,
                                    inj_r ? p.inject_reset + ((t0 + j) * p.N + env) * 4 : nullptr);
            } else {
                const long long a_in = lane_value64(my_a, j);
                step_core<CPL, SNAKE, false>(e, g, nullptr, a_in, out, p.seed, call, env_id, inj_f, inj_cell, lds);
                if (obs_mode != WURM_OBS_NONE)
                    write_obs<CPL, SNAKE>(e, g, out.headcell, obs_t, obs_mode, p.obs_n, lds);
                if (out.done) {
                    const int *inj = inj_r ? p.inject_reset + ((t0 + j) * p.N + env) * (SNAKE ? 4 : 1) : nullptr;
                    reset_core<CPL, SNAKE>(e, g, p.seed, call + 1ull, env_id, inj, p.start_y, p.start_x);
                }
            }
            if (g.lane == j) {
                my_out = (int)out.action;
                my_flags = out.done | (out.selfc << 1) | (out.edgec << 2) | (out.reward != 0.0f ? 8 : 0);
            }
        }
        if (g.lane < nt) {
            const long long i = my_t * p.N + env;
            if (SNAKE) {
                store_action(p.actions, p.act_dtype, i, (long long)my_out);
                p.selfc[i] = (uint8_t)((my_flags >> 1) & 1);
            }
            p.reward[i] = (my_flags & 8) ? 1.0f : 0.0f;
            p.done[i] = (uint8_t)(my_flags & 1);
            p.edgec[i] = (uint8_t)((my_flags >> 2) & 1);
        }
    }
    if (SNAKE && fast) fast_sync_bits<CPL>(e, g, f);
    store_state<CPL, SNAKE>(envp, g, e);
}

template <int CPL, bool SNAKE, int OBSK = -1, bool INJ = true>
__global__ __launch_bounds__(256) void rollout_kernel(StepArgs p)
{
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    if (p.only_flagged && uniform((int)p.done[env]) != (int)GRID_SKIPPED) return; // the grid kernel rolled this env out
    signed char *lds = wurm_lds + wave * p.lds_per_wave;
    const Geo g = make_geo<CPL>(p.S);
    float *envp = p.envs + env * (SNAKE ? 3 : 2) * g.C;
    Env<CPL> e;
    load_state<CPL, SNAKE>(envp, g, e);
    rollout_generic<CPL, SNAKE, OBSK, INJ>(p, env, envp, g, e, lds);
}

// ---- the envs a grid / lane kernel left to the one-env-per-wave code (done[env] == GRID_SKIPPED; for a rollout the flag is in
// done[0][env]): ONE WAVE PER 64 ENVS reads their flags — one coalesced load, one ballot — and serves the flagged ones in
// turn.  Launching the one-env-per-wave kernels for EVERY env just to read its flag (only_flagged, rounds 3-5) cost 3.9 us at
// 8 192 envs and 6.8-9 us at 65 536: a fifth of the per-call step of 8 192 x 36 x 36 that it stood behind.
// ROLL: rollout_kernel's body; else fused_step_env (which is step_kernel's body when the call carries no reset).
template <int CPL, bool SNAKE, bool ROLL>
__global__ __launch_bounds__(256) void flagged_kernel(StepArgs p)
{
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);
    const long long base = ((long long)blockIdx.x * wpb + wave) * 64;
    if (base >= p.N) return;
    const bool flag = base + lane < p.N && p.done[base + lane] == GRID_SKIPPED;
    signed char *lds = wurm_lds + wave * p.lds_per_wave;
    for (u64 m = ballot(flag); m != 0; m &= m - 1) {
        const long long env = base + first_bit(m);
        if constexpr (ROLL) {
            const Geo g = make_geo<CPL>(p.S);
            float *envp = p.envs + env * (SNAKE ? 3 : 2) * g.C;
            Env<CPL> e;
            load_state<CPL, SNAKE>(envp, g, e);
            rollout_generic<CPL, SNAKE, -1, true>(p, env, envp, g, e, lds);
        } else {
            fused_step_env<CPL, SNAKE>(p, env, lds);
        }
        wave_lds_sync(); // (the next env reuses the wave's LDS)
    }
}

// ---------------------------------------------------------------------------------------------- lean rollout
// The headline shape — SingleSnake on a grid of at most 128 cells (S <= 11), partial_n crop of at most 64 window
// cells (n <= 3) or no observation, RNG mode — runs 512 envs as 512 lone waves on 1024 SIMDs: nothing hides
// latency, every instruction costs its full issue slot and every taken branch an instruction-fetch bubble, so the
// loop below is written for instruction count (PMC: 205 -> ~95 instructions per env-step).  On top of the scalar
// carry of `Fast` it uses
//   * a virtual clock for the body channel: cell value = max(expire - T, 0).  A step that does not eat advances T
//     (= "every body cell decays by one", single_snake.py:246-249) and a step that eats leaves it alone; only the new
//     head cell is written (expire = T + L), so the decay costs no per-cell work;
//   * everything a reset needs (single_snake.py:344-387: seed cell, direction, the three body cells, the food cell)
//     depends only on (seed, env, call), never on the state: lane j of the wave computes the would-be reset of step
//     t0+j — both Philox blocks included — once per 64-step chunk, in parallel, and a reset is two readlanes;
//   * the food draw of an eating step (purpose RNG_FOOD) is precomputed the same way; only the rank select over the
//     free cells, which does depend on the state, runs when food is eaten;
//   * observation masks combined as 64-bit scalar lane masks; lanes past the window duplicate its last cell
//     instead of being masked off.
// Preconditions (else the kernel runs rollout_generic): fast_init's well-formed snake, a head strictly inside the
// border ring, and no food on a body cell.  They are preserved by step + reset, so they are checked once.
// A state that self-collided is rebuilt by the reset of the same iteration, so the value under the new head is not
// accumulated (only its occupancy is observable, through the crop).

struct LeanReset {
    int a; // hy | hx << 4 | d << 8 | food cell << 10
    int b; // head cell | seed cell << 7 | tail cell << 14
};

// would-be reset of (env, call): reset_core / fast_reset in closed form.  After a rebuild the free interior cells
// are the (S-2)^2 interior cells minus the three collinear snake cells, so the K-th free cell in row-major order is
// the K-th interior cell pushed past the snake cells' interior ranks in ascending order.
__device__ __forceinline__ LeanReset lean_reset_draw(u64 seed, u64 call, u64 env_id, int S, float rcpSm2)
{
    const Words w = rng_words(seed, call, env_id, RNG_RESET, 0);
    const int Sm2 = S - 2;
    const int sy = 4 + (int)mulhi_range(w.w[0], (u32)(S - 8));
    const int sx = 4 + (int)mulhi_range(w.w[1], (u32)(S - 8));
    const int d = (int)(w.w[2] >> 30);
    const int ty = tap_y(d), tx = tap_x(d);
    const int hy = sy + ty, hx = sx + tx;
    const int rs = (sy - 1) * Sm2 + sx - 1, dr = ty * Sm2 + tx; // interior rank of the seed cell; head = rs + dr
    const int lo = rs - abs(dr), hi = rs + abs(dr);
    int K = (int)mulhi_range(w.w[3], (u32)(Sm2 * Sm2 - 3));
    K += K >= lo;
    K += K >= rs;
    K += K >= hi;
    const int qy = div_size(K, rcpSm2), qx = K - qy * Sm2;
    const int food = (qy + 1) * S + qx + 1;
    const int sc = sy * S + sx, dc = ty * S + tx;
    LeanReset r;
    r.a = hy | (hx << 4) | (d << 8) | (food << 10);
    r.b = (sc + dc) | (sc << 7) | ((sc - dc) << 14);
    return r;
}

// K-th free interior cell (K = mulhi(word, n_free)) given the occupancy masks of cells 0..63 / 64..127; -1 if none
__device__ __forceinline__ int lean_food_cell(u64 m0, u64 m1, u64 int0, u64 int1, u32 word, int lane)
{
    const u64 F0 = int0 & ~m0, F1 = int1 & ~m1;
    const int n0 = popc64(F0), n_free = n0 + popc64(F1);
    if (n_free == 0) return -1;
    const int K = (int)mulhi_range(word, (u32)n_free);
    const bool second = K >= n0;
    const u64 F = second ? F1 : F0;
    const int K2 = second ? K - n0 : K;
    const u64 hit = ballot((int)((F >> lane) & 1) & (int)(rank_below(F) == K2));
    return (second ? 64 : 0) + first_bit(hit);
}

// lane mask of a per-lane predicate (folds with the compares / logic that produce it; __ballot goes through an int)
__device__ __forceinline__ u64 lane_mask(bool b) { return __builtin_amdgcn_ballot_w64(b); }

template <int OBSK, bool COMPACT>
__global__ __launch_bounds__(256) void rollout_lean_kernel(StepArgs p)
{
    constexpr int CPL = 2;
    static_assert(OBSK == WURM_OBS_PARTIAL || OBSK == WURM_OBS_NONE, "lean rollout: partial_n or no observation");
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const Geo g = make_geo<CPL>(p.S);
    const int S = g.S, Sm2 = S - 2, lane = g.lane;
    float *envp = p.envs + env * 3 * g.C;
    Env<CPL> e;
    load_state<CPL, true>(envp, g, e);

    Fast f = {-1, 0, 0, 0, 0, -1};
    bool lean = fast_init<CPL>(e, g, f);
    if (lean) {
        const bool head_in = f.hc >= 0 && (unsigned)(f.hy - 1) < (unsigned)Sm2 && (unsigned)(f.hx - 1) < (unsigned)Sm2;
        int under_food = 0;
        if (f.food >= 0) under_food = lane_value(f.food >= 64 ? e.body[1] : e.body[0], f.food & 63);
        lean = head_in && under_food == 0;
    }
    if (!uniform((int)lean)) {
        rollout_generic<CPL, true, OBSK, false>(p, env, envp, g, e, wurm_lds + wave * p.lds_per_wave);
        return;
    }

    const u64 env_id = (u64)(p.env_offset + env);
    const u64 int0 = ballot((g.interior & 1) != 0), int1 = ballot((g.interior & 2) != 0);
    const float rcpSm2 = 1.0f / (float)Sm2;
    // carried scalars: head row - 1, head column - 1, head cell, length, orientation * 16, food cell
    int hy1 = uniform(f.hy) - 1, hx1 = uniform(f.hx) - 1, hc = uniform(f.hc), L = uniform(f.L), o16 = uniform(f.o) << 4;
    int food = uniform(f.food);
    int foodc = food - (COMPACT ? S + 1 : 0); // the food cell in the crop's cell numbering
    int G = L;                                // G = T + L: the expiry clock a new head cell gets; +1 every step
    int ex0 = e.body[0], ex1 = e.body[1];     // expiry clock of cells lane, lane + 64
    // cells whose entry makes a step "eventful" besides body cells: the border ring (edge collision) and the food cell
    const u64 ring0 = lane_mask((g.valid & 1) != 0) & ~int0, ring1 = lane_mask((g.valid & 2) != 0) & ~int1;
    u64 X0 = ring0 | ((unsigned)food < 64u ? 1ull << food : 0), X1 = ring1 | (food >= 64 ? 1ull << (food - 64) : 0);

    // partial_n crop: lane owns window cell w (lanes past the window repeat its last cell: same address, same value).
    // COMPACT (S <= 9): the interior cells S+1 .. S*S-S-2 fit one 64-bit occupancy mask M with bit 63 to spare;
    // bit 63 is kept set and the centre lane (the head cell itself) looks at it, the others at head cell + offset.
    const int n = OBSK == WURM_OBS_PARTIAL ? p.obs_n : 0, W = 2 * n + 1, W2 = W * W;
    const int w = min(lane, W2 - 1), wy = div_size(w, 1.0f / (float)W), wx = w - wy * W;
    const int dy0 = wy - n, dx0 = wx - n;                                // window row / column offset from the head
    const bool centre = w == n * W + n;
    const int cell_d = dy0 * S + dx0 - (COMPACT ? S + 1 : 0);           // crop cell = head cell + cell_d
    const float green = centre ? 1.0f : 127.0f / 255.0f;                 // what an occupied cell shows in channel 1
    const u32 off_r = (u32)w * 4u, off_g = (u32)(W2 + w) * 4u, off_b = (u32)(2 * W2 + w) * 4u;
    const long long obs_stride = p.N * p.obs_elems;
    float *obs_t = p.obs + env * p.obs_elems;

    for (long long t0 = 0; t0 < p.T; t0 += 64) {
        const int nt = (int)min((long long)64, p.T - t0);
        const long long my_t = t0 + lane;
        long long my_a = lane < nt ? load_action(p.actions, p.act_dtype, my_t * p.N + env) : 0;
        asm volatile("" : "+v"(my_a)); // retire the load here, not in front of the first readlane of the step loop
        // Moves of step t0 + lane for each of the four orientations the snake may have by then, 16 bits each:
        // sanitised action (single_snake.py:221-222) & 7 | next orientation << 4 | (row step & 3) << 6
        // | (column step & 3) << 8 | (cell step & 63) << 10, the step being -TAP[action] (:225-233).  The step loop
        // reads both words with readlanes that do not depend on the state (so they are off its critical path) and
        // picks one with the carried orientation.
        int my_mov01, my_mov23;
        {
            const bool in_range = my_a >= 0 && my_a < 4;
            const int a_small = in_range ? (int)my_a : 7, a_mod = (int)(my_a % 4);
            int ent[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int a_out = o == a_small ? (o ^ 2) : a_mod;
                const int ai = a_out & 3, dy = -tap_y(ai), dx = -tap_x(ai);
                ent[o] = (a_out & 7) | ((ai ^ 2) << 4) | ((dy & 3) << 6) | ((dx & 3) << 8) | (((dy * S + dx) & 63) << 10);
            }
            my_mov01 = ent[0] | (ent[1] << 16);
            my_mov23 = ent[2] | (ent[3] << 16);
        }
        const u64 my_call = p.call + 2ull * (u64)my_t; // step t uses call0 + 2t, its reset call0 + 2t + 1
        const LeanReset my_reset = lean_reset_draw(p.seed, my_call + 1ull, env_id, S, rcpSm2);
        const int my_food = (int)rng_words(p.seed, my_call, env_id, RNG_FOOD, 0).w[0];
        // what lane j keeps of step t0 + j: its move entry, whether it ate, self collision | edge collision << 1
        int my_rec = 0, my_ate = 0, my_fl = 0;
        {   // re-base the clocks so that they cannot overflow however long the tape is
            const int T = G - L;
            ex0 = max(ex0 - T, 0);
            ex1 = max(ex1 - T, 0);
            G = L;
        }

        for (int j = 0; j < nt; ++j) {
            // ---- step (single_snake.py:197-304; same line references as step_core / fast_step)
            const int m01 = lane_value(my_mov01, j), m23 = lane_value(my_mov23, j);
            const int ent = ((o16 & 32) ? m23 : m01) >> (o16 & 16);
            o16 = ent & 48;
            hy1 += (ent << 24) >> 30;                // the head is off the border ring: the move stays on the grid
            hx1 += (ent << 22) >> 30;
            hc += (ent << 16) >> 26;
            G += 1;
            // L += (head cell == food cell)  (:242; spelled out: the compiler detours through a 64-bit lane mask)
            asm("s_cmp_eq_u32 %1, %2\n\ts_addc_u32 %0, %0, 0" : "+s"(L) : "s"(hc), "s"(food) : "scc");
            const int T = G - L;                     // :246-249: the clock stands still on the step that eats
            const u64 p0 = lane_mask(ex0 > T), p1 = lane_mask(ex1 > T); // body after the decay, before the head is written
            ex0 = lane == hc ? G : ex0;              // :258-262
            ex1 = lane + 64 == hc ? G : ex1;
            // One test for everything that is not a plain move: the head entered a body cell (:252), the border ring
            // (:290-295) or the food cell (:242).
            const u64 special = hc >= 64 ? (p1 | X1) : (p0 | X0);
            bool finished = false, selfc = false, edgec = false;
            if (__builtin_expect(((special >> (hc & 63)) & 1) != 0, 0)) {
                if (hc == food) {                    // :270-282
                    my_ate = lane == j ? 1 : my_ate;
                    const u64 head0 = (unsigned)hc < 64u ? 1ull << hc : 0, head1 = hc >= 64 ? 1ull << (hc - 64) : 0;
                    food = lean_food_cell(p0 | head0, p1 | head1, int0, int1, (u32)lane_value(my_food, j), lane);
                    foodc = food - (COMPACT ? S + 1 : 0);
                    X0 = ring0 | ((unsigned)food < 64u ? 1ull << food : 0);
                    X1 = ring1 | (food >= 64 ? 1ull << (food - 64) : 0);
                }
                selfc = (((hc >= 64 ? p1 : p0) >> (hc & 63)) & 1) != 0;
                edgec = max((unsigned)hy1, (unsigned)hx1) >= (unsigned)Sm2;
                finished = selfc | edgec;
            }

            // ---- observation of the stepped state (single_snake.py:166-193): a window cell that is off the grid or
            // on the border ring is (0,0,0); food (1,0,0), head (0,1,0), body (0,127/255,0), background (1,1,1).
            // The class logic stays in the VALU (compare -> select through vcc): every detour of a lane mask through
            // scalar logic and back costs a lone wave about two extra issue slots.
            if (OBSK == WURM_OBS_PARTIAL) {
                const unsigned off_grid = max((unsigned)(hy1 + dy0), (unsigned)(hx1 + dx0)); // < S-2: a live cell
                const int cell = (COMPACT && centre) ? 63 : hc + cell_d;
                unsigned occ;
                if (COMPACT) {
                    const u64 M = (p0 >> (S + 1)) | (p1 << (63 - S)) | (1ull << 63);
                    u64 sh; // M >> cell; spelled out: the compiler prefers (1 << cell) & M, two 64-bit VALU ops more
                    asm("v_lshrrev_b64 %0, %1, %2" : "=v"(sh) : "v"(cell), "s"(M));
                    occ = (u32)sh & 1u;
                } else {
                    occ = ((u32)((cell < 64 ? p0 : p1) >> (cell & 63)) & 1u) | (centre ? 1u : 0u);
                }
                const float vr = (off_grid | (occ << 30)) < (unsigned)Sm2 ? 1.0f : 0.0f;          // live and free
                const float vb = cell == foodc ? 0.0f : vr;                                       // ... and not food
                const float vg = (off_grid | ((occ ^ 1u) << 30)) < (unsigned)Sm2 ? green : vb;    // live and occupied
                // scalar base + 32-bit lane offset form, spelled out: the compiler hoists the zero-extension of the
                // lane offsets out of the loop and then pays a 64-bit VALU add per store.  (Untracked stores are
                // harmless for its vmcnt bookkeeping: nothing is read back and waits only become conservative.)
                asm volatile("global_store_dword %0, %1, %2" : : "v"(off_r), "v"(vr), "s"(obs_t) : "memory");
                asm volatile("global_store_dword %0, %1, %2" : : "v"(off_g), "v"(vg), "s"(obs_t) : "memory");
                asm volatile("global_store_dword %0, %1, %2" : : "v"(off_b), "v"(vb), "s"(obs_t) : "memory");
                obs_t += obs_stride;
            }
            my_rec = lane == j ? ent : my_rec;

            // ---- reset of a finished env (single_snake.py:322-387)
            if (__builtin_expect(finished, 0)) {
                my_fl = lane == j ? (selfc ? 1 : 0) | (edgec ? 2 : 0) : my_fl;
                const int ra = lane_value(my_reset.a, j), rb = lane_value(my_reset.b, j);
                hy1 = (ra & 15) - 1; hx1 = ((ra >> 4) & 15) - 1; o16 = ((ra >> 8) & 3) << 4; food = ra >> 10;
                foodc = food - (COMPACT ? S + 1 : 0);
                X0 = ring0 | ((unsigned)food < 64u ? 1ull << food : 0);
                X1 = ring1 | (food >= 64 ? 1ull << (food - 64) : 0);
                hc = rb & 127;
                const int sc = (rb >> 7) & 127, tc = rb >> 14;
                const int c1 = lane + 64;
                ex0 = lane == tc ? T + 1 : 0; ex0 = lane == sc ? T + 2 : ex0; ex0 = lane == hc ? T + 3 : ex0;
                ex1 = c1 == tc ? T + 1 : 0;   ex1 = c1 == sc ? T + 2 : ex1;   ex1 = c1 == hc ? T + 3 : ex1;
                L = 3;
                G = T + 3;
            }
        }
        if (lane < nt) {
            const long long i = my_t * p.N + env;
            store_action(p.actions, p.act_dtype, i, (long long)((my_rec << 29) >> 29));
            p.reward[i] = my_ate ? 1.0f : 0.0f;
            p.done[i] = (uint8_t)(my_fl != 0);
            p.selfc[i] = (uint8_t)(my_fl & 1);
            p.edgec[i] = (uint8_t)(my_fl >> 1);
        }
    }
    const int T = G - L;
    e.body[0] = max(ex0 - T, 0);
    e.body[1] = max(ex1 - T, 0);
    f.hc = hc; f.hy = hy1 + 1; f.hx = hx1 + 1; f.L = L; f.o = o16 >> 4; f.food = food;
    fast_sync_bits<CPL>(e, g, f);
    store_state<CPL, true>(envp, g, e);
}

// ------------------------------------------------------------------------------------------- 9 x 9 rollout
// The reference's default grid (size 9: 7 x 7 interior cells) gets one more specialisation of the lean loop.
// Cells are numbered code = 8 * row + column.  The 49 interior cells have distinct codes in 9..63, so ONE lane per
// interior cell holds the whole body channel (one expiry clock per lane, one 64-bit occupancy mask); the border
// ring's codes alias only each other, also modulo 64 (column 8 of a row = column 0 of the next, row 8 = row 0), so
//   * "the head entered the ring, a body cell or the food cell" is one bit test on (occupancy | RING | food bit);
//   * which window cells of the crop are inside the ring is a per-lane constant 64-bit table indexed by the head
//     code (valid while the head is inside the ring; the final observation of an env whose head is on the ring is
//     computed from row / column arithmetic on the rare path).
// Extra preconditions (else rollout_generic): no body and no food on the ring.

struct S9Reset {
    int a; // orientation | food code << 2
    int b; // head code | seed code << 7 | tail code << 14
};

// lean_reset_draw in code numbering (single_snake.py:344-387)
__device__ __forceinline__ S9Reset s9_reset_draw(u64 seed, u64 call, u64 env_id)
{
    const Words w = rng_words(seed, call, env_id, RNG_RESET, 0);
    const int sy = 4 + (int)mulhi_range(w.w[0], 1u), sx = 4 + (int)mulhi_range(w.w[1], 1u); // S - 8 = 1
    const int d = (int)(w.w[2] >> 30);
    const int ty = tap_y(d), tx = tap_x(d);
    const int rs = (sy - 1) * 7 + sx - 1, dr = ty * 7 + tx; // interior rank of the seed cell; head = rs + dr
    const int lo = rs - abs(dr), hi = rs + abs(dr);
    int K = (int)mulhi_range(w.w[3], 46u); // 49 interior cells - 3 snake cells
    K += K >= lo;
    K += K >= rs;
    K += K >= hi;
    const int qy = div_size(K, 1.0f / 7.0f), qx = K - qy * 7;
    const int sc = sy * 8 + sx, dc = ty * 8 + tx;
    S9Reset r;
    r.a = d | (((qy + 1) * 8 + qx + 1) << 2);
    r.b = (sc + dc) | (sc << 7) | ((sc - dc) << 14);
    return r;
}

// v = value in the lanes of `lanes`, unchanged elsewhere — with the lane mask taken from an SGPR pair as it is (the
// compiler has no way to say that; `lane == j` costs a VALU compare and drags the scalar j into a VGPR)
__device__ __forceinline__ int keep_in_lane(int v, int value, u64 lanes)
{
    asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v) : "v"(value), "s"(lanes));
    return v;
}

// INJ: the random outcomes (food cell of an eating step, seed cell / direction / food cell of a reset) come from
// p.inject_food / p.inject_reset instead of Philox, so that the tapes recorded from the reference (tests/golden) run
// through THIS kernel; the RNG instantiation is compiled without any of it.
template <int OBSK, bool INJ = false>
__global__ __launch_bounds__(256) void rollout_s9_kernel(StepArgs p)
{
    constexpr int CPL = 2, S = 9;
    static_assert(OBSK == WURM_OBS_PARTIAL || OBSK == WURM_OBS_NONE, "9x9 rollout: partial_n or no observation");
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const Geo g = make_geo<CPL>(S);
    const int lane = g.lane;
    float *envp = p.envs + env * 3 * (S * S);
    Env<CPL> e;
    load_state<CPL, true>(envp, g, e);

    Fast f = {-1, 0, 0, 0, 0, -1};
    bool lean = fast_init<CPL>(e, g, f);
    if (lean) {
        const bool head_in = f.hc >= 0 && (unsigned)(f.hy - 1) < 7u && (unsigned)(f.hx - 1) < 7u;
        const int fy = f.food >= 0 ? div_size(f.food, g.rcpS) : 1, fx = f.food >= 0 ? f.food - fy * S : 1;
        const bool food_in = (unsigned)(fy - 1) < 7u && (unsigned)(fx - 1) < 7u;
        int under_food = 0;
        if (f.food >= 0) under_food = lane_value(f.food >= 64 ? e.body[1] : e.body[0], f.food & 63);
        const bool ring_body = lane_mask((e.body[0] > 0 && !(g.interior & 1)) || (e.body[1] > 0 && !(g.interior & 2))) != 0;
        lean = head_in && food_in && under_food == 0 && !ring_body;
    }
    if (!uniform((int)lean)) {
        rollout_generic<CPL, true, OBSK, INJ>(p, env, envp, g, e, wurm_lds + wave * p.lds_per_wave);
        return;
    }

    const u64 env_id = (u64)(p.env_offset + env);
    // code layout: lane = 8 * row + column holds that cell if it is inside the ring
    const int ly = lane >> 3, lx = lane & 7;
    const bool lane_in = ly >= 1 && lx >= 1;
    const int my_cell = ly * S + lx;
    const u64 RING = ~lane_mask(lane_in);
    int ex = lane_in ? __float2int_rn(envp[2 * S * S + my_cell]) : 0; // expiry clock of the lane's cell (body, re-read)
    // carried scalars: head code, length, orientation * 16, food code (-1: none), G = clock + length
    int c = uniform(f.hy) * 8 + uniform(f.hx), L = uniform(f.L), o16 = uniform(f.o) << 4;
    int foodc = -1;
    if (f.food >= 0) {
        const int fy = uniform(div_size(f.food, g.rcpS)); // (float arithmetic: a VALU result, back to an SGPR)
        foodc = fy * 8 + (uniform(f.food) - fy * S);
    }
    u64 XF = RING | (foodc >= 0 ? 1ull << foodc : 0); // ring + food: the non-body cells that make a step eventful
    int G = L;

    // partial_n crop: lane owns window cell w (lanes past the window repeat its last cell: same address, same value)
    const int n = OBSK == WURM_OBS_PARTIAL ? p.obs_n : 0, W = 2 * n + 1, W2 = W * W;
    const int w = min(lane, W2 - 1), wy = div_size(w, 1.0f / (float)W), wx = w - wy * W;
    const int dy0 = wy - n, dx0 = wx - n;      // window row / column offset from the head
    const int code_d = dy0 * 8 + dx0;          // code of the window cell = head code + code_d
    const float green = (w == n * W + n) ? 1.0f : 127.0f / 255.0f; // what an occupied cell shows in channel 1
    u64 live_tab = 0;                          // bit (8 * row + column): with the head there, this window cell is inside the ring
#pragma unroll
    for (int y = 1; y <= 7; ++y) {
        u32 cols = 0;
#pragma unroll
        for (int x = 1; x <= 7; ++x)
            if ((unsigned)(x + dx0 - 1) < 7u) cols |= 1u << x;
        if ((unsigned)(y + dy0 - 1) < 7u) live_tab |= (u64)cols << (8 * y);
    }
    const u32 off_r = (u32)w * 4u, off_g = (u32)(W2 + w) * 4u, off_b = (u32)(2 * W2 + w) * 4u;
    const long long obs_stride = p.N * p.obs_elems;
    float *obs_t = p.obs + env * p.obs_elems;

    for (long long t0 = 0; t0 < p.T; t0 += 64) {
        const int nt = (int)min((long long)64, p.T - t0);
        const long long my_t = t0 + lane;
        long long my_a = lane < nt ? load_action(p.actions, p.act_dtype, my_t * p.N + env) : 0;
        asm volatile("" : "+v"(my_a)); // retire the load here, not in front of the first readlane of the step loop
        // Moves of step t0 + lane for each of the four orientations the snake may have by then, 16 bits each:
        // sanitised action (single_snake.py:221-222) & 7 | next orientation << 4 | (code step & 63) << 6, the step
        // being -TAP[action] (:225-233).  The step loop reads both words with readlanes that do not depend on the
        // state and picks one with the carried orientation.
        int my_mov01, my_mov23;
        {
            const bool in_range = my_a >= 0 && my_a < 4;
            const int a_small = in_range ? (int)my_a : 7, a_mod = (int)(my_a % 4);
            int ent[4];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const int a_out = o == a_small ? (o ^ 2) : a_mod;
                const int ai = a_out & 3;
                ent[o] = (a_out & 7) | ((ai ^ 2) << 4) | (((-tap_y(ai) * 8 - tap_x(ai)) & 63) << 6);
            }
            my_mov01 = ent[0] | (ent[1] << 16);
            my_mov23 = ent[2] | (ent[3] << 16);
        }
        const u64 my_call = p.call + 2ull * (u64)my_t; // step t uses call0 + 2t, its reset call0 + 2t + 1
        S9Reset my_reset;
        int my_food; // RNG: the word the food cell is drawn with; INJ: the food code itself (-1: none)
        if constexpr (INJ) {
            // recorded outcomes of step t0 + lane, converted to cell codes 8 * row + column (reset_core's layout of inj[])
            int sy = 4, sx = 4, d = 0, fc = -1, fe = -1;
            if (lane < nt) {
                const int *ir = p.inject_reset + (my_t * p.N + env) * 4;
                sy = ir[0]; sx = ir[1]; d = ir[2]; fc = ir[3];
                fe = p.inject_food[my_t * p.N + env];
            }
            asm volatile("" : "+v"(sy), "+v"(sx), "+v"(d), "+v"(fc), "+v"(fe)); // retire the loads before the step loop
            d &= 3;
            const int sc = sy * 8 + sx, dc = tap_y(d) * 8 + tap_x(d);
            const int fcy = div_size(max(fc, 0), g.rcpS), fey = div_size(max(fe, 0), g.rcpS);
            const int fcode = (fc >= 0 && fc < S * S) ? fcy * 8 + (fc - fcy * S) : -1;
            my_reset.a = d | (fcode << 2);
            my_reset.b = (sc + dc) | (sc << 7) | ((sc - dc) << 14);
            my_food = (fe >= 0 && fe < S * S) ? fey * 8 + (fe - fey * S) : -1;
        } else {
            my_reset = s9_reset_draw(p.seed, my_call + 1ull, env_id);
            my_food = (int)rng_words(p.seed, my_call, env_id, RNG_FOOD, 0).w[0];
        }
        // what lane j keeps of step t0 + j: its move entry, whether it ate, self collision | edge collision << 1
        int my_rec = 0, my_ate = 0, my_fl = 0;
        {   // re-base the clocks so that they cannot overflow however long the tape is
            const int T = G - L;
            ex = max(ex - T, 0);
            G = L;
        }

        for (int j = 0; j < nt; ++j) {
            const u64 lane_j = 1ull << j; // lane j keeps the record of step t0 + j
            // ---- step (single_snake.py:197-304; same line references as step_core / fast_step)
            const int m01 = lane_value(my_mov01, j), m23 = lane_value(my_mov23, j);
            const int ent = ((o16 & 32) ? m23 : m01) >> (o16 & 16);
            o16 = ent & 48;
            c += (ent << 20) >> 26;                  // the head is inside the ring: the move stays on the grid
            G += 1;
            int ate; // head == food (:242; spelled out: the compiler detours through a 64-bit lane mask)
            asm("s_cmp_eq_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(ate) : "s"(c), "s"(foodc) : "scc");
            L += ate;
            const int T = G - L;                     // :246-249: the clock stands still on the step that eats
            const u64 body = lane_mask(ex > T);      // after the decay, before the head is written
            const u64 head = 1ull << (c & 63);
            ex = keep_in_lane(ex, G, head);          // :258-262 (a head on the ring may land in a wrong lane: it is reset below)
            const u64 occ = body | head;

            // what the crop shows: per lane, is the window cell inside the ring (0 / 1), its code, the occupancy mask
            unsigned inside = 0;
            int code = c + code_d;
            u64 mask = occ;
            if (OBSK == WURM_OBS_PARTIAL) {
                u64 lv; // live_tab >> head code
                asm("v_lshrrev_b64 %0, %1, %2" : "=v"(lv) : "s"(c), "v"(live_tab));
                inside = (u32)lv & 1u;
            }

            // One test for everything that is not a plain move: the head entered a body cell (:252), the border ring
            // (:290-295) or the food cell (:242).
            int event = 0; // 1: self collision, 2: edge collision
            if (__builtin_expect((((body | XF) >> (c & 63)) & 1) != 0, 0)) {
                if (c == foodc) {                    // :270-282: K-th free interior cell in row-major order
                    my_ate = keep_in_lane(my_ate, 1, lane_j);
                    if constexpr (INJ) {
                        foodc = lane_value(my_food, j);
                    } else {
                        const u64 fr = ~(occ | RING);
                        const int n_free = popc64(fr);
                        foodc = -1;
                        if (n_free > 0) {
                            const int K = (int)mulhi_range((u32)lane_value(my_food, j), (u32)n_free);
                            foodc = first_bit(lane_mask((int)((fr >> lane) & 1) & (int)(rank_below(fr) == K)));
                        }
                    }
                    XF = RING | (foodc >= 0 ? 1ull << foodc : 0);
                }
                event = ((RING >> (c & 63)) & 1) ? 2 : ((body >> (c & 63)) & 1) ? 1 : 0;
                if (OBSK == WURM_OBS_PARTIAL && event == 2) {
                    // the head is on the ring and its code may have wrapped: row / column arithmetic from the cell it left
                    const int ai = ent & 3, pc = c - ((ent << 20) >> 26);
                    const int hy = (pc >> 3) - tap_y(ai), hx = (pc & 7) - tap_x(ai);
                    inside = max((unsigned)(hy + dy0 - 1), (unsigned)(hx + dx0 - 1)) < 7u ? 1u : 0u;
                    code = (hy + dy0) * 8 + hx + dx0;
                    mask = body;
                }
            }

            // ---- crop of the stepped state (single_snake.py:166-193): a window cell that is off the grid or on the
            // ring is (0,0,0); food (1,0,0), head (0,1,0), body (0,127/255,0), background (1,1,1).  `inside` / `taken`
            // are 0 / 1 per lane; the class logic stays in the VALU (compare -> select through vcc).
            if (OBSK == WURM_OBS_PARTIAL) {
                u64 sh; // mask >> code; spelled out: the compiler prefers (1 << code) & mask, two 64-bit VALU ops more
                asm("v_lshrrev_b64 %0, %1, %2" : "=v"(sh) : "v"(code), "s"(mask));
                const unsigned taken = (u32)sh & 1u;
                // vr = inside > taken ? 1 : 0        (inside the ring and free)
                // vb = code == food ? 0 : vr         (... and not the food)
                // vg = inside & taken ? green : vb   (inside the ring and occupied)
                // Spelled out: three compares into three SGPR pairs, then three selects.  The compiler funnels all of
                // them through vcc and pads each compare -> select pair with s_nop (gfx950 needs two wait states there).
                const unsigned both = inside & taken;
                float vr, vb, vg;
                u64 m_free, m_not_food, m_taken;
                asm("v_cmp_gt_u32_e64 %3, %6, %7\n\t"
                    "v_cmp_ne_u32_e64 %4, %8, %9\n\t"
                    "v_cmp_ne_u32_e64 %5, 0, %10\n\t"
                    "v_cndmask_b32_e64 %0, 0, 1.0, %3\n\t"
                    "v_cndmask_b32_e64 %1, 0, %0, %4\n\t"
                    "v_cndmask_b32_e64 %2, %1, %11, %5"
                    : "=&v"(vr), "=&v"(vb), "=&v"(vg), "=&s"(m_free), "=&s"(m_not_food), "=&s"(m_taken)
                    : "v"(inside), "v"(taken), "s"(foodc), "v"(code), "v"(both), "v"(green));
                // scalar base + 32-bit lane offset form, spelled out: the compiler hoists the zero-extension of the
                // lane offsets out of the loop and then pays a 64-bit VALU add per store.  (Untracked stores are
                // harmless for its vmcnt bookkeeping: nothing is read back and waits only become conservative.)
                asm volatile("global_store_dword %0, %1, %6\n\tglobal_store_dword %2, %3, %6\n\tglobal_store_dword %4, %5, %6"
                             : : "v"(off_r), "v"(vr), "v"(off_g), "v"(vg), "v"(off_b), "v"(vb), "s"(obs_t) : "memory");
                obs_t += obs_stride;
            }
            my_rec = keep_in_lane(my_rec, ent, lane_j);

            // ---- reset of a finished env (single_snake.py:322-387)
            if (__builtin_expect(event != 0, 0)) {
                my_fl = keep_in_lane(my_fl, event, lane_j);
                const int ra = lane_value(my_reset.a, j), rb = lane_value(my_reset.b, j);
                o16 = (ra & 3) << 4;
                foodc = ra >> 2;
                XF = INJ ? (RING | (foodc >= 0 ? 1ull << foodc : 0)) : (RING | (1ull << foodc));
                c = rb & 127;
                const int sc = (rb >> 7) & 127, tc = rb >> 14;
                ex = lane == tc ? T + 1 : 0; ex = lane == sc ? T + 2 : ex; ex = lane == c ? T + 3 : ex;
                L = 3;
                G = T + 3;
            }
        }
        if (lane < nt) {
            const long long i = my_t * p.N + env;
            store_action(p.actions, p.act_dtype, i, (long long)((my_rec << 29) >> 29));
            p.reward[i] = my_ate ? 1.0f : 0.0f;
            p.done[i] = (uint8_t)(my_fl != 0);
            p.selfc[i] = (uint8_t)(my_fl & 1);
            p.edgec[i] = (uint8_t)(my_fl >> 1);
        }
    }
    if (lane_in) { // the ring was empty and still is
        const int T = G - L;
        envp[my_cell] = lane == foodc ? 1.0f : 0.0f;
        envp[S * S + my_cell] = lane == c ? 1.0f : 0.0f;
        envp[2 * S * S + my_cell] = (float)max(ex - T, 0);
    }
}

// wurm.utils.env_consistency (wurm/utils.py:113-178) per env, as an error bitmask
template <int CPL>
__global__ __launch_bounds__(256) void check_kernel(const float *__restrict__ envs, uint32_t *__restrict__ err,
                                                    long long N, int S)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= N) return;
    const Geo g = make_geo<CPL>(S);
    const float *envp = envs + env * 3 * g.C;
    // in fp32 like the reference's sums (wurm/utils.py:113-178) — a state that holds non-integers (food 0.5, only ever by
    // hand) gets the reference's verdict on every check, not only on the first one (tests/test_checker_failing_states.py).
    // Sums of integer-valued floats are exact in any order WHILE THEY STAY BELOW 2^24 (so are sums of a few halves and
    // quarters): a body sum reaches 2^24 only for a snake of more than 5 792 cells (S = 64 has 4 096), so for every snake a
    // grid of the supported sizes can hold the lane-then-butterfly order here and torch's .sum() agree bit for bit; with
    // hand-made non-dyadic values the two orders may round differently, and so may the verdict.
    int bad_food = 0;
    float hs = 0.0f, bs = 0.0f, bm = -INFINITY, hb = 0.0f, hf = 0.0f, fs = 0.0f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        int c = g.lane + 64 * k;
        if ((g.valid >> k) & 1) {
            float f = envp[c], h = envp[g.C + c], b = envp[2 * g.C + c];
            bad_food |= !(f == 0.0f || f == 1.0f);
            hs += h; bs += b; hb += h * b; hf += h * f; fs += f;
            bm = fmaxf(bm, b);
        }
    }
    bad_food = ballot(bad_food != 0) != 0;
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) {
        hs += __shfl_xor(hs, sh); bs += __shfl_xor(bs, sh); hb += __shfl_xor(hb, sh); hf += __shfl_xor(hf, sh);
        fs += __shfl_xor(fs, sh); bm = fmaxf(bm, __shfl_xor(bm, sh));
    }
    uint32_t m = 0;
    if (bad_food) m |= WURM_CHK_FOOD_VALUE;
    if (hs != 1.0f) m |= WURM_CHK_ONE_HEAD;
    if (!(bs > 0.0f)) m |= WURM_CHK_HAS_SNAKE;
    if (bm != hb) m |= WURM_CHK_HEAD_AT_END;
    if ((__fsqrt_rn(8.0f * bs + 1.0f) - 1.0f) / 2.0f != bm) m |= WURM_CHK_BODY_RANGE; // bs is the bm-th triangular number
    if (!(bs >= 6.0f)) m |= WURM_CHK_MIN_LENGTH;
    if (hf != 0.0f) m |= WURM_CHK_HEAD_ON_FOOD;
    if (fs != 1.0f) m |= WURM_CHK_ONE_FOOD;
    if (g.lane == 0) err[env] = m;
}

// wurm.utils.determine_orientations over a (n,3,S,S) batch
template <int CPL>
__global__ __launch_bounds__(256) void orientations_kernel(const float *__restrict__ envs, long long *__restrict__ out,
                                                           long long N, int S, int lds_per_wave)
{
    const int wave = (int)(threadIdx.x >> 6), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= N) return;
    signed char *lds = wurm_lds + wave * lds_per_wave;
    const Geo g = make_geo<CPL>(S);
    Env<CPL> e;
    load_state<CPL, true>(envs + env * 3 * g.C, g, e);
    int lm = 0;
#pragma unroll
    for (int k = 0; k < CPL; ++k) lm = max(lm, e.body[k]);
    const int L = uniform(wave_max_i32(lm));
    const int o = orientation_of<CPL>(e, g, L, lds);
    if (g.lane == 0) out[env] = o;
}

#ifndef WURM_SINGLE_SNAKE_DEVICE_CODE_ONLY // lane_rollout.hip includes this file for the device code above only
// ------------------------------------------------------------------------------------------------ host side

enum Kind { K_STEP, K_RESET, K_OBSERVE, K_ROLLOUT, K_FUSED };

static int pick_cpl(int S)
{
    int need = (S * S + 63) / 64;
    const int opts[] = {2, 4, 8, 16, 24, 32, 48, 64};
    for (int o : opts)
        if (need <= o) return o;
    return -1;
}

// ---- which kernel serves a call.  ONE table (route_of), read top to bottom: the first row whose condition holds wins.
//   kind             | condition                                                                    | route
//   step / fused     | snake, S >= 12, grid_step_eligible, (N S^2 >= grid_step_min_cells or a mirror) | R_GRID_STEP   grid_rollout.hip (+ generic for the rest)
//   step / fused     | snake, S <= 11, N >= lane_step_min_envs, lane_step_eligible                    | R_LANE_STEP   lane_step.hpp
//   step/reset/observe/fused | otherwise                                                              | R_GENERIC     one env per wave
//   rollout          | snake, S >= 12, grid_rollout_eligible                                          | R_GRID_ROLLOUT
//   rollout          | snake, S <= 11, N >= lane_rollout_min_envs, lane_rollout_eligible              | R_LANE_ROLLOUT lane_rollout.hpp (9 x 9)
//   rollout          | snake, S = 10 / 11, N >= lane_rollout_min_envs, lane_wide_eligible              | R_LANE_WIDE   lane_wide.hpp (default, one_channel, partial_2 / 3, positions, none)
//   rollout          | snake, S == 9, both inject arrays, partial_n (n <= 3) or none                  | R_S9_INJ      rollout_s9_kernel<., true>
//   rollout          | snake, S == 9, RNG mode, partial_n (n <= 3) or none                            | R_S9          rollout_s9_kernel
//   rollout          | snake, S = 10 / 11, RNG mode, partial_n (n <= 3) or none                       | R_LEAN        rollout_lean_kernel
//   rollout          | snake, S <= 11, RNG mode, partial_n (n <= 6) / none                            | R_GENERIC_PARTIAL / R_GENERIC_NONE (mode as template argument)
//   rollout          | gridworld, RNG mode, N >= lane_rollout_min_envs, gridworld_lane_eligible        | R_GRIDWORLD_LANE gridworld_lane.hip (+ generic for the rest)
//   step / fused     | gridworld, RNG mode, no immediate reset, N >= lane_step_min_envs                | R_GRIDWORLD_LANE_STEP gridworld_lane.hip (+ generic for the rest)
//   rollout          | otherwise                                                                      | R_GENERIC
// (the resident 9 x 9 step, lane_resident.hpp, is chosen by fused_entry: it needs the caller's mirror)
enum Route { R_GENERIC, R_GRID_STEP, R_LANE_STEP, R_GRID_ROLLOUT, R_LANE_ROLLOUT, R_LANE_WIDE, R_S9_INJ, R_S9, R_LEAN, R_GENERIC_PARTIAL, R_GENERIC_NONE, R_LANE_RESIDENT, R_LANE_WIDE_RESIDENT, R_GRIDWORLD_LANE, R_GRIDWORLD_LANE_STEP };
// (wurm_single_last_route: the route of the CALLING THREAD's last launch — a diagnostic the tests and bench.py name a launch by; no
// state that a later call depends on.  One object for both translation units of this file: see WURM_TU_GRID below.)
extern thread_local Route last_route;
#ifndef WURM_TU_GRID
thread_local Route last_route = R_GENERIC;
#endif

static const char *route_name(Route r)
{
    switch (r) {
    case R_GRID_STEP: return "grid_step";
    case R_LANE_STEP: return "lane_step";
    case R_GRID_ROLLOUT: return "grid_rollout";
    case R_LANE_ROLLOUT: return "lane_rollout";
    case R_LANE_WIDE: return "lane_wide";
    case R_S9_INJ: return "rollout_s9_injected";
    case R_S9: return "rollout_s9";
    case R_LEAN: return "rollout_lean";
    case R_GENERIC_PARTIAL: return "rollout_generic_partial";
    case R_GENERIC_NONE: return "rollout_generic_none";
    case R_LANE_RESIDENT: return "lane_resident";
    case R_LANE_WIDE_RESIDENT: return "lane_wide_resident";
    case R_GRIDWORLD_LANE: return "gridworld_lane";
    case R_GRIDWORLD_LANE_STEP: return "gridworld_lane_step";
    default: return "generic";
    }
}

static Route route_of(Kind kind, bool snake, int cpl, const StepArgs &p)
{
    const bool stepish = kind == K_STEP || kind == K_FUSED;
    if (snake && cpl >= 4 && stepish && grid_step_eligible(p) &&
        (p.N * (long long)p.S * p.S >= opt.grid_step_min_cells || p.resident != nullptr)) return R_GRID_STEP;
    if (snake && cpl == 2 && stepish && p.N >= opt.lane_step_min_envs && lane_step_eligible(p)) return R_LANE_STEP;
    if (kind == K_ROLLOUT && !snake && gridworld_lane_eligible(p)) return R_GRIDWORLD_LANE;
    if (stepish && !snake && gridworld_lane_step_eligible(p)) return R_GRIDWORLD_LANE_STEP;
    if (kind != K_ROLLOUT || !snake) return R_GENERIC;
    if (cpl >= 4) return grid_rollout_eligible(p) ? R_GRID_ROLLOUT : R_GENERIC;
    if (p.N >= opt.lane_rollout_min_envs && lane_rollout_eligible(p)) return R_LANE_ROLLOUT;
    if (p.N >= opt.lane_rollout_min_envs && lane_wide_eligible(p)) return R_LANE_WIDE;
    const bool rng_mode = p.inject_food == nullptr && p.inject_reset == nullptr;
    const bool injected = p.inject_food != nullptr && p.inject_reset != nullptr;
    const bool small_crop_or_none = (p.obs_mode == WURM_OBS_PARTIAL && p.obs_n <= 3) || p.obs_mode == WURM_OBS_NONE;
    if (injected && p.S == 9 && small_crop_or_none) return R_S9_INJ;
    if (rng_mode && p.S == 9 && small_crop_or_none) return R_S9;
    if (rng_mode && p.S > 9 && small_crop_or_none) return R_LEAN;
    if (rng_mode && p.obs_mode == WURM_OBS_PARTIAL && p.obs_n <= 6) return R_GENERIC_PARTIAL;
    if (rng_mode && p.obs_mode == WURM_OBS_NONE) return R_GENERIC_NONE;
    return R_GENERIC;
}

template <int CPL, bool SNAKE>
static hipError_t launch_one(Kind kind, const StepArgs &p, dim3 grid, dim3 block, size_t lds, hipStream_t st)
{
    (void)hipGetLastError(); // drop any stale error left by earlier runtime calls of this thread
    const Route route = route_of(kind, SNAKE, CPL, p);
    last_route = route;
    // (the one-env-per-wave code behind a grid / lane kernel, for the envs that one could not take: flagged_kernel, a wave per 64 envs)
    const unsigned wpb_f = block.x / 64u;
    const dim3 fgrid((unsigned)((p.N + 64ll * wpb_f - 1) / (64ll * wpb_f)));
    switch (route) {
    case R_GRID_STEP:
        if constexpr (SNAKE && CPL >= 4) {
            hipError_t err = launch_grid_step(p, st);
            if (err != hipSuccess) return err;
            WURM_LAUNCH((flagged_kernel<CPL, SNAKE, false>), fgrid, block, lds, st, p);
        }
        break;
    case R_LANE_STEP:
        if constexpr (SNAKE && CPL == 2) return launch_lane_step(p, st);
        break;
    case R_GRID_ROLLOUT:
        if constexpr (SNAKE && CPL >= 4) {
            hipError_t err = launch_grid_rollout(p, st);
            if (err != hipSuccess) return err;
            WURM_LAUNCH((flagged_kernel<CPL, SNAKE, true>), fgrid, block, lds, st, p);
        }
        break;
    case R_LANE_ROLLOUT:
        if constexpr (SNAKE && CPL == 2) return launch_lane_rollout(p, st);
        break;
    case R_LANE_WIDE:
        if constexpr (SNAKE && CPL == 2) return launch_lane_wide(p, st);
        break;
    case R_GRIDWORLD_LANE:
        if constexpr (!SNAKE) {
            hipError_t err = launch_gridworld_lane_rollout(p, st);
            if (err != hipSuccess) return err;
            if (!(p.resident != nullptr && p.resident_valid)) // (a mirror that was current describes every env: see R_GRIDWORLD_LANE_STEP)
                WURM_LAUNCH((flagged_kernel<CPL, SNAKE, true>), fgrid, block, lds, st, p); // (the envs outside the lane kernel's domain)
        }
        break;
    case R_GRIDWORLD_LANE_STEP:
        if constexpr (!SNAKE) {
            hipError_t err = launch_gridworld_lane_step(p, st);
            if (err != hipSuccess) return err;
            // (a mirror that was current describes every env — the library reports a mirror valid only if the launch that
            // built it found nothing outside the lane kernel's domain, and that domain is closed under the library's own
            // launches — so nothing can be flagged: ONE launch per call)
            if (!(p.resident != nullptr && p.resident_valid))
                WURM_LAUNCH((flagged_kernel<CPL, SNAKE, false>), fgrid, block, lds, st, p);
        }
        break;
    case R_S9_INJ:
        if constexpr (SNAKE && CPL == 2) {
            if (p.obs_mode == WURM_OBS_NONE) WURM_LAUNCH((rollout_s9_kernel<WURM_OBS_NONE, true>), grid, block, lds, st, p);
            else WURM_LAUNCH((rollout_s9_kernel<WURM_OBS_PARTIAL, true>), grid, block, lds, st, p);
        }
        break;
    case R_S9:
        if constexpr (SNAKE && CPL == 2) {
            if (p.obs_mode == WURM_OBS_NONE) WURM_LAUNCH((rollout_s9_kernel<WURM_OBS_NONE>), grid, block, lds, st, p);
            else WURM_LAUNCH((rollout_s9_kernel<WURM_OBS_PARTIAL>), grid, block, lds, st, p);
        }
        break;
    case R_LEAN:
        if constexpr (SNAKE && CPL == 2) {
            if (p.obs_mode == WURM_OBS_NONE) WURM_LAUNCH((rollout_lean_kernel<WURM_OBS_NONE, false>), grid, block, lds, st, p);
            else WURM_LAUNCH((rollout_lean_kernel<WURM_OBS_PARTIAL, false>), grid, block, lds, st, p);
        }
        break;
    case R_GENERIC_PARTIAL:
        if constexpr (SNAKE && CPL == 2) WURM_LAUNCH((rollout_kernel<CPL, SNAKE, WURM_OBS_PARTIAL, false>), grid, block, lds, st, p);
        break;
    case R_GENERIC_NONE:
        if constexpr (SNAKE && CPL == 2) WURM_LAUNCH((rollout_kernel<CPL, SNAKE, WURM_OBS_NONE, false>), grid, block, lds, st, p);
        break;
    case R_LANE_RESIDENT: // (chosen by fused_entry, which launches it itself)
    case R_LANE_WIDE_RESIDENT:
    case R_GENERIC:
        switch (kind) {
        case K_STEP: WURM_LAUNCH((step_kernel<CPL, SNAKE>), grid, block, lds, st, p); break;
        case K_RESET: WURM_LAUNCH((reset_kernel<CPL, SNAKE>), grid, block, lds, st, p); break;
        case K_OBSERVE: WURM_LAUNCH((observe_kernel<CPL, SNAKE>), grid, block, lds, st, p); break;
        case K_FUSED: WURM_LAUNCH((fused_step_kernel<CPL, SNAKE>), grid, block, lds, st, p); break;
        case K_ROLLOUT: WURM_LAUNCH((rollout_kernel<CPL, SNAKE>), grid, block, lds, st, p); break;
        }
        break;
    }
    return hipGetLastError();
}

// This file is compiled TWICE (round 6: the build's longest translation unit, 4.7 of its 6 minutes): as itself with the
// SingleSnake half of the kernels — launch<true> — and everything else in it, and through single_grid.hip (WURM_TU_GRID) with
// the SimpleGridworld half — launch<false>, which this unit then only declares.  Same source, two compilers at once.
template <bool SNAKE>
int launch(Kind kind, StepArgs p, void *stream)
{
    if (p.N == 0) return WURM_OK;
    const int cpl = pick_cpl(p.S);
    if (cpl < 0) return WURM_ERR_UNSUPPORTED;
    // small batches: one wave per workgroup so the envs spread over all 256 CUs; large: 4 waves per workgroup
    const int wpb = p.N <= 4096 ? 1 : 4;
    p.lds_per_wave = ((p.S * p.S + 15) / 16) * 16;
    dim3 block(64 * wpb), grid((unsigned)((p.N + wpb - 1) / wpb));
    size_t lds = (size_t)p.lds_per_wave * wpb;
    hipStream_t st = (hipStream_t)stream;
    hipError_t err;
    switch (cpl) {
    case 2: err = launch_one<2, SNAKE>(kind, p, grid, block, lds, st); break;
    case 4: err = launch_one<4, SNAKE>(kind, p, grid, block, lds, st); break;
    case 8: err = launch_one<8, SNAKE>(kind, p, grid, block, lds, st); break;
    case 16: err = launch_one<16, SNAKE>(kind, p, grid, block, lds, st); break;
    case 24: err = launch_one<24, SNAKE>(kind, p, grid, block, lds, st); break;
    case 32: err = launch_one<32, SNAKE>(kind, p, grid, block, lds, st); break;
    case 48: err = launch_one<48, SNAKE>(kind, p, grid, block, lds, st); break;
    default: err = launch_one<64, SNAKE>(kind, p, grid, block, lds, st); break;
    }
    return err == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

#ifdef WURM_TU_GRID
template int launch<false>(Kind, StepArgs, void *);
} // namespace wurm
#else
extern template int launch<false>(Kind, StepArgs, void *);

static long long obs_elems(bool snake, int mode, int n, int S)
{
    const long long C = (long long)S * S;
    switch (mode) {
    case WURM_OBS_DEFAULT: return 3 * C;
    case WURM_OBS_RAW: return (snake ? 3 : 2) * C;
    case WURM_OBS_ONE_CHANNEL: return snake ? C : 0;
    case WURM_OBS_POSITIONS: return 4;
    case WURM_OBS_PARTIAL: return (snake && n >= 0) ? 3ll * (2 * n + 1) * (2 * n + 1) : 0;
    default: return 0;
    }
}

static int check_common(bool snake, const void *envs, long long N, int S, const void *obs, int mode, int n, int dtype)
{
    if (N < 0 || S < 3 || S > 64) return S > 64 ? WURM_ERR_UNSUPPORTED : WURM_ERR_INVALID_ARG;
    if (N > 0 && envs == nullptr) return WURM_ERR_INVALID_ARG;
    if (dtype != WURM_ACT_I64 && dtype != WURM_ACT_I32) return WURM_ERR_DTYPE;
    if (mode != WURM_OBS_NONE) {
        if (obs_elems(snake, mode, n, S) == 0) return WURM_ERR_INVALID_ARG;
        if (N > 0 && obs == nullptr) return WURM_ERR_INVALID_ARG;
    }
    return WURM_OK;
}

} // namespace wurm

#include "policy_rollout.hpp"

using namespace wurm;

extern "C" {

const char *wurm_version(void) { return "wurm_hip 0.1 gfx950"; }
const char *wurm_single_last_route(void) { return route_name(last_route); }

int64_t wurm_single_obs_elems(int obs_mode, int obs_n, int size) { return obs_elems(true, obs_mode, obs_n, size); }
int64_t wurm_grid_obs_elems(int obs_mode, int obs_n, int size) { return obs_elems(false, obs_mode, obs_n, size); }

int wurm_single_step(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                     uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                     int64_t num_envs, int size, uint64_t seed, uint64_t call, int64_t env_offset,
                     const int32_t *inject_food, void *stream)
{
    int rc = check_common(true, envs, num_envs, size, obs, obs_mode, obs_n, actions_dtype);
    if (rc) return rc;
    if (num_envs > 0 && (!actions || !reward || !done || !self_collision || !edge_collision)) return WURM_ERR_INVALID_ARG;
    StepArgs p = {};
    p.envs = envs; p.actions = actions; p.act_dtype = actions_dtype; p.reward = reward; p.done = done;
    p.selfc = self_collision; p.edgec = edge_collision; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(true, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.seed = seed; p.call = call;
    p.env_offset = env_offset; p.inject_food = inject_food;
    return launch<true>(K_STEP, p, stream);
}

int wurm_single_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                      int size, uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_reset,
                      void *stream)
{
    int rc = check_common(true, envs, num_envs, size, obs, obs_mode, obs_n, WURM_ACT_I64);
    if (rc) return rc;
    if (size <= 8) return WURM_ERR_UNSUPPORTED; // single_snake.py:346-347
    if (num_envs > 0 && !done) return WURM_ERR_INVALID_ARG;
    StepArgs p = {};
    p.envs = envs; p.done_in = done; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(true, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.seed = seed; p.call = call;
    p.env_offset = env_offset; p.inject_reset = inject_reset;
    return launch<true>(K_RESET, p, stream);
}

int wurm_single_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                        void *stream)
{
    if (obs_mode == WURM_OBS_NONE) return WURM_ERR_INVALID_ARG;
    int rc = check_common(true, envs, num_envs, size, obs, obs_mode, obs_n, WURM_ACT_I64);
    if (rc) return rc;
    StepArgs p = {};
    p.envs = const_cast<float *>(envs); p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(true, obs_mode, obs_n, size); p.N = num_envs; p.S = size;
    return launch<true>(K_OBSERVE, p, stream);
}

int wurm_single_rollout(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                        uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                        int64_t num_envs, int size, int64_t num_steps, uint64_t seed, uint64_t call0,
                        int64_t env_offset, const int32_t *inject_food, const int32_t *inject_reset, void *stream)
{
    int rc = check_common(true, envs, num_envs, size, obs, obs_mode, obs_n, actions_dtype);
    if (rc) return rc;
    if (num_steps < 0) return WURM_ERR_INVALID_ARG;
    if (size <= 8) return WURM_ERR_UNSUPPORTED;
    if (num_envs > 0 && num_steps > 0 && (!actions || !reward || !done || !self_collision || !edge_collision))
        return WURM_ERR_INVALID_ARG;
    if (num_steps == 0) return WURM_OK;
    StepArgs p = {};
    p.envs = envs; p.actions = actions; p.act_dtype = actions_dtype; p.reward = reward; p.done = done;
    p.selfc = self_collision; p.edgec = edge_collision; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(true, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.T = num_steps; p.seed = seed;
    p.call = call0; p.env_offset = env_offset; p.inject_food = inject_food; p.inject_reset = inject_reset;
    return launch<true>(K_ROLLOUT, p, stream);
}

// *mirror_state (nullable): 1 = c->resident describes the state once this call has run, 0 = another path wrote envs (the
// mirror is stale), -1 = nothing was launched
static int fused_entry(bool snake, const wurm_single_call *c, void *stream, int *mirror_state = nullptr)
{
    if (mirror_state) *mirror_state = -1;
    if (!c) return WURM_ERR_INVALID_ARG;
    int rc = check_common(snake, c->envs, c->num_envs, c->size, c->obs, c->obs_mode, c->obs_n, c->actions_dtype);
    if (rc) return rc;
    const int64_t N = c->num_envs;
    if (N > 0 && (!c->actions || !c->reward || !c->done || !c->edge_collision || (snake && !c->self_collision)))
        return WURM_ERR_INVALID_ARG;
    const bool resets = c->pre_done || c->post_reset || c->obs_after;
    if (resets) { // the same limits as wurm_single_reset / wurm_grid_reset
        if (snake && c->size <= 8) return WURM_ERR_UNSUPPORTED;
        if (!snake && (c->size <= 4 || c->start_y < 0 || c->start_x < 0 || c->start_y >= c->size || c->start_x >= c->size))
            return WURM_ERR_UNSUPPORTED;
    }
    StepArgs p = {};
    p.envs = c->envs; p.actions = c->actions; p.act_dtype = c->actions_dtype; p.reward = c->reward; p.done = c->done;
    p.selfc = c->self_collision; p.edgec = c->edge_collision; p.obs = c->obs; p.obs_mode = c->obs_mode;
    p.obs_n = c->obs_n; p.obs_elems = obs_elems(snake, c->obs_mode, c->obs_n, c->size); p.N = N; p.S = c->size;
    // check_mask: only the resident 9 x 9 step computes it (below); any other kernel leaves "not computed" for every env
    auto no_mask = [&]() -> int {
        if (c->check_mask == nullptr || N == 0) return WURM_OK;
        return hipMemsetAsync(c->check_mask, 0xFF, (size_t)N * 4, (hipStream_t)stream) == hipSuccess ? WURM_OK : WURM_ERR_HIP;
    };
    p.start_y = c->start_y; p.start_x = c->start_x; p.seed = c->seed; p.call = c->call; p.env_offset = c->env_offset;
    p.inject_food = c->inject_food; p.inject_reset = c->inject_reset; p.done_in = c->pre_done;
    p.obs_after = c->obs_after; p.done_copy = c->done_copy; p.inject_pre_reset = c->inject_pre_reset;
    p.pre_call = c->pre_call; p.post_reset = c->post_reset;
    if (snake && c->resident != nullptr && N > 0) {
        p.lds_per_wave = ((p.S * p.S + 15) / 16) * 16;
        if (lane_resident_eligible(p)) {
            // the caller keeps a compact mirror of the state: the step reads that instead of envs (lane_resident.hpp)
            if (launch_lane_resident(p, c->resident, c->resident_valid != 0, c->resident_lazy != 0, c->check_mask,
                                     (hipStream_t)stream) != hipSuccess)
                return WURM_ERR_HIP;
            last_route = R_LANE_RESIDENT;
            if (mirror_state) *mirror_state = 1;
            return WURM_OK;
        }
        if (lane_wide_resident_eligible(p) && c->resident_lazy) {
            // 10 x 10 / 11 x 11: the same on lane_wide.hpp's state (lane_wide_resident.hpp), lazy form only — a caller that
            // wants envs written every call gets the kernels without a mirror below, and the mirror reported stale
            if (launch_lane_wide_resident(p, c->resident, c->resident_valid != 0, c->check_mask, (hipStream_t)stream) != hipSuccess)
                return WURM_ERR_HIP;
            last_route = R_LANE_WIDE_RESIDENT;
            if (mirror_state) *mirror_state = 1;
            return WURM_OK;
        }
        if (no_mask() != WURM_OK) return WURM_ERR_HIP;
        if (grid_resident_eligible(p)) {
            // 12 x 12 and larger: the LDS clock-grid step keeps its grids in the mirror (grid_rollout.hip)
            p.resident = c->resident;
            p.resident_valid = c->resident_valid != 0;
            p.resident_lazy = c->resident_lazy != 0;
            rc = launch<true>(resets ? K_FUSED : K_STEP, p, stream);
            if (mirror_state) *mirror_state = rc == WURM_OK ? 1 : 0;
            return rc;
        }
        // this call cannot use the mirror: a lazy one is written out to envs before the ordinary kernels read them
        if (c->resident_lazy && c->resident_valid) {
            hipError_t err = hipSuccess;
            if (p.S == 9) err = launch_lane_resident_flush(p, c->resident, (hipStream_t)stream);
            else if (p.S == 10 || p.S == 11) err = launch_lane_wide_resident_flush(p, c->resident, (hipStream_t)stream);
            else if (grid_step_eligible(p)) { StepArgs q = p; q.resident = c->resident; err = launch_grid_resident_flush(q, (hipStream_t)stream); }
            if (err != hipSuccess) return WURM_ERR_HIP;
        }
    }
    if (!snake && c->resident != nullptr && N > 0) {
        // SimpleGridworld's mirror (gridworld_lane.hip): one record per env.  resident_valid: 0 = build it in this launch,
        // 1 = current, 2 = refused (the launch that built it found envs outside the lane kernel's domain: the planes stay
        // the state until the caller clears resident_valid again).
        if (no_mask() != WURM_OK) return WURM_ERR_HIP;
        if (c->resident_valid != 2 && gridworld_lane_step_eligible(p)) {
            p.resident = c->resident;
            p.resident_valid = c->resident_valid == 1;
            p.resident_lazy = c->resident_lazy != 0;
            if (!p.resident_valid && hipMemsetAsync(c->resident, 0, 16, (hipStream_t)stream) != hipSuccess) return WURM_ERR_HIP;
            rc = launch<false>(resets ? K_FUSED : K_STEP, p, stream);
            if (rc != WURM_OK) { if (mirror_state) *mirror_state = 0; return rc; }
            int state = 1;
            if (!p.resident_valid) {
                // the launch built the mirror (and wrote the planes whatever `lazy` says): valid only if it could describe every
                // env.  One synchronous 4-byte read per BUILD — the first step of an env object, and the step after something
                // else wrote the state — is what lets every other call be a single launch
                int odd = 0;
                if (hipMemcpyAsync(&odd, c->resident, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
                    hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                    return WURM_ERR_HIP;
                if (odd != 0) state = 2;
            }
            if (mirror_state) *mirror_state = state;
            return WURM_OK;
        }
        // this call cannot use the mirror: a lazy one is written out to envs before the ordinary kernels read them
        if (c->resident_lazy && c->resident_valid == 1) {
            StepArgs q = p;
            q.resident = c->resident;
            if (launch_gridworld_lane_flush(q, (hipStream_t)stream) != hipSuccess) return WURM_ERR_HIP;
        }
        if (mirror_state) *mirror_state = c->resident_valid == 2 ? 2 : 0;
        const Kind kind_g = resets ? K_FUSED : K_STEP;
        return launch<false>(kind_g, p, stream);
    }
    if (!(snake && c->resident != nullptr && N > 0) && no_mask() != WURM_OK) return WURM_ERR_HIP;
    if (mirror_state) *mirror_state = 0;
    // nothing to rebuild and no second observation: the plain step kernel (lighter on registers for large grids)
    const Kind kind = resets ? K_FUSED : K_STEP;
    return snake ? launch<true>(kind, p, stream) : launch<false>(kind, p, stream);
}

int wurm_single_resident_flush(const wurm_single_call *c, void *stream)
{
    if (!c) return WURM_ERR_INVALID_ARG;
    if (!c->resident || !c->resident_lazy || !c->resident_valid || c->num_envs <= 0) return WURM_OK;
    if (!c->envs) return WURM_ERR_INVALID_ARG;
    StepArgs p = {};
    p.envs = c->envs; p.N = c->num_envs; p.S = c->size; p.resident = c->resident;
    hipError_t err;
    if (c->size == 9) err = launch_lane_resident_flush(p, c->resident, (hipStream_t)stream);
    else if (c->size == 10 || c->size == 11) err = launch_lane_wide_resident_flush(p, c->resident, (hipStream_t)stream);
    else if (grid_step_eligible(p)) err = launch_grid_resident_flush(p, (hipStream_t)stream);
    else return WURM_ERR_INVALID_ARG;
    return err == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

int64_t wurm_single_resident_size(int64_t num_envs, int size, int obs_mode, int obs_n)
{
    if (num_envs <= 0) return 0;
    if (lane_resident_shape(size, obs_mode, obs_n)) return num_envs * 32; // 9 x 9: 32 bytes per env (lane_resident.hpp)
    if (lane_wide_resident_shape(size, obs_mode, obs_n)) return num_envs * 48; // 10 x 10 / 11 x 11: 48 (lane_wide_resident.hpp)
    StepArgs p = {};
    p.S = size;
    if (grid_step_eligible(p) && obs_elems(true, obs_mode, obs_n, size) >= 0) // 12 x 12 and larger: grid + record per env
        return grid_resident_bytes(num_envs, size);
    return 0;
}

int64_t wurm_single_resident_bytes(int64_t num_envs, int size, int obs_mode, int obs_n)
{
    if (num_envs <= 0) return 0;
    const long long e = opt.resident_min_envs; // -1: by shape
    const bool big = e >= 0 ? num_envs >= e
                            : ((lane_resident_shape(size, obs_mode, obs_n) || lane_wide_resident_shape(size, obs_mode, obs_n)) ? num_envs >= 4096
                                                                          : num_envs * (long long)size * size >= (1ll << 20));
    return big ? wurm_single_resident_size(num_envs, size, obs_mode, obs_n) : 0;
}

int64_t wurm_grid_resident_size(int64_t num_envs, int size, int obs_mode)
{
    if (num_envs <= 0 || size < 5 || size > 64) return 0;
    if (!(obs_mode == WURM_OBS_DEFAULT || obs_mode == WURM_OBS_RAW || obs_mode == WURM_OBS_POSITIONS || obs_mode == WURM_OBS_NONE)) return 0;
    return gridworld_resident_bytes(num_envs, size, obs_mode, obs_elems(false, obs_mode, 0, size));
}

int64_t wurm_grid_resident_bytes(int64_t num_envs, int size, int obs_mode)
{
    const long long e = opt.resident_min_envs; // -1: where the per-call lane kernel takes over
    return num_envs >= (e >= 0 ? e : opt.lane_step_min_envs) ? wurm_grid_resident_size(num_envs, size, obs_mode) : 0;
}

int wurm_grid_resident_flush(const wurm_single_call *c, void *stream)
{
    if (!c) return WURM_ERR_INVALID_ARG;
    if (!c->resident || !c->resident_lazy || c->resident_valid != 1 || c->num_envs <= 0) return WURM_OK;
    if (!c->envs) return WURM_ERR_INVALID_ARG;
    StepArgs p = {};
    p.envs = c->envs; p.N = c->num_envs; p.S = c->size; p.resident = c->resident;
    return launch_gridworld_lane_flush(p, (hipStream_t)stream) == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

int wurm_single_step_reset(const wurm_single_call *c, void *stream) { return fused_entry(true, c, stream); }

int wurm_grid_step_reset(const wurm_single_call *c, void *stream)
{
    int mirror = -1;
    const int rc = fused_entry(false, c, stream, &mirror);
    // (the block is const here: a caller that keeps the mirror learns of a refusal from the return value)
    return (rc == WURM_OK && c && c->resident && c->resident_valid == 0 && mirror == 2) ? WURM_MIRROR_REFUSED : rc;
}

static int step_slot(bool snake, wurm_single_call *c, const wurm_single_slabs *s, int64_t slot, void *actions,
                     int actions_dtype, uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after,
                     void *stream)
{
    if (!c || !s || slot < 0 || slot >= s->steps) return WURM_ERR_INVALID_ARG;
    const int64_t N = c->num_envs, elems = obs_elems(snake, c->obs_mode, c->obs_n, c->size);
    c->actions = actions;
    c->actions_dtype = actions_dtype;
    c->call = call;
    c->obs = s->obs ? s->obs + slot * N * elems : nullptr;
    c->obs_after = (want_obs_after && s->obs_after) ? s->obs_after + slot * N * elems : nullptr;
    c->reward = s->reward + slot * N;
    c->done = s->flags + slot * N;
    c->self_collision = s->flags + (s->steps + slot) * N;
    c->edge_collision = s->flags + (2 * s->steps + slot) * N;
    if (apply_pending) {
        if (!c->done_copy) return WURM_ERR_INVALID_ARG;
        c->pre_done = c->done_copy;
        c->pre_call = pre_call;
    } else {
        c->pre_done = nullptr;
    }
    int mirror = -1;
    const int rc = fused_entry(snake, c, stream, &mirror);
    if (c->resident && mirror >= 0) c->resident_valid = rc != WURM_OK ? 0 : mirror; // (2: SimpleGridworld's mirror refused)
    return rc;
}

int wurm_single_step_slot(wurm_single_call *c, const wurm_single_slabs *s, int64_t slot, void *actions,
                          int actions_dtype, uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after,
                          void *stream)
{
    return step_slot(true, c, s, slot, actions, actions_dtype, call, apply_pending, pre_call, want_obs_after, stream);
}

int wurm_grid_step_slot(wurm_single_call *c, const wurm_single_slabs *s, int64_t slot, void *actions,
                        int actions_dtype, uint64_t call, int apply_pending, uint64_t pre_call, int want_obs_after,
                        void *stream)
{
    return step_slot(false, c, s, slot, actions, actions_dtype, call, apply_pending, pre_call, want_obs_after, stream);
}

int wurm_single_policy_rollout(float *envs, const float *obs0, const float *params, int64_t *actions, float *probs,
                               float *values, float *reward, uint8_t *done, uint8_t *self_collision,
                               uint8_t *edge_collision, float *obs, uint8_t *status, int obs_n, int64_t num_envs,
                               int size, int64_t num_steps, uint64_t seed, uint64_t call0, int64_t env_offset,
                               void *stream)
{
    if (num_envs < 0 || num_steps < 0 || size < 3) return WURM_ERR_INVALID_ARG;
    if (size <= 8 || size > 11 || obs_n < 0 || obs_n > 3) return WURM_ERR_UNSUPPORTED;
    if (num_envs == 0 || num_steps == 0) return WURM_OK;
    if (!envs || !obs0 || !params || !actions || !probs || !values || !reward || !done || !self_collision ||
        !edge_collision || !obs || !status)
        return WURM_ERR_INVALID_ARG;
    PolicyArgs p = {};
    p.envs = envs; p.obs0 = obs0; p.params = params; p.actions = (long long *)actions; p.probs = probs;
    p.values = values; p.reward = reward; p.done = done; p.selfc = self_collision; p.edgec = edge_collision;
    p.obs = obs; p.status = status; p.N = num_envs; p.T = num_steps; p.S = size; p.seed = seed; p.call = call0;
    p.env_offset = env_offset;
    return launch_policy_rollout(p, obs_n, stream);
}

int wurm_single_check(const float *envs, uint32_t *err, int64_t num_envs, int size, void *stream)
{
    if (num_envs < 0 || size < 3) return WURM_ERR_INVALID_ARG;
    if (num_envs == 0) return WURM_OK;
    if (!envs || !err) return WURM_ERR_INVALID_ARG;
    const int cpl = pick_cpl(size);
    if (cpl < 0) return WURM_ERR_UNSUPPORTED;
    const int wpb = 4;
    dim3 block(64 * wpb), grid((unsigned)((num_envs + wpb - 1) / wpb));
    hipStream_t st = (hipStream_t)stream;
    long long N = num_envs;
    (void)hipGetLastError();
    switch (cpl) {
    case 2: WURM_LAUNCH(check_kernel<2>, grid, block, 0, st, envs, err, N, size); break;
    case 4: WURM_LAUNCH(check_kernel<4>, grid, block, 0, st, envs, err, N, size); break;
    case 8: WURM_LAUNCH(check_kernel<8>, grid, block, 0, st, envs, err, N, size); break;
    case 16: WURM_LAUNCH(check_kernel<16>, grid, block, 0, st, envs, err, N, size); break;
    case 24: WURM_LAUNCH(check_kernel<24>, grid, block, 0, st, envs, err, N, size); break;
    case 32: WURM_LAUNCH(check_kernel<32>, grid, block, 0, st, envs, err, N, size); break;
    case 48: WURM_LAUNCH(check_kernel<48>, grid, block, 0, st, envs, err, N, size); break;
    default: WURM_LAUNCH(check_kernel<64>, grid, block, 0, st, envs, err, N, size); break;
    }
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

int wurm_orientations(const float *envs, int64_t *out, int64_t n, int size, void *stream)
{
    if (n < 0 || size < 3) return WURM_ERR_INVALID_ARG;
    if (n == 0) return WURM_OK;
    if (!envs || !out) return WURM_ERR_INVALID_ARG;
    const int cpl = pick_cpl(size);
    if (cpl < 0) return WURM_ERR_UNSUPPORTED;
    const int wpb = 4, lpw = ((size * size + 15) / 16) * 16;
    dim3 block(64 * wpb), grid((unsigned)((n + wpb - 1) / wpb));
    hipStream_t st = (hipStream_t)stream;
    long long N = n;
    long long *o = (long long *)out;
    size_t lds = (size_t)lpw * wpb;
    (void)hipGetLastError();
    switch (cpl) {
    case 2: WURM_LAUNCH(orientations_kernel<2>, grid, block, lds, st, envs, o, N, size, lpw); break;
    case 4: WURM_LAUNCH(orientations_kernel<4>, grid, block, lds, st, envs, o, N, size, lpw); break;
    case 8: WURM_LAUNCH(orientations_kernel<8>, grid, block, lds, st, envs, o, N, size, lpw); break;
    case 16: WURM_LAUNCH(orientations_kernel<16>, grid, block, lds, st, envs, o, N, size, lpw); break;
    case 24: WURM_LAUNCH(orientations_kernel<24>, grid, block, lds, st, envs, o, N, size, lpw); break;
    case 32: WURM_LAUNCH(orientations_kernel<32>, grid, block, lds, st, envs, o, N, size, lpw); break;
    case 48: WURM_LAUNCH(orientations_kernel<48>, grid, block, lds, st, envs, o, N, size, lpw); break;
    default: WURM_LAUNCH(orientations_kernel<64>, grid, block, lds, st, envs, o, N, size, lpw); break;
    }
    return hipGetLastError() == hipSuccess ? WURM_OK : WURM_ERR_HIP;
}

/* ---------------------------------------------------------------------------------------- SimpleGridworld */

int wurm_grid_step(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                   uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                   uint64_t seed, uint64_t call, int64_t env_offset, const int32_t *inject_food, void *stream)
{
    int rc = check_common(false, envs, num_envs, size, obs, obs_mode, obs_n, actions_dtype);
    if (rc) return rc;
    if (num_envs > 0 && (!actions || !reward || !done || !edge_collision)) return WURM_ERR_INVALID_ARG;
    StepArgs p = {};
    p.envs = envs; p.actions = const_cast<void *>(actions); p.act_dtype = actions_dtype; p.reward = reward;
    p.done = done; p.edgec = edge_collision; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(false, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.seed = seed; p.call = call;
    p.env_offset = env_offset; p.inject_food = inject_food;
    return launch<false>(K_STEP, p, stream);
}

int wurm_grid_reset(float *envs, const uint8_t *done, float *obs, int obs_mode, int obs_n, int64_t num_envs,
                    int size, int start_y, int start_x, uint64_t seed, uint64_t call, int64_t env_offset,
                    const int32_t *inject_reset, void *stream)
{
    int rc = check_common(false, envs, num_envs, size, obs, obs_mode, obs_n, WURM_ACT_I64);
    if (rc) return rc;
    if (size <= 4) return WURM_ERR_UNSUPPORTED;                                                  // simple_gridworld.py:249-250
    if (start_y < 0 || start_x < 0 || start_y >= size || start_x >= size) return WURM_ERR_UNSUPPORTED; // :254-260
    if (num_envs > 0 && !done) return WURM_ERR_INVALID_ARG;
    StepArgs p = {};
    p.envs = envs; p.done_in = done; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(false, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.start_y = start_y;
    p.start_x = start_x; p.seed = seed; p.call = call; p.env_offset = env_offset; p.inject_reset = inject_reset;
    return launch<false>(K_RESET, p, stream);
}

int wurm_grid_observe(const float *envs, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                      void *stream)
{
    if (obs_mode == WURM_OBS_NONE) return WURM_ERR_INVALID_ARG;
    int rc = check_common(false, envs, num_envs, size, obs, obs_mode, obs_n, WURM_ACT_I64);
    if (rc) return rc;
    StepArgs p = {};
    p.envs = const_cast<float *>(envs); p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(false, obs_mode, obs_n, size); p.N = num_envs; p.S = size;
    return launch<false>(K_OBSERVE, p, stream);
}

int wurm_grid_rollout(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                      uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                      int64_t num_steps, int start_y, int start_x, uint64_t seed, uint64_t call0,
                      int64_t env_offset, const int32_t *inject_food, const int32_t *inject_reset, void *stream)
{
    int rc = check_common(false, envs, num_envs, size, obs, obs_mode, obs_n, actions_dtype);
    if (rc) return rc;
    if (num_steps < 0) return WURM_ERR_INVALID_ARG;
    if (size <= 4) return WURM_ERR_UNSUPPORTED;
    if (start_y < 0 || start_x < 0 || start_y >= size || start_x >= size) return WURM_ERR_UNSUPPORTED;
    if (num_envs > 0 && num_steps > 0 && (!actions || !reward || !done || !edge_collision)) return WURM_ERR_INVALID_ARG;
    if (num_steps == 0) return WURM_OK;
    StepArgs p = {};
    p.envs = envs; p.actions = const_cast<void *>(actions); p.act_dtype = actions_dtype; p.reward = reward;
    p.done = done; p.edgec = edge_collision; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(false, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.T = num_steps;
    p.start_y = start_y; p.start_x = start_x; p.seed = seed; p.call = call0; p.env_offset = env_offset;
    p.inject_food = inject_food; p.inject_reset = inject_reset;
    return launch<false>(K_ROLLOUT, p, stream);
}

/* wurm_single_rollout (RNG mode) for a caller that keeps the mirror of wurm_single_call.resident: grids of 12 x 12 and larger
 * roll out on the clock grids and records of the per-call step (grid_rollout.hip) — an env its record describes is read from
 * the mirror (2 bytes per cell instead of 12) and written back there, the planes only while the mirror is not lazy; the
 * mirror describes the final state afterwards (*resident_valid = 1).  9 x 9 (another mirror format, a launch that costs 11 us
 * besides its steps) and every other case run wurm_single_rollout on the planes after writing a lazy valid mirror out;
 * *resident_valid is then 0. */
int wurm_single_rollout_resident(float *envs, void *actions, int actions_dtype, float *reward, uint8_t *done,
                                 uint8_t *self_collision, uint8_t *edge_collision, float *obs, int obs_mode, int obs_n,
                                 int64_t num_envs, int size, int64_t num_steps, uint64_t seed, uint64_t call0,
                                 int64_t env_offset, void *resident, int *resident_valid, int resident_lazy, void *stream)
{
    if (!resident || !resident_valid)
        return wurm_single_rollout(envs, actions, actions_dtype, reward, done, self_collision, edge_collision, obs, obs_mode, obs_n,
                                   num_envs, size, num_steps, seed, call0, env_offset, nullptr, nullptr, stream);
    int rc = check_common(true, envs, num_envs, size, obs, obs_mode, obs_n, actions_dtype);
    if (rc) return rc;
    if (num_steps < 0) return WURM_ERR_INVALID_ARG;
    if (size <= 8) return WURM_ERR_UNSUPPORTED;
    if (num_envs > 0 && num_steps > 0 && (!actions || !reward || !done || !self_collision || !edge_collision))
        return WURM_ERR_INVALID_ARG;
    if (num_steps == 0 || num_envs == 0) return WURM_OK;
    StepArgs p = {};
    p.envs = envs; p.actions = actions; p.act_dtype = actions_dtype; p.reward = reward; p.done = done;
    p.selfc = self_collision; p.edgec = edge_collision; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(true, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.T = num_steps; p.seed = seed;
    p.call = call0; p.env_offset = env_offset;
    if (grid_rollout_eligible(p) && grid_resident_eligible(p)) {
        p.resident = resident;
        p.resident_valid = *resident_valid != 0;
        p.resident_lazy = resident_lazy != 0;
        rc = launch<true>(K_ROLLOUT, p, stream);
        *resident_valid = rc == WURM_OK ? 1 : 0;
        return rc;
    }
    if (resident_lazy && *resident_valid) {
        hipError_t err = hipSuccess;
        if (size == 9) err = launch_lane_resident_flush(p, resident, (hipStream_t)stream);
        else if (size == 10 || size == 11) err = launch_lane_wide_resident_flush(p, resident, (hipStream_t)stream);
        else if (grid_step_eligible(p)) { StepArgs q = p; q.resident = resident; err = launch_grid_resident_flush(q, (hipStream_t)stream); }
        if (err != hipSuccess) return WURM_ERR_HIP;
    }
    *resident_valid = 0;
    return launch<true>(K_ROLLOUT, p, stream);
}

/* wurm_grid_rollout (RNG mode) for a caller that keeps SimpleGridworld's mirror (wurm_grid_resident_bytes; meaning of
 * *resident_valid / resident_lazy as in wurm_single_call): where the lane kernel serves the launch the state is read from the
 * records when *resident_valid == 1 — no scan of the planes, no flag pass behind the launch — and the records describe the final
 * state afterwards; the planes are written unless the mirror is lazy and was current.  A launch that builds the mirror reads
 * its verdict back (one stream synchronisation): *resident_valid = 1, or 2 = refused.  Any other launch (a batch or
 * observation the lane kernel does not serve, a refused mirror) runs wurm_grid_rollout on the planes, after writing a lazy
 * valid mirror out; *resident_valid is then 0 (2 stays 2). */
int wurm_grid_rollout_resident(float *envs, const void *actions, int actions_dtype, float *reward, uint8_t *done,
                               uint8_t *edge_collision, float *obs, int obs_mode, int obs_n, int64_t num_envs, int size,
                               int64_t num_steps, int start_y, int start_x, uint64_t seed, uint64_t call0, int64_t env_offset,
                               void *resident, int *resident_valid, int resident_lazy, void *stream)
{
    if (!resident || !resident_valid)
        return wurm_grid_rollout(envs, actions, actions_dtype, reward, done, edge_collision, obs, obs_mode, obs_n, num_envs, size,
                                 num_steps, start_y, start_x, seed, call0, env_offset, nullptr, nullptr, stream);
    int rc = check_common(false, envs, num_envs, size, obs, obs_mode, obs_n, actions_dtype);
    if (rc) return rc;
    if (num_steps < 0) return WURM_ERR_INVALID_ARG;
    if (size <= 4) return WURM_ERR_UNSUPPORTED;
    if (start_y < 0 || start_x < 0 || start_y >= size || start_x >= size) return WURM_ERR_UNSUPPORTED;
    if (num_envs > 0 && num_steps > 0 && (!actions || !reward || !done || !edge_collision)) return WURM_ERR_INVALID_ARG;
    if (num_steps == 0 || num_envs == 0) return WURM_OK;
    StepArgs p = {};
    p.envs = envs; p.actions = const_cast<void *>(actions); p.act_dtype = actions_dtype; p.reward = reward;
    p.done = done; p.edgec = edge_collision; p.obs = obs; p.obs_mode = obs_mode; p.obs_n = obs_n;
    p.obs_elems = obs_elems(false, obs_mode, obs_n, size); p.N = num_envs; p.S = size; p.T = num_steps;
    p.start_y = start_y; p.start_x = start_x; p.seed = seed; p.call = call0; p.env_offset = env_offset;
    if (*resident_valid != 2 && gridworld_lane_eligible(p)) {
        p.resident = resident;
        p.resident_valid = *resident_valid == 1;
        p.resident_lazy = resident_lazy != 0;
        if (!p.resident_valid && hipMemsetAsync(resident, 0, 16, (hipStream_t)stream) != hipSuccess) return WURM_ERR_HIP;
        rc = launch<false>(K_ROLLOUT, p, stream);
        if (rc != WURM_OK) { *resident_valid = 0; return rc; }
        if (!p.resident_valid) { // the launch built the mirror: valid only if it could describe every env (see fused_entry)
            int odd = 0;
            if (hipMemcpyAsync(&odd, resident, 4, hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
                hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
                return WURM_ERR_HIP;
            *resident_valid = odd != 0 ? 2 : 1;
        }
        return WURM_OK;
    }
    if (resident_lazy && *resident_valid == 1) {
        StepArgs q = p;
        q.resident = resident;
        if (launch_gridworld_lane_flush(q, (hipStream_t)stream) != hipSuccess) return WURM_ERR_HIP;
    }
    if (*resident_valid != 2) *resident_valid = 0;
    return launch<false>(K_ROLLOUT, p, stream);
}

} // extern "C"
#endif // WURM_TU_GRID

#else
} // namespace wurm
#endif // WURM_SINGLE_SNAKE_DEVICE_CODE_ONLY
