// grid_rollout.hip — SingleSnake rollout for grids of 12 x 12 and larger with the env resident in LDS as a CLOCK GRID.
//
// Same contract as rollout_kernel (single_snake.hip): T fused iterations of the caller loop
//     obs, reward, done, info = env.step(actions[t]);  env.reset(done)
// (tests/test_single_snake_env.py:24-31, experiments/main.py:212-227; step = wurm/envs/single_snake.py:197-304, reset
// = :322-387), bit-identical to T calls of wurm_single_step / wurm_single_reset.  What differs is where the env lives.
// The register-resident kernel keeps ceil(S^2/64) body values per lane and touches every one of them on every step
// (decay, class, colour): 171 VGPRs + ~500 spilled SGPRs at 36 x 36, two waves per SIMD, ~1000 instructions per
// env-step — its observation stores and its arithmetic add up instead of overlapping (round 1: 3.7 of the 5.3 TB/s a pure
// store stream reaches).  Here:
//   * one 16-bit clock per cell in LDS (2.5 KB per 36 x 36 env), lane l owning cells 256 i + 4 l .. + 3 (one ds_read_b64 /
//     three global_store_dwordx4 per 256 cells): few VGPRs, 8 waves per SIMD — all 8192 envs of BASELINE configs[4]
//     resident in ONE round — so the store stream of one wave hides behind the others' work;
//   * the cell holds an EXPIRY CLOCK: body value = max(ex - T, 0).  A step that does not eat advances T ("every body
//     cell decays", :246-249), one that eats leaves it; only the new head cell is written (ex = T + L = G).  A reset
//     jumps T past every clock (T = G) instead of clearing the grid.  Nothing is per-cell except the observation;
//   * ring cells hold the marker EX_RING and the food cell EX_FOOD, so ONE broadcast LDS read of the cell the head
//     moves to classifies the step: <= T plain move, EX_FOOD eat (:242), EX_RING edge collision (:290-295), anything
//     else > T self collision (:252); and the head is the one cell with ex == G — the whole observation is a function
//     of (ex, T, G).  Clocks are re-based (ex -= T) between 64-step chunks before they can reach the markers;
//   * per-step outputs are kept by lane j for step t0 + j and flushed every 64 steps, as in rollout_kernel.
// Domain (checked per env at entry; anything else is left untouched, marked GRID_SKIPPED and rolled out by
// rollout_kernel in a second launch): a well-formed snake — one head, on the unique maximum L >= 2 of the body channel,
// a unique cell L - 1, at most one food cell, not under the body — strictly inside an empty border ring.  The domain
// is closed under step + reset.  All observation modes; RNG mode and injected outcomes.
#include "step_args.hpp"
#include <algorithm>
#include <cstdlib>

namespace wurm {

extern __shared__ __attribute__((aligned(16))) unsigned char grid_lds_raw[];

namespace {

constexpr int NO_CELL_G = 1 << 20;
constexpr int EX_RING = 0xffff;      // border ring and the padding behind the last cell
constexpr int EX_FOOD = 0xfffe;
#ifdef GRID_W32
typedef int cell_t;
#else
typedef unsigned short cell_t;
#endif
constexpr int EX_MAX_BODY = 0x7000;  // a longer snake (impossible on a 64 x 64 grid) goes to the generic kernel
constexpr int EX_REBASE = 0xc000;    // re-base the clocks once G passes this (a 64-step chunk adds at most 4 * 64)

typedef unsigned short ushort4v __attribute__((ext_vector_type(4)));
typedef int int4v __attribute__((ext_vector_type(4)));

struct Grid;
#ifdef GRID_W32
__device__ __forceinline__ int4v read4(const cell_t *ex, int c0) { return *(const int4v *)(ex + c0); }
__device__ __forceinline__ void write4(cell_t *ex, int c0, const int4v &e) { *(int4v *)(ex + c0) = e; }
#else
__device__ __forceinline__ int4v read4(const cell_t *ex, int c0)
{
    const ushort4v v = *(const ushort4v *)(ex + c0);
    int4v r = {(int)v.x, (int)v.y, (int)v.z, (int)v.w};
    return r;
}
__device__ __forceinline__ void write4(cell_t *ex, int c0, const int4v &e)
{
    const ushort4v e16 = {(cell_t)e.x, (cell_t)e.y, (cell_t)e.z, (cell_t)e.w};
    *(ushort4v *)(ex + c0) = e16;
}
#endif
typedef float float4v __attribute__((ext_vector_type(4)));

struct Grid {
    cell_t *ex; // this wave's clock grid, iters * 256 cells
    int S, C, iters, lane;
    float rcpS;
    int rot;    // grid_observe starts its rows of 256 cells at row `rot` (and wraps): see make_grid
    bool linear; // 'default' observations plane by plane, front to back (WURM_GRID_ROTATE = -1)
};

__device__ __forceinline__ bool ring_cell(const Grid &g, int c)
{
    const int y = div_size(c, g.rcpS), x = c - y * g.S;
    return y == 0 || x == 0 || y == g.S - 1 || x == g.S - 1;
}

// K-th cell (row-major) with ex <= T — a free interior cell: markers and live clocks are > T — with K = mulhi(word,
// number of such cells): the choice add_food / fast_food_cell make (single_snake.py:306-320).  -1 if none is free.
__device__ __forceinline__ int grid_pick_free(const Grid &g, int T, u32 word)
{
    int n_free = 0;
    for (int it = 0; it < g.iters; ++it) {
        const int4v e = read4(g.ex, it * 256 + 4 * g.lane);
        n_free += popc64(ballot(e.x <= T)) + popc64(ballot(e.y <= T)) + popc64(ballot(e.z <= T)) + popc64(ballot(e.w <= T));
    }
    if (n_free == 0) return -1;
    int K = (int)mulhi_range(word, (u32)n_free);
    for (int it = 0; it < g.iters; ++it) {
        const int4v e = read4(g.ex, it * 256 + 4 * g.lane);
        const bool f0 = e.x <= T, f1 = e.y <= T, f2 = e.z <= T, f3 = e.w <= T;
        const u64 b0 = ballot(f0), b1 = ballot(f1), b2 = ballot(f2), b3 = ballot(f3);
        const int cnt = popc64(b0) + popc64(b1) + popc64(b2) + popc64(b3);
        if (K < cnt) {
            // order inside the 256 cells: lane-major, then the lane's four cells
            const int below = rank_below(b0) + rank_below(b1) + rank_below(b2) + rank_below(b3);
            const int mine = (int)f0 + (int)f1 + (int)f2 + (int)f3;
            const int tl = first_bit(ballot(K >= below && K < below + mine));
            const int nib = lane_value((int)f0 | ((int)f1 << 1) | ((int)f2 << 2) | ((int)f3 << 3), tl);
            int r = K - lane_value(below, tl), j = 0;
            for (; j < 4; ++j) {
                if ((nib >> j) & 1) {
                    if (r == 0) break;
                    --r;
                }
            }
            return it * 256 + 4 * tl + j;
        }
        K -= cnt;
    }
    return -1;
}

// what one step leaves for the observation besides the grid itself
struct StepView {
    int hc, hy, hx; // head cell (on an edge collision: the ring cell it moved to) and its row / column
    int food;       // food cell, -1 = none
    int T, G;
    int head_body;  // body value under the head: L (+ the value it ran into on a self collision, :252-262)
};

template <bool VEC>
__device__ __forceinline__ void store3(float *__restrict__ o, int C, int c0, const float4v &a, const float4v &b, const float4v &c)
{
    if (VEC) {
        if (c0 < C) {
            *(float4v *)(o + c0) = a;
            *(float4v *)(o + C + c0) = b;
            *(float4v *)(o + 2 * C + c0) = c;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < C) {
                o[c0 + j] = a[j];
                o[C + c0 + j] = b[j];
                o[2 * C + c0 + j] = c[j];
            }
    }
}

// _observe of the stepped (pre-reset) state (single_snake.py:130-195) from the clock grid
template <bool VEC>
__device__ __forceinline__ void grid_observe(const Grid &g, const StepView &s, float *__restrict__ o, int mode, int n)
{
    const int C = g.C, lane = g.lane;
    if (mode == WURM_OBS_DEFAULT) {
        // _get_rgb (:104-128): body (0,127,0), head (0,255,0), food (255,0,0) on white, ring black, / 255
        const float c127 = 127.0f / 255.0f;
        if (VEC && g.linear) {
            // WURM_GRID_ROTATE = -1: the run FRONT TO BACK — all of the red plane, then green, then blue (the clocks are read
            // three times; LDS is not what this kernel waits for).  A pure store kernel of this shape writes the linear order
            // 4-5 % faster than 1 KiB of each plane in turn (profiles/r06_store_flavours_microbench.txt)
            for (int ch = 0; ch < 3; ++ch) {
                for (int i = 0; i < g.iters; ++i) {
                    const int c0 = i * 256 + 4 * lane;
                    const int4v e = read4(g.ex, c0);
                    float4v v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int x = e[j];
                        const bool occ = x > s.T;
                        v[j] = ch == 0 ? (occ ? (x == EX_FOOD ? 1.0f : 0.0f) : 1.0f)
                             : ch == 1 ? (occ ? (x == s.G ? 1.0f : (x >= EX_FOOD ? 0.0f : c127)) : 1.0f)
                                       : (occ ? 0.0f : 1.0f);
                    }
                    if (c0 < C) *(float4v *)(o + ch * C + c0) = v;
                }
            }
            return;
        }
        for (int i = 0; i < g.iters; ++i) {
            const int it = i + g.rot < g.iters ? i + g.rot : i + g.rot - g.iters;
            const int c0 = it * 256 + 4 * lane;
            const int4v e = read4(g.ex, c0);
            float4v r, gr, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = e[j];
                const bool occ = x > s.T;
                r[j] = occ ? (x == EX_FOOD ? 1.0f : 0.0f) : 1.0f;
                gr[j] = occ ? (x == s.G ? 1.0f : (x >= EX_FOOD ? 0.0f : c127)) : 1.0f;
                b[j] = occ ? 0.0f : 1.0f;
            }
            store3<VEC>(o, C, c0, r, gr, b);
        }
    } else if (mode == WURM_OBS_RAW) { // clone of the state: food, head, body
        for (int i = 0; i < g.iters; ++i) {
            const int it = i + g.rot < g.iters ? i + g.rot : i + g.rot - g.iters;
            const int c0 = it * 256 + 4 * lane;
            const int4v e = read4(g.ex, c0);
            float4v f, h, b;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = e[j];
                const bool head = c0 + j == s.hc;
                f[j] = x == EX_FOOD ? 1.0f : 0.0f;
                h[j] = head ? 1.0f : 0.0f;
                b[j] = head ? (float)s.head_body : ((x > s.T && x < EX_FOOD) ? (float)(x - s.T) : 0.0f);
            }
            store3<VEC>(o, C, c0, f, h, b);
        }
    } else if (mode == WURM_OBS_ONE_CHANNEL) { // :142-151: 0.5 body + 0.5 head + 1.5 food, ring -1
        for (int i = 0; i < g.iters; ++i) {
            const int it = i + g.rot < g.iters ? i + g.rot : i + g.rot - g.iters;
            const int c0 = it * 256 + 4 * lane;
            const int4v e = read4(g.ex, c0);
            float4v v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = e[j];
                v[j] = x == EX_RING ? -1.0f : x == EX_FOOD ? 1.5f : x == s.G ? 1.0f : x > s.T ? 0.5f : 0.0f;
            }
            if (VEC) {
                if (c0 < C) *(float4v *)(o + c0) = v;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c0 + j < C) o[c0 + j] = v[j];
            }
        }
    } else if (mode == WURM_OBS_PARTIAL) {
        // (2n+1)^2 crop of the zero-padded RGB image around the head, channel-major (:166-193)
        const int W = 2 * n + 1, W2 = W * W;
        const float rcpW = 1.0f / (float)W, c127 = 127.0f / 255.0f;
        for (int w = lane; w < W2; w += 64) {
            const int wy = div_size(w, rcpW), wx = w - wy * W;
            const int y = s.hy - n + wy, x = s.hx - n + wx;
            int v = EX_RING; // off the grid: F.pad zeros (:179)
            if (y >= 0 && y < g.S && x >= 0 && x < g.S) v = (int)g.ex[y * g.S + x];
            const bool occ = v > s.T;
            o[w] = occ ? (v == EX_FOOD ? 1.0f : 0.0f) : 1.0f;
            o[W2 + w] = occ ? (v == s.G ? 1.0f : (v >= EX_FOOD ? 0.0f : c127)) : 1.0f;
            o[2 * W2 + w] = occ ? 0.0f : 1.0f;
        }
    } else if (mode == WURM_OBS_POSITIONS) { // head y, x, food y, x (first maximum of an empty channel: cell 0)
        const int f = s.food < 0 ? 0 : s.food, fy = div_size(f, g.rcpS);
        if (lane < 4) o[lane] = (float)(lane == 0 ? s.hy : lane == 1 ? s.hx : lane == 2 ? fy : f - fy * g.S);
    }
}

// carried wave-uniform scalars of one env
struct Snk {
    int hc, hy, hx; // head cell, row, column
    int L, o;       // length, orientation
    int food;       // food cell, -1 = none
    int G, T;       // clocks: G = T + L
};

// env: the env this wave works on (-1: none yet).  Its observation is written row by row of 256 cells, every wave of the chip
// in the SAME order: all concurrent stores lie on one lattice of addresses (env stride 12 S^2 bytes, same offset), and how that
// lattice falls on the HBM channels depends on the physical pages the 2 GB output happens to lie on — the same cfg5 launch
// takes 0.37 to 0.48 ms by allocation within one process, while a linear fill of the same blocks runs at 7.0-7.2 TB/s in each
// (tools/placement_probe.py, profiles/r04_placement_probe.txt).  Option WURM_GRID_ROTATE = 1 starts every env's rows at an
// env-dependent row (the same bytes): that removes the dependence — and is as slow as the worst allocation everywhere
// (0.47-0.50 ms): it is the compact lattice that is fast.  Off by default; kept for the record.
__device__ __forceinline__ Grid make_grid(const StepArgs &p, int wave, long long env = -1)
{
    Grid g;
    g.S = p.S;
    g.C = p.S * p.S;
    g.iters = (g.C + 255) >> 8;
    g.lane = (int)(threadIdx.x & 63u);
    g.rcpS = 1.0f / (float)p.S;
    g.ex = (cell_t *)grid_lds_raw + wave * (g.iters * 256);
    // WURM_GRID_ROTATE: 0 = every env's rows in the same order (default), 1 = start row env % iters, k >= 2 = the coarser
    // skew (env % k) * iters / k — k start rows spread over the env's image by id modulo k (round 5 probe: k = 2, 4, 256)
    g.rot = 0;
    g.linear = p.grid_rotate < 0;
    if (p.grid_rotate == 1 && env >= 0) g.rot = (int)((unsigned long long)env % (unsigned)g.iters);
    else if (p.grid_rotate >= 2 && env >= 0)
        g.rot = (int)(((unsigned long long)env % (unsigned)p.grid_rotate) * (unsigned)g.iters / (unsigned)p.grid_rotate);
    return g;
}

// the state, fp32 [food, head, body] -> clock grid (T = 0: ex = body value), markers on ring and food.  false: the env
// is outside the domain (see the head of this file) and must be left to the generic kernels.
template <bool VEC>
__device__ __forceinline__ bool grid_load(const Grid &g, const float *__restrict__ envp, Snk &s)
{
    const int S = g.S, C = g.C, lane = g.lane;
    int lmax = 0, counts = 0, myh = NO_CELL_G, myf = NO_CELL_G;
    bool bad_l = false;
    for (int it = 0; it < g.iters; ++it) {
        const int c0 = it * 256 + 4 * lane;
        float4v f = {0, 0, 0, 0}, h = {0, 0, 0, 0}, b = {0, 0, 0, 0};
        if (VEC) {
            if (c0 < C) {
                f = *(const float4v *)(envp + c0);
                h = *(const float4v *)(envp + C + c0);
                b = *(const float4v *)(envp + 2 * C + c0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c0 + j < C) {
                    f[j] = envp[c0 + j];
                    h[j] = envp[C + c0 + j];
                    b[j] = envp[2 * C + c0 + j];
                }
        }
        int4v e;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + j;
            const bool inb = c < C, ring = !inb || ring_cell(g, c);
            const bool fd = f[j] > 0.5f, hd = h[j] > 0.5f; // as load_state reads the channels
            const int bi = __float2int_rn(b[j]);
            bad_l |= ring && inb && (fd || hd || bi != 0);   // the ring must be empty
            bad_l |= fd && (hd || bi != 0);                  // nothing under the food
            bad_l |= bi < 0 || bi > EX_MAX_BODY;
            e[j] = ring ? EX_RING : fd ? EX_FOOD : bi;
            if (!ring && !fd) lmax = max(lmax, bi);
            counts += (int)hd + ((int)fd << 16);
            if (hd) myh = min(myh, c);
            if (fd) myf = min(myf, c);
        }
        write4(g.ex, c0, e);
    }
    wave_lds_sync();
    counts = wave_sum_i32(counts);
    const int nhead = counts & 0xffff, nfood = counts >> 16;
    int hc = wave_min_i32(myh), food = wave_min_i32(myf), L = wave_max_i32(lmax);
    if (food >= NO_CELL_G) food = -1;
    int packed = 0, cL = NO_CELL_G, cN = NO_CELL_G;
    for (int it = 0; it < g.iters; ++it) {
        const int c0 = it * 256 + 4 * lane;
        const int4v e = read4(g.ex, c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (e[j] == L) { packed += 1; cL = min(cL, c0 + j); }
            if (e[j] == L - 1) { packed += 1 << 16; cN = min(cN, c0 + j); }
        }
    }
    packed = wave_sum_i32(packed);
    const int cellL = wave_min_i32(cL), cellN = wave_min_i32(cN);
    const bool ok = ballot(bad_l) == 0 && nhead == 1 && nfood <= 1 && L >= 2 && (packed & 0xffff) == 1 &&
                    (packed >> 16) == 1 && cellL == hc;
    if (!uniform((int)ok)) return false;
    const int hy = div_size(hc, g.rcpS), hx = hc - hy * S;
    // orientation from the two newest cells, as orientation_of / fast_init (wurm/utils.py:36-65)
    const int yN = div_size(cellN, g.rcpS), xN = cellN - yN * S;
    const int dy = hy - yN, dx = hx - xN;
    const int o = (dy == 0 && dx == 1) ? 1 : (dy == 1 && dx == 0) ? 2 : (dy == 0 && dx == -1) ? 3 : 0;
    s.hc = uniform(hc); s.hy = uniform(hy); s.hx = uniform(hx); s.L = uniform(L); s.o = uniform(o);
    s.food = uniform(food);
    s.G = s.L;
    s.T = 0;
    return true;
}

// an empty grid: markers on the ring and the padding, 0 elsewhere; no snake, no food, all clocks at 0
__device__ __forceinline__ void grid_clear(const Grid &g, Snk &s)
{
    for (int it = 0; it < g.iters; ++it) {
        const int c0 = it * 256 + 4 * g.lane;
        int4v e;
#pragma unroll
        for (int j = 0; j < 4; ++j) e[j] = (c0 + j >= g.C || ring_cell(g, c0 + j)) ? EX_RING : 0;
        write4(g.ex, c0, e);
    }
    s.hc = s.hy = s.hx = 0;
    s.L = 0; s.o = 0; s.food = -1; s.G = 0; s.T = 0;
    wave_lds_sync();
}

struct StepEv {
    int a_out;   // sanitised action
    bool eat, selfc, edgec;
    int v;       // what the head ran into (the clock of the cell before the step)
    int old_hc;  // head cell before the move
};

// one transition (single_snake.py:197-304; line references as step_core / fast_step).  a_small: the action if it is
// one of 0..3 else -1; a_mod: action % 4 (C semantics).  use_inject: the food cell of an eating step is inject_cell.
__device__ __forceinline__ void grid_step(const Grid &g, Snk &s, int a_small, int a_mod, bool use_inject,
                                          int inject_cell, u64 seed, u64 call, u64 env_id, StepEv &ev)
{
    const int lane = g.lane;
    const int a_out = s.o == a_small ? ((s.o + 2) & 3) : a_mod;                    // :221-222
    const int ai = a_out & 3, dy = -tap_y(ai), dx = -tap_x(ai);
    ev.old_hc = s.hc;
    s.hy += dy; s.hx += dx; s.hc += dy * g.S + dx;                                 // :225-233 (the head was inside the ring)
    s.o = (ai + 2) & 3;
    const int v = uniform((int)g.ex[s.hc]);                                        // what the head runs into
    const bool eat = v == EX_FOOD, edgec = v == EX_RING;                           // :242, :290-295
    s.G += 1;
    s.L += (int)eat;
    s.T = s.G - s.L;                                                               // :246-249: no decay on the step that eats
    const bool selfc = !eat && !edgec && v > s.T;                                  // :252
    if (!edgec && lane == 0) g.ex[s.hc] = (cell_t)s.G;                             // :258-262 (a ring cell keeps its marker)
    wave_lds_sync();
    if (eat) {                                                                     // :270-282
        if (use_inject) s.food = (inject_cell >= 0 && inject_cell < g.C && uniform((int)g.ex[inject_cell]) <= s.T) ? inject_cell : -1;
        else s.food = grid_pick_free(g, s.T, rng_words(seed, call, env_id, RNG_FOOD, 0).w[0]);
        if (s.food >= 0 && lane == 0) g.ex[s.food] = (cell_t)EX_FOOD;
        wave_lds_sync();
    }
    ev.a_out = a_out; ev.eat = eat; ev.selfc = selfc; ev.edgec = edgec; ev.v = v;
}

// reset of one env (single_snake.py:322-387): the clock jumps past every live cell, then a 3-segment snake and food.
// inj: nullable {seed_y, seed_x, direction, food_cell}
__device__ __forceinline__ void grid_reset(const Grid &g, Snk &s, u64 seed, u64 call, u64 env_id, const int *inj)
{
    const int S = g.S, lane = g.lane;
    wave_lds_sync();
    s.T = s.G; // every clock of the dead snake is <= G: the grid is empty without touching it
    // the old food marker becomes a dead clock (1 <= T), not 0: "not 0" keeps meaning "has held something since the load"
    if (s.food >= 0 && lane == 0) g.ex[s.food] = 1;
    int sy, sx, d, fc = -1;
    Words w;
    w.w[0] = w.w[1] = w.w[2] = w.w[3] = 0;
    if (inj) {
        sy = uniform(inj[0]); sx = uniform(inj[1]); d = uniform(inj[2]); fc = uniform(inj[3]);
    } else { // randint(4, S-4) twice, randint(4) (:358-359,366)
        w = rng_words(seed, call, env_id, RNG_RESET, 0);
        sy = 4 + (int)mulhi_range(w.w[0], (u32)(S - 8));
        sx = 4 + (int)mulhi_range(w.w[1], (u32)(S - 8));
        d = (int)(w.w[2] >> 30);
    }
    sy = uniform(sy); sx = uniform(sx); d = uniform(d);
    s.hy = sy + tap_y(d); s.hx = sx + tap_x(d);
    s.hc = s.hy * S + s.hx;
    const int sc = sy * S + sx, tc = (sy - tap_y(d)) * S + sx - tap_x(d);
    if (lane == 0) { // conv2d(seed, LENGTH_3_SNAKES[d]) (:372-376)
        g.ex[tc] = (cell_t)(s.T + 1);
        g.ex[sc] = (cell_t)(s.T + 2);
        g.ex[s.hc] = (cell_t)(s.T + 3);
    }
    s.L = 3;
    s.G = s.T + 3;
    s.o = d;
    wave_lds_sync();
    if (inj) s.food = (fc >= 0 && fc < g.C && uniform((int)g.ex[fc]) <= s.T) ? fc : -1;
    else s.food = grid_pick_free(g, s.T, w.w[3]);                                  // :384-385
    if (s.food >= 0 && lane == 0) g.ex[s.food] = (cell_t)EX_FOOD;
    wave_lds_sync();
}

__device__ __forceinline__ StepView view_of(const Snk &s, int head_body)
{
    StepView v;
    v.hc = s.hc; v.hy = s.hy; v.hx = s.hx; v.food = s.food; v.T = s.T; v.G = s.G; v.head_body = head_body;
    return v;
}

// ---- the caller's mirror of the state (wurm_single_call.resident for S >= 12): per env the clock grid as it sits in
// LDS (iters * 256 cells of 16 bits) and a 48-byte record — what grid_load would produce, kept between calls so that the
// per-call step does not read 12 S^2 bytes of fp32 to find one snake.  Plane 0: grids [N][iters * 256]; plane 1: records.
constexpr int MR_INTS = 12;   // hc, hy, hx, L, o, food, G, T, head_body, flags, -, -
constexpr int MR_ACT = 1;     // the record and the grid describe the env (else: envs is authoritative for it)
constexpr int MR_TERMINAL = 2; // the last step finished the env: outside the domain unless the next call rebuilds it

__device__ __forceinline__ cell_t *mirror_grid(const StepArgs &p, const Grid &g, long long env)
{
    return (cell_t *)p.resident + env * (long long)(g.iters * 256);
}
__device__ __forceinline__ int *mirror_rec(const StepArgs &p, const Grid &g, long long env)
{
    return (int *)((cell_t *)p.resident + p.N * (long long)(g.iters * 256)) + env * MR_INTS;
}

// mirror -> LDS / LDS -> mirror: 8 bytes per lane and instruction, the layout is the same on both sides
__device__ __forceinline__ void mirror_load_grid(const Grid &g, const cell_t *src)
{
    for (int it0 = 0; it0 < g.iters; it0 += 8) {
        uint2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = *(const uint2 *)(src + min(it0 + j, g.iters - 1) * 256 + 4 * g.lane);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (it0 + j < g.iters) *(uint2 *)(g.ex + (it0 + j) * 256 + 4 * g.lane) = v[j];
    }
    wave_lds_sync();
}
__device__ __forceinline__ void mirror_store_grid(const Grid &g, cell_t *dst)
{
    wave_lds_sync();
    for (int it = 0; it < g.iters; ++it)
        *(uint2 *)(dst + it * 256 + 4 * g.lane) = *(const uint2 *)(g.ex + it * 256 + 4 * g.lane);
}
__device__ __forceinline__ void mirror_store_rec(const Grid &g, int *rec, const Snk &s, int head_body, int flags)
{
    const int lane = g.lane;
    const int v = lane == 0 ? s.hc : lane == 1 ? s.hy : lane == 2 ? s.hx : lane == 3 ? s.L : lane == 4 ? s.o
                : lane == 5 ? s.food : lane == 6 ? s.G : lane == 7 ? s.T : lane == 8 ? head_body : lane == 9 ? flags : 0;
    if (lane < MR_INTS) rec[lane] = v;
}
// returns the flags; s and head_body from the record
// (v: the lane's word of the record, lanes 0 .. MR_INTS - 1)
__device__ __forceinline__ int mirror_parse_rec(int v, Snk &s, int &head_body)
{
    s.hc = lane_value(v, 0); s.hy = lane_value(v, 1); s.hx = lane_value(v, 2); s.L = lane_value(v, 3);
    s.o = lane_value(v, 4); s.food = lane_value(v, 5); s.G = lane_value(v, 6); s.T = lane_value(v, 7);
    head_body = lane_value(v, 8);
    return lane_value(v, 9);
}

__device__ __forceinline__ int mirror_load_rec(const Grid &g, const int *rec, Snk &s, int &head_body)
{
    return mirror_parse_rec(g.lane < MR_INTS ? rec[g.lane] : 0, s, head_body);
}

// mirror_load_grid in two halves: the loads of the first eight rows of 256 cells are REQUESTED (grid_step_kernel asks for
// everything it will need in one go — its record, its action, its grid — instead of one memory round trip after the other),
// later they are written to LDS (and the rows beyond the eighth, S > 45, fetched the plain way)
__device__ __forceinline__ void mirror_grid_request(const Grid &g, const cell_t *src, uint2 (&v)[8])
{
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = *(const uint2 *)(src + min(j, g.iters - 1) * 256 + 4 * g.lane);
}

__device__ __forceinline__ void mirror_grid_commit(const Grid &g, const cell_t *src, const uint2 (&v)[8])
{
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j < g.iters) *(uint2 *)(g.ex + j * 256 + 4 * g.lane) = v[j];
    for (int it0 = 8; it0 < g.iters; it0 += 8) {
        uint2 w[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) w[j] = *(const uint2 *)(src + min(it0 + j, g.iters - 1) * 256 + 4 * g.lane);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (it0 + j < g.iters) *(uint2 *)(g.ex + (it0 + j) * 256 + 4 * g.lane) = w[j];
    }
    wave_lds_sync();
}

template <bool VEC>
__global__ __launch_bounds__(256) void grid_rollout_kernel(StepArgs p)
{
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const Grid g = make_grid(p, wave, env);
    const int lane = g.lane;
    float *envp = p.envs + env * 3 * g.C;
    const u64 env_id = (u64)(p.env_offset + env);
    Snk s;
    // (round 6) the caller's mirror of the per-call step (wurm_single_rollout_resident; grid_step_kernel's records and clock
    // grids): an env its record describes comes from there — 2 bytes per cell instead of 12 — and goes back there; the
    // planes are written only while the mirror is not lazy.  An env the record does not describe is read from the planes
    // as ever (they are authoritative for it), one that finished in the last per-call step and was not rebuilt is outside
    // the domain (its last state written out first if the mirror is lazy).
    const bool mirrored = p.resident != nullptr, lazy = mirrored && p.resident_lazy != 0;
    cell_t *const mgrid = mirrored ? mirror_grid(p, g, env) : nullptr;
    int *const mrec = mirrored ? mirror_rec(p, g, env) : nullptr;
    bool from_rec = false;
    if (mirrored && p.resident_valid) {
        int hb = 0;
        const int flags = mirror_load_rec(g, mrec, s, hb);
        if ((flags & (MR_ACT | MR_TERMINAL)) == MR_ACT) {
            mirror_load_grid(g, mgrid);
            from_rec = true;
        } else if (flags & MR_ACT) {
            if (lazy) {
                mirror_load_grid(g, mgrid);
                grid_observe<VEC>(g, view_of(s, hb), envp, WURM_OBS_RAW, 0);
            }
            if (lane == 0) { mrec[9] = 0; p.done[env] = GRID_SKIPPED; }
            return;
        }
    }
    if (!from_rec && !grid_load<VEC>(g, envp, s)) { // outside the domain: rollout_kernel takes this env (second launch, only_flagged)
        if (lane == 0) {
            if (mirrored) mrec[9] = 0;
            p.done[env] = GRID_SKIPPED;
        }
        return;
    }

    const bool inj_f = p.inject_food != nullptr, inj_r = p.inject_reset != nullptr;
    const long long obs_stride = p.N * p.obs_elems;
    float *obs_t = p.obs + env * p.obs_elems;
    u64 call = p.call; // step t uses call0 + 2t, its reset call0 + 2t + 1
    const int hc0 = s.hc, food0 = s.food; // what HBM holds: for the sparse write-back at the end
    bool rebased = false;

    for (long long t0 = 0; t0 < p.T; t0 += 64) {
        const int nt = (int)min((long long)64, p.T - t0);
        const long long my_t = t0 + lane;
        long long my_a = lane < nt ? load_action(p.actions, p.act_dtype, my_t * p.N + env) : 0;
        int my_inj = (inj_f && lane < nt) ? p.inject_food[my_t * p.N + env] : -1;
        asm volatile("" : "+v"(my_a), "+v"(my_inj)); // retire the prefetch here, not in front of the first readlane
        const int my_small = (my_a >= 0 && my_a < 4) ? (int)my_a : -1, my_mod = (int)(my_a % 4);
        int my_out = 0, my_flags = 0; // of step t0 + lane: sanitised action; done | selfc << 1 | edgec << 2 | reward << 3
        if (s.G > EX_REBASE) { // keep the 16-bit clocks away from the markers: ex -= T for the live cells, 0 for the rest
            for (int it = 0; it < g.iters; ++it) {
                const int c0 = it * 256 + 4 * lane;
                int4v e = read4(g.ex, c0);
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = e[q] >= EX_FOOD ? e[q] : max(e[q] - s.T, 0);
                write4(g.ex, c0, e);
            }
            s.G -= s.T;
            s.T = 0;
            rebased = true; // dead cells were zeroed: "ex != 0" no longer marks every cell that ever held a value
            wave_lds_sync();
        }

        for (int j = 0; j < nt; ++j, obs_t += obs_stride, call += 2) {
            StepEv ev;
            grid_step(g, s, lane_value(my_small, j), lane_value(my_mod, j), inj_f, inj_f ? lane_value(my_inj, j) : -1,
                      p.seed, call, env_id, ev);
            if (p.obs_mode != WURM_OBS_NONE)
                grid_observe<VEC>(g, view_of(s, s.L + (ev.selfc ? ev.v - s.T : 0)), obs_t, p.obs_mode, p.obs_n);
            if (lane == j) {
                my_out = ev.a_out;
                my_flags = (int)(ev.selfc | ev.edgec) | ((int)ev.selfc << 1) | ((int)ev.edgec << 2) | ((int)ev.eat << 3);
            }
            if (ev.selfc | ev.edgec)
                grid_reset(g, s, p.seed, call + 1ull, env_id, inj_r ? p.inject_reset + ((t0 + j) * p.N + env) * 4 : nullptr);
        }
        if (lane < nt) {
            const long long i = my_t * p.N + env;
            store_action(p.actions, p.act_dtype, i, (long long)my_out);
            p.reward[i] = (my_flags & 8) ? 1.0f : 0.0f;
            p.done[i] = (uint8_t)(my_flags & 1);
            p.selfc[i] = (uint8_t)((my_flags >> 1) & 1);
            p.edgec[i] = (uint8_t)((my_flags >> 2) & 1);
        }
    }

    // ---- the mirror: the grid whole, the record (every done env was reset: never terminal here)
    if (mirrored) {
        mirror_store_grid(g, mgrid);
        mirror_store_rec(g, mrec, s, s.L, MR_ACT);
        if (lazy) return; // (the planes are written out by wurm_single_resident_flush when something looks at them)
    }
    // ---- back to the reference layout (every done env was reset: the head is on the grid, inside the ring).
    // Only what may differ from HBM is written: a cell whose clock is not 0 has held a body value or the food since the
    // load (clocks and markers only ever overwrite each other; nothing is cleared to 0 but by a re-base), the head and
    // food planes hold a single 1 each.  After a re-base that bookkeeping is gone and the state is stored whole — as is
    // the state of an env that came from the mirror (its clocks say nothing about what the planes hold).
    wave_lds_sync();
    if (rebased || from_rec) {
        grid_observe<VEC>(g, view_of(s, s.L), envp, WURM_OBS_RAW, 0);
        return;
    }
    const int C = g.C;
    for (int it = 0; it < g.iters; ++it) {
        const int c0 = it * 256 + 4 * lane;
        const int4v e = read4(g.ex, c0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e[j] != 0 && e[j] != EX_RING) envp[2 * C + c0 + j] = e[j] == EX_FOOD ? 0.0f : (float)max(e[j] - s.T, 0);
    }
    if (lane == 0) {
        if (hc0 != s.hc) {
            envp[C + hc0] = 0.0f;
            envp[C + s.hc] = 1.0f;
        }
        if (food0 != s.food) {
            if (food0 >= 0) envp[food0] = 0.0f;
            if (s.food >= 0) envp[s.food] = 1.0f;
        }
    }
}

// ---------------------------------------------------------------------------------------------- per-call
// fused_step_kernel (single_snake.hip) for grids >= 12 x 12 on the clock grid: [reset of the envs flagged in
// p.done_in with call = p.pre_call], step (call = p.call), observation, [p.obs_after: the observation reset(done)
// returns], [p.post_reset: the rebuilt state stored].  The state is read with 16-byte loads into LDS and only the
// cells the step changed are written back: the decayed body cells, the two head cells, the food cells (a rebuilt env
// is stored whole).  Envs outside the domain are flagged in done[env] for fused_step_kernel (second launch).
template <bool VEC>
__global__ __launch_bounds__(256) void grid_step_kernel(StepArgs p)
{
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const Grid g = make_grid(p, wave, env);
    const int lane = g.lane, C = g.C;
    float *envp = p.envs + env * 3 * C;
    const u64 env_id = (u64)(p.env_offset + env);
    Snk s;
    WURM_TL_DECL;
    WURM_TL(0); // entry
    const bool mirrored = p.resident != nullptr;
    const bool lazy = mirrored && p.resident_lazy != 0;
    cell_t *mgrid = mirrored ? mirror_grid(p, g, env) : nullptr;
    int *mrec = mirrored ? mirror_rec(p, g, env) : nullptr;
    // everything the step reads before it can start, requested in one go: "was this env finished by the last call", its
    // record in the mirror, its action, its grid (speculatively: a finished env does not use it).  One after the other they
    // were four dependent memory round trips at the head of a 32 us launch.
    const bool from_mirror = mirrored && p.resident_valid != 0;
    const int pre_byte = p.done_in != nullptr ? (int)p.done_in[env] : 0;
    const int rec_word = (from_mirror && lane < MR_INTS) ? mrec[lane] : 0;
    const long long a_word = load_action(p.actions, p.act_dtype, env);
    const bool inj_f = p.inject_food != nullptr;
    const int inj_word = inj_f ? p.inject_food[env] : -1;
    uint2 gv[8];
    if (from_mirror) mirror_grid_request(g, mgrid, gv);
    const bool pre = uniform(pre_byte) != 0;
    bool whole = true; // the mirror's grid has to be stored whole (else: the cells this step changed)
    if (pre) {
        grid_clear(g, s);
        grid_reset(g, s, p.seed, p.pre_call, env_id, p.inject_pre_reset ? p.inject_pre_reset + env * 4 : nullptr);
    } else {
        int flags = 0, hb = 0;
        if (from_mirror) flags = mirror_parse_rec(rec_word, s, hb);
        if ((flags & (MR_ACT | MR_TERMINAL)) == MR_ACT) {
            mirror_grid_commit(g, mgrid, gv);
            whole = false;
        } else if (flags & MR_ACT) {
            // finished by the last step and not rebuilt: outside the domain.  The generic kernel steps it on envs (second
            // launch) — which the lazy form has not been writing: its last state goes out first
            if (lazy) {
                mirror_grid_commit(g, mgrid, gv);
                grid_observe<VEC>(g, view_of(s, hb), envp, WURM_OBS_RAW, 0);
            }
            if (lane == 0) { mrec[9] = 0; p.done[env] = GRID_SKIPPED; }
            return;
        } else if (!grid_load<VEC>(g, envp, s)) {
            if (lane == 0) {
                if (mirrored) mrec[9] = 0;
                p.done[env] = GRID_SKIPPED;
            }
            return;
        }
    }
    WURM_TL(1); // the env's grid is in LDS (or rebuilt)
    const long long a_in = uniform64(a_word);
    const int inj_cell = inj_f ? uniform(inj_word) : -1;
    const int food0 = s.food, T0 = s.T;
    StepEv ev;
    WURM_TL(2); // action loaded
    grid_step(g, s, (a_in >= 0 && a_in < 4) ? (int)a_in : -1, (int)(a_in % 4), inj_f, inj_cell, p.seed, p.call, env_id, ev);
    WURM_TL(3); // stepped
    const bool done = ev.selfc | ev.edgec;
    const int head_body = s.L + (ev.selfc ? ev.v - s.T : 0);
    if (lane == 0) {
        store_action(p.actions, p.act_dtype, env, (long long)ev.a_out);
        p.selfc[env] = (uint8_t)ev.selfc;
        p.reward[env] = ev.eat ? 1.0f : 0.0f;
        p.done[env] = (uint8_t)done;
        p.edgec[env] = (uint8_t)ev.edgec;
        if (p.done_copy) p.done_copy[env] = (uint8_t)done;
    }
    WURM_TL(4); // outputs stored
    if (p.obs_mode != WURM_OBS_NONE)
        grid_observe<VEC>(g, view_of(s, head_body), p.obs + env * p.obs_elems, p.obs_mode, p.obs_n);
    WURM_TL(5); // observation issued

    // ---- the post-step state back to HBM
    const bool rebuild_after = done && p.post_reset;
    if (!rebuild_after && !lazy) {
        if (pre) {
            grid_observe<VEC>(g, view_of(s, head_body), envp, WURM_OBS_RAW, 0); // the env was rebuilt: everything changed
        } else {
            if (s.T != T0) { // the body decayed (:246-249): every cell that held a value holds one less
                for (int it = 0; it < g.iters; ++it) {
                    const int c0 = it * 256 + 4 * lane;
                    const int4v e = read4(g.ex, c0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (e[j] > T0 && e[j] < EX_FOOD && c0 + j != s.hc) envp[2 * C + c0 + j] = (float)max(e[j] - s.T, 0);
                }
            }
            if (lane == 0) {
                envp[C + ev.old_hc] = 0.0f;            // :225-233 the head moved
                envp[C + s.hc] = 1.0f;
                envp[2 * C + s.hc] = (float)head_body; // :258-262 (on the ring too: the reference grows the body there)
                if (ev.eat) {                          // :270-282
                    envp[food0] = 0.0f;
                    if (s.food >= 0) envp[s.food] = 1.0f;
                }
            }
        }
    }
    if (mirrored) { // (never with post_reset: grid_resident_eligible)
        if (s.G > EX_REBASE) { // keep the 16-bit clocks away from the markers (as the rollout does between chunks)
            wave_lds_sync();
            for (int it = 0; it < g.iters; ++it) {
                const int c0 = it * 256 + 4 * lane;
                int4v e = read4(g.ex, c0);
#pragma unroll
                for (int q = 0; q < 4; ++q) e[q] = e[q] >= EX_FOOD ? e[q] : max(e[q] - s.T, 0);
                write4(g.ex, c0, e);
            }
            s.G -= s.T;
            s.T = 0;
            whole = true;
        }
        if (whole) {
            mirror_store_grid(g, mgrid);
        } else if (lane == 0) { // what grid_step wrote: the clock of the new head cell, the marker of a respawned food
            if (!ev.edgec) mgrid[s.hc] = (cell_t)s.G;
            if (ev.eat && s.food >= 0) mgrid[s.food] = (cell_t)EX_FOOD;
        }
        mirror_store_rec(g, mrec, s, head_body, MR_ACT | (done ? MR_TERMINAL : 0));
        wave_lds_sync();
    }
    WURM_TL(6); // state / mirror stored; WURM_TL_STORE: drained
#ifdef WURM_TIMELINE
    if (!p.post_reset && p.obs_after == nullptr && p.obs_mode != WURM_OBS_NONE) WURM_TL_STORE(p.obs + env * p.obs_elems, lane);
#endif
    if (!p.post_reset && p.obs_after == nullptr) return;
    if (done) grid_reset(g, s, p.seed, p.call + 1ull, env_id, p.inject_reset ? p.inject_reset + env * 4 : nullptr);
    if (rebuild_after) grid_observe<VEC>(g, view_of(s, s.L), envp, WURM_OBS_RAW, 0);
    if (p.obs_after != nullptr && p.obs_mode != WURM_OBS_NONE)
        grid_observe<VEC>(g, view_of(s, done ? s.L : head_body), p.obs_after + env * p.obs_elems, p.obs_mode, p.obs_n);
}

// envs from the mirror (lazy form): every env its record describes is written whole
template <bool VEC>
__global__ __launch_bounds__(256) void grid_flush_kernel(StepArgs p)
{
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6);
    const long long env = xcd_block(blockIdx.x, gridDim.x) * wpb + wave;
    if (env >= p.N) return;
    const Grid g = make_grid(p, wave, env);
    Snk s;
    int hb = 0;
    const int flags = mirror_load_rec(g, mirror_rec(p, g, env), s, hb);
    if (!(flags & MR_ACT)) return;
    mirror_load_grid(g, mirror_grid(p, g, env));
    grid_observe<VEC>(g, view_of(s, hb), p.envs + env * 3 * g.C, WURM_OBS_RAW, 0);
}

} // namespace

// From which grid size on the clock-grid rollout beats the one-env-per-wave kernels it replaced depends on how much of a
// step is observation: measured at 65 536 and 8 192 envs, 16 steps per launch (tools/grid_vs_generic_probe.py,
// profiles/r06_grid_vs_generic.txt; ratio = clock grid / one env per wave): 'default' 1.09-1.38 at 12 / 13, 0.78-0.97 from 14
// on; 'partial_2' 1.01-1.21 up to 16, 0.78-0.82 at 20; 'one_channel' 1.31-1.75 up to 20, 1.03-1.06 at 24.
// WURM_GRID_ROLLOUT_MIN_SIZE >= 12 forces one threshold for every mode (tests keep the kernel covered at every size).
bool grid_rollout_eligible(const StepArgs &p)
{
    int min_size = (int)opt.grid_rollout_min_size;
    if (min_size < 12)
        min_size = p.obs_mode == WURM_OBS_ONE_CHANNEL ? 26 : (p.obs_mode == WURM_OBS_DEFAULT || p.obs_mode == WURM_OBS_RAW) ? 14 : 18;
    return p.S >= min_size && p.S <= 64 && p.T <= (1ll << 26);
}

bool grid_step_eligible(const StepArgs &p) { return p.S >= 12 && p.S <= 64; }

bool grid_resident_eligible(const StepArgs &p)
{
    return grid_step_eligible(p) && !p.inject_food && !p.inject_reset && !p.inject_pre_reset && !p.post_reset && !p.only_flagged;
}

long long grid_resident_bytes(long long N, int S)
{
    const long long iters = ((long long)S * S + 255) >> 8;
    return N * (iters * 256 * (long long)sizeof(cell_t) + MR_INTS * 4);
}

static bool grid_aligned(const StepArgs &p)
{
    const bool plain = p.obs_mode == WURM_OBS_PARTIAL || p.obs_mode == WURM_OBS_POSITIONS || p.obs_mode == WURM_OBS_NONE;
    return (p.S * p.S) % 4 == 0 && ((uintptr_t)p.envs % 16 == 0) && ((uintptr_t)p.obs % 16 == 0) &&
           ((uintptr_t)p.obs_after % 16 == 0) && (p.obs_elems % 4 == 0 || plain);
}

hipError_t launch_grid_rollout(const StepArgs &p_in, hipStream_t stream)
{
    StepArgs p = p_in;
    p.grid_rotate = (int)opt.grid_rotate;
    const int C = p.S * p.S, iters = (C + 255) >> 8;
    const int wpb = p.N <= 4096 ? 1 : 4;
    dim3 block(64 * wpb), grid((unsigned)((p.N + wpb - 1) / wpb));
    size_t lds = (size_t)iters * 256 * sizeof(cell_t) * wpb;
    // Residency.  Dynamic LDS is the only per-launch handle on how many waves share a CU.  Measured on MI355X (BASELINE
    // configs[4], 16-step launches, median of 5 x 20 launches): 0.38-0.39 ms at 8-12 waves per CU, 0.39-0.46 ms with all
    // 32 resident — with every wave resident the launch runs in lockstep (all probing, then all storing) more often.
    // The effect is at the edge of the run-to-run noise; 12 is kept, WURM_GRID_WAVES_PER_CU overrides it.
    const int waves_per_cu = (int)std::max(1ll, opt.grid_waves_per_cu); // tuning knob
    const size_t per_wave_target = (160u * 1024u / (unsigned)waves_per_cu) & ~255u;
    lds = std::min<size_t>(std::max(lds, per_wave_target * wpb), 64u * 1024u);
    (void)hipGetLastError();
    if (grid_aligned(p)) WURM_LAUNCH(grid_rollout_kernel<true>, grid, block, lds, stream, p);
    else WURM_LAUNCH(grid_rollout_kernel<false>, grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_grid_step(const StepArgs &p_in, hipStream_t stream)
{
    StepArgs p = p_in;
    p.grid_rotate = (int)opt.grid_rotate;
    const int C = p.S * p.S, iters = (C + 255) >> 8;
    const int wpb = p.N <= 4096 ? 1 : 4;
    dim3 block(64 * wpb), grid((unsigned)((p.N + wpb - 1) / wpb));
    const size_t lds = (size_t)iters * 256 * sizeof(cell_t) * wpb; // one step per launch: as many waves per CU as fit
    (void)hipGetLastError();
    if (grid_aligned(p)) WURM_LAUNCH(grid_step_kernel<true>, grid, block, lds, stream, p);
    else WURM_LAUNCH(grid_step_kernel<false>, grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_grid_resident_flush(const StepArgs &p_in, hipStream_t stream)
{
    StepArgs p = p_in;
    const int C = p.S * p.S, iters = (C + 255) >> 8;
    const int wpb = p.N <= 4096 ? 1 : 4;
    dim3 block(64 * wpb), grid((unsigned)((p.N + wpb - 1) / wpb));
    const size_t lds = (size_t)iters * 256 * sizeof(cell_t) * wpb;
    (void)hipGetLastError();
    if (C % 4 == 0 && (uintptr_t)p.envs % 16 == 0) WURM_LAUNCH(grid_flush_kernel<true>, grid, block, lds, stream, p);
    else WURM_LAUNCH(grid_flush_kernel<false>, grid, block, lds, stream, p);
    return hipGetLastError();
}

} // namespace wurm
