// lane_rollout.hpp — the fused rollout (T iterations of `step(a[t]); reset(done)`) for LARGE batches of 9 x 9 SingleSnake
// envs: ONE ENV PER LANE.
//
// rollout_s9_kernel (one env per wave) spends a whole wave's 30 SALU + 28 VALU + 3 stores on every env-step; with 8 192 or
// 65 536 envs resident it is bound by instruction issue at 0.40-0.44 of the HBM peak although it moves no byte twice
// (profiles/r02_cfg2_rollout_instmix_pmc.json).  Here a wave owns EPW consecutive envs (EPW = 4 ... 64) and
// works on CHUNKS of TC = 64 / EPW steps, so that every lane is one (step, env) PAIR of the chunk wherever the work is
// not inherently serial:
//   (1) pair lanes: the action of (t, env) -> sanitise inputs; both Philox blocks of (t, env) — the food draw and the
//       complete would-be reset, in closed form as in rollout_s9_kernel — into an LDS record;
//   (2) env lanes (lanes 0 .. EPW-1), TC steps one after another: the transition of single_snake.py:197-304 on per-lane
//       state — a 64-bit occupancy mask over cell codes 8 * row + column, the body as a QUEUE OF MOVES (2 bits per
//       segment, 96 bits: the tail cell is found by popping the oldest move, so "every body cell decays" costs nothing per
//       cell), head / tail codes, length, orientation, food code; food respawn = n-th set bit of `interior & ~occupied`;
//       reset = unpacking the precomputed draw.  Each step leaves a 12-byte record of the stepped (pre-reset) state;
//   (3) pair lanes: record -> the 5 x 5 window of the occupancy mask around the head (one 64-bit shift), window planes
//       "value is 1" / "value is 127/255" for the three channels, compacted to 75 bits in the crop's own (c, y, x) order
//       and OR-ed into the chunk's flat bit string in LDS; the per-step outputs (sanitised action, reward, done and its
//       two causes) go out as coalesced stores;
//   (4) all lanes, 19 iterations per chunk: 16-byte group j of the chunk's crops (EPW * 300 contiguous bytes per step) =
//       one aligned nibble of each flat bit string -> one 16-byte read of a 256-entry table of float4 -> one
//       global_store_dwordx4, 1 KB per wave instruction, no address arithmetic.
// The state is read once (cooperatively: three coalesced dword loads per (env, cell) pair, non-zero elements scattered into
// a per-env value -> cell table) and written once per launch.
// Domain: S = 9, observation 'partial_2' or none; snakes whose body values are exactly 1..L once each on edge-adjacent
// interior cells with the head on L and at most one food on a free interior cell (closed under step + reset).  Any other
// env is left alone and rolled out by rollout_generic — the one-env-per-wave code — at the end of the same launch.
// INJ: recorded random outcomes instead of Philox (tests/golden tapes through this kernel).
#pragma once

#include <type_traits>

#include "lane_load.hpp"

namespace wurm {

constexpr int LR_C = 81, LR_C3 = 243, LR_E = 75; // cells, state floats and crop floats per env
constexpr int LR_VS = 68;                        // bytes per env of the value -> cell table (17 dwords)
constexpr int LR_BM = 84;                        // bytes per env of the body map written back (21 dwords)
constexpr int LR_TAB = 4096 + 656;               // workgroup tables: 256 x float4, 81 x u64 window-interior masks

// per-wave LDS (bytes)
template <int EPW>
struct LaneRollLds {
    static constexpr int IO = 0;                           // uint4 [64]   step inputs of pair (s, e), then its state record
    static constexpr int BITS = IO + 1024;                 // u32 [2][152] flat bit strings: value is 1 / value is 127/255
    static constexpr int SCR = BITS + 2 * 608;             // start / end of launch scratch (and rollout_generic's class map)
    static constexpr int VM = SCR;                         // u32 [2][EPW] bit set of body values present
    static constexpr int STAT = VM + 8 * EPW;              // u32 [EPW]    count | heads << 8 | foods << 16 | bad << 24
    static constexpr int HPOS = STAT + 4 * EPW;            // u8  [EPW]
    static constexpr int FPOS = HPOS + EPW;                // u8  [EPW]
    static constexpr int VALPOS = FPOS + EPW;              // u8  [EPW][LR_VS]
    static constexpr int QUEUE = (VALPOS + EPW * LR_VS + 15) & ~15; // non-zero float4s found by the state read (lane_load.hpp)
    static constexpr int START_END = QUEUE + LANE_QUEUE_BYTES;
    static constexpr int BMAP = SCR;                       // u8  [EPW][LR_BM] body values by cell (end of launch, slow path)
    static constexpr int HC = BMAP + EPW * LR_BM;          // s16 [EPW] head cell (-1: env not written back)
    static constexpr int FC = HC + 2 * EPW;                // s16 [EPW] food cell (-1: none)
    static constexpr int SLAB = SCR;                       // u8  [EPW * 243] the block's state, one byte per float (fast path)
    static constexpr int END_SLOW = FC + 2 * EPW, END_FAST = SLAB + EPW * LR_C3 + 16;
    static constexpr int END_END = END_SLOW > END_FAST ? END_SLOW : END_FAST;
    static constexpr int BYTES = ((START_END > END_END ? START_END : END_END) + 15) & ~15;
    static_assert(BYTES - SCR >= 96, "scratch must hold rollout_generic's class map");
};

// row / column / code step of move index ai (= sanitised action & 3): -TAP[ai] (single_snake.py:225-233)
__device__ __forceinline__ int lr_dcode(int ai) { return (int)(signed char)(0x01F8FF08u >> (8 * ai)); } // +8, -1, -8, +1
__device__ __forceinline__ int lr_dy(int ai) { return (int)(signed char)(0x00FF0001u >> (8 * ai)); }    // +1, 0, -1, 0
__device__ __forceinline__ int lr_dx(int ai) { return (int)(signed char)(0x0100FF00u >> (8 * ai)); }    // 0, -1, 0, +1

// rows of 5 bits at byte stride (window layout, bit 8 i + j) -> 25 contiguous bits (bit 5 i + j)
__device__ __forceinline__ u32 lr_compact(u32 lo, u32 hi)
{
    return (lo & 31u) | ((lo >> 3) & (31u << 5)) | ((lo >> 6) & (31u << 10)) | ((lo >> 9) & (31u << 15)) | ((hi & 31u) << 20);
}

// Per-lane state of one env (plain scalars, passed by reference — kept out of a struct: the compiler turns a select
// between adjacent struct members into an indexed load and then keeps the whole struct in scratch): occ = occupancy over
// cell codes 8 * row + column, q0..q2 = the body as a queue of moves (newest in bits 0-1 of q0), c / tc = head / tail codes,
// L = length, o = orientation, food = food code (-1: none), act = the env is in the kernel's domain.
constexpr u64 LR_RING = 0x01010101010101FFull; // codes of the border ring, modulo 64 (row 0 / 8: 0..8, columns 0 / 8: 8 r)
constexpr u64 LR_INTERIOR = ~LR_RING;          // codes 8 r + c, r, c in 1..7

// One transition (single_snake.py:197-304) and, for a finished env, the reset that follows it (:322-387).
// in:  x = action bits (the action if it is 0..3 else 7 | (action % 4 & 7) << 3), y = would-be reset, z = food word;
// out: x, y = occupancy of the stepped state, z = head code before the move | sanitised action << 8,
//      w = food code + 1 | ate << 8 | self collision << 9 | edge collision << 10 | valid << 15
// RESET = false: the step alone (the per-call kernel of lane_resident.hpp rebuilds finished envs in the NEXT launch)
// A state as one byte per float of its 'raw' observation (single_snake.py:139-140: the state itself), 243 bytes at `raw`,
// zeroed by the caller: food code, head (hy, hx) — true coordinates, also on the ring —, the neck's code c_neck and the
// queue of moves BEHIND the newest one (w0..w2), length L.  Body: the neck has L - 1, ... the tail 1 (walking the queue from
// the neck); the head cell ADDS L (:258-262) — after a self collision it shows the sum with the segment it ran into.
__device__ __forceinline__ void lr_raw_bytes(unsigned char *raw, int food, int hy, int hx, int c_neck, u32 w0, u32 w1, u32 w2, int L)
{
    if (food >= 0) raw[9 * (food >> 3) + (food & 7)] = 1;
    raw[LR_C + 9 * hy + hx] = 1;
    int code = c_neck;
    for (int v = L - 1; v >= 1; --v) {
        raw[2 * LR_C + 9 * (code >> 3) + (code & 7)] = (unsigned char)v;
        code -= lr_dcode((int)(w0 & 3u));
        w0 = (w0 >> 2) | (w1 << 30); w1 = (w1 >> 2) | (w2 << 30); w2 >>= 2;
    }
    raw[2 * LR_C + 9 * hy + hx] += (unsigned char)L;
}

// RAWB: the stepped (pre-reset) state also as one byte per float of the 'raw' observation (lr_raw_bytes)
template <bool INJ, bool RESET = true, bool RAWB = false>
__device__ __forceinline__ uint4 lr_transition(u64 &occ, u32 &q0, u32 &q1, u32 &q2, int &c, int &tc, int &L, int &o, int &food,
                                               const bool act, const uint4 &cur, unsigned char *raw = nullptr)
{
    u32 rz = 0, rw = 0;
    u64 occ_rec = 0;
    if (act) {
        const int a_small = (int)(cur.x & 7u), a_mod = ((int)(cur.x << 26)) >> 29;
        const int a_out = o == a_small ? (o ^ 2) : a_mod;        // :221-222
        const int ai = a_out & 3;
        const int cn = c + lr_dcode(ai), cb = cn & 63;               // :225-233
        const bool eat = cn == food;                                 // :242
        const int pos = 2 * L - 4;                                   // the oldest move
        const u32 qs = pos < 32 ? q0 : (pos < 64 ? q1 : q2);
        const int m = (int)((qs >> (pos & 31)) & 3u);
        const u64 occ_d = eat ? occ : occ & ~(1ull << tc);     // :246-249 (only the tail cell expires)
        tc = eat ? tc : tc + lr_dcode(m);
        const u32 selfc = (u32)(occ_d >> cb) & 1u;                      // :252
        const u32 edge = (u32)(LR_RING >> cb) & 1u;                     // :290-295
        q2 = (q2 << 2) | (q1 >> 30); q1 = (q1 << 2) | (q0 >> 30); q0 = (q0 << 2) | (u32)ai;
        occ = occ_d | (1ull << cb);                                  // :258-262
        L += eat ? 1 : 0;
        const int c_prev = c;
        c = cn;
        o = ai ^ 2;
        if (eat) {                                                      // :270-282
            if constexpr (INJ) {
                food = (int)cur.z - 1;
            } else {
                const u64 fr = LR_INTERIOR & ~occ;
                const int n_free = __popcll(fr);
                food = n_free ? nth_bit64(fr, (int)mulhi_range(cur.z, (u32)n_free)) : -1;
            }
        }
        occ_rec = occ;
        rz = (u32)c_prev | (((u32)a_out & 0xffu) << 8);
        rw = (u32)(food + 1) | ((u32)eat << 8) | (selfc << 9) | (edge << 10) | 0x8000u;
        if constexpr (RAWB)
            lr_raw_bytes(raw, food, (c_prev >> 3) + lr_dy(ai), (c_prev & 7) + lr_dx(ai), c_prev, (q0 >> 2) | (q1 << 30),
                         (q1 >> 2) | (q2 << 30), q2 >> 2, L);
        if (RESET && (selfc | edge)) {                                  // :322-387
            const u32 r = cur.y;
            const int hc = (int)(r & 127u), sc = (int)((r >> 7) & 127u), d = (int)((r >> 21) & 3u);
            tc = (int)((r >> 14) & 127u);
            c = hc; o = d; L = 3;
            food = (int)((r >> 23) & 127u) - 1;
            occ = (1ull << hc) | (1ull << sc) | (1ull << tc);
            q0 = (u32)((d ^ 2) * 5); q1 = 0; q2 = 0;
        }
    }
    return make_uint4((u32)occ_rec, (u32)(occ_rec >> 32), rz, rw);
}

// workgroup tables (LR_TAB bytes): tab[i] = the four floats of nibble pair i (low nibble: "value is 1", high nibble: "value
// is 127/255"); wint[head cell] = the cells of the 5 x 5 window (bit 8 i + j) that lie inside the border ring
__device__ __forceinline__ void lr_build_tables(float4 *tab, u64 *wint)
{
    constexpr int S = 9;
    for (int i = (int)threadIdx.x; i < 256; i += (int)blockDim.x) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ((i >> j) & 1) ? 1.0f : ((i >> (4 + j)) & 1) ? 127.0f / 255.0f : 0.0f;
        tab[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
    for (int i = (int)threadIdx.x; i < LR_C; i += (int)blockDim.x) {
        const int hy = i / S, hx = i - hy * S;
        // window column j shows grid column hx - 2 + j: inside the ring for max(0, 3 - hx) <= j <= min(4, 9 - hx); rows alike
        const int j0 = max(0, 3 - hx), j1 = min(4, 9 - hx);
        const u64 cols = j1 >= j0 ? (u64)(((2u << j1) - 1u) & ~((1u << j0) - 1u)) : 0ull;
        u64 m = 0;
#pragma unroll
        for (int wy = 0; wy < 5; ++wy)
            if ((unsigned)(hy - 2 + wy - 1) < 7u) m |= cols << (8 * wy);
        wint[i] = m;
    }
}

// ---- every other observation of the 9 x 9 state (round 4): 'default' (243 floats), 'one_channel' (81, the reference's
// constructor default: single_snake.py:55-65), 'positions' (4) and 'partial_n' with n != 2 (3 (2n+1)^2) — written FLOAT BY
// FLOAT from the step records: lane = one float of a step's row of EPW consecutive observations (256 contiguous bytes per
// store instruction), its (channel, row, column) read from a workgroup table, its value a function of the record's
// occupancy mask / head / food codes (single_snake.py:104-195).  ('raw' needs the body VALUES: LR_OBS_RAW below, round 5;
// 'default' / 'one_channel' / 'partial_3' have their own bit-plane forms since.)
constexpr int LR_OBS_GENERIC = -1; // OBSK of lane_rollout_kernel: p.obs_mode / p.obs_n at run time

// lut[r] = channel | a << 2 | b << 6 of float r of one env's observation: (a, b) = (row, column) of the grid for the grid
// modes, of the window for a crop
__device__ __forceinline__ void lr_build_lut(unsigned short *lut, int mode, int n, int E)
{
    for (int r = (int)threadIdx.x; r < E; r += (int)blockDim.x) {
        int ch, a, b;
        if (mode == WURM_OBS_PARTIAL) {
            const int W = 2 * n + 1, W2 = W * W;
            ch = r / W2;
            const int w = r - ch * W2;
            a = w / W; b = w - a * W;
        } else if (mode == WURM_OBS_POSITIONS) {
            ch = r; a = b = 0;
        } else {
            ch = r / LR_C;
            const int cell = r - ch * LR_C;
            a = cell / 9; b = cell - a * 9;
        }
        lut[r] = (unsigned short)(ch | (a << 2) | (b << 6));
    }
}

// value of float (ch, a, b) of the observation of a state given as occupancy mask `oc` (cell codes 8 y + x), head (hy, hx)
// — also when it is on the ring — and food code fc (-1: none)
__device__ __forceinline__ float lr_obs_value(u64 oc, int hy, int hx, int fc, u32 l, int mode, int n)
{
    const int ch = (int)(l & 3u), a = (int)((l >> 2) & 15u), b = (int)((l >> 6) & 15u);
    if (mode == WURM_OBS_POSITIONS) // argmax of the head / food channels (:153-163): (0, 0) when there is no food
        return (float)(ch == 0 ? hy : ch == 1 ? hx : ch == 2 ? (fc >= 0 ? fc >> 3 : 0) : (fc >= 0 ? fc & 7 : 0));
    const bool crop = mode == WURM_OBS_PARTIAL;
    const int y = crop ? hy - n + a : a, x = crop ? hx - n + b : b;
    const bool inside = (unsigned)y < 9u && (unsigned)x < 9u;          // F.pad zeros around the grid (:179)
    const bool ring = y == 0 || y == 8 || x == 0 || x == 8;
    const bool interior = inside && !ring;
    const int code = 8 * y + x;
    const bool occ = interior && ((oc >> (code & 63)) & 1ull) != 0;
    const bool head = y == hy && x == hx, food = interior && fc == code;
    if (mode == WURM_OBS_ONE_CHANNEL) // :142-151
        return ring ? -1.0f : (occ ? 0.5f : 0.0f) + (head ? 0.5f : 0.0f) + (food ? 1.5f : 0.0f);
    // 'default' / crops: background white, body (0,127,0), head (0,255,0), food (255,0,0), ring / padding black (:104-128)
    if (!interior) return 0.0f;
    if (food) return ch == 0 ? 1.0f : 0.0f;
    if (head) return ch == 1 ? 1.0f : 0.0f;
    if (occ) return ch == 1 ? 127.0f / 255.0f : 0.0f;
    return 1.0f;
}

// the observations of a chunk's nt steps x nenv envs from their records io[step * EPW + env]
template <int EPW>
__device__ __forceinline__ void lr_write_generic(const StepArgs &p, const uint4 *io, const unsigned short *lut, float *obs_c,
                                                 int nt, int nenv, int lane)
{
    const int E = (int)p.obs_elems, mode = p.obs_mode, n = p.obs_n;
    const float rcpE = 1.0f / (float)E;
    const int row = nenv * E;                               // < 2^16: div_size is exact
    for (int s = 0; s < nt; ++s) {
        float *orow = obs_c + (long long)s * p.N * E;
        const uint4 *recs = io + s * EPW;
#pragma unroll 2
        for (int f = lane; f < row; f += 64) {
            const int e = div_size(f, rcpE), r = f - e * E;
            const uint4 rec = recs[e];
            const u32 l = lut[r];
            const int cp = (int)(rec.z & 63u), ai = (int)((rec.z >> 8) & 3u);   // head code before the move, the move
            if (rec.w & 0x8000u) // (else an env outside the domain: the fallback writes it)
                orow[f] = lr_obs_value((u64)rec.x | ((u64)rec.y << 32), (cp >> 3) + lr_dy(ai), (cp & 7) + lr_dx(ai),
                                       (int)(rec.w & 127u) - 1, l, mode, n);
        }
    }
}

// ---- the two WHOLE-GRID observations through bit planes (round 4): 'one_channel' (81 floats: the reference's constructor
// default) and 'default' (3 x 81 floats).  The float-by-float writer above costs ~80 VALU per 64 floats and was SLOWER
// than the one-env-per-wave kernels at 65 536 envs (one_channel 3.9e9 against 4.4e9 env-steps/s, default 1.6e9 against
// 3.1e9); here, as for the 'partial_2' crops: the (step, env) pair lane turns its record into 81-bit row-major planes
// (bit 9 y + x), ORs them into the chunk's flat bit strings in LDS, and every lane then turns aligned nibbles of those
// strings into four floats with a table read and stores 16 bytes.
//   'default' (single_snake.py:104-128, :134-136): two planes of 243 bits per pair — "value is 1" = R | G << 81 | B << 162
//       with R = free or food, G = free or head, B = free (all inside the ring), "value is 127/255" = body << 81 — and the
//       nibble-pair table of the crops;
//   'one_channel' (:142-151: 0.5 body + 0.5 head + 1.5 food, ring -1): four planes of 81 bits — body without the head
//       (0.5), head (1.0 = 0.5 + 0.5), food (1.5), ring (-1) — and two tables whose results are ADDED: the planes exclude each
//       other, so one of the two addends is always +0 and the sum is exact.
constexpr int LR_OBS_GRID1 = -2, LR_OBS_GRID3 = -3;   // OBSK of the kernels: one_channel / default through bit planes
constexpr int LR_OBS_CROP3 = -4;                      // 'partial_3' (round 5): 7 x 7 crops through the same bit planes, 147 floats per env
constexpr int LR_OBS_RAW = -5;                        // 'raw' (round 5): the state itself, through one byte per float
constexpr int LR_E3 = 147;                            // floats of a 7 x 7 crop
constexpr int LR_RAW_SLAB = 64 * LR_C3 + 16;          // bytes of a chunk's 'raw' observations, one byte per float (64 pairs)
constexpr int LR_GRID_BITS = 4096;                    // bytes per wave of their flat bit strings (behind LaneRollLds::BYTES)
constexpr int LR_TAB_GRID = 8192 + 512;               // their workgroup tables: 2 x 256 float4, the float-by-float lut

constexpr u64 LR_I9_LO = (0x7Full << 10) | (0x7Full << 19) | (0x7Full << 28) | (0x7Full << 37) | (0x7Full << 46) | (0x7Full << 55);
constexpr u32 LR_I9_HI = 0x7Fu;                       // interior cells of the 9 x 9 grid, bit 9 y + x (bits 64 .. 80 in HI)

// interior cells of an occupancy mask over codes 8 y + x -> bit 9 y + x
__device__ __forceinline__ void lr_rows9(u64 oc, u64 &lo, u32 &hi)
{
    lo = 0;
#pragma unroll
    for (int y = 1; y <= 6; ++y) lo |= ((oc >> (8 * y + 1)) & 0x7Full) << (9 * y + 1);
    hi = (u32)(oc >> 57) & 0x7Fu;
}

// ORs the 81 bits (lo, hi) into plane k of NPL interleaved flat bit strings at bit offset off (word w of plane k: bits[NPL w + k])
template <int NPL>
__device__ __forceinline__ void lr_or81(u32 *bits, int k, int off, u64 lo, u32 hi)
{
    const int w = off >> 5, sb = off & 31;
    const u64 a = (u64)(u32)lo << sb, b = (u64)(u32)(lo >> 32) << sb, c = (u64)hi << sb;
    u32 *P = bits + NPL * w + k;
    atomicOr(&P[0], (u32)a);
    atomicOr(&P[NPL], (u32)(a >> 32) | (u32)b);
    atomicOr(&P[2 * NPL], (u32)(b >> 32) | (u32)c);
    if ((u32)(c >> 32)) atomicOr(&P[3 * NPL], (u32)(c >> 32));
}

// the planes of pair p (bit offset E * p of every string) of a state: occupancy oc, head (hy, hx), food code fc (-1: none)
template <int OBSK>
__device__ __forceinline__ void lr_grid_planes(u32 *bits, int p, u64 oc, int hy, int hx, int fc)
{
    u64 olo; u32 ohi;
    lr_rows9(oc, olo, ohi);
    const bool hin = (unsigned)(hy - 1) < 7u && (unsigned)(hx - 1) < 7u;
    const int hp = 9 * hy + hx, fp = fc >= 0 ? 9 * (fc >> 3) + (fc & 7) : -1;
    const u64 hlo = hin && hp < 64 ? 1ull << hp : 0ull, flo = fp >= 0 && fp < 64 ? 1ull << fp : 0ull;
    const u32 hhi = hin && hp >= 64 ? 1u << (hp - 64) : 0u, fhi = fp >= 64 ? 1u << (fp - 64) : 0u;
    if (OBSK == LR_OBS_GRID1) {
        const int off = LR_C * p;
        lr_or81<4>(bits, 0, off, olo & ~hlo, ohi & ~hhi);           // body without the head: 0.5
        lr_or81<4>(bits, 1, off, hlo, hhi);                         // head: 1.0
        lr_or81<4>(bits, 2, off, flo, fhi);                         // food: 1.5
        lr_or81<4>(bits, 3, off, ~LR_I9_LO, ~LR_I9_HI & 0x1FFFFu);  // ring: -1
    } else {
        const int off = LR_C3 * p;
        const u64 free_lo = LR_I9_LO & ~olo & ~flo;
        const u32 free_hi = LR_I9_HI & ~ohi & ~fhi;
        lr_or81<2>(bits, 0, off, free_lo | flo, free_hi | fhi);               // R: free or food
        lr_or81<2>(bits, 0, off + LR_C, free_lo | hlo, free_hi | hhi);        // G: free or head
        lr_or81<2>(bits, 0, off + 2 * LR_C, free_lo, free_hi);                // B: free
        lr_or81<2>(bits, 1, off + LR_C, olo & ~hlo, ohi & ~hhi);              // G = 127/255: body
    }
}

// tables of the grid modes: tabA as lr_build_tables's for 'default'; for 'one_channel' tabA[low nibble: 0.5, high: 1.0],
// tabB[low nibble: 1.5, high: -1.0]
template <int OBSK>
__device__ __forceinline__ void lr_build_grid_tables(float4 *tabA, float4 *tabB)
{
    for (int i = (int)threadIdx.x; i < 256; i += (int)blockDim.x) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool lo = ((i >> j) & 1) != 0, hi = ((i >> (4 + j)) & 1) != 0;
            if (OBSK == LR_OBS_GRID1) { a[j] = lo ? 0.5f : hi ? 1.0f : 0.0f; b[j] = lo ? 1.5f : hi ? -1.0f : 0.0f; }
            else { a[j] = lo ? 1.0f : hi ? 127.0f / 255.0f : 0.0f; b[j] = 0.0f; }
        }
        tabA[i] = make_float4(a[0], a[1], a[2], a[3]);
        if (OBSK == LR_OBS_GRID1) tabB[i] = make_float4(b[0], b[1], b[2], b[3]);
    }
}

// 16-byte group j of a flat run of floats whose bits start at bit 0 of the strings -> four floats
template <int OBSK>
__device__ __forceinline__ float4 lr_grid_group(const u32 *bits, const float4 *tabA, const float4 *tabB, int j)
{
    const int w = j >> 3, sh = (j & 7) * 4;
    if (OBSK == LR_OBS_GRID1) {
        const uint4 q = ((const uint4 *)bits)[w];
        const float4 a = tabA[((q.x >> sh) & 15u) | (((q.y >> sh) & 15u) << 4)];
        const float4 b = tabB[((q.z >> sh) & 15u) | (((q.w >> sh) & 15u) << 4)];
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    const uint2 q = ((const uint2 *)bits)[w];
    return tabA[((q.x >> sh) & 15u) | (((q.y >> sh) & 15u) << 4)];
}

// ---- 'partial_3' (round 5): the 7 x 7 window of the occupancy mask around the head is still ONE 64-bit shift (window bit
// 8 i + j <-> code sh + 8 i + j, i, j < 7: 55 bits); the planes of the 'default' colours restricted to the window cells
// inside the ring, rows of 7 bits compacted to 49 contiguous bits per channel, 147 bits per pair in the crop's (c, y, x) order
__device__ __forceinline__ void lr_build_wint7(u64 *wint)
{
    for (int i = (int)threadIdx.x; i < LR_C; i += (int)blockDim.x) {
        const int hy = i / 9, hx = i - hy * 9;
        u64 m = 0;
#pragma unroll
        for (int wy = 0; wy < 7; ++wy)
#pragma unroll
            for (int wx = 0; wx < 7; ++wx)
                if ((unsigned)(hy - 3 + wy - 1) < 7u && (unsigned)(hx - 3 + wx - 1) < 7u) m |= 1ull << (8 * wy + wx);
        wint[i] = m;
    }
}

__device__ __forceinline__ u64 lr_compact7(u64 x)
{
    return (x & 0x7Full) | ((x >> 1) & (0x7Full << 7)) | ((x >> 2) & (0x7Full << 14)) | ((x >> 3) & (0x7Full << 21)) |
           ((x >> 4) & (0x7Full << 28)) | ((x >> 5) & (0x7Full << 35)) | ((x >> 6) & (0x7Full << 42));
}

// ORs 49 bits into plane k of NPL interleaved flat bit strings at bit offset off
template <int NPL>
__device__ __forceinline__ void lr_or49(u32 *bits, int k, int off, u64 v)
{
    const int w = off >> 5, sb = off & 31;
    const u64 a = (u64)(u32)v << sb, b = (u64)(u32)(v >> 32) << sb;
    u32 *P = bits + NPL * w + k;
    atomicOr(&P[0], (u32)a);
    atomicOr(&P[NPL], (u32)(a >> 32) | (u32)b);
    if ((u32)(b >> 32)) atomicOr(&P[2 * NPL], (u32)(b >> 32));
}

// planes of pair p: occupancy oc of the stepped state, head (hy, hx) — also on the ring —, food code fc (-1: none)
__device__ __forceinline__ void lr_crop3_planes(u32 *bits, const u64 *wint7, int p, u64 oc, int hy, int hx, int fc)
{
    const int sh = 8 * hy + hx - 27;
    const u64 V = sh >= 0 ? oc >> sh : oc << (-sh);
    const u64 W = wint7[hy * 9 + hx];
    const int fpos_w = fc - sh;
    const u64 F = fc >= 0 && (unsigned)fpos_w < 56u ? (1ull << fpos_w) & W : 0ull;
    const u64 R = W & ~V;                         // free or food: red
    const u64 B = R & ~F;                         // free: blue (and green)
    const u64 CENTRE = 1ull << 27;
    const u64 G1 = B | (W & CENTRE);              // green 1: free, or the head inside the ring
    const u64 GH = V & W & ~CENTRE;               // green 127/255: body
    const int off = LR_E3 * p;
    lr_or49<2>(bits, 0, off, lr_compact7(R));
    lr_or49<2>(bits, 0, off + 49, lr_compact7(G1));
    lr_or49<2>(bits, 0, off + 98, lr_compact7(B));
    lr_or49<2>(bits, 1, off + 49, lr_compact7(GH));
}

// per-wave LDS of lane_rollout_kernel<EPW, OBSK, ·>: the grid / crop modes keep their flat bit strings behind
// LaneRollLds::BYTES, 'raw' its byte slab from LaneRollLds::SCR on (that scratch is free between the state read and the
// write-back)
template <int EPW, int OBSK>
constexpr int lr_wave_bytes()
{
    typedef LaneRollLds<EPW> Lds;
    if (OBSK == LR_OBS_RAW) return Lds::BYTES > Lds::SCR + LR_RAW_SLAB ? Lds::BYTES : ((Lds::SCR + LR_RAW_SLAB + 15) & ~15);
    if (OBSK == LR_OBS_GRID1 || OBSK == LR_OBS_GRID3 || OBSK == LR_OBS_CROP3) return Lds::BYTES + LR_GRID_BITS;
    return Lds::BYTES;
}
constexpr bool lr_grid_tables(int OBSK) { return OBSK == LR_OBS_GRID1 || OBSK == LR_OBS_GRID3 || OBSK == LR_OBS_CROP3 || OBSK == LR_OBS_RAW; }

// The state of a block of EPW consecutive envs (`block` = its first float), read cooperatively — lanes = (env, cell) pairs,
// the few non-zero elements scattered into a per-env value -> cell table in LDS — then, per env lane: validation and the
// state as occupancy mask + queue of moves.  act = the env is in the domain (header of this file).
template <int EPW>
__device__ __forceinline__ void lr_read_block(const float *block, const bool whole, const int nenv, const int lane,
                                              unsigned char *lds, u64 &occ, u32 &q0, u32 &q1, u32 &q2, int &c, int &tc, int &L,
                                              int &o, int &food, bool &act)
{
    typedef LaneRollLds<EPW> Lds;
    constexpr int S = 9;
    // ---- cooperative read of the state: lanes = (env, cell) pairs of the block, three dwords each (food, head, body)
    u32 *vm = (u32 *)(lds + Lds::VM), *stat = (u32 *)(lds + Lds::STAT);
    unsigned char *hpos = lds + Lds::HPOS, *fpos = lds + Lds::FPOS, *valpos = lds + Lds::VALPOS;
    if (lane < EPW) { vm[lane] = 0; vm[EPW + lane] = 0; stat[lane] = 0; }
    wave_lds_sync();
    if (whole) {
        lane_load_block<EPW, LR_C, 2, LR_VS>(block, lane, vm, stat, hpos, fpos, valpos,
                                             lds + Lds::QUEUE);
    } else {
        constexpr int LOADS = 9;
        const char *base = (const char *)(block);
        const int pairs = nenv * LR_C;
        int e = 0, cell = lane, idx = lane;
        unsigned off = 4u * (unsigned)lane;
        for (int i0 = 0; i0 < pairs; i0 += 64 * LOADS) {
            float f[LOADS], h[LOADS], b[LOADS];
            int es[LOADS], cs[LOADS];
#pragma unroll
            for (int j = 0; j < LOADS; ++j) {
                es[j] = idx < pairs ? e : -1;
                cs[j] = cell;
                const char *q = base + (idx < pairs ? off : 0u);
                f[j] = *(const float *)q;
                h[j] = *(const float *)(q + 4 * LR_C);
                b[j] = *(const float *)(q + 8 * LR_C);
                idx += 64; cell += 64; off += 256;
                if (cell >= LR_C) { cell -= LR_C; ++e; off += 8 * LR_C; }
            }
#pragma unroll
            for (int j = 0; j < LOADS; ++j) {
                const int ej = es[j], cj = cs[j];
                if (ej < 0) continue;
                if (f[j] > 0.5f) { fpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 16); }
                if (h[j] > 0.5f) { hpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 8); }
                const int bi = __float2int_rn(b[j]);
                if (bi > 0 && bi < 64) {
                    valpos[ej * LR_VS + bi] = (unsigned char)cj;
                    atomicOr(&vm[(bi >> 5) * EPW + ej], 1u << (bi & 31));
                    atomicAdd(&stat[ej], 1u);
                } else if (bi != 0) {
                    atomicAdd(&stat[ej], 1u << 24);
                }
            }
        }
    }
    wave_lds_sync();

    // ---- env lanes: validation, and the state as occupancy mask + queue of moves
    occ = 0; q0 = q1 = q2 = 0; c = tc = L = o = 0; food = -1; act = false;
    const bool mine = lane < nenv;
    if (mine) {
        const u32 st = stat[lane];
        const int cnt = (int)(st & 0xffu), nhd = (int)((st >> 8) & 0xffu), nfd = (int)((st >> 16) & 0xffu);
        const u64 vset = (u64)vm[lane] | ((u64)vm[EPW + lane] << 32);
        L = vset ? 63 - __clzll((long long)vset) : 0;
        act = (st >> 24) == 0 && nhd == 1 && nfd <= 1 && L >= 2 && cnt == L && vset == (2ull << L) - 2ull;
        if (act) act = (int)valpos[lane * LR_VS + L] == (int)hpos[lane];
        if (act && nfd) {
            const int fc = fpos[lane], fy = fc / S, fx = fc - fy * S;
            act = (unsigned)(fy - 1) < 7u && (unsigned)(fx - 1) < 7u;
            food = fy * 8 + fx;
        }
    }
    {
        int prev = 0;
        for (int v = 1; ballot(act && v <= L) != 0; ++v) {
            if (act && v <= L) {
                const int cell = valpos[lane * LR_VS + v], y = cell / S, x = cell - y * S;
                const int code = y * 8 + x;
                if (!((unsigned)(y - 1) < 7u && (unsigned)(x - 1) < 7u)) act = false;
                occ |= 1ull << (code & 63);
                if (v == 1) {
                    tc = code;
                } else {
                    const int d = code - prev;
                    const int m = d == 8 ? 0 : d == -1 ? 1 : d == -8 ? 2 : d == 1 ? 3 : -1;
                    if (m < 0) act = false;
                    q2 = (q2 << 2) | (q1 >> 30); q1 = (q1 << 2) | (q0 >> 30); q0 = (q0 << 2) | (u32)(m & 3);
                }
                prev = code;
            }
        }
        c = prev;
        o = (int)(q0 & 3u) ^ 2;                       // head = neck + TAP[o], the last move was -TAP[o ^ 2]
        if (act && food >= 0 && ((occ >> food) & 1)) act = false;
    }
}

// an env outside the domain: the one-env-per-wave rollout, whole wave
template <int OBSK, bool INJ>
__device__ __forceinline__ void lane_rollout_fallback(const StepArgs &p, long long env, signed char *lds)
{
    const Geo g = make_geo<2>(9);
    float *envp = p.envs + env * LR_C3;
    Env<2> e;
    load_state<2, true>(envp, g, e);
    rollout_generic<2, true, (OBSK < 0 ? -1 : OBSK), INJ>(p, env, envp, g, e, lds); // (-1: the mode at run time)
}

template <int EPW, int OBSK, bool INJ>
__global__ __launch_bounds__(256) void lane_rollout_kernel(StepArgs p)
{
    typedef LaneRollLds<EPW> Lds;
    static_assert(EPW == 4 || EPW == 8 || EPW == 16 || EPW == 32 || EPW == 64, "envs per wave");
    static_assert(OBSK == WURM_OBS_PARTIAL || OBSK == WURM_OBS_NONE || OBSK == LR_OBS_GENERIC || OBSK == LR_OBS_GRID1 ||
                  OBSK == LR_OBS_GRID3 || OBSK == LR_OBS_CROP3 || OBSK == LR_OBS_RAW,
                  "lane rollout: partial_2, partial_3, one_channel, default, raw, no observation, or any other mode at run time");
    constexpr bool GRID = OBSK == LR_OBS_GRID1 || OBSK == LR_OBS_GRID3 || OBSK == LR_OBS_CROP3; // flat bit strings -> table
    constexpr bool RAW = OBSK == LR_OBS_RAW;                                                    // byte slab -> floats
    constexpr bool GTAB = lr_grid_tables(OBSK);                 // LDS layout of the grid modes
    constexpr int GE = OBSK == LR_OBS_GRID1 ? LR_C : OBSK == LR_OBS_CROP3 ? LR_E3 : LR_C3; // floats per env of such a mode
    constexpr int GPL = OBSK == LR_OBS_GRID1 ? 4 : 2;           // its interleaved bit planes
    constexpr int TC = 64 / EPW;                  // steps per chunk
    constexpr int LOG_EPW = EPW == 4 ? 2 : EPW == 8 ? 3 : EPW == 16 ? 4 : EPW == 32 ? 5 : 6;
    constexpr int GS = EPW * LR_E / 4;            // 16-byte groups per step of the wave's crops
    constexpr int SUPER = 16;                     // chunks per batch of action loads
    constexpr int S = 9;
    extern __shared__ __attribute__((aligned(16))) unsigned char lr_lds[];
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);

    // ---- workgroup tables
    float4 *tab = (float4 *)lr_lds;               // nibble pair -> four floats
    u64 *wint = (u64 *)(lr_lds + 4096);           // head (row, column) -> window cells that lie inside the border ring
    // (generic observations: float -> channel / row / column, in place of tab; grid modes: behind their two tables)
    unsigned short *lut = (unsigned short *)(lr_lds + (GTAB ? 8192 : 0));
    float4 *tabB = (float4 *)(lr_lds + 4096);
    if (OBSK == LR_OBS_GENERIC || GRID) lr_build_lut(lut, p.obs_mode, p.obs_n, (int)p.obs_elems);
    if (GRID) lr_build_grid_tables<OBSK>(tab, tabB);
    else if (OBSK != LR_OBS_GENERIC && !RAW) lr_build_tables(tab, wint);
    if (OBSK == LR_OBS_CROP3) lr_build_wint7(wint); // (in the place of 'one_channel's second table)
    __syncthreads();

    const long long env0 = (xcd_block(blockIdx.x, gridDim.x) * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    unsigned char *lds = lr_lds + (GTAB ? LR_TAB_GRID : LR_TAB) + wave * lr_wave_bytes<EPW, OBSK>();
    u32 *gbits = (u32 *)(lds + Lds::BYTES);      // (grid modes: the flat bit strings of a chunk)
    unsigned char *gb = lds + Lds::SCR;          // ('raw': the chunk's observations, one byte per float)
    const int nenv = (int)min((long long)EPW, p.N - env0);
    const bool mine = lane < nenv;                // env lanes: lane e owns env0 + e
    const int ps = lane >> LOG_EPW, pe = lane & (EPW - 1); // pair lanes: step ps of the chunk, env env0 + pe

    // Actions: one load per chunk and pair lane, SUPER chunks at a time and one batch AHEAD (loads and stores share vmcnt
    // and retire in order: the wait for a batch issued a whole super-chunk earlier only drains the last few stores, and the
    // chunk loop itself never waits on memory).  Unconditional loads (index clamped into the tape) so that all of a batch
    // are in flight together.
    const long long a_last = p.T * p.N - 1;
    long long av[SUPER];
    auto load_batch = [&](long long t_first) { // the dtype test outside the unrolled loads: SUPER loads back to back
        if (p.act_dtype == WURM_ACT_I64) {
#pragma unroll
            for (int k = 0; k < SUPER; ++k)
                av[k] = ((const long long *)p.actions)[min((t_first + (long long)k * TC + ps) * p.N + env0 + pe, a_last)];
        } else {
            int a32[SUPER];
#pragma unroll
            for (int k = 0; k < SUPER; ++k)
                a32[k] = ((const int *)p.actions)[min((t_first + (long long)k * TC + ps) * p.N + env0 + pe, a_last)];
#pragma unroll
            for (int k = 0; k < SUPER; ++k) av[k] = (long long)a32[k];
        }
    };
    load_batch(0);

    // ---- the state: cooperative read, validation, occupancy mask + queue of moves per env lane
    u64 occ = 0;
    u32 q0 = 0, q1 = 0, q2 = 0;
    int c = 0, tc = 0, L = 0, o = 0, food = -1;
    bool act = false;
    const bool whole = nenv == EPW && (((size_t)p.envs) & 15u) == 0; // full block, 16-byte aligned (env0 is a multiple of 4)
    lr_read_block<EPW>(p.envs + env0 * LR_C3, whole, nenv, lane, lds, occ, q0, q1, q2, c, tc, L, o, food, act);
    const u64 odd = ballot(mine && !act);             // envs outside the domain: rollout_generic below
    wave_lds_sync();

    // ---- the chunks
    uint4 *io = (uint4 *)(lds + Lds::IO);
    u32 *bits = (u32 *)(lds + Lds::BITS);
    const u64 env_id = (u64)(p.env_offset + env0 + pe); // of the pair lane
    const bool pair_env = pe < nenv;
    const int E = OBSK == LR_OBS_GENERIC ? (int)p.obs_elems : (GRID || RAW) ? GE : LR_E;
    float *obs_c = p.obs + env0 * E;                  // observations of the chunk's first step, this wave's envs
    const unsigned obs_step_bytes = (unsigned)(p.N * (LR_E * 4)); // (the launcher keeps TC * N * 300 below 2^32)

    for (long long T0 = 0; T0 < p.T; T0 += SUPER * TC) {
        // 4 bits per action: the action if it is 0..3, else 8 | (action % 4 & 7) (single_snake.py:221-222 needs "equals the
        // orientation" and the C remainder)
        u64 apk = 0;
#pragma unroll
        for (int k = 0; k < SUPER; ++k) {
            const long long a = av[k];
            const u32 code = (a >= 0 && a < 4) ? (u32)a : (8u | ((u32)(int)(a % 4) & 7u));
            apk |= (u64)code << (4 * k);
        }
        if (T0 + SUPER * TC < p.T) load_batch(T0 + SUPER * TC);

        for (int k = 0; k < SUPER; ++k) {
            const long long t0 = T0 + (long long)k * TC;
            if (t0 >= p.T) break;
            const int nt = (int)min((long long)TC, p.T - t0);
            const long long t = t0 + ps;
            const bool pv = pair_env && ps < nt;
            const long long oi = t * p.N + env0 + pe;  // index of the pair's per-step outputs

            // (1) pair lanes: step inputs
            uint4 rec; // first the inputs of pair (ps, pe), then the record of its stepped state
            {
                const u32 acode = (u32)(apk >> (4 * k)) & 15u;
                const u32 a_small = (acode & 8u) ? 7u : acode, a_mod = acode & 7u;
                const u64 call = p.call + 2ull * (u64)t; // step t uses call0 + 2t, its reset call0 + 2t + 1
                u32 rpack, fword;
                if constexpr (INJ) {
                    int sy = 4, sx = 4, d = 0, fc = -1, fe = -1;
                    if (pv) {
                        const int *ir = p.inject_reset + oi * 4;
                        sy = ir[0]; sx = ir[1]; d = ir[2]; fc = ir[3];
                        fe = p.inject_food[oi];
                    }
                    d &= 3;
                    const int sc = sy * 8 + sx, dc = tap_y(d) * 8 + tap_x(d);
                    const int fcy = max(fc, 0) / S, fey = max(fe, 0) / S;
                    const int fcode = (fc >= 0 && fc < LR_C) ? fcy * 8 + (fc - fcy * S) : -1;
                    const int ecode = (fe >= 0 && fe < LR_C) ? fey * 8 + (fe - fey * S) : -1;
                    rpack = (u32)((sc + dc) | (sc << 7) | ((sc - dc) << 14) | (d << 21)) | ((u32)(fcode + 1) << 23);
                    fword = (u32)(ecode + 1);
                } else {
                    const S9Reset r = s9_reset_draw(p.seed, call + 1ull, env_id);
                    rpack = (u32)r.b | ((u32)(r.a & 3) << 21) | ((u32)((r.a >> 2) + 1) << 23);
                    fword = rng_words(p.seed, call, env_id, RNG_FOOD, 0).w[0];
                }
                rec = make_uint4(a_small | (a_mod << 3), rpack, fword, 0u);
            }

            // (2) env lanes: TC transitions (single_snake.py:197-304, then the reset of :322-387 for a finished env).
            // in: x = action bits, y = would-be reset, z = food word;  out: occupancy, head code before the move |
            // sanitised action << 8, food code + 1 | ate << 8 | self collision << 9 | edge collision << 10 | valid << 15
            if (RAW) { // the chunk's byte slab: zero, then the env lanes leave every stepped state in it
                for (int i = lane; i < LR_RAW_SLAB / 16; i += 64) ((uint4 *)gb)[i] = make_uint4(0, 0, 0, 0);
                wave_lds_sync();
            }
            if constexpr (EPW == 64) {
                rec = lr_transition<INJ, true, RAW>(occ, q0, q1, q2, c, tc, L, o, food, act, rec, gb + lane * LR_C3); // pair lane == env lane: the record never leaves its registers
                if (OBSK == LR_OBS_GENERIC || GRID || RAW) io[lane] = rec;                             // (the float-by-float writer reads records by env)
            } else {
                io[lane] = rec;
                wave_lds_sync();
                if (lane < EPW) {
                    uint4 in = io[lane];
                    for (int s2 = 0; s2 < nt; ++s2) {
                        const uint4 cur = in;
                        if (s2 + 1 < nt) in = io[(s2 + 1) * EPW + lane];
                        io[s2 * EPW + lane] = lr_transition<INJ, true, RAW>(occ, q0, q1, q2, c, tc, L, o, food, act, cur,
                                                                            gb + (s2 * EPW + lane) * LR_C3);
                    }
                }
            }
            if (OBSK == WURM_OBS_PARTIAL) { // clear the flat bit strings (the previous chunk's reads are done: LDS is in order)
#pragma unroll
                for (int i = lane; i < 2 * 152; i += 64) bits[i] = 0;
            }
            if (GRID) {
                constexpr int NWORDS = GPL * ((64 * GE + 31) / 32 + 4);
                static_assert(NWORDS * 4 + 16 <= LR_GRID_BITS, "flat bit strings of a grid mode");
                for (int i = lane; i < (NWORDS + 3) / 4; i += 64) ((uint4 *)gbits)[i] = make_uint4(0, 0, 0, 0);
            }
            wave_lds_sync();
            if constexpr (EPW != 64) rec = io[lane];

            // (3) pair lanes: outputs of (t, env) and its crop as bit planes
            {
                const u32 rz = rec.z, rw = rec.w;
                const bool valid = pv && (rw & 0x8000u) != 0;
                if (valid) {
                    store_action(p.actions, p.act_dtype, oi, (long long)(int)(signed char)(rz >> 8));
                    p.reward[oi] = (rw & 0x100u) ? 1.0f : 0.0f;
                    p.done[oi] = (uint8_t)((rw & 0x600u) != 0);
                    p.selfc[oi] = (uint8_t)((rw >> 9) & 1u);
                    p.edgec[oi] = (uint8_t)((rw >> 10) & 1u);
                }
                if (GRID && valid) { // planes of the stepped state (the head also when it is on the ring)
                    const int cp = (int)(rz & 63u), ai = (int)((rz >> 8) & 3u);
                    if constexpr (OBSK == LR_OBS_CROP3)
                        lr_crop3_planes(gbits, wint, lane, (u64)rec.x | ((u64)rec.y << 32), (cp >> 3) + lr_dy(ai), (cp & 7) + lr_dx(ai),
                                        (int)(rw & 127u) - 1);
                    else
                        lr_grid_planes<OBSK>(gbits, lane, (u64)rec.x | ((u64)rec.y << 32), (cp >> 3) + lr_dy(ai), (cp & 7) + lr_dx(ai),
                                             (int)(rw & 127u) - 1);
                }
                if (OBSK == WURM_OBS_PARTIAL && valid) {
                    // crop of the stepped state (single_snake.py:166-193): a window cell that is off the grid or on the
                    // ring is (0,0,0); food (1,0,0), head (0,1,0), body (0,127/255,0), background (1,1,1)
                    const int cp = (int)(rz & 63u), ai = (int)((rz >> 8) & 3u);
                    const int hy = (cp >> 3) + lr_dy(ai), hx = (cp & 7) + lr_dx(ai); // the head, also when it is on the ring
                    const int sh = 8 * hy + hx - 18;              // window bit 8 i + j <-> code sh + 8 i + j
                    const u64 oc = (u64)rec.x | ((u64)rec.y << 32);
                    const u64 V = sh >= 0 ? oc >> sh : oc << (-sh);
                    const u64 W = wint[hy * S + hx];
                    const int fpos_w = (int)(rw & 127u) - 1 - sh;
                    const u64 F = (rw & 127u) != 0 && (unsigned)fpos_w < 40u ? (1ull << fpos_w) & W : 0ull;
                    const u64 R = W & ~V;                         // free or food: red
                    const u64 B = R & ~F;                         // free: blue (and green)
                    const u64 CENTRE = 1ull << 18;
                    const u64 G1 = B | (W & CENTRE);              // green 1: free, or the head inside the ring
                    const u64 GH = V & W & ~CENTRE;               // green 127/255: body
                    const u32 r25 = lr_compact((u32)R, (u32)(R >> 32)), b25 = lr_compact((u32)B, (u32)(B >> 32));
                    const u32 g25 = lr_compact((u32)G1, (u32)(G1 >> 32)), h25 = lr_compact((u32)GH, (u32)(GH >> 32));
                    // 75 bits in (c, y, x) order, shifted to bit 75 * pair of the flat strings (dword w of "value is 1" at
                    // bits[2 w], of "value is 127/255" at bits[2 w + 1]: one 8-byte read gets both in phase 4)
                    const u32 d0 = r25 | (g25 << 25), d1 = (g25 >> 7) | (b25 << 18), d2 = b25 >> 14;
                    const u32 e0 = h25 << 25, e1 = h25 >> 7;
                    const int bit = LR_E * lane, w0 = bit >> 5, sb = bit & 31;
                    const u64 x01 = ((u64)d0 << sb), x12 = (((u64)d2 << 32) | d1) << sb;
                    const u64 y01 = ((u64)e0 << sb), y1 = ((u64)e1 << sb);
                    u32 *P = bits + 2 * w0;
                    atomicOr(&P[0], (u32)x01);
                    atomicOr(&P[2], (u32)(x01 >> 32) | (u32)x12);
                    atomicOr(&P[4], (u32)(x12 >> 32));
                    atomicOr(&P[6], (u32)(((u64)d2 << sb) >> 32));
                    atomicOr(&P[1], (u32)y01);
                    atomicOr(&P[3], (u32)(y01 >> 32) | (u32)y1);
                    atomicOr(&P[5], (u32)(y1 >> 32));
                }
            }

            // (4) all lanes: the chunk's crops, 16 bytes per lane and instruction — staged so that the 19 reads of the bit
            // strings, the 19 table reads and the 19 stores are each in flight together (a lone wave pays one LDS round
            // trip per stage, not per group)
            if (OBSK == WURM_OBS_PARTIAL) {
                wave_lds_sync();
                if (nenv == EPW && nt == TC) {
                    const int shn = (lane & 7) * 4;
                    const uint2 *b2 = (const uint2 *)bits + (lane >> 3);
                    char *ob = (char *)obs_c;
                    // one straight-line block (no per-group branch): the scheduler keeps the reads of the bit strings, the
                    // table reads and the stores of several groups in flight (no local arrays: they would go to scratch)
#pragma unroll
                    for (int i = 0; i < 19; ++i) {
                        const uint2 w = b2[8 * i];
                        const float4 v = tab[((w.x >> shn) & 15u) | (((w.y >> shn) & 15u) << 4)];
                        const int j = 64 * i + lane;               // group j = floats 4 j .. 4 j + 3 of the chunk
                        const unsigned s = EPW == 64 ? 0u : ((unsigned)j * (unsigned)(((1 << 20) + GS - 1) / GS)) >> 20; // j / GS
                        const unsigned off = s * obs_step_bytes + 16u * ((unsigned)j - s * (unsigned)GS);
                        if (i < 18 || lane < 48) *(float4 *)(ob + off) = v; // 1200 groups
                    }
                } else { // the ragged last wave, the last chunk of a tape that is not a multiple of TC: float by float
                    for (int f = lane; f < 64 * LR_E; f += 64) {
                        const int pr = f / LR_E, k2 = f - pr * LR_E, s = pr >> LOG_EPW, e = pr & (EPW - 1);
                        if (s < nt && e < nenv) {
                            const u32 w1 = bits[2 * (f >> 5)], wh = bits[2 * (f >> 5) + 1];
                            const float v = ((w1 >> (f & 31)) & 1u) ? 1.0f : ((wh >> (f & 31)) & 1u) ? 127.0f / 255.0f : 0.0f;
                            obs_c[(long long)s * p.N * LR_E + e * LR_E + k2] = v;
                        }
                    }
                }
            }
            if (OBSK == LR_OBS_GENERIC) {
                wave_lds_sync();
                lr_write_generic<EPW>(p, io, lut, obs_c, nt, nenv, lane);
                wave_lds_sync(); // (the next chunk's inputs go into io)
            }
            if (GRID) {
                wave_lds_sync();
                if (nenv == EPW && nt == TC) {
                    constexpr int GSG = EPW * GE / 4, NGRP = TC * GSG;   // 16-byte groups per step / per chunk
                    char *ob = (char *)obs_c;
                    const size_t step_bytes = (size_t)p.N * (GE * 4);
#pragma unroll 4
                    for (int j = lane; j < NGRP; j += 64) {
                        const float4 v = lr_grid_group<OBSK>(gbits, tab, tabB, j);
                        const int s = EPW == 64 ? 0 : j / GSG;
                        // (a float4 pointer indexed by group: through `char * + 16 j` the EPW = 64 instantiations came out with
                        // every 16-byte store split into four dword stores — see lane_resident.hpp)
                        ((float4 *)(ob + (size_t)s * step_bytes))[j - s * GSG] = v;
                    }
                } else { // the ragged last wave, the last chunk of a tape that is not a multiple of TC: float by float
                    lr_write_generic<EPW>(p, io, lut, obs_c, nt, nenv, lane);
                }
                wave_lds_sync();
            }
            if (RAW) {
                wave_lds_sync();
                if (nenv == EPW && nt == TC) {
                    constexpr int GSG = EPW * LR_C3 / 4, NGRP = TC * GSG;   // 16-byte groups per step / per chunk
                    char *ob = (char *)obs_c;
                    const size_t step_bytes = (size_t)p.N * (LR_C3 * 4);
                    const u32 *g4 = (const u32 *)gb;
#pragma unroll 4
                    for (int j = lane; j < NGRP; j += 64) {
                        const u32 b = g4[j];
                        const int s = EPW == 64 ? 0 : j / GSG;
                        ((float4 *)(ob + (size_t)s * step_bytes))[j - s * GSG] =
                            make_float4((float)(b & 0xffu), (float)((b >> 8) & 0xffu), (float)((b >> 16) & 0xffu), (float)(b >> 24));
                    }
                } else { // the ragged last wave, the last chunk of a tape that is not a multiple of TC: float by float
                    for (int f = lane; f < 64 * LR_C3; f += 64) {
                        const int pr = f / LR_C3, k2 = f - pr * LR_C3, s = pr >> LOG_EPW, e = pr & (EPW - 1);
                        if (s < nt && e < nenv && (io[pr].w & 0x8000u))
                            obs_c[(long long)s * p.N * LR_C3 + e * LR_C3 + k2] = (float)gb[f];
                    }
                }
                wave_lds_sync();
            }
            obs_c += (long long)TC * p.N * E;
        }
    }
    wave_lds_sync();

    // ---- write the state back.  Fast path (full block, every env in the domain): the block as one byte per float in LDS —
    // zeroed, the food / head cells and the body values (walking the queue from the head) scattered by the env lanes —
    // then 4 bytes -> float4 per lane and instruction.
    if (whole && odd == 0) {
        constexpr int N4 = EPW * LR_C3 / 4;
        u32 *slab = (u32 *)(lds + Lds::SLAB);
        unsigned char *sb8 = (unsigned char *)slab;
        for (int i = lane; i < N4 + 4; i += 64) slab[i] = 0;
        wave_lds_sync();
        if (lane < EPW) {
            unsigned char *my = sb8 + lane * LR_C3;
            if (food >= 0) my[(food >> 3) * S + (food & 7)] = 1;
            my[LR_C + (c >> 3) * S + (c & 7)] = 1;
            int code = c;
            u32 w0 = q0, w1 = q1, w2 = q2;
            for (int v = L; v >= 1; --v) {
                my[2 * LR_C + (code >> 3) * S + (code & 7)] = (unsigned char)v;
                code -= lr_dcode((int)(w0 & 3u));
                w0 = (w0 >> 2) | (w1 << 30); w1 = (w1 >> 2) | (w2 << 30); w2 >>= 2;
            }
        }
        wave_lds_sync();
        float4 *out4 = (float4 *)(p.envs + env0 * LR_C3);
#pragma unroll 4
        for (int g = lane; g < N4; g += 64) {
            const u32 b = slab[g];
            out4[g] = make_float4((float)(b & 0xffu), (float)((b >> 8) & 0xffu), (float)((b >> 16) & 0xffu), (float)(b >> 24));
        }
    } else {
        unsigned char *bm = lds + Lds::BMAP;
        short *hcs = (short *)(lds + Lds::HC), *fcs = (short *)(lds + Lds::FC);
        for (int i = lane; i < EPW * LR_BM / 4; i += 64) ((u32 *)bm)[i] = 0;
        wave_lds_sync();
        if (lane < EPW) {
            hcs[lane] = (short)(act ? (c >> 3) * S + (c & 7) : -1);
            fcs[lane] = (short)(food >= 0 ? (food >> 3) * S + (food & 7) : -1);
        }
        {
            int code = c;
            u32 w0 = q0, w1 = q1, w2 = q2;
            for (int v = L; ballot(act && v >= 1) != 0; --v) {
                if (act && v >= 1) {
                    bm[lane * LR_BM + (code >> 3) * S + (code & 7)] = (unsigned char)v;
                    code -= lr_dcode((int)(w0 & 3u));
                    w0 = (w0 >> 2) | (w1 << 30); w1 = (w1 >> 2) | (w2 << 30); w2 >>= 2;
                }
            }
        }
        wave_lds_sync();
        float *sb = p.envs + env0 * LR_C3;
        const int total = nenv * LR_C3;
        for (int i = lane; i < total; i += 64) {
            const int e = i / LR_C3, r = i - e * LR_C3, ch = r / LR_C, cell = r - ch * LR_C;
            const int hc = hcs[e];
            if (hc < 0) continue; // outside the domain: untouched
            const float v = ch == 0 ? (cell == (int)fcs[e] ? 1.0f : 0.0f)
                          : ch == 1 ? (cell == hc ? 1.0f : 0.0f) : (float)bm[e * LR_BM + cell];
            sb[i] = v;
        }
    }

    // ---- envs outside the domain: the one-env-per-wave code, whole wave per env (it reads and writes their state,
    // action tape, outputs and crops itself; nothing above touched them except crop bytes, which it overwrites)
    if (odd != 0) {
        __threadfence();
        wave_lds_sync();
        for (u64 m = odd; m != 0; m &= m - 1)
            lane_rollout_fallback<OBSK, INJ>(p, env0 + first_bit(m), (signed char *)(lds + Lds::SCR));
    }
}

bool lane_rollout_eligible(const StepArgs &p)
{
    if (p.S != 9 || p.only_flagged) return false;
    // every observation but crops of 9 x 9 and more: the float-by-float writer that serves partial_0 / partial_1 / positions
    // loses to the one-env-per-wave kernels there (partial_3 that way at 65 536 envs: 2.5e9 against 5.9e9 env-steps/s); the
    // bit-plane form of the crops is written for the 5 x 5 and (round 5) the 7 x 7 window, 'raw' goes through bytes (round 5)
    if (p.obs_mode == WURM_OBS_PARTIAL && (p.obs_n < 0 || p.obs_n > 3)) return false;
    if ((p.inject_food == nullptr) != (p.inject_reset == nullptr)) return false;
    return true;
}

// envs per wave: automatic unless the option WURM_LANE_ROLLOUT_EPW forces it (tests and the tuning sweep).
static int lane_rollout_epw(long long N, int obs_mode)
{
    {
        const int v = (int)opt.lane_rollout_epw;
        if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) return v;
    }
    // Round 6, measured again by observation mode once the 64-envs-per-wave instantiations stored 16 bytes per instruction
    // (they had been writing dword by dword: tools/check_split_stores.py) — tools/lane_modes_probe.py, 64 steps per launch,
    // profiles/r06_lane_epw.txt: 65 536 envs 64 per wave in every mode (partial_2 0.301 -> 0.287 ms, one_channel 0.336 -> 0.287,
    // default 0.806 -> 0.784); 49 152 and 40 960: 64 but for one_channel (32); 24 576: 32 (partial_2 0.132 -> 0.110 ms) but for
    // one_channel (16); 16 384: 32 for default / raw (0.217 -> 0.196), 16 for the rest.  (Round 3's sweep had 16 / 32 at 65 536.)
    if (obs_mode == WURM_OBS_ONE_CHANNEL) return N >= 57344 ? 64 : N >= 40960 ? 32 : N >= 12288 ? 16 : N >= 6144 ? 8 : 4;
    if (obs_mode == WURM_OBS_DEFAULT || obs_mode == WURM_OBS_RAW) return N >= 40960 ? 64 : N >= 16384 ? 32 : N >= 12288 ? 16 : N >= 6144 ? 8 : 4;
    return N >= 40960 ? 64 : N >= 24576 ? 32 : N >= 12288 ? 16 : N >= 6144 ? 8 : 4;
}

// the kernel addresses the crops of a chunk (64 / epw steps) with 32-bit byte offsets
static int lane_rollout_epw_checked(long long N, int obs_mode)
{
    int epw = lane_rollout_epw(N, obs_mode);
    while (epw < 64 && (64 / epw) * N * (LR_E * 4) >= (1ll << 32)) epw *= 2; // (partial_2 only: the generic writer uses 64-bit rows)
    return epw;
}

template <int OBSK>
static hipError_t launch_lane_rollout_obs(const StepArgs &p, hipStream_t stream)
{
    const bool inj = p.inject_food != nullptr;
    const int epw = inj ? 16 : lane_rollout_epw_checked(p.N, p.obs_mode);
    const long long waves = (p.N + epw - 1) / epw;
    const int wpb = waves >= 2048 ? 4 : 1;
    dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
    (void)hipGetLastError();
    auto go = [&](auto kernel, int lds_per_wave) {
        const size_t lds_bytes = (size_t)((lr_grid_tables(OBSK) ? LR_TAB_GRID : LR_TAB) + lds_per_wave * wpb);
        // ('raw' with four waves per workgroup: 80 KB — beyond the 64 KB a launch gets without the kernel's opt-in)
        if (lds_bytes > 65536) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        WURM_LAUNCH(kernel, grid, block, lds_bytes, stream, p);
    };
    if (inj) go(lane_rollout_kernel<16, OBSK, true>, lr_wave_bytes<16, OBSK>());
    else if (epw == 4) go(lane_rollout_kernel<4, OBSK, false>, lr_wave_bytes<4, OBSK>());
    else if (epw == 8) go(lane_rollout_kernel<8, OBSK, false>, lr_wave_bytes<8, OBSK>());
    else if (epw == 16) go(lane_rollout_kernel<16, OBSK, false>, lr_wave_bytes<16, OBSK>());
    else if (epw == 32) go(lane_rollout_kernel<32, OBSK, false>, lr_wave_bytes<32, OBSK>());
    else go(lane_rollout_kernel<64, OBSK, false>, lr_wave_bytes<64, OBSK>());
    return hipGetLastError();
}

hipError_t launch_lane_rollout(const StepArgs &p, hipStream_t stream)
{
    if (p.obs_mode == WURM_OBS_NONE) return launch_lane_rollout_obs<WURM_OBS_NONE>(p, stream);
    if (p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 2) return launch_lane_rollout_obs<WURM_OBS_PARTIAL>(p, stream);
    if (p.obs_mode == WURM_OBS_ONE_CHANNEL) return launch_lane_rollout_obs<LR_OBS_GRID1>(p, stream);
    if (p.obs_mode == WURM_OBS_DEFAULT) return launch_lane_rollout_obs<LR_OBS_GRID3>(p, stream);
    if (p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 3) return launch_lane_rollout_obs<LR_OBS_CROP3>(p, stream);
    if (p.obs_mode == WURM_OBS_RAW) return launch_lane_rollout_obs<LR_OBS_RAW>(p, stream);
    return launch_lane_rollout_obs<LR_OBS_GENERIC>(p, stream);
}

} // namespace wurm
