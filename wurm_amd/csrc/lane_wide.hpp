// lane_wide.hpp — the fused rollout (T iterations of `step(a[t]); reset(done)`) for LARGE batches of 10 x 10 and 11 x 11
// SingleSnake envs: ONE ENV PER LANE, the sizes between lane_rollout.hpp's 9 x 9 (a 64-bit occupancy mask over 8 x 8
// cell codes) and grid_rollout.hip's LDS clock grids (S >= 12).  Until round 6 these sizes ran one env per wave
// (rollout_lean_kernel / rollout_kernel) at about half of what 9 x 9 gets: 65 536 envs of 10 x 10 'partial_2' 6.4e9
// env-steps/s against 1.15e10, 11 x 11 'default' 2.3e9 against 5.1e9 (tools/s10_s11_probe.py).
//
// Same four phases per chunk of TC = 64 / EPW steps as lane_rollout.hpp:
//   (1) pair lanes ((step, env) pairs): the action -> sanitise inputs; both Philox blocks of (t, env) — the food draw and
//       the complete would-be reset (lean_reset_draw) — or the recorded outcomes (INJ);
//   (2) env lanes, TC steps one after another: the transition of single_snake.py:197-304 on per-lane state — a 128-bit
//       occupancy mask over the cells y * S + x of the WHOLE grid (so it is also the row-major bit plane the grid
//       observations are made of), the body as a queue of moves (2 bits per segment, 192 bits), head / tail cells, length,
//       orientation, food cell — and the reset of :322-387 for a finished env; each step leaves a record of the stepped
//       (pre-reset) state;
//   (3) pair lanes: the per-step outputs as coalesced stores, and the observation of the record as BIT PLANES —
//       'default' (:104-128): "value is 1" = R | G << C | B << 2C with R = free or food, G = free or head, B = free (inside
//       the ring), "value is 127/255" = body << C; 'one_channel' (:142-151): body without the head (0.5), head (1.0), food
//       (1.5), ring (-1); 'partial_n', n = 2 / 3 (:166-193): the same colours on the (2n+1)^2 window around the head, rows
//       cut out of the 128-bit mask — OR-ed into the chunk's flat bit strings in LDS;
//   (4) all lanes: aligned nibbles of the flat strings -> four floats through a 256-entry table -> one 16-byte store.
// Domain: S = 10 / 11; observations 'default', 'one_channel', 'partial_2', 'partial_3', 'positions', none; snakes whose body values are
// exactly 1..L once each on edge-adjacent interior cells with the head on L and at most one food on a free interior cell
// (closed under step + reset).  Any other env is left alone and rolled out by rollout_generic — the one-env-per-wave code —
// at the end of the same launch.  Every other observation mode stays with the one-env-per-wave kernels.
#pragma once

#include "lane_load.hpp"

namespace wurm {

constexpr int LW_OBS_GRID1 = -2, LW_OBS_GRID3 = -3; // OBSK: 'one_channel' / 'default'; WURM_OBS_PARTIAL (+ NW) / WURM_OBS_NONE
constexpr int LW_TAB = 8192;                        // workgroup tables: 256 x float4, then 256 x float4 or the window masks
constexpr int LW_VS = 128;                          // bytes per env of the value -> cell table
constexpr int LW_QCAP = 128;                        // entries of lane_load_block's queue of non-zero float4s (20 bytes each)

template <int S>
struct LwGeo {
    static constexpr int C = S * S, C3 = 3 * C, BM = (C + 3) & ~3;
    static constexpr u64 ring_half(int half)
    {
        u64 m = 0;
        for (int y = 0; y < S; ++y)
            for (int x = 0; x < S; ++x) {
                const int cell = y * S + x;
                if ((y == 0 || y == S - 1 || x == 0 || x == S - 1) && cell / 64 == half) m |= 1ull << (cell % 64);
            }
        return m;
    }
    static constexpr u64 RING0 = ring_half(0), RING1 = ring_half(1);
    static constexpr u64 ALL1 = (1ull << (C - 64)) - 1ull;
    static constexpr u64 INT0 = ~RING0, INT1 = ALL1 & ~RING1;
    static_assert(C > 64 && C < 128, "two 64-bit words");
};

// floats per env of the observation / interleaved bit planes of its flat strings
template <int S, int OBSK, int NW>
constexpr int lw_elems() { return OBSK == LW_OBS_GRID1 ? S * S : OBSK == LW_OBS_GRID3 ? 3 * S * S : OBSK == WURM_OBS_PARTIAL ? 3 * NW * NW : 0; } // (bit-plane modes)
template <int OBSK>
constexpr int lw_planes() { return OBSK == LW_OBS_GRID1 ? 4 : 2; }

// per-wave LDS (bytes)
template <int EPW, int S, int OBSK, int NW>
struct LwLds {
    static constexpr int E = lw_elems<S, OBSK, NW>(), NPL = lw_planes<OBSK>();
    static constexpr int IO0 = 0;                          // uint4 [64] step inputs of pair (s, e), then the occupancy of its stepped state
    static constexpr int IO1 = IO0 + 1024;                 // uint2 [64] the rest of the record
    static constexpr int SCR = IO1 + 512;                  // flat bit strings of a chunk; start / end of launch scratch
    static constexpr int WORDS = E ? NPL * ((64 * E + 31) / 32 + 5) : 0;
    static constexpr int BITS_END = SCR + ((WORDS * 4 + 15) & ~15);
    static constexpr int VM = SCR;                         // u32 [4][EPW] bit set of body values present
    static constexpr int STAT = VM + 16 * EPW;             // u32 [EPW]    count | heads << 8 | foods << 16 | bad << 24
    static constexpr int HPOS = STAT + 4 * EPW;            // u8  [EPW]
    static constexpr int FPOS = HPOS + EPW;                // u8  [EPW]
    static constexpr int VALPOS = FPOS + EPW;              // u8  [EPW][LW_VS]
    static constexpr int QUEUE = (VALPOS + EPW * LW_VS + 15) & ~15; // lane_load_block's queue of non-zero float4s
    static constexpr int START_END = QUEUE + 20 * LW_QCAP;
    static constexpr int BMAP = SCR;                       // u8  [EPW][BM] body values by cell (end of launch)
    static constexpr int HC = BMAP + EPW * LwGeo<S>::BM;   // s16 [EPW] head cell (-1: env not written back)
    static constexpr int FC = HC + 2 * EPW;                // s16 [EPW] food cell (-1: none)
    static constexpr int END_END = FC + 2 * EPW;
    static constexpr int M1 = BITS_END > START_END ? BITS_END : START_END;
    static constexpr int BYTES = ((M1 > END_END ? M1 : END_END) + 15) & ~15;
};

// cell step of move index ai (= sanitised action & 3): -TAP[ai] (single_snake.py:225-233)
template <int S>
__device__ __forceinline__ int lw_dcell(int ai) { return ai == 0 ? S : ai == 1 ? -1 : ai == 2 ? -S : 1; }

__device__ __forceinline__ u32 lw_bit(u64 a0, u64 a1, int c) { return (u32)(((c & 64) ? a1 : a0) >> (c & 63)) & 1u; }
__device__ __forceinline__ void lw_set(u64 &a0, u64 &a1, int c)
{
    const u64 b = 1ull << (c & 63);
    a0 |= (c & 64) ? 0ull : b;
    a1 |= (c & 64) ? b : 0ull;
}
__device__ __forceinline__ void lw_clear(u64 &a0, u64 &a1, int c)
{
    const u64 b = 1ull << (c & 63);
    a0 &= (c & 64) ? ~0ull : ~b;
    a1 &= (c & 64) ? ~b : ~0ull;
}
// the 64 bits of (a1:a0) from bit pos on; pos may be negative (zeros come in from below) or beyond the mask
__device__ __forceinline__ u64 lw_window(u64 a0, u64 a1, int pos)
{
    if (pos <= -64 || pos >= 128) return 0ull;
    if (pos < 0) return a0 << (-pos);
    if (pos == 0) return a0;
    if (pos < 64) return (a0 >> pos) | (a1 << (64 - pos));
    return a1 >> (pos - 64);
}

// One transition (single_snake.py:197-304) and, for a finished env, the reset that follows it (:322-387).
// in:  cur.x = action bits (the action if it is 0..3 else 7 | (action % 4 & 7) << 3), cur.y = the would-be reset (head cell |
//      seed cell << 7 | tail cell << 14 | direction << 21 | food cell + 1 << 23), cur.z = food word (INJ: food cell + 1);
// out: occ_rec = occupancy of the stepped state, rz = head cell before the move | sanitised action << 8,
//      rw = food cell + 1 | ate << 8 | self collision << 9 | edge collision << 10 | valid << 15
// RESET = false: the step alone (the per-call kernel of lane_wide_resident.hpp rebuilds finished envs in the NEXT launch)
template <int S, bool INJ, bool RESET = true>
__device__ __forceinline__ void lw_transition(u64 &o0, u64 &o1, u64 &qa, u64 &qb, u64 &qc, int &c, int &tc, int &L, int &o, int &food,
                                              const bool act, const uint4 &cur, u64 &rec0, u64 &rec1, u32 &rz, u32 &rw)
{
    typedef LwGeo<S> G;
    rec0 = rec1 = 0;
    rz = rw = 0;
    if (!act) return;
    const int a_small = (int)(cur.x & 7u), a_mod = ((int)(cur.x << 26)) >> 29;
    const int a_out = o == a_small ? (o ^ 2) : a_mod;            // :221-222
    const int ai = a_out & 3;
    const int cn = c + lw_dcell<S>(ai);                          // :225-233 (the head is inside the ring: the move stays on the grid)
    const bool eat = cn == food;                                 // :242
    const int pos = 2 * L - 4;                                   // the oldest move
    const u64 qs = pos < 64 ? qa : (pos < 128 ? qb : qc);
    const int m = (int)((qs >> (pos & 63)) & 3ull);
    if (!eat) {                                                  // :246-249 (only the tail cell expires)
        lw_clear(o0, o1, tc);
        tc += lw_dcell<S>(m);
    }
    const u32 selfc = lw_bit(o0, o1, cn);                        // :252
    const u32 edge = lw_bit(G::RING0, G::RING1, cn);             // :290-295
    qc = (qc << 2) | (qb >> 62); qb = (qb << 2) | (qa >> 62); qa = (qa << 2) | (u64)ai;
    lw_set(o0, o1, cn);                                          // :258-262
    L += eat ? 1 : 0;
    const int c_prev = c;
    c = cn;
    o = ai ^ 2;
    if (eat) {                                                   // :270-282: the K-th free interior cell in row-major order
        if constexpr (INJ) {
            food = (int)cur.z - 1;
        } else {
            const u64 f0 = G::INT0 & ~o0, f1 = G::INT1 & ~o1;
            const int n0 = __popcll(f0), n_free = n0 + __popcll(f1);
            if (n_free == 0) {
                food = -1;
            } else {
                const int K = (int)mulhi_range(cur.z, (u32)n_free);
                food = K < n0 ? nth_bit64(f0, K) : 64 + nth_bit64(f1, K - n0);
            }
        }
    }
    rec0 = o0; rec1 = o1;
    rz = (u32)c_prev | (((u32)a_out & 0xffu) << 8);
    rw = (u32)(food + 1) | ((u32)eat << 8) | (selfc << 9) | (edge << 10) | 0x8000u;
    if (RESET && (selfc | edge)) {                               // :322-387
        const u32 r = cur.y;
        const int hc = (int)(r & 127u), sc = (int)((r >> 7) & 127u), d = (int)((r >> 21) & 3u);
        tc = (int)((r >> 14) & 127u);
        c = hc; o = d; L = 3;
        food = (int)(r >> 23) - 1;
        o0 = o1 = 0;
        lw_set(o0, o1, hc); lw_set(o0, o1, sc); lw_set(o0, o1, tc);
        qa = (u64)((d ^ 2) * 5); qb = 0; qc = 0;
    }
}

// tables of the bit-plane writers: tabA[low nibble: "value is 1", high nibble: "value is 127/255"]; for 'one_channel'
// tabA[low nibble: 0.5, high: 1.0], tabB[low nibble: 1.5, high: -1.0] (the planes exclude each other: the sum is exact)
template <int OBSK>
__device__ __forceinline__ void lw_build_tables(float4 *tabA, float4 *tabB)
{
    for (int i = (int)threadIdx.x; i < 256; i += (int)blockDim.x) {
        float a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool lo = ((i >> j) & 1) != 0, hi = ((i >> (4 + j)) & 1) != 0;
            if (OBSK == LW_OBS_GRID1) { a[j] = lo ? 0.5f : hi ? 1.0f : 0.0f; b[j] = lo ? 1.5f : hi ? -1.0f : 0.0f; }
            else { a[j] = lo ? 1.0f : hi ? 127.0f / 255.0f : 0.0f; b[j] = 0.0f; }
        }
        tabA[i] = make_float4(a[0], a[1], a[2], a[3]);
        if (OBSK == LW_OBS_GRID1) tabB[i] = make_float4(b[0], b[1], b[2], b[3]);
    }
}

// wint[head cell] = the cells of the NW x NW window around it (bit NW i + j) that lie inside the border ring
template <int S, int NW>
__device__ __forceinline__ void lw_build_wint(u64 *wint)
{
    constexpr int n = NW / 2;
    for (int i = (int)threadIdx.x; i < S * S; i += (int)blockDim.x) {
        const int hy = i / S, hx = i - hy * S;
        u64 m = 0;
        for (int wy = 0; wy < NW; ++wy)
            for (int wx = 0; wx < NW; ++wx)
                if ((unsigned)(hy - n + wy - 1) < (unsigned)(S - 2) && (unsigned)(hx - n + wx - 1) < (unsigned)(S - 2)) m |= 1ull << (NW * wy + wx);
        wint[i] = m;
    }
}

// ORs the bits (v1:v0) into plane k of NPL interleaved flat bit strings at bit offset off (word w of plane k: bits[NPL w + k])
template <int NPL>
__device__ __forceinline__ void lw_or128(u32 *bits, int k, int off, u64 v0, u64 v1)
{
    const int w = off >> 5, sb = off & 31;
    const u64 x0 = (u64)(u32)v0 << sb, x1 = (u64)(u32)(v0 >> 32) << sb, x2 = (u64)(u32)v1 << sb, x3 = (u64)(u32)(v1 >> 32) << sb;
    u32 *P = bits + NPL * w + k;
    const u32 d0 = (u32)x0, d1 = (u32)(x0 >> 32) | (u32)x1, d2 = (u32)(x1 >> 32) | (u32)x2, d3 = (u32)(x2 >> 32) | (u32)x3, d4 = (u32)(x3 >> 32);
    if (d0) atomicOr(&P[0], d0);
    if (d1) atomicOr(&P[NPL], d1);
    if (d2) atomicOr(&P[2 * NPL], d2);
    if (d3) atomicOr(&P[3 * NPL], d3);
    if (d4) atomicOr(&P[4 * NPL], d4);
}

// ... a value of at most 64 bits (the crops: 25 / 49 bits per channel)
template <int NPL>
__device__ __forceinline__ void lw_or64(u32 *bits, int k, int off, u64 v)
{
    const int w = off >> 5, sb = off & 31;
    const u64 a = (u64)(u32)v << sb, b = (u64)(u32)(v >> 32) << sb;
    u32 *P = bits + NPL * w + k;
    atomicOr(&P[0], (u32)a);
    atomicOr(&P[NPL], (u32)(a >> 32) | (u32)b);
    if ((u32)(b >> 32)) atomicOr(&P[2 * NPL], (u32)(b >> 32));
}

// the planes of pair `pair` of a stepped state: occupancy (o1:o0), head cell hc — also when it is on the ring —, food cell
// fc (-1: none)
template <int S, int OBSK, int NW>
__device__ __forceinline__ void lw_planes_of(u32 *bits, const u64 *wint, int pair, u64 o0, u64 o1, int hc, int fc)
{
    typedef LwGeo<S> G;
    constexpr int C = G::C;
    u64 h0 = 0, h1 = 0, f0 = 0, f1 = 0;
    lw_set(h0, h1, hc);
    h0 &= G::INT0; h1 &= G::INT1;                         // a head on the ring shows the ring
    if (fc >= 0) lw_set(f0, f1, fc);
    if constexpr (OBSK == LW_OBS_GRID1) {
        const int off = C * pair;
        lw_or128<4>(bits, 0, off, o0 & G::INT0 & ~h0, o1 & G::INT1 & ~h1); // body without the head: 0.5
        lw_or128<4>(bits, 1, off, h0, h1);                                   // head: 1.0
        lw_or128<4>(bits, 2, off, f0, f1);                                   // food: 1.5
        lw_or128<4>(bits, 3, off, G::RING0, G::RING1);                       // ring: -1
    } else if constexpr (OBSK == LW_OBS_GRID3) {
        const int off = 3 * C * pair;
        const u64 fr0 = G::INT0 & ~o0 & ~f0, fr1 = G::INT1 & ~o1 & ~f1;
        lw_or128<2>(bits, 0, off, fr0 | f0, fr1 | f1);                       // R: free or food
        lw_or128<2>(bits, 0, off + C, fr0 | h0, fr1 | h1);                   // G: free or head
        lw_or128<2>(bits, 0, off + 2 * C, fr0, fr1);                         // B: free
        lw_or128<2>(bits, 1, off + C, o0 & G::INT0 & ~h0, o1 & G::INT1 & ~h1); // G = 127/255: body
    } else if constexpr (OBSK == WURM_OBS_PARTIAL) {
        // crop of the stepped state (:166-193): a window cell that is off the grid or on the ring is (0,0,0); food (1,0,0),
        // head (0,1,0), body (0,127/255,0), background (1,1,1)
        constexpr int n = NW / 2, W2 = NW * NW;
        const int hy = hc / S, hx = hc - hy * S;
        u64 V = 0;                                        // occupancy of the window, bit NW i + j
#pragma unroll
        for (int i = 0; i < NW; ++i) V |= (lw_window(o0, o1, (hy - n + i) * S + hx - n) & ((1ull << NW) - 1ull)) << (NW * i);
        const u64 W = wint[hc];
        const int fy = fc >= 0 ? fc / S : -99, fx = fc - fy * S;
        const int wy = fy - (hy - n), wx = fx - (hx - n);
        const u64 F = (fc >= 0 && (unsigned)wy < (unsigned)NW && (unsigned)wx < (unsigned)NW) ? (1ull << (NW * wy + wx)) & W : 0ull;
        const u64 R = W & ~V;                             // free or food: red
        const u64 B = R & ~F;                             // free: blue (and green)
        const u64 CENTRE = 1ull << (NW * n + n);
        const u64 G1 = B | (W & CENTRE);                  // green 1: free, or the head inside the ring
        const u64 GH = V & W & ~CENTRE;                   // green 127/255: body
        const int off = 3 * W2 * pair;
        lw_or64<2>(bits, 0, off, R);
        lw_or64<2>(bits, 0, off + W2, G1);
        lw_or64<2>(bits, 0, off + 2 * W2, B);
        lw_or64<2>(bits, 1, off + W2, GH);
    }
}

// 16-byte group j of the flat run of floats whose bits start at bit 0 of the strings -> four floats
template <int OBSK>
__device__ __forceinline__ float4 lw_group(const u32 *bits, const float4 *tabA, const float4 *tabB, int j)
{
    const int w = j >> 3, sh = (j & 7) * 4;
    if (OBSK == LW_OBS_GRID1) {
        const uint4 q = ((const uint4 *)bits)[w];
        const float4 a = tabA[((q.x >> sh) & 15u) | (((q.y >> sh) & 15u) << 4)];
        const float4 b = tabB[((q.z >> sh) & 15u) | (((q.w >> sh) & 15u) << 4)];
        return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
    const uint2 q = ((const uint2 *)bits)[w];
    return tabA[((q.x >> sh) & 15u) | (((q.y >> sh) & 15u) << 4)];
}

// float f of the flat run, bit by bit (the ragged last wave, the last chunk of a tape that is not a multiple of TC)
template <int OBSK>
__device__ __forceinline__ float lw_float(const u32 *bits, int f)
{
    constexpr int NPL = lw_planes<OBSK>();
    const u32 *P = bits + NPL * (f >> 5);
    const int b = f & 31;
    if (OBSK == LW_OBS_GRID1)
        return ((P[0] >> b) & 1u) ? 0.5f : ((P[1] >> b) & 1u) ? 1.0f : ((P[2] >> b) & 1u) ? 1.5f : ((P[3] >> b) & 1u) ? -1.0f : 0.0f;
    return ((P[0] >> b) & 1u) ? 1.0f : ((P[1] >> b) & 1u) ? 127.0f / 255.0f : 0.0f;
}

// The state of a block of EPW consecutive envs, read cooperatively (lane_load_block for a whole aligned block, else three
// dwords per (env, cell) pair), then per env lane: validation and the state as occupancy mask + queue of moves.
template <int EPW, int S, typename Lds>
__device__ __forceinline__ void lw_read_block(const float *block, const bool whole, const int nenv, const int lane, unsigned char *lds,
                                              u64 &o0, u64 &o1, u64 &qa, u64 &qb, u64 &qc, int &c, int &tc, int &L, int &o, int &food,
                                              bool &act)
{
    typedef LwGeo<S> G;
    constexpr int C = G::C;
    u32 *vm = (u32 *)(lds + Lds::VM), *stat = (u32 *)(lds + Lds::STAT);
    unsigned char *hpos = lds + Lds::HPOS, *fpos = lds + Lds::FPOS, *valpos = lds + Lds::VALPOS;
    if (lane < EPW) { vm[lane] = 0; vm[EPW + lane] = 0; vm[2 * EPW + lane] = 0; vm[3 * EPW + lane] = 0; stat[lane] = 0; }
    wave_lds_sync();
    if (whole) {
        lane_load_block<EPW, C, 4, LW_VS, LW_QCAP>(block, lane, vm, stat, hpos, fpos, valpos, lds + Lds::QUEUE);
    } else {
        constexpr int LOADS = 8;
        const char *base = (const char *)(block);
        const int pairs = nenv * C;
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
                h[j] = *(const float *)(q + 4 * C);
                b[j] = *(const float *)(q + 8 * C);
                idx += 64; cell += 64; off += 256;
                if (cell >= C) { cell -= C; ++e; off += 8 * C; }
            }
#pragma unroll
            for (int j = 0; j < LOADS; ++j) {
                const int ej = es[j], cj = cs[j];
                if (ej < 0) continue;
                if (f[j] > 0.5f) { fpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 16); }
                if (h[j] > 0.5f) { hpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 8); }
                const int bi = __float2int_rn(b[j]);
                if (bi > 0 && bi < 128) {
                    valpos[ej * LW_VS + bi] = (unsigned char)cj;
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
    o0 = o1 = 0; qa = qb = qc = 0; c = tc = L = o = 0; food = -1; act = false;
    const bool mine = lane < nenv;
    if (mine) {
        const u32 st = stat[lane];
        const int cnt = (int)(st & 0xffu), nhd = (int)((st >> 8) & 0xffu), nfd = (int)((st >> 16) & 0xffu);
        const u64 v0 = (u64)vm[lane] | ((u64)vm[EPW + lane] << 32), v1 = (u64)vm[2 * EPW + lane] | ((u64)vm[3 * EPW + lane] << 32);
        L = v1 ? 127 - __clzll((long long)v1) : v0 ? 63 - __clzll((long long)v0) : 0;
        const u64 want0 = L >= 63 ? ~1ull : (2ull << L) - 2ull, want1 = L >= 64 ? (2ull << (L - 64)) - 1ull : 0ull;
        act = (st >> 24) == 0 && nhd == 1 && nfd <= 1 && L >= 2 && cnt == L && v0 == want0 && v1 == want1;
        if (act) act = (int)valpos[lane * LW_VS + L] == (int)hpos[lane];
        if (act && nfd) {
            food = fpos[lane];
            act = lw_bit(G::INT0, G::INT1, food) != 0;
        }
    }
    {
        int prev = 0;
        for (int v = 1; ballot(act && v <= L) != 0; ++v) {
            if (act && v <= L) {
                const int cell = valpos[lane * LW_VS + v];
                if (!lw_bit(G::INT0, G::INT1, cell)) act = false;
                lw_set(o0, o1, cell);
                if (v == 1) {
                    tc = cell;
                } else {
                    const int d = cell - prev;
                    const int m = d == S ? 0 : d == -1 ? 1 : d == -S ? 2 : d == 1 ? 3 : -1;
                    if (m < 0) act = false;
                    qc = (qc << 2) | (qb >> 62); qb = (qb << 2) | (qa >> 62); qa = (qa << 2) | (u64)(m & 3);
                }
                prev = cell;
            }
        }
        c = prev;
        o = (int)(qa & 3ull) ^ 2;                     // head = neck + TAP[o], the last move was -TAP[o ^ 2]
        if (act && food >= 0 && lw_bit(o0, o1, food)) act = false;
    }
}

// an env outside the domain: the one-env-per-wave rollout, whole wave
template <int S, int OBSK, bool INJ>
__device__ __forceinline__ void lane_wide_fallback(const StepArgs &p, long long env, signed char *lds)
{
    const Geo g = make_geo<2>(S);
    float *envp = p.envs + env * (3 * S * S);
    Env<2> e;
    load_state<2, true>(envp, g, e);
    rollout_generic<2, true, (OBSK < 0 ? -1 : OBSK), INJ>(p, env, envp, g, e, lds); // (-1: the mode at run time)
}

template <int EPW, int S, int OBSK, int NW, bool INJ>
__global__ __launch_bounds__(256) void lane_wide_rollout_kernel(StepArgs p)
{
    typedef LwLds<EPW, S, OBSK, NW> Lds;
    typedef LwGeo<S> G;
    static_assert(EPW == 8 || EPW == 16 || EPW == 32, "envs per wave");
    static_assert(S == 10 || S == 11, "grid size");
    static_assert(OBSK == WURM_OBS_PARTIAL || OBSK == WURM_OBS_NONE || OBSK == LW_OBS_GRID1 || OBSK == LW_OBS_GRID3 ||
                  OBSK == WURM_OBS_POSITIONS, "observation");
    constexpr bool OBS = OBSK != WURM_OBS_NONE && OBSK != WURM_OBS_POSITIONS; // through bit planes
    constexpr bool POS = OBSK == WURM_OBS_POSITIONS;                          // four floats per pair, straight from its lane
    constexpr int C = G::C, C3 = G::C3, E = Lds::E;
    constexpr int TC = 64 / EPW;                  // steps per chunk
    constexpr int LOG_EPW = EPW == 8 ? 3 : EPW == 16 ? 4 : 5;
    constexpr int SUPER = 16;                     // chunks per batch of action loads
    extern __shared__ __attribute__((aligned(16))) unsigned char lw_lds[];
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);

    // ---- workgroup tables
    float4 *tab = (float4 *)lw_lds, *tabB = (float4 *)(lw_lds + 4096);
    u64 *wint = (u64 *)(lw_lds + 4096);           // (crops: in the place of 'one_channel's second table)
    if (OBS) lw_build_tables<OBSK>(tab, tabB);
    if (OBSK == WURM_OBS_PARTIAL) lw_build_wint<S, NW>(wint);
    __syncthreads();

    const long long env0 = (xcd_block(blockIdx.x, gridDim.x) * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    unsigned char *lds = lw_lds + LW_TAB + wave * Lds::BYTES;
    u32 *bits = (u32 *)(lds + Lds::SCR);
    const int nenv = (int)min((long long)EPW, p.N - env0);
    const bool mine = lane < nenv;                // env lanes: lane e owns env0 + e
    const int ps = lane >> LOG_EPW, pe = lane & (EPW - 1); // pair lanes: step ps of the chunk, env env0 + pe

    // Actions: one load per chunk and pair lane, SUPER chunks at a time and one batch AHEAD (lane_rollout.hpp)
    const long long a_last = p.T * p.N - 1;
    long long av[SUPER];
    auto load_batch = [&](long long t_first) {
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
    u64 o0 = 0, o1 = 0, qa = 0, qb = 0, qc = 0;
    int c = 0, tc = 0, L = 0, o = 0, food = -1;
    bool act = false;
    const bool whole = nenv == EPW && (((size_t)p.envs) & 15u) == 0; // full block, 16-byte aligned (env0 is a multiple of 8)
    lw_read_block<EPW, S, Lds>(p.envs + env0 * C3, whole, nenv, lane, lds, o0, o1, qa, qb, qc, c, tc, L, o, food, act);
    const u64 odd = ballot(mine && !act);         // envs outside the domain: rollout_generic below
    wave_lds_sync();

    // ---- the chunks
    uint4 *io0 = (uint4 *)(lds + Lds::IO0);
    uint2 *io1 = (uint2 *)(lds + Lds::IO1);
    const u64 env_id = (u64)(p.env_offset + env0 + pe); // of the pair lane
    const bool pair_env = pe < nenv;
    float *obs_c = p.obs + env0 * E;              // observations of the chunk's first step, this wave's envs
    const float rcpSm2 = 1.0f / (float)(S - 2);
    const bool aligned = OBS && ((p.N * E) & 3) == 0 && (((size_t)p.obs) & 15u) == 0; // every step's row of observations starts on 16 bytes

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
            uint4 in;
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
                    const int sc = sy * S + sx, dc = tap_y(d) * S + tap_x(d);
                    const int fcode = (fc >= 0 && fc < C) ? fc : -1, ecode = (fe >= 0 && fe < C) ? fe : -1;
                    rpack = (u32)(((sc + dc) & 127) | ((sc & 127) << 7) | (((sc - dc) & 127) << 14) | (d << 21)) | ((u32)(fcode + 1) << 23);
                    fword = (u32)(ecode + 1);
                } else {
                    const LeanReset r = lean_reset_draw(p.seed, call + 1ull, env_id, S, rcpSm2);
                    rpack = (u32)r.b | ((u32)((r.a >> 8) & 3) << 21) | ((u32)((r.a >> 10) + 1) << 23);
                    fword = rng_words(p.seed, call, env_id, RNG_FOOD, 0).w[0];
                }
                in = make_uint4(a_small | (a_mod << 3), rpack, fword, 0u);
            }

            // (2) env lanes: TC transitions, each leaving the record of its stepped state in io0 / io1
            u64 r0 = 0, r1 = 0;
            u32 rz = 0, rw = 0;
            io0[lane] = in;
            wave_lds_sync();
            if (lane < EPW) {
                uint4 nxt = io0[lane];
                for (int s2 = 0; s2 < nt; ++s2) {
                    const uint4 cur = nxt;
                    if (s2 + 1 < nt) nxt = io0[(s2 + 1) * EPW + lane];
                    lw_transition<S, INJ>(o0, o1, qa, qb, qc, c, tc, L, o, food, act, cur, r0, r1, rz, rw);
                    io0[s2 * EPW + lane] = make_uint4((u32)r0, (u32)(r0 >> 32), (u32)r1, (u32)(r1 >> 32));
                    io1[s2 * EPW + lane] = make_uint2(rz, rw);
                }
            }
            if (OBS) { // clear the flat bit strings (the previous chunk's reads are done: LDS is in order)
                for (int i = lane; i < (Lds::WORDS + 3) / 4; i += 64) ((uint4 *)bits)[i] = make_uint4(0, 0, 0, 0);
            }
            wave_lds_sync();
            {
                const uint4 a = io0[lane];
                const uint2 b = io1[lane];
                r0 = (u64)a.x | ((u64)a.y << 32); r1 = (u64)a.z | ((u64)a.w << 32);
                rz = b.x; rw = b.y;
            }

            // (3) pair lanes: outputs of (t, env) and its observation as bit planes
            {
                const bool valid = pv && (rw & 0x8000u) != 0;
                if (valid) {
                    store_action(p.actions, p.act_dtype, oi, (long long)(int)(signed char)(rz >> 8));
                    p.reward[oi] = (rw & 0x100u) ? 1.0f : 0.0f;
                    p.done[oi] = (uint8_t)((rw & 0x600u) != 0);
                    p.selfc[oi] = (uint8_t)((rw >> 9) & 1u);
                    p.edgec[oi] = (uint8_t)((rw >> 10) & 1u);
                }
                if (OBS && valid) {
                    const int cp = (int)(rz & 127u), ai = (int)((rz >> 8) & 3u);
                    lw_planes_of<S, OBSK, NW>(bits, wint, lane, r0, r1, cp + lw_dcell<S>(ai), (int)(rw & 0xffu) - 1);
                }
                if (POS && valid) { // head y, x, food y, x (:153-163: the first maximum of an empty channel is cell 0)
                    const int cp = (int)(rz & 127u), ai = (int)((rz >> 8) & 3u), hc = cp + lw_dcell<S>(ai);
                    const int fc = max((int)(rw & 0xffu) - 1, 0);
                    const int hy = hc / S, fy = fc / S;
                    float *o4 = p.obs + oi * 4;
                    o4[0] = (float)hy; o4[1] = (float)(hc - hy * S); o4[2] = (float)fy; o4[3] = (float)(fc - fy * S);
                }
            }

            // (4) all lanes: the chunk's observations, 16 bytes per lane and instruction
            if (OBS) {
                wave_lds_sync();
                if (nenv == EPW && nt == TC && aligned) {
                    constexpr int GSG = EPW * E / 4;                    // 16-byte groups per step of the wave's envs
                    float4 *ob = (float4 *)obs_c;
                    const size_t step_f4 = (size_t)p.N * E / 4;          // (N E is a multiple of 4 wherever this path runs: below)
#pragma unroll
                    for (int s = 0; s < TC; ++s, ob += step_f4) {
#pragma unroll 4
                        for (int g = lane; g < GSG; g += 64) ob[g] = lw_group<OBSK>(bits, tab, tabB, s * GSG + g);
                    }
                } else { // the ragged last wave, the last chunk of a tape that is not a multiple of TC: float by float
                    for (int f = lane; f < 64 * E; f += 64) {
                        const int pr = f / E, k2 = f - pr * E, s = pr >> LOG_EPW, e = pr & (EPW - 1);
                        if (s < nt && e < nenv && (io1[pr].y & 0x8000u))
                            obs_c[(long long)s * p.N * E + e * E + k2] = lw_float<OBSK>(bits, f);
                    }
                }
                wave_lds_sync(); // (the next chunk's inputs go into io0)
            }
            obs_c += (long long)TC * p.N * E;
        }
    }
    wave_lds_sync();

    // ---- write the state back: body values by cell in LDS (walking the queue from the head), then float by float
    {
        unsigned char *bm = lds + Lds::BMAP;
        short *hcs = (short *)(lds + Lds::HC), *fcs = (short *)(lds + Lds::FC);
        for (int i = lane; i < EPW * G::BM / 4; i += 64) ((u32 *)bm)[i] = 0;
        wave_lds_sync();
        if (lane < EPW) {
            hcs[lane] = (short)(act ? c : -1);
            fcs[lane] = (short)food;
        }
        {
            int cell = c;
            u64 w0 = qa, w1 = qb, w2 = qc;
            for (int v = L; ballot(act && v >= 1) != 0; --v) {
                if (act && v >= 1) {
                    bm[lane * G::BM + cell] = (unsigned char)v;
                    cell -= lw_dcell<S>((int)(w0 & 3ull));
                    w0 = (w0 >> 2) | (w1 << 62); w1 = (w1 >> 2) | (w2 << 62); w2 >>= 2;
                }
            }
        }
        wave_lds_sync();
        float *sb = p.envs + env0 * C3;
        const int total = nenv * C3;
        for (int i = lane; i < total; i += 64) {
            const int e = i / C3, r = i - e * C3, ch = r / C, cell = r - ch * C;
            const int hc = hcs[e];
            if (hc < 0) continue; // outside the domain: untouched
            const float v = ch == 0 ? (cell == (int)fcs[e] ? 1.0f : 0.0f)
                          : ch == 1 ? (cell == hc ? 1.0f : 0.0f) : (float)bm[e * G::BM + cell];
            sb[i] = v;
        }
    }

    // ---- envs outside the domain: the one-env-per-wave code, whole wave per env (it reads and writes their state,
    // action tape, outputs and observations itself; nothing above touched them except observation bytes, which it overwrites)
    if (odd != 0) {
        __threadfence();
        wave_lds_sync();
        for (u64 m = odd; m != 0; m &= m - 1)
            lane_wide_fallback<S, OBSK, INJ>(p, env0 + first_bit(m), (signed char *)(lds + Lds::SCR));
    }
}

bool lane_wide_eligible(const StepArgs &p)
{
    if ((p.S != 10 && p.S != 11) || p.only_flagged) return false;
    if ((p.inject_food == nullptr) != (p.inject_reset == nullptr)) return false;
    if (p.obs_mode == WURM_OBS_PARTIAL) return p.obs_n == 2 || p.obs_n == 3;
    return p.obs_mode == WURM_OBS_NONE || p.obs_mode == WURM_OBS_DEFAULT || p.obs_mode == WURM_OBS_ONE_CHANNEL ||
           p.obs_mode == WURM_OBS_POSITIONS;
}

template <int S, int OBSK, int NW>
static hipError_t launch_lane_wide_obs(const StepArgs &p, hipStream_t stream)
{
    const bool inj = p.inject_food != nullptr;
    int epw = (int)opt.lane_rollout_epw;
    // (64 envs per wave, instantiated and measured at 65 536 envs, is no faster here than 32 — one_channel 0.257 against 0.238 ms
    // per 32 steps, default 0.323 against 0.303 per 16 — unlike lane_rollout.hpp's after its round-6 sweep: profiles/r06_lane_epw.txt)
    if (!(epw == 8 || epw == 16 || epw == 32)) epw = p.N >= 40960 ? 32 : p.N >= 12288 ? 16 : 8;
    if (inj) epw = 16;
    const long long waves = (p.N + epw - 1) / epw;
    const int wpb = waves >= 2048 ? 4 : 1;
    dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
    (void)hipGetLastError();
    auto go = [&](auto kernel, int lds_per_wave) {
        const size_t lds_bytes = (size_t)(LW_TAB + lds_per_wave * wpb);
        if (lds_bytes > 65536) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        WURM_LAUNCH(kernel, grid, block, lds_bytes, stream, p);
    };
    if (inj) go(lane_wide_rollout_kernel<16, S, OBSK, NW, true>, LwLds<16, S, OBSK, NW>::BYTES);
    else if (epw == 8) go(lane_wide_rollout_kernel<8, S, OBSK, NW, false>, LwLds<8, S, OBSK, NW>::BYTES);
    else if (epw == 16) go(lane_wide_rollout_kernel<16, S, OBSK, NW, false>, LwLds<16, S, OBSK, NW>::BYTES);
    else go(lane_wide_rollout_kernel<32, S, OBSK, NW, false>, LwLds<32, S, OBSK, NW>::BYTES);
    return hipGetLastError();
}

template <int S>
static hipError_t launch_lane_wide_size(const StepArgs &p, hipStream_t stream)
{
    if (p.obs_mode == WURM_OBS_NONE) return launch_lane_wide_obs<S, WURM_OBS_NONE, 0>(p, stream);
    if (p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 2) return launch_lane_wide_obs<S, WURM_OBS_PARTIAL, 5>(p, stream);
    if (p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 3) return launch_lane_wide_obs<S, WURM_OBS_PARTIAL, 7>(p, stream);
    if (p.obs_mode == WURM_OBS_ONE_CHANNEL) return launch_lane_wide_obs<S, LW_OBS_GRID1, 0>(p, stream);
    if (p.obs_mode == WURM_OBS_POSITIONS) return launch_lane_wide_obs<S, WURM_OBS_POSITIONS, 0>(p, stream);
    return launch_lane_wide_obs<S, LW_OBS_GRID3, 0>(p, stream);
}

hipError_t launch_lane_wide(const StepArgs &p, hipStream_t stream)
{
    return p.S == 10 ? launch_lane_wide_size<10>(p, stream) : launch_lane_wide_size<11>(p, stream);
}

} // namespace wurm
