// lane_step.hpp — the per-call SingleSnake step for LARGE batches of SMALL grids: ONE ENV PER LANE.
//
// fused_step_kernel<2> (one env per wave) spends 311 VALU + 310 SALU instructions per env at 9 x 9 and is issue bound:
// 65 536 envs take ~49 us where their 93 MB of traffic need ~20 us (profiles/).  Almost all of those
// instructions are per-env scalars computed by a whole wave.  Here a wave owns EPW (4, 8 or 16) CONSECUTIVE envs and works in
// two alternating shapes:
//   * cooperative, lanes = (env, cell) pairs of the block: the state (EPW x 3 x S x S floats, one contiguous run) is
//     read with coalesced dword loads and the few non-zero elements are scattered into a per-env summary in LDS (head
//     cell, food cell, position of every body value, a bit set of the body values present); the observation crops
//     (EPW x 3 x W x W floats, also one contiguous run) are produced from per-env descriptors by (env, window cell) pairs;
//   * one env per lane: validation, orientation, move, eat / decay / grow, food respawn, reset draws — the code of
//     small_step() with per-lane values, so its instructions are shared by EPW envs; the handful of cells that change
//     (the body cells of a decaying snake, two head cells, the food) are written straight from the lanes, the per-env
//     outputs (reward, done, flags, sanitised action) are coalesced stores.
// EPW trades the per-wave cost of the per-lane phase against parallelism: the whole batch is resident at once, so the
// launch takes as long as one wave (65 536 x 9 x 9: 64 envs per wave 82 us, 32: 34 us, 16: 26 us); smaller batches take
// 8 or 4 envs per wave (launch_lane_step).
// Domain: well-formed snakes (body values exactly 1..L once each, one head on L, at most one food) on 9 <= S <= 11
// (S*S <= 128: occupancy is a 128-bit mask per lane), RNG mode, observation 'partial_n' or none, the contract of
// fused_step_kernel without post_reset.  An env outside the domain is stepped by fused_step_env() — the one-env-per-wave
// code — inside the same launch, so results are bit-identical by construction wherever this path gives up.
// Follows single_snake.py:197-304 (step), :322-387 (reset), :130-195 (_observe) like the kernels it stands in for.
#pragma once

#include "lane_load.hpp"

namespace wurm {

struct LaneArgs {
    StepArgs p;
    u64 int_lo, int_hi; // interior cells (not on the border ring), bit = row-major cell index
};

constexpr int LANE_VS = 132;  // bytes per env of the value -> cell table (33 dwords: lanes fall on distinct banks)

// per-wave LDS layout (bytes) for EPW envs per wave
template <int EPW>
struct LaneLds {
    static constexpr int VM = 0;                          // u32 [4][EPW]  bit set of body values present
    static constexpr int STAT = VM + 16 * EPW;            // u32 [EPW]     count | heads << 8 | foods << 16 | bad << 24
    static constexpr int HPOS = STAT + 4 * EPW;           // u8  [EPW]
    static constexpr int FPOS = HPOS + EPW;               // u8  [EPW]
    static constexpr int DESC = FPOS + EPW;               // 2 x { short h[EPW], short f[EPW], u64 B[2][EPW] }
    static constexpr int DESC_BYTES = 20 * EPW;
    static constexpr int VALPOS = DESC + 2 * DESC_BYTES;  // u8  [EPW][LANE_VS]
    static constexpr int QUEUE = (VALPOS + EPW * LANE_VS + 15) & ~15; // non-zero float4s of the state read (lane_load.hpp)
    static constexpr int BYTES = (QUEUE + LANE_QUEUE_BYTES + 15) & ~15;
};

struct Mask128 {
    u64 lo, hi;
};

__device__ __forceinline__ void mset(Mask128 &m, int c)
{
    if (c < 64) m.lo |= 1ull << c;
    else m.hi |= 1ull << (c - 64);
}
__device__ __forceinline__ void mclr(Mask128 &m, int c)
{
    if (c < 64) m.lo &= ~(1ull << c);
    else m.hi &= ~(1ull << (c - 64));
}
__device__ __forceinline__ bool mtest(const Mask128 &m, int c)
{
    return c < 64 ? (m.lo >> c) & 1 : (m.hi >> (c - 64)) & 1;
}

// the K-th free cell in row-major order, K = mulhi(word, n_free) (add_food above); -1 if nothing is free
__device__ __forceinline__ int lane_pick_free(const Mask128 &fr, u32 word)
{
    const int n_lo = __popcll(fr.lo), n = n_lo + __popcll(fr.hi);
    if (n == 0) return -1;
    const int K = (int)mulhi_range(word, (u32)n);
    return K < n_lo ? nth_bit64(fr.lo, K) : 64 + nth_bit64(fr.hi, K - n_lo);
}

struct LaneSnake { // a freshly built env (reset_core above): body 3, 2, 1 on c3, c2, c1, the head on c3, food on fc
    int c3, c2, c1, fc;
};

__device__ __forceinline__ LaneSnake lane_reset(u64 seed, u64 call, u64 env_id, int S, const Mask128 &interior)
{
    const Words w = rng_words(seed, call, env_id, RNG_RESET, 0);
    const int sy = 4 + (int)mulhi_range(w.w[0], (u32)(S - 8));
    const int sx = 4 + (int)mulhi_range(w.w[1], (u32)(S - 8));
    const int d = (int)(w.w[2] >> 30);
    LaneSnake r;
    r.c3 = (sy + tap_y(d)) * S + sx + tap_x(d);
    r.c2 = sy * S + sx;
    r.c1 = (sy - tap_y(d)) * S + sx - tap_x(d);
    Mask128 fr = interior;
    mclr(fr, r.c3);
    mclr(fr, r.c2);
    mclr(fr, r.c1);
    r.fc = lane_pick_free(fr, w.w[3]);
    return r;
}

// (2n+1)^2 crops of a block of envs, one contiguous run of nenv * 3 * W * W floats, from the descriptors in LDS
template <int EPW, int S>
__device__ __forceinline__ void lane_observe(const LaneArgs &a, unsigned char *lds, int which, float *__restrict__ out,
                                             int nenv, int lane)
{
    typedef LaneLds<EPW> Lds;
    const StepArgs &p = a.p;
    const int n = p.obs_n, W = 2 * n + 1, W2 = W * W, E = 3 * W2;
    const float rcpW2 = 1.0f / (float)W2, rcpW = 1.0f / (float)W, rcpS = 1.0f / (float)S;
    const short *dh = (const short *)(lds + Lds::DESC + which * Lds::DESC_BYTES), *df = dh + EPW;
    const u64 *dB = (const u64 *)(dh + 2 * EPW);
    // lanes = (env, window cell) pairs: the cell is classified once and its three channel values are stored
    const int pairs = nenv * W2;
    char *ob = (char *)out;
    for (int idx = lane; idx < pairs; idx += 64) {
        const int e = div_size(idx, rcpW2), w = idx - e * W2;
        const int wy = div_size(w, rcpW), wx = w - wy * W;
        const int h = dh[e];
        if (h < -1) continue; // an env outside the domain: fused_step_env() writes its crop, nothing is stored here
        const int hy = div_size(max(h, 0), rcpS), hx = h - hy * S;
        const int y = hy - n + wy, x = hx - n + wx;
        float r = 0.0f, g = 0.0f, b = 0.0f; // zero padding (single_snake.py:179), the border ring, no head
        if (h >= 0 && y >= 1 && y <= S - 2 && x >= 1 && x <= S - 2) {
            const int c = y * S + x;
            const u64 bw = dB[(c >> 6) * EPW + e];
            // class priority of cell_class(): food, head, body, background
            if (c == (int)df[e]) r = 1.0f;
            else if (c == h) g = 1.0f;
            else if ((bw >> (c & 63)) & 1) g = 127.0f / 255.0f;
            else r = g = b = 1.0f;
        }
        const unsigned o = 4u * (unsigned)(e * E + w); // scalar base + 32-bit lane offset stores
        *(float *)(ob + o) = r;
        *(float *)(ob + o + 4u * (unsigned)W2) = g;
        *(float *)(ob + o + 8u * (unsigned)W2) = b;
    }
}

template <int CPL, bool SNAKE>
__device__ __forceinline__ void fused_step_env(const StepArgs &p, long long env, signed char *lds);

template <int EPW, int S>
__global__ __launch_bounds__(256) void lane_step_kernel(LaneArgs a)
{
    typedef LaneLds<EPW> Lds;
    extern __shared__ __attribute__((aligned(16))) unsigned char lane_lds[];
    const StepArgs &p = a.p;
    // the wave index through readfirstlane: per-wave pointers then live in SGPRs (scalar base + 32-bit lane offset loads)
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);
    const long long env0 = ((long long)blockIdx.x * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    unsigned char *lds = lane_lds + wave * Lds::BYTES;
    constexpr int C = S * S, C3 = 3 * C;
    const int nenv = (int)min((long long)EPW, p.N - env0);
    const long long env = env0 + lane;
    const bool mine = lane < nenv;
    WURM_TL_DECL;
    WURM_TL(0); // entry

    u32 *vm = (u32 *)(lds + Lds::VM), *stat = (u32 *)(lds + Lds::STAT);
    unsigned char *hpos = lds + Lds::HPOS, *fpos = lds + Lds::FPOS, *valpos = lds + Lds::VALPOS;
    if (lane < EPW) {
        vm[lane] = 0; vm[EPW + lane] = 0; vm[2 * EPW + lane] = 0; vm[3 * EPW + lane] = 0;
        stat[lane] = 0;
    }
    wave_lds_sync();

    // ---- cooperative read of the block's state into the per-env summaries (lane_load.hpp): float4 loads with the non-zero
    // elements compacted through an LDS queue when the block is whole and 16-byte aligned, else (the ragged last block,
    // an odd base address) lanes = (env, cell) pairs, three dwords each (food, head, body)
    constexpr int LANE_LOADS = EPW >= 16 ? 11 : 8; // pairs (= 3 loads each) in flight per lane: EPW * C / 64 pairs in all
    if (nenv == EPW && (((size_t)p.envs) & 15u) == 0) {
        lane_load_block<EPW, C, 4, LANE_VS>(p.envs + env0 * C3, lane, vm, stat, hpos, fpos, valpos, lds + Lds::QUEUE);
    } else
    {
        const char *base = (const char *)(p.envs + env0 * C3);
        const int pairs = nenv * C;
        int e = 0, cell = lane, idx = lane; // pair idx = e * C + cell; C > 63: a step of 64 wraps at most once
        unsigned off = 4u * (unsigned)lane; // byte offset of element e * C3 + cell (scalar base + 32-bit lane offset)
        for (int i0 = 0; i0 < pairs; i0 += 64 * LANE_LOADS) {
            float f[LANE_LOADS], h[LANE_LOADS], b[LANE_LOADS];
            int es[LANE_LOADS], cs[LANE_LOADS];
#pragma unroll
            for (int j = 0; j < LANE_LOADS; ++j) { // unconditional loads (past the end: pair 0), all in flight together
                es[j] = idx < pairs ? e : -1;
                cs[j] = cell;
                const unsigned o = idx < pairs ? off : 0u;
                const char *q = base + o;
                f[j] = *(const float *)q;
                h[j] = *(const float *)(q + 4 * C);
                b[j] = *(const float *)(q + 8 * C);
                idx += 64; cell += 64; off += 256;
                if (cell >= C) { cell -= C; ++e; off += 8 * C; }
            }
#pragma unroll
            for (int j = 0; j < LANE_LOADS; ++j) {
                const int ej = es[j], cj = cs[j];
                if (ej < 0) continue;
                if (f[j] > 0.5f) { fpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 16); }
                if (h[j] > 0.5f) { hpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 8); }
                const int bi = __float2int_rn(b[j]); // body (single_snake.py:210): position of every value, values present
                if (bi > 0 && bi < 128) {
                    valpos[ej * LANE_VS + bi] = (unsigned char)cj;
                    atomicOr(&vm[(bi >> 5) * EPW + ej], 1u << (bi & 31));
                    atomicAdd(&stat[ej], 1u);
                } else if (bi != 0) {
                    atomicAdd(&stat[ej], 1u << 24);
                }
            }
        }
    }
    wave_lds_sync();

    WURM_TL(1); // state read
    // ---- one env per lane
    const Mask128 interior = {a.int_lo, a.int_hi};
    const u64 env_id = (u64)(p.env_offset + env);
    const bool pre = mine && p.done_in != nullptr && p.done_in[env] != 0;
    bool regular = false;
    int L = 0, hc = -1, fc = -1;
    LaneSnake fresh = {0, 0, 0, -1};
    if (pre) { // the postponed reset(done) (reset_kernel with call = pre_call): the env is rebuilt, whatever it held
        fresh = lane_reset(p.seed, p.pre_call, env_id, S, interior);
        valpos[lane * LANE_VS + 3] = (unsigned char)fresh.c3;
        valpos[lane * LANE_VS + 2] = (unsigned char)fresh.c2;
        valpos[lane * LANE_VS + 1] = (unsigned char)fresh.c1;
        L = 3; hc = fresh.c3; fc = fresh.fc;
        regular = true;
    } else if (mine) {
        const u32 st = stat[lane];
        const int cnt = (int)(st & 0xffu), nhd = (int)((st >> 8) & 0xffu), nfd = (int)((st >> 16) & 0xffu);
        const u64 vlo = (u64)vm[lane] | ((u64)vm[EPW + lane] << 32);
        const u64 vhi = (u64)vm[2 * EPW + lane] | ((u64)vm[3 * EPW + lane] << 32);
        L = vhi ? 127 - __clzll((long long)vhi) : (vlo ? 63 - __clzll((long long)vlo) : 0);
        // values exactly 1..L, once each: L non-zero cells and every bit 1..L present
        const u64 want_lo = L >= 64 ? ~1ull : (2ull << L) - 2ull, want_hi = L >= 64 ? (2ull << (L - 64)) - 1ull : 0ull;
        regular = (st >> 24) == 0 && nhd == 1 && nfd <= 1 && L >= 2 && cnt == L && vlo == want_lo && vhi == want_hi;
        if (regular) {
            hc = hpos[lane];
            fc = nfd ? (int)fpos[lane] : -1;
            regular = (int)valpos[lane * LANE_VS + L] == hc; // the head sits on the largest body value
        }
    }
    wave_lds_sync();

    WURM_TL(2); // validated
    // transition (small_step above, per lane)
    constexpr float rcpS = 1.0f / (float)S;
    long long act = 0;
    int nh = -1, grow = 0, dec = 0, fc_after = fc;
    bool EAT = false, SELFC = false, EDGEC = false, inside = false;
    Mask128 B = {0, 0};
    if (regular) {
        const int neck = valpos[lane * LANE_VS + L - 1];
        const int hy = div_size(hc, rcpS), hx = hc - hy * S;
        const int yN = div_size(neck, rcpS), xN = neck - yN * S;
        const int dy = hy - yN, dx = hx - xN;
        const int o = (dy == 0 && dx == 1) ? 1 : (dy == 1 && dx == 0) ? 2 : (dy == 0 && dx == -1) ? 3 : 0;
        act = load_action(p.actions, p.act_dtype, env);
        if ((long long)o == act) act += 2;                                         // :221-222
        act = act % 4;
        const int ai = (int)(((act % 4) + 4) % 4);
        const int ny = hy - tap_y(ai), nx = hx - tap_x(ai);                        // :225-233
        inside = ny >= 0 && ny < S && nx >= 0 && nx < S;
        nh = inside ? ny * S + nx : -1;
        EAT = inside && nh == fc;                                                  // :242
        dec = EAT ? 0 : 1;                                                         // :246-249
        grow = L + (EAT ? 1 : 0);                                                  // :258-262
        EDGEC = !(inside && ny >= 1 && ny <= S - 2 && nx >= 1 && nx <= S - 2);     // :290-295
    }
    WURM_TL(3); // action loaded, move decided
    // body: value v sits on valpos[v]; every one decays unless food was eaten, the new head cell grows
    char *sb = (char *)(p.envs + env0 * C3);                 // scalar base of the block's state ...
    const unsigned so = 4u * (unsigned)(lane * C3);           // ... + this lane's env (32-bit byte offsets)
    auto put = [&](int elem, float v) { *(float *)(sb + (so + 4u * (unsigned)elem)) = v; };
    int under = 0;
    for (int v = 1; ballot(regular && v <= L) != 0; ++v) {
        if (regular && v <= L) {
            const int cell = valpos[lane * LANE_VS + v];
            int nv = v - dec;
            if (cell == nh) { under = v; nv += grow; }
            if (nv > 0) mset(B, cell);
            if (!pre && nv != v) put(2 * C + cell, (float)nv);
        }
    }
    if (regular) {
        SELFC = inside && under - dec > 0;                                         // :252 (after the decay)
        if (inside) mset(B, nh);
        if (!pre) {
            if (inside && under == 0) put(2 * C + nh, (float)grow);
            put(C + hc, 0.0f);
            if (inside) put(C + nh, 1.0f);
        }
        if (EAT) {                                                                 // :270-282
            Mask128 fr = {interior.lo & ~B.lo, interior.hi & ~B.hi};
            fc_after = lane_pick_free(fr, rng_words(p.seed, p.call, env_id, RNG_FOOD, 0).w[0]);
            if (!pre) {
                put(nh, 0.0f);
                if (fc_after >= 0) put(fc_after, 1.0f);
            }
        }
        const int done = SELFC | EDGEC;
        store_action(p.actions, p.act_dtype, env, act);
        p.selfc[env] = (uint8_t)SELFC;
        p.reward[env] = EAT ? 1.0f : 0.0f;
        p.done[env] = (uint8_t)done;
        p.edgec[env] = (uint8_t)EDGEC;
        if (p.done_copy) p.done_copy[env] = (uint8_t)done;
    }

    WURM_TL(4); // changed cells and outputs stored
    // a rebuilt env is stored whole (its old contents are unrelated): cooperative, one env at a time
    for (u64 m = ballot(pre); m != 0; m &= m - 1) {
        const int src = first_bit(m);
        const int c3 = lane_value(fresh.c3, src), c2 = lane_value(fresh.c2, src), c1 = lane_value(fresh.c1, src);
        const int s_nh = lane_value(nh, src), s_dec = lane_value(dec, src), s_grow = lane_value(grow, src);
        const int s_f = lane_value(fc_after, src);
        float *ep = p.envs + (env0 + src) * C3;
        constexpr float rcpC = 1.0f / (float)C;
        for (int i = lane; i < C3; i += 64) {
            const int ch = div_size(i, rcpC), cell = i - ch * C;
            float v;
            if (ch == 0) v = cell == s_f ? 1.0f : 0.0f;
            else if (ch == 1) v = cell == s_nh ? 1.0f : 0.0f;
            else v = (float)((cell == c3 ? 3 - s_dec : cell == c2 ? 2 - s_dec : cell == c1 ? 1 - s_dec : 0) +
                             (cell == s_nh ? s_grow : 0));
            ep[i] = v;
        }
    }

    WURM_TL(5); // rebuilt envs stored
    // ---- observations: descriptors per env, then the crops of the whole block
    if (p.obs_mode == WURM_OBS_PARTIAL) {
        short *dh = (short *)(lds + Lds::DESC), *df = dh + EPW;
        u64 *dB = (u64 *)(dh + 2 * EPW);
        if (mine) {
            dh[lane] = (short)(regular ? nh : -2);
            df[lane] = (short)fc_after;
            dB[lane] = B.lo;
            dB[EPW + lane] = B.hi;
        }
        if (p.obs_after != nullptr) { // what reset(done) returns: done envs rebuilt with call + 1 (not stored here)
            short *ah = (short *)(lds + Lds::DESC + Lds::DESC_BYTES), *af = ah + EPW;
            u64 *aB = (u64 *)(ah + 2 * EPW);
            int h2 = regular ? nh : -2, f2 = fc_after;
            Mask128 B2 = B;
            if (regular && (SELFC || EDGEC)) {
                const LaneSnake r = lane_reset(p.seed, p.call + 1ull, env_id, S, interior);
                B2.lo = B2.hi = 0;
                mset(B2, r.c3); mset(B2, r.c2); mset(B2, r.c1);
                h2 = r.c3; f2 = r.fc;
            }
            if (mine) {
                ah[lane] = (short)h2;
                af[lane] = (short)f2;
                aB[lane] = B2.lo;
                aB[EPW + lane] = B2.hi;
            }
        }
        wave_lds_sync();
        lane_observe<EPW, S>(a, lds, 0, p.obs + env0 * p.obs_elems, nenv, lane);
        if (p.obs_after != nullptr) lane_observe<EPW, S>(a, lds, 1, p.obs_after + env0 * p.obs_elems, nenv, lane);
    }

    WURM_TL(6); // crops issued; WURM_TL_STORE: drained
    if (p.obs_mode == WURM_OBS_PARTIAL) WURM_TL_STORE(p.obs + env0 * p.obs_elems, lane);
    // ---- envs outside the domain: the one-env-per-wave code writes their state, crops and outputs (nothing above did)
    u64 odd = ballot(mine && !regular);
    if (odd != 0) {
        wave_lds_sync();
        for (; odd != 0; odd &= odd - 1)
            fused_step_env<2, true>(p, env0 + first_bit(odd), (signed char *)lds);
    }
}

#ifndef WURM_SINGLE_SNAKE_DEVICE_CODE_ONLY
static bool lane_step_eligible(const StepArgs &p)
{
    if (p.S < 9 || p.S * p.S > 128) return false;
    if (p.inject_food || p.inject_reset || p.inject_pre_reset || p.post_reset || p.only_flagged) return false;
    if (p.obs_mode != WURM_OBS_NONE && p.obs_mode != WURM_OBS_PARTIAL) return false;
    if (p.obs_mode == WURM_OBS_PARTIAL && 3 * (2 * p.obs_n + 1) * (2 * p.obs_n + 1) > 256) return false;
    return true;
}

static hipError_t launch_lane_step(const StepArgs &p, hipStream_t stream)
{
    LaneArgs a;
    a.p = p;
    a.int_lo = a.int_hi = 0;
    for (int y = 1; y <= p.S - 2; ++y)
        for (int x = 1; x <= p.S - 2; ++x) {
            const int c = y * p.S + x;
            if (c < 64) a.int_lo |= 1ull << c;
            else a.int_hi |= 1ull << (c - 64);
        }
    // Envs per wave.  The whole batch is resident at once, so a launch takes as long as ONE wave: fewer envs per wave
    // = more, shorter waves, but the per-lane phase is paid once per wave.  Measured (9 x 9 partial_2, us per launch;
    // the one-env-per-wave kernel in brackets): 8192 envs 9.2 / 10.8 / 13.9 at 4 / 8 / 16 envs per wave [9.6],
    // 16 384: 11.2 / 12.0 / 14.9 [15.1], 32 768: 17.4 / 15.6 / 17.6 [26.6], 65 536: 28.7 / 27.3 / 26.0 [48.3].
    const int epw = p.N >= 49152 ? 16 : (p.N >= 24576 ? 8 : 4);
    (void)hipGetLastError();
    auto go = [&](auto k9, auto k10, auto k11, int epw, int lds_per_wave) {
        const int wpb = p.N <= 8192 ? 1 : 4;
        const long long waves = (p.N + epw - 1) / epw;
        dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
        const size_t lds = (size_t)lds_per_wave * wpb;
        if (p.S == 9) WURM_LAUNCH(k9, grid, block, lds, stream, a);
        else if (p.S == 10) WURM_LAUNCH(k10, grid, block, lds, stream, a);
        else WURM_LAUNCH(k11, grid, block, lds, stream, a);
    };
    if (epw == 4) go(lane_step_kernel<4, 9>, lane_step_kernel<4, 10>, lane_step_kernel<4, 11>, 4, LaneLds<4>::BYTES);
    else if (epw == 8) go(lane_step_kernel<8, 9>, lane_step_kernel<8, 10>, lane_step_kernel<8, 11>, 8, LaneLds<8>::BYTES);
    else go(lane_step_kernel<16, 9>, lane_step_kernel<16, 10>, lane_step_kernel<16, 11>, 16, LaneLds<16>::BYTES);
    return hipGetLastError();
}
#endif // WURM_SINGLE_SNAKE_DEVICE_CODE_ONLY

} // namespace wurm
