// lane_wide_resident.hpp — the per-call SingleSnake step (`obs, r, d, info = env.step(a); env.reset(d)`, one launch per
// iteration) for LARGE batches of 10 x 10 and 11 x 11 envs with the state RESIDENT in compact form between calls: what
// lane_resident.hpp is for 9 x 9, on lane_wide.hpp's per-lane state.  Until round 6 these sizes read the whole (N, 3, S, S) fp32
// state every call (lane_step_kernel: 31 us per iteration of 65 536 envs with 'partial_2', the one-env-per-wave kernels 46-52 us
// with 'one_channel' / 'default'; 9 x 9 on its mirror: 11 / 11 / 26 us — tools/s10_s11_percall_probe.py).
//
// The MIRROR (wurm_single_call.resident, caller-owned): three planes of N uint4, 48 bytes per env —
//   plane 0 [env]: the 128-bit occupancy mask over the cells y * S + x
//   plane 1 [env]: the queue of moves, bits 0 .. 127
//   plane 2 [env]: bits 128 .. 191 of the queue, head cell | tail cell << 7 | length << 14 | orientation << 21 |
//                  (food cell + 1) << 23, flags (RES_ACT / RES_TERMINAL as in lane_resident.hpp)
// (cells are whole-grid indices: a head on the border ring needs no second encoding).
// LAZY FORM ONLY: `envs` is not written by the step; lane_wide_resident_flush_kernel (wurm_single_resident_flush) brings it up
// to date before anything else looks at it.  A caller that wants `envs` written every call (resident_lazy = 0) gets the kernels
// without a mirror, as before — the entry point reports the mirror stale.
// Contract: fused_step_kernel's without post_reset (deferred reset: envs flagged in p.done_in are rebuilt in front of the step
// with call = p.pre_call); `obs` of the stepped state and, when asked for, `obs_after` of the state once the finished envs are
// rebuilt with call + 1, as (which, env) pair lanes -> bit planes -> table -> 16-byte stores (lane_wide.hpp phases 3 and 4).
// Domain: lane_wide.hpp's, RNG mode.  An env outside it — or one that finished and is stepped again without the reset — is
// stepped by fused_step_env (the one-env-per-wave code) on `envs` inside the same launch, and stays there until it is rebuilt.
// Follows single_snake.py:197-304 (step), :322-387 (reset), :130-195 (_observe) like the kernels it stands in for.
#pragma once

#include "lane_wide.hpp"

namespace wurm {

constexpr u32 LWR_ACT = 1u;       // the env is in the lane kernels' domain and the record describes it
constexpr u32 LWR_TERMINAL = 2u;  // the last step finished the env: the record is void unless the next call rebuilds it
constexpr int LWR_BYTES = 48;

struct WideResArgs {
    StepArgs p;
    uint4 *res;
    uint32_t *check_mask; // nullable (N): wurm_single_call.check_mask
};

__device__ __forceinline__ u32 lwr_pack(int c, int tc, int L, int o, int food)
{
    return (u32)(c & 127) | ((u32)(tc & 127) << 7) | ((u32)(L & 127) << 14) | ((u32)(o & 3) << 21) | ((u32)((food + 1) & 127) << 23);
}

__device__ __forceinline__ void lwr_store(uint4 *res, long long N, long long env, u64 o0, u64 o1, u64 qa, u64 qb, u64 qc, u32 pk, u32 flags)
{
    res[env] = make_uint4((u32)o0, (u32)(o0 >> 32), (u32)o1, (u32)(o1 >> 32));
    res[N + env] = make_uint4((u32)qa, (u32)(qa >> 32), (u32)qb, (u32)(qb >> 32));
    res[2 * N + env] = make_uint4((u32)qc, (u32)(qc >> 32), pk, flags);
}

template <int EPW, int S>
__global__ __launch_bounds__(256) void lane_wide_resident_build_kernel(WideResArgs a)
{
    typedef LwLds<EPW, S, WURM_OBS_NONE, 0> Lds;
    typedef LwGeo<S> G;
    extern __shared__ __attribute__((aligned(16))) unsigned char lwr_lds[];
    const StepArgs &p = a.p;
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);
    const long long env0 = ((long long)blockIdx.x * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    unsigned char *lds = lwr_lds + wave * Lds::BYTES;
    const int nenv = (int)min((long long)EPW, p.N - env0);
    u64 o0, o1, qa, qb, qc;
    int c, tc, L, o, food;
    bool act;
    const bool whole = nenv == EPW && (((size_t)p.envs) & 15u) == 0;
    lw_read_block<EPW, S, Lds>(p.envs + env0 * G::C3, whole, nenv, lane, lds, o0, o1, qa, qb, qc, c, tc, L, o, food, act);
    if (lane < nenv) lwr_store(a.res, p.N, env0 + lane, o0, o1, qa, qb, qc, lwr_pack(c, tc, L, o, food), act ? LWR_ACT : 0u);
}

// `envs` of the envs of a block whose bit is set in `which` from their per-lane state (env lanes: c = the head cell, also on
// the ring; L; the queue; food): body values by walking the queue from the head — a cell the walk visits twice (the snake ran
// into itself: single_snake.py:252-262 adds the new head's value on top of what the cell held) gets the sum — then float by
// float.  bm: EPW * BM bytes, hcs / fcs: EPW shorts each, of LDS.
template <int EPW, int S>
__device__ __forceinline__ void lwr_write_planes(float *block, int nenv, int lane, unsigned char *bm, short *hcs, short *fcs, bool sel,
                                                 int c, int L, u64 qa, u64 qb, u64 qc, int food)
{
    typedef LwGeo<S> G;
    for (int i = lane; i < EPW * G::BM / 4; i += 64) ((u32 *)bm)[i] = 0;
    wave_lds_sync();
    if (lane < EPW) {
        hcs[lane] = (short)(sel ? c : -1);
        fcs[lane] = (short)food;
    }
    {
        int cell = c;
        u64 w0 = qa, w1 = qb, w2 = qc;
        for (int v = L; ballot(sel && v >= 1) != 0; --v) {
            if (sel && v >= 1) {
                bm[lane * G::BM + cell] += (unsigned char)v;
                cell -= lw_dcell<S>((int)(w0 & 3ull));
                w0 = (w0 >> 2) | (w1 << 62); w1 = (w1 >> 2) | (w2 << 62); w2 >>= 2;
            }
        }
    }
    wave_lds_sync();
    const int total = nenv * G::C3;
    for (int i = lane; i < total; i += 64) {
        const int e = i / G::C3, r = i - e * G::C3, ch = r / G::C, cell = r - ch * G::C;
        const int hc = hcs[e];
        if (hc < 0) continue;
        block[i] = ch == 0 ? (cell == (int)fcs[e] ? 1.0f : 0.0f) : ch == 1 ? (cell == hc ? 1.0f : 0.0f) : (float)bm[e * G::BM + cell];
    }
    wave_lds_sync();
}

// `envs` from the mirror: every env whose record describes it (LWR_ACT) is written whole; the others are the ones the
// one-env-per-wave code steps on `envs` itself, which is current for them.
template <int EPW, int S>
__global__ __launch_bounds__(256) void lane_wide_resident_flush_kernel(WideResArgs a)
{
    typedef LwGeo<S> G;
    constexpr int WB = EPW * G::BM + 4 * EPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char lwr_lds[];
    const StepArgs &p = a.p;
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);
    const long long env0 = ((long long)blockIdx.x * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    const int nenv = (int)min((long long)EPW, p.N - env0);
    unsigned char *bm = lwr_lds + wave * WB;
    short *hcs = (short *)(bm + EPW * G::BM), *fcs = hcs + EPW;
    uint4 r1 = make_uint4(0, 0, 0, 0), r2 = r1;
    if (lane < nenv) {
        r1 = a.res[p.N + env0 + lane];
        r2 = a.res[2 * p.N + env0 + lane];
    }
    const bool sel = lane < nenv && (r2.w & LWR_ACT) != 0;
    lwr_write_planes<EPW, S>(p.envs + env0 * G::C3, nenv, lane, bm, hcs, fcs, sel, (int)(r2.z & 127u), (int)((r2.z >> 14) & 127u),
                             (u64)r1.x | ((u64)r1.y << 32), (u64)r1.z | ((u64)r1.w << 32), (u64)r2.x | ((u64)r2.y << 32),
                             (int)((r2.z >> 23) & 127u) - 1);
}

// per-wave LDS of the step kernel: lane_wide.hpp's (the flat bit strings of up to 64 pairs, the records), and the end-of-launch
// scratch of lwr_write_planes where an env has to be written out
template <int EPW, int S, int OBSK, int NW>
constexpr int lwr_wave_bytes()
{
    typedef LwLds<EPW, S, OBSK, NW> Lds;
    constexpr int a = Lds::BITS_END, b = Lds::SCR + EPW * LwGeo<S>::BM + 4 * EPW + 16, c = Lds::SCR + 128;
    return ((a > b ? (a > c ? a : c) : (b > c ? b : c)) + 15) & ~15;
}

// EPW envs per wave; NOBS = 2: observations `obs` and `obs_after` (EPW * 2 <= 64 pair lanes), NOBS = 1: `obs` only
template <int EPW, int S, int OBSK, int NW, int NOBS>
__global__ __launch_bounds__(256) void lane_wide_resident_step_kernel(WideResArgs a)
{
    typedef LwLds<EPW, S, OBSK, NW> Lds;
    typedef LwGeo<S> G;
    static_assert(EPW == 16 || EPW == 32, "envs per wave");
    static_assert(EPW * NOBS <= 64 && (NOBS == 1 || NOBS == 2), "pair lanes");
    constexpr bool OBS = OBSK != WURM_OBS_NONE && OBSK != WURM_OBS_POSITIONS; // through bit planes
    constexpr bool POS = OBSK == WURM_OBS_POSITIONS;
    constexpr int C3 = G::C3, E = Lds::E;
    constexpr int LOG_EPW = EPW == 16 ? 4 : 5;
    constexpr int NP = EPW * NOBS;               // pair lanes
    extern __shared__ __attribute__((aligned(16))) unsigned char lwr_lds[];
    const StepArgs &p = a.p;
    const int wave = uniform((int)(threadIdx.x >> 6)), wpb = (int)(blockDim.x >> 6), lane = (int)(threadIdx.x & 63u);

    float4 *tab = (float4 *)lwr_lds, *tabB = (float4 *)(lwr_lds + 4096);
    u64 *wint = (u64 *)(lwr_lds + 4096);
    if (OBS) lw_build_tables<OBSK>(tab, tabB);
    if (OBSK == WURM_OBS_PARTIAL) lw_build_wint<S, NW>(wint);
    __syncthreads();

    const long long env0 = (xcd_block(blockIdx.x, gridDim.x) * wpb + wave) * EPW;
    if (env0 >= p.N) return;
    unsigned char *lds = lwr_lds + LW_TAB + wave * lwr_wave_bytes<EPW, S, OBSK, NW>();
    u32 *bits = (u32 *)(lds + Lds::SCR);
    const int nenv = (int)min((long long)EPW, p.N - env0);
    const bool mine = lane < nenv;                // env lanes: lane e owns env0 + e
    const long long env = env0 + lane;
    const u64 env_id = (u64)(p.env_offset + env);
    const float rcpSm2 = 1.0f / (float)(S - 2);

    // ---- env lanes: the record, the action, the postponed reset
    uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0, r2 = r0;
    long long a_in = 0;
    bool pre = false;
    if (mine) {
        r0 = a.res[env];
        r1 = a.res[p.N + env];
        r2 = a.res[2 * p.N + env];
        a_in = load_action(p.actions, p.act_dtype, env);
        pre = p.done_in != nullptr && p.done_in[env] != 0;
    }
    u64 o0 = (u64)r0.x | ((u64)r0.y << 32), o1 = (u64)r0.z | ((u64)r0.w << 32);
    u64 qa = (u64)r1.x | ((u64)r1.y << 32), qb = (u64)r1.z | ((u64)r1.w << 32), qc = (u64)r2.x | ((u64)r2.y << 32);
    int c = (int)(r2.z & 127u), tc = (int)((r2.z >> 7) & 127u), L = (int)((r2.z >> 14) & 127u), o = (int)((r2.z >> 21) & 3u);
    int food = (int)((r2.z >> 23) & 127u) - 1;
    const u32 flags_in = r2.w;
    bool act = mine && (flags_in & (LWR_ACT | LWR_TERMINAL)) == LWR_ACT;
    // (kept for a finished env that is stepped again without its reset: `envs` has to show its last state first)
    const bool stale_terminal = mine && (flags_in & (LWR_ACT | LWR_TERMINAL)) == (LWR_ACT | LWR_TERMINAL) && !pre;
    const int t_c = c, t_L = L, t_food = food;
    const u64 t_qa = qa, t_qb = qb, t_qc = qc;
    if (pre) {                  // reset_kernel with call = pre_call (single_snake.py:322-387): the env is rebuilt, whatever it held
        const LeanReset r = lean_reset_draw(p.seed, p.pre_call, env_id, S, rcpSm2);
        const int hc = r.b & 127, sc = (r.b >> 7) & 127;
        tc = (r.b >> 14) & 127;
        c = hc; o = (r.a >> 8) & 3; L = 3;
        food = r.a >> 10;
        o0 = o1 = 0;
        lw_set(o0, o1, hc); lw_set(o0, o1, sc); lw_set(o0, o1, tc);
        qa = (u64)((o ^ 2) * 5); qb = 0; qc = 0;
        act = true;
    }

    // ---- the transition (single_snake.py:197-304) on the env lanes
    u64 q0r = 0, q1r = 0;
    u32 rz = 0, rw = 0;
    {
        const u32 a_small = (a_in >= 0 && a_in < 4) ? (u32)a_in : 7u;
        const u32 a_mod = (a_in >= 0 && a_in < 4) ? (u32)a_in : ((u32)(int)(a_in % 4) & 7u);
        u32 fword = 0;
        if (act) fword = rng_words(p.seed, p.call, env_id, RNG_FOOD, 0).w[0];
        lw_transition<S, false, false>(o0, o1, qa, qb, qc, c, tc, L, o, food, act, make_uint4(a_small | (a_mod << 3), 0u, fword, 0u),
                                       q0r, q1r, rz, rw);
    }
    const bool eat = (rw & 0x100u) != 0, selfc = (rw & 0x200u) != 0, edgec = (rw & 0x400u) != 0;
    const bool fin = selfc || edgec;
    const int a_out = (int)(signed char)(rz >> 8);

    if (act) { // per-env outputs
        store_action(p.actions, p.act_dtype, env, (long long)a_out);
        p.selfc[env] = (uint8_t)selfc;
        p.reward[env] = eat ? 1.0f : 0.0f;
        p.done[env] = (uint8_t)fin;
        p.edgec[env] = (uint8_t)edgec;
        if (p.done_copy) p.done_copy[env] = (uint8_t)fin;
    }
    if (mine) // the mirror
        lwr_store(a.res, p.N, env, o0, o1, qa, qb, qc, lwr_pack(c, tc, L, o, food),
                  act ? (LWR_ACT | (fin ? LWR_TERMINAL : 0u)) : (flags_in & ~LWR_ACT));
    if (mine && a.check_mask != nullptr) // wurm_single_check's mask of the stepped state: a live env of the domain is a well-formed snake
        a.check_mask[env] = act && !fin ? ((L < 3 ? WURM_CHK_MIN_LENGTH : 0u) | (food < 0 ? WURM_CHK_ONE_FOOD : 0u)) : WURM_CHK_NOT_COMPUTED;

    // ---- observations: records of the (which, env) pairs — which = 0: the stepped state, 1: that state once a finished env is
    // rebuilt with call + 1 (not stored: the next launch's postponed reset recreates it)
    const u64 odd = ballot(mine && !act);
    if (OBS || POS) {
        uint4 *io0 = (uint4 *)(lds + Lds::IO0);
        uint2 *io1 = (uint2 *)(lds + Lds::IO1);
        // record: occupancy; head cell | valid << 8, food cell + 1
        u64 s0 = o0, s1 = o1;
        u32 hw = (u32)c | (act ? 0x100u : 0u), fw = (u32)(food + 1);
        if (lane < EPW) {
            io0[lane] = make_uint4((u32)s0, (u32)(s0 >> 32), (u32)s1, (u32)(s1 >> 32));
            io1[lane] = make_uint2(hw, fw);
        }
        if (NOBS == 2) {
            if (act && fin) {
                const LeanReset r = lean_reset_draw(p.seed, p.call + 1ull, env_id, S, rcpSm2);
                const int hc = r.b & 127, sc = (r.b >> 7) & 127, t2 = (r.b >> 14) & 127;
                s0 = s1 = 0;
                lw_set(s0, s1, hc); lw_set(s0, s1, sc); lw_set(s0, s1, t2);
                hw = (u32)hc | 0x100u;
                fw = (u32)(r.a >> 10) + 1u;
            }
            if (lane < EPW) {
                io0[EPW + lane] = make_uint4((u32)s0, (u32)(s0 >> 32), (u32)s1, (u32)(s1 >> 32));
                io1[EPW + lane] = make_uint2(hw, fw);
            }
        }
        if (OBS)
            for (int i = lane; i < (Lds::WORDS + 3) / 4; i += 64) ((uint4 *)bits)[i] = make_uint4(0, 0, 0, 0);
        wave_lds_sync();
        const int pw = lane >> LOG_EPW, pe = lane & (EPW - 1); // pair lane (which, env)
        uint4 q = make_uint4(0, 0, 0, 0);
        uint2 h = make_uint2(0, 0);
        if (lane < NP) { q = io0[lane]; h = io1[lane]; }
        const bool pv = lane < NP && pe < nenv && (h.x & 0x100u) != 0;
        float *ob0 = p.obs + env0 * E, *ob1 = NOBS == 2 ? p.obs_after + env0 * E : nullptr;
        if (POS) {
            if (pv) { // head y, x, food y, x (:153-163: the first maximum of an empty channel is cell 0)
                const int hc = (int)(h.x & 127u), fc = max((int)h.y - 1, 0), hy = hc / S, fy = fc / S;
                float *o4 = (pw ? p.obs_after : p.obs) + (env0 + pe) * 4;
                o4[0] = (float)hy; o4[1] = (float)(hc - hy * S); o4[2] = (float)fy; o4[3] = (float)(fc - fy * S);
            }
        } else {
            if (pv) lw_planes_of<S, OBSK, NW>(bits, wint, lane, (u64)q.x | ((u64)q.y << 32), (u64)q.z | ((u64)q.w << 32), (int)(h.x & 127u), (int)h.y - 1);
            wave_lds_sync();
            const bool aligned = (((size_t)p.obs) & 15u) == 0 && (NOBS == 1 || (((size_t)p.obs_after) & 15u) == 0);
            if (nenv == EPW && aligned) {
                constexpr int GSG = EPW * E / 4;   // 16-byte groups of one observation of the wave's envs
#pragma unroll 4
                for (int j = lane; j < NOBS * GSG; j += 64) {
                    const float4 v = lw_group<OBSK>(bits, tab, tabB, j);
                    const bool second = NOBS == 2 && j >= GSG;
                    ((float4 *)(second ? ob1 : ob0))[second ? j - GSG : j] = v;
                }
            } else { // the ragged last wave: float by float
                for (int f = lane; f < NP * E; f += 64) {
                    const int pr = f / E, k2 = f - pr * E, w = pr >> LOG_EPW, e = pr & (EPW - 1);
                    if (e < nenv && (io1[pr].x & 0x100u)) (w ? ob1 : ob0)[e * E + k2] = lw_float<OBSK>(bits, f);
                }
            }
        }
        wave_lds_sync();
    }

    // ---- envs outside the domain: the one-env-per-wave code reads and writes their state, outputs and observations itself
    // (nothing above touched them except observation bytes, which it overwrites)
    if (odd != 0) {
        if (ballot(stale_terminal) != 0) { // (their last state as the mirror held it at entry)
            unsigned char *bm = lds + Lds::SCR;
            short *hcs = (short *)(bm + EPW * G::BM), *fcs = hcs + EPW;
            lwr_write_planes<EPW, S>(p.envs + env0 * C3, nenv, lane, bm, hcs, fcs, stale_terminal, t_c, t_L, t_qa, t_qb, t_qc, t_food);
        }
        __threadfence();
        wave_lds_sync();
        for (u64 m = odd; m != 0; m &= m - 1)
            fused_step_env<2, true>(p, env0 + first_bit(m), (signed char *)(lds + Lds::SCR));
    }
}

// ------------------------------------------------------------------------------------------------ host side

// the shapes the resident step serves (the caller opts in per call by passing wurm_single_call.resident)
bool lane_wide_resident_shape(int S, int obs_mode, int obs_n)
{
    if (S != 10 && S != 11) return false;
    if (obs_mode == WURM_OBS_PARTIAL) return obs_n == 2 || obs_n == 3;
    return obs_mode == WURM_OBS_NONE || obs_mode == WURM_OBS_DEFAULT || obs_mode == WURM_OBS_ONE_CHANNEL || obs_mode == WURM_OBS_POSITIONS;
}

bool lane_wide_resident_eligible(const StepArgs &p)
{
    if (!lane_wide_resident_shape(p.S, p.obs_mode, p.obs_n)) return false;
    if (p.inject_food || p.inject_reset || p.inject_pre_reset || p.post_reset || p.only_flagged) return false;
    if (p.N * (long long)LWR_BYTES >= (1ll << 40)) return false;
    return true;
}

template <int S>
static hipError_t launch_lane_wide_resident_flush_size(const StepArgs &p, void *resident, hipStream_t stream)
{
    WideResArgs a;
    a.p = p;
    a.res = (uint4 *)resident;
    a.check_mask = nullptr;
    constexpr int EPW = 16;
    const long long waves = (p.N + EPW - 1) / EPW;
    const int wpb = waves >= 1024 ? 4 : 1;
    dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
    (void)hipGetLastError();
    WURM_LAUNCH((lane_wide_resident_flush_kernel<EPW, S>), grid, block, (size_t)((EPW * LwGeo<S>::BM + 4 * EPW) * wpb), stream, a);
    return hipGetLastError();
}

hipError_t launch_lane_wide_resident_flush(const StepArgs &p, void *resident, hipStream_t stream)
{
    return p.S == 10 ? launch_lane_wide_resident_flush_size<10>(p, resident, stream) : launch_lane_wide_resident_flush_size<11>(p, resident, stream);
}

template <int S, int OBSK, int NW>
static hipError_t launch_lane_wide_resident_obs(const WideResArgs &a, hipStream_t stream)
{
    const StepArgs &p = a.p;
    const int nobs = (p.obs_mode != WURM_OBS_NONE && p.obs_after != nullptr) ? 2 : 1;
    // envs per wave (automatic unless the option WURM_RESIDENT_EPW forces it: tests and the tuning sweep)
    int epw = (int)opt.resident_epw;
    if (epw != 16 && epw != 32) epw = nobs == 1 ? (p.N >= 16384 ? 32 : 16) : (p.N >= 49152 ? 32 : 16);
    auto go = [&](auto kernel, int e, int wave_bytes) {
        const long long waves = (p.N + e - 1) / e;
        const int wpb = waves >= 1024 ? 4 : 1;
        dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
        const size_t lds_bytes = (size_t)(LW_TAB + wave_bytes * wpb);
        if (lds_bytes > 65536) (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        WURM_LAUNCH(kernel, grid, block, lds_bytes, stream, a);
    };
    if constexpr (OBSK == WURM_OBS_NONE) {
        if (epw == 16) go(lane_wide_resident_step_kernel<16, S, OBSK, NW, 1>, 16, lwr_wave_bytes<16, S, OBSK, NW>());
        else go(lane_wide_resident_step_kernel<32, S, OBSK, NW, 1>, 32, lwr_wave_bytes<32, S, OBSK, NW>());
    } else if (nobs == 2) {
        if (epw == 16) go(lane_wide_resident_step_kernel<16, S, OBSK, NW, 2>, 16, lwr_wave_bytes<16, S, OBSK, NW>());
        else go(lane_wide_resident_step_kernel<32, S, OBSK, NW, 2>, 32, lwr_wave_bytes<32, S, OBSK, NW>());
    } else {
        if (epw == 16) go(lane_wide_resident_step_kernel<16, S, OBSK, NW, 1>, 16, lwr_wave_bytes<16, S, OBSK, NW>());
        else go(lane_wide_resident_step_kernel<32, S, OBSK, NW, 1>, 32, lwr_wave_bytes<32, S, OBSK, NW>());
    }
    return hipGetLastError();
}

template <int S>
static hipError_t launch_lane_wide_resident_size(const StepArgs &p, void *resident, bool valid, uint32_t *check_mask, hipStream_t stream)
{
    WideResArgs a;
    a.p = p;
    a.res = (uint4 *)resident;
    a.check_mask = check_mask;
    (void)hipGetLastError();
    if (!valid) {
        constexpr int EPW = 16;
        const long long waves = (p.N + EPW - 1) / EPW;
        const int wpb = waves >= 1024 ? 4 : 1;
        dim3 block(64 * wpb), grid((unsigned)((waves + wpb - 1) / wpb));
        WURM_LAUNCH((lane_wide_resident_build_kernel<EPW, S>), grid, block, (size_t)(LwLds<EPW, S, WURM_OBS_NONE, 0>::BYTES * wpb), stream, a);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) return err;
    }
    if (p.obs_mode == WURM_OBS_NONE) return launch_lane_wide_resident_obs<S, WURM_OBS_NONE, 0>(a, stream);
    if (p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 2) return launch_lane_wide_resident_obs<S, WURM_OBS_PARTIAL, 5>(a, stream);
    if (p.obs_mode == WURM_OBS_PARTIAL && p.obs_n == 3) return launch_lane_wide_resident_obs<S, WURM_OBS_PARTIAL, 7>(a, stream);
    if (p.obs_mode == WURM_OBS_ONE_CHANNEL) return launch_lane_wide_resident_obs<S, LW_OBS_GRID1, 0>(a, stream);
    if (p.obs_mode == WURM_OBS_POSITIONS) return launch_lane_wide_resident_obs<S, WURM_OBS_POSITIONS, 0>(a, stream);
    return launch_lane_wide_resident_obs<S, LW_OBS_GRID3, 0>(a, stream);
}

// (lazy form only: see the header of this file)
hipError_t launch_lane_wide_resident(const StepArgs &p, void *resident, bool valid, uint32_t *check_mask, hipStream_t stream)
{
    return p.S == 10 ? launch_lane_wide_resident_size<10>(p, resident, valid, check_mask, stream)
                     : launch_lane_wide_resident_size<11>(p, resident, valid, check_mask, stream);
}

} // namespace wurm
