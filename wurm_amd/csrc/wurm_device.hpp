// wurm_device.hpp — device-side helpers shared by the gfx950 env-step kernels.
//
// Execution model used throughout: ONE ENVIRONMENT PER 64-LANE WAVEFRONT.  Lane l owns the cells
// c = l + 64*k (k = 0..CPL-1) of the row-major S*S grid, so every global access of a channel is a run of
// consecutive dwords across the wave.  Per-env scalars (head cell, snake length, collision flags) are
// wave-uniform values produced with ballots / wave reductions and live in SGPRs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "options.hpp"

namespace wurm {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int WAVE = 64;

// RNG purposes: one Philox stream per (env, call, purpose, sub-block).  Must match oracle/oracle_common.h.
enum : u32 {
    RNG_FOOD = 0,
    RNG_RESET = 1,
    RNG_DEATH_FOOD_A = 2,
    RNG_DEATH_FOOD_B = 3,
    RNG_BOOST_COST = 4,
    RNG_RATE_FOOD = 5,
    RNG_SPAWN = 6,
    RNG_COLOUR = 7,
    RNG_POLICY = 8
};

struct Words {
    u32 w[4];
};

// Philox4x32-10, counter = (env_id, call_lo, call_hi, purpose | sub << 8), key = seed.
__device__ __forceinline__ Words rng_words(u64 seed, u64 call, u64 env_id, u32 purpose, u32 sub)
{
    const u32 M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    u32 c0 = (u32)env_id, c1 = (u32)call, c2 = (u32)(call >> 32), c3 = (purpose & 0xffu) | (sub << 8);
    u32 k0 = (u32)seed, k1 = (u32)(seed >> 32);
    // (the key is wave-uniform and loop-invariant in every rollout kernel: left alone, the compiler hoists the ten round keys
    // out of the step loop — twenty more live SGPRs in kernels that already spill a hundred — for the sake of eighteen
    // scalar adds per evaluation; the empty asm makes the key opaque at each evaluation)
    asm volatile("" : "+s"(k0), "+s"(k1));
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const u64 p0 = (u64)M0 * c0, p1 = (u64)M1 * c2; // one 32x32->64 multiply each (v_mad_u64_u32 per lane)
        const u32 hi0 = (u32)(p0 >> 32), lo0 = (u32)p0, hi1 = (u32)(p1 >> 32), lo1 = (u32)p1;
        u32 n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    Words o;
    o.w[0] = c0; o.w[1] = c1; o.w[2] = c2; o.w[3] = c3;
    return o;
}

__device__ __forceinline__ u32 mulhi_range(u32 w, u32 n) { return __umulhi(w, n); } // uniform in [0,n)
__device__ __forceinline__ float u01(u32 w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// q^n in fp32 by binary exponentiation — this exact operation order is the specification (oracle/oracle_common.h:
// oracle_pow_n does the same); 0 <= q <= 1
__device__ __forceinline__ float pow_n(float q, int n)
{
    float pw = 1.0f, base = q;
    for (int e = n; e; e >>= 1) {
        if (e & 1) pw = pw * base;
        base = base * base;
    }
    return pw;
}

// Number of successes among n independent Bernoulli(p) trials from ONE uniform u, by inversion of the Binomial(n, p)
// distribution function (oracle/oracle_common.h: oracle_binomial_inverse, the same fp32 operations in the same order).
// pw = pow_n(1 - p, n) = P(0 successes), which the caller has compared against BINOMIAL_MIN_P0 (below it the recurrence
// starts from a number with too few bits, or from an underflowed 0: the caller draws cell by cell instead).
constexpr float BINOMIAL_MIN_P0 = 1e-6f;
__device__ __forceinline__ int binomial_inverse(int n, float p, float pw, float u)
{
    const float r = p / (1.0f - p);
    float pmf = pw, cdf = pw;
    int k = 0;
    while (u >= cdf && k < n) {
        pmf = pmf * ((float)(n - k) * r) / (float)(k + 1);
        const float nc = cdf + pmf;
        ++k;
        if (nc == cdf) break; // the tail beyond is below the rounding of the sum
        cdf = nc;
    }
    return k;
}

// per-cell uniform: cells c, c+64, c+128, c+192 share one Philox block (sub = (c >> 8) * 64 + (c & 63),
// word = (c >> 6) & 3), so the lane that owns them (c & 63) computes one block for four of its cells.
__device__ __forceinline__ float cell_u01(u64 seed, u64 call, u64 env_id, u32 purpose, u32 cell)
{
    Words o = rng_words(seed, call, env_id, purpose, ((cell >> 8) << 6) | (cell & 63u));
    u32 j = (cell >> 6) & 3u;
    u32 w = j == 0 ? o.w[0] : j == 1 ? o.w[1] : j == 2 ? o.w[2] : o.w[3];
    return u01(w);
}

// ---- wave-level primitives -------------------------------------------------------------------------------

__device__ __forceinline__ u64 ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int popc64(u64 m) { return __popcll(m); }
__device__ __forceinline__ int first_bit(u64 m) { return __ffsll((long long)m) - 1; } // -1 when empty

// number of set bits of m below this lane
__device__ __forceinline__ int rank_below(u64 m)
{
    return (int)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
}

// index of the k-th (0-based) set bit of w; k < popc(w)
__device__ __forceinline__ int nth_bit64(u64 w, int k)
{
    int pos = 0;
    u32 x = (u32)w;
    int c = __popc(x);
    if (k >= c) { k -= c; pos = 32; x = (u32)(w >> 32); }
#pragma unroll
    for (int sh = 16; sh >= 1; sh >>= 1) {
        c = __popc(x & ((1u << sh) - 1u));
        if (k >= c) { k -= c; x >>= sh; pos += sh; }
    }
    return pos;
}

// DPP lane exchange inside a row of 16 lanes (VALU latency; no LDS round trip like ds_bpermute / __shfl).
template <int CTRL>
__device__ __forceinline__ int dpp_row(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}

constexpr int DPP_QUAD_XOR1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_XOR2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7-i within each 8
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15-i within each 16

// Wave-wide reductions, result wave-uniform (SGPR): 4 DPP steps inside each 16-lane row, then 4 readlanes.
// The butterfly is hand-written: for `v = op(v, dpp_row<..>(v))` the compiler emits v_mov + s_nop + v_mov_dpp + op per
// step (~20 instructions per reduction), the DPP form of the op itself is one instruction per step.  Rules that keep it
// safe (tools/microbench/reduce_test.hip exercises them, including partially active waves):
//   * IN PLACE — destination = second source = %0.  With bound_ctrl:0 a lane whose DPP source lane is disabled by EXEC
//     is not written at all; with a fresh destination register it would keep whatever that register held (for a
//     minimum: possibly a small stale value, i.e. a wrong cell index), in place it keeps its own value, which is the
//     identity of min / max and what __builtin_amdgcn_update_dpp(v, v, ...) gives;  for the SUM the untouched lane must
//     contribute nothing twice, so sums are only taken with all lanes active (every call site is wave-uniform code);
//   * s_nop 1 in front of every DPP read of a VGPR written by the previous VALU instruction (2 wait states on gfx9;
//     the hazard recogniser does not look inside an asm block).
#define WURM_DPP_BUTTERFLY(OP)                                                                         \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                  \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                  \
    "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                      \
    "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                           \
    "s_nop 0"

__device__ __forceinline__ int wave_max_i32(int v)
{
    asm(WURM_DPP_BUTTERFLY("v_max_i32_dpp") : "+v"(v));
    int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}

__device__ __forceinline__ int wave_min_i32(int v)
{
    asm(WURM_DPP_BUTTERFLY("v_min_i32_dpp") : "+v"(v));
    int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
    int r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
    return min(min(r0, r1), min(r2, r3));
}

__device__ __forceinline__ int wave_sum_i32(int v)
{
    asm(WURM_DPP_BUTTERFLY("v_add_u32_dpp") : "+v"(v));
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

__device__ __forceinline__ long long uniform64(long long v)
{
    u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v);
    u32 hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)((u64)v >> 32));
    return (long long)(((u64)hi << 32) | lo);
}

// value held by lane `src`; src must be wave-uniform (v_readlane_b32 with an SGPR lane select)
__device__ __forceinline__ int lane_value(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

__device__ __forceinline__ long long lane_value64(long long v, int src)
{
    u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, src);
    u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)v >> 32), src);
    return (long long)(((u64)hi << 32) | lo);
}

// LDS written by some lanes of a wave and read by other lanes of the SAME wave: the hardware runs one wave's
// LDS instructions in order; this only stops the compiler from moving accesses across the hand-off.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup -> first env of the workgroup, XCD-aware.  The hardware deals workgroups to the 8 XCDs round-robin (workgroup
// b runs on XCD b % 8) and every XCD has its own L2.  Neighbouring envs share cache lines: the 300-byte observation
// records of 9 x 9 'partial_2' envs and their 972-byte state slabs are packed back to back, so with env = b the two halves
// of almost every line are written from two different L2s and go to HBM as two partial-line writes (round 1 measured
// WRITE_SIZE = 1.32 x the bytes stored).  Giving each XCD one CONTIGUOUS range of envs keeps both halves of a line in one
// L2, where they merge before the write-back.  Pure relabelling: every random draw is keyed by the env id, not by the
// workgroup, so results do not change.
__device__ __forceinline__ long long xcd_block(unsigned b, unsigned nblocks)
{
    const unsigned q = nblocks >> 3; // workgroups per XCD; the nblocks % 8 stragglers keep their index
    return b < 8u * q ? (long long)(b & 7u) * q + (b >> 3) : (long long)b;
}

// exact floor(c / S) for 0 <= c < 2^16, 1 <= S <= 256, via one fp32 multiply (rcpS = 1.0f / S)
__device__ __forceinline__ int div_size(int c, float rcpS) { return (int)(((float)c + 0.5f) * rcpS); }

// taps of the reference's ORIENTATION_FILTERS (wurm/_filters.py:7-28): orientation i <=> head = neck + TAP[i];
// action a moves the head by -TAP[a] = [(+1,0),(0,-1),(-1,0),(0,+1)] (row, col).
__device__ __forceinline__ int tap_y(int i) { return i == 0 ? -1 : (i == 2 ? 1 : 0); }
__device__ __forceinline__ int tap_x(int i) { return i == 1 ? 1 : (i == 3 ? -1 : 0); }

// In-kernel timeline (tools/kernel_timeline.py): a build with -DWURM_TIMELINE (make -C wurm_amd/csrc timeline ->
// libwurm_hip_timeline.so, loaded with WURM_HIP_LIBRARY=...) stamps s_memtime at up to eight points of the per-call
// lane kernels and overwrites the first 64 bytes of the wave's own observation block with the stamps once its stores have
// drained.  Scalars, not an array (an array would live in scratch).  In the shipped build the macros are empty.
#ifdef WURM_TIMELINE
#define WURM_TL_DECL unsigned long long tl_0 = 0, tl_1 = 0, tl_2 = 0, tl_3 = 0, tl_4 = 0, tl_5 = 0, tl_6 = 0, tl_7 = 0
#define WURM_TL(k) tl_##k = __builtin_amdgcn_s_memtime()
#define WURM_TL_STORE(ptr, lane)                                                                                         \
    do {                                                                                                                 \
        __builtin_amdgcn_s_waitcnt(0);                                                                                   \
        WURM_TL(7);                                                                                                      \
        unsigned long long *tl_o = (unsigned long long *)(ptr);                                                          \
        if ((lane) == 0) {                                                                                               \
            tl_o[0] = tl_0; tl_o[1] = tl_1; tl_o[2] = tl_2; tl_o[3] = tl_3;                                              \
            tl_o[4] = tl_4; tl_o[5] = tl_5; tl_o[6] = tl_6; tl_o[7] = tl_7;                                              \
        }                                                                                                                \
    } while (0)
// MultiSnake kernels (tools/multi_timeline.py): 32 eight-byte slots of LDS per env (Ctx::tl; multi_layout reserves them in the
// timeline build only) so that device functions (step_middle, multi_step_body, class_write) can stamp too.  WURM_TLS(cx, k)
// writes the stamp into slot k (per-call kernels: a wave passes each point once) AND adds the time since the wave's previous
// stamp to slot 16 + k (rollouts: per-segment totals over the launch; slot 15's accumulator place, 31, holds the previous
// stamp).  WURM_TLS_STORE copies slots 0 .. 15, WURM_TLA_STORE the totals 16 .. 31, over the first 128 bytes of `ptr` once
// the wave's stores have drained.
#define WURM_TLS(cx, k)                                                                                                  \
    do {                                                                                                                 \
        const unsigned long long tls_t = __builtin_amdgcn_s_memtime();                                                   \
        if ((cx).lane == 0) {                                                                                            \
            (cx).tl[k] = tls_t;                                                                                          \
            (cx).tl[16 + (k)] += tls_t - (cx).tl[31];                                                                    \
            (cx).tl[31] = tls_t;                                                                                         \
        }                                                                                                                \
    } while (0)
#define WURM_TLS_INIT(cx)                                                                                                \
    do {                                                                                                                 \
        if ((cx).lane < 32) (cx).tl[(cx).lane] = (cx).lane == 31 ? __builtin_amdgcn_s_memtime() : 0ull;                  \
    } while (0)
#define WURM_TLS_STORE(cx, ptr)                                                                                          \
    do {                                                                                                                 \
        __builtin_amdgcn_s_waitcnt(0);                                                                                   \
        WURM_TLS(cx, 15);                                                                                                \
        __builtin_amdgcn_s_waitcnt(0);                                                                                   \
        if ((cx).lane < 16) ((unsigned long long *)(ptr))[(cx).lane] = (cx).tl[(cx).lane];                               \
    } while (0)
#define WURM_TLA_STORE(cx, ptr)                                                                                          \
    do {                                                                                                                 \
        __builtin_amdgcn_s_waitcnt(0);                                                                                   \
        if ((cx).lane < 15) ((unsigned long long *)(ptr))[(cx).lane] = (cx).tl[16 + (cx).lane];                          \
    } while (0)
#else
#define WURM_TL_DECL
#define WURM_TL(k)
#define WURM_TL_STORE(ptr, lane)
#define WURM_TLS(cx, k)
#define WURM_TLS_INIT(cx)
#define WURM_TLS_STORE(cx, ptr)
#define WURM_TLA_STORE(cx, ptr)
#endif

} // namespace wurm
