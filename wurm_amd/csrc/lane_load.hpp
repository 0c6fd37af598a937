// lane_load.hpp — the cooperative state read shared by the one-env-per-LANE kernels (lane_step.hpp: one call; lane_rollout.hpp:
// a fused rollout): a wave's block of EPW consecutive SingleSnake envs (EPW * 3 * S * S floats, one contiguous, 16-byte
// aligned run) is read as float4 and summarised per env in LDS — head cell, food cell, the cell of every body value, the
// bit set of the body values present, a packed counter (cells | heads << 8 | foods << 16 | out-of-range values << 24).
// About 2 % of the elements are non-zero: each wave-wide slot (j, k) compacts its non-zero (element index, value) pairs
// into an LDS queue with one ballot + prefix count, and the queue is decoded 64 entries at a time — the division-heavy
// decode runs once per 64 non-zero elements instead of once per slot (round 2 read three predicated dwords per
// (env, cell) pair: about half of lane_step_kernel's 109 VALU instructions per env).
#pragma once

namespace wurm {

constexpr int LANE_QCAP = 512; // entries of the non-zero queue (drained when fewer than 256 are free)

// vm: u32 [VW][EPW] bit set of body values (values 1 .. 32 * VW - 1 are in range), stat: u32 [EPW], hpos / fpos: u8 [EPW],
// valpos: u8 [EPW][VS] cell of each body value, queue: u64 [LANE_QCAP].  All of the block's EPW envs are present.
template <int EPW, int C, int VW, int VS>
__device__ __forceinline__ void lane_load_block(const float *__restrict__ block, int lane, u32 *vm, u32 *stat,
                                                unsigned char *hpos, unsigned char *fpos, unsigned char *valpos, u64 *queue)
{
    constexpr int C3 = 3 * C, N4 = EPW * C3 / 4, B4 = 16; // B4 loads in flight per lane
    static_assert((EPW * C3) % 4 == 0, "the block is a whole number of float4");
    const float4 *base4 = (const float4 *)block;
    int qn = 0; // wave-uniform
    auto drain = [&]() {
        wave_lds_sync();
        for (int i = lane; i < qn; i += 64) {
            const u64 ent = queue[i];
            const float val = __uint_as_float((u32)ent);
            const int f = (int)(ent >> 32), ej = f / C3, r = f - ej * C3, ch = r / C, cj = r - ch * C;
            if (ch == 0) {
                if (val > 0.5f) { fpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 16); }
            } else if (ch == 1) {
                if (val > 0.5f) { hpos[ej] = (unsigned char)cj; atomicAdd(&stat[ej], 1u << 8); }
            } else {
                const int bi = __float2int_rn(val); // body (single_snake.py:210): position of every value, values present
                if (bi > 0 && bi < 32 * VW) {
                    valpos[ej * VS + bi] = (unsigned char)cj;
                    atomicOr(&vm[(bi >> 5) * EPW + ej], 1u << (bi & 31));
                    atomicAdd(&stat[ej], 1u);
                } else if (bi != 0) {
                    atomicAdd(&stat[ej], 1u << 24);
                }
            }
        }
        wave_lds_sync();
        qn = 0;
    };
    for (int i0 = 0; i0 < N4; i0 += 64 * B4) {
        float x[B4], y[B4], z[B4], w[B4];
#pragma unroll
        for (int j = 0; j < B4; ++j) {
            const int g = i0 + 64 * j + lane;
            const float4 v = base4[min(g, N4 - 1)];
            x[j] = v.x; y[j] = v.y; z[j] = v.z; w[j] = v.w;
        }
#pragma unroll
        for (int j = 0; j < B4; ++j) {
            const int g = i0 + 64 * j + lane;
            if (i0 + 64 * j >= N4) break;
            if (qn > LANE_QCAP - 256) drain();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float val = k == 0 ? x[j] : k == 1 ? y[j] : k == 2 ? z[j] : w[j];
                const bool nz = g < N4 && val != 0.0f;
                const u64 m = ballot(nz);
                if (m != 0) {
                    if (nz) queue[qn + rank_below(m)] = ((u64)(u32)(4 * g + k) << 32) | (u64)__float_as_uint(val);
                    qn += popc64(m);
                }
            }
        }
    }
    drain();
}

} // namespace wurm
