// Micro-benchmark: cycles per instruction for ONE wave alone on a SIMD (the situation of the 512-env rollout):
// dependent / independent SALU and VALU chains, the SALU<->VALU hand-offs the lean rollout loop is made of
// (v_readlane -> SALU -> v_readlane, v_cmp -> s_and -> v_cndmask), taken branches, stores.
// build: hipcc --offload-arch=gfx950 -O2 -o issue_latency issue_latency.hip ; run: ./issue_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP32(x) REP16(x) REP16(x)

#define BENCH(NAME, NINSTR, BODY)                                                                   \
    __global__ void NAME(long long *out, int iters, float *sink)                                    \
    {                                                                                               \
        int v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3;                                               \
        int s0 = iters, s1 = 1, s2 = 2, s3 = 3;                                                     \
        asm volatile("" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3));                                  \
        long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int i = 0; i < iters; ++i) {                                                           \
            asm volatile(BODY : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) \
                         : "s"(sink) : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "v10", "v11", "v12", "v13", "v14", "v15", "memory");      \
        }                                                                                           \
        long long t1 = __builtin_amdgcn_s_memtime();                                                \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = NINSTR; }                                \
        if (v0 + v1 + v2 + v3 + s0 + s1 + s2 + s3 == 0x7fffffff) sink[0] = 1.0f;                    \
    }

BENCH(salu_dep, 32, REP32("s_add_i32 %4, %4, 1\n"))
BENCH(salu_indep, 32, REP16("s_add_i32 %4, %4, 1\n s_add_i32 %5, %5, 1\n"))
BENCH(valu_dep, 32, REP32("v_add_u32 %0, %0, 1\n"))
BENCH(valu_indep, 32, REP16("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n"))
BENCH(valu_indep4, 32, REP4(REP4("v_add_u32 %0, %0, 1\n") REP4("v_add_u32 %1, %1, 1\n")))
BENCH(salu_valu_alt_indep, 32, REP16("s_add_i32 %4, %4, 1\n v_add_u32 %0, %0, 1\n"))
// VALU reads an SGPR the previous SALU wrote, SALU independent of VALU
BENCH(salu_to_valu, 32, REP16("s_add_i32 %4, %4, 1\n v_add_u32 %0, %4, %0\n"))
// v_cmp writes an SGPR pair, s_and reads it, v_cndmask reads the result: the mask logic of the crop
BENCH(vcmp_sand_vcnd, 48, REP16("v_cmp_lt_i32 s[40:41], %4, %0\n s_and_b64 s[42:43], s[40:41], exec\n v_cndmask_b32 %0, %0, %1, s[42:43]\n"))
BENCH(vcmp_vcnd_vcc, 32, REP16("v_cmp_lt_i32 vcc, %4, %0\n v_cndmask_b32 %0, %0, %1, vcc\n"))
// readlane with an SGPR lane select computed by SALU from the previous readlane (the move-table lookup)
BENCH(readlane_chain, 48, REP16("v_readlane_b32 s44, %0, %4\n s_and_b32 %4, s44, 63\n s_nop 3\n"))
BENCH(readlane_const, 32, REP16("v_readlane_b32 s44, %0, 3\n s_add_i32 %4, %4, s44\n"))
BENCH(readlane_only, 32, REP32("v_readlane_b32 s44, %0, 3\n"))
BENCH(snop0, 32, REP32("s_nop 0\n"))
BENCH(scmp_cselect, 32, REP16("s_cmp_lt_i32 %4, %5\n s_cselect_b32 %4, %4, %6\n"))
BENCH(branch_taken, 32, REP16("s_branch 0\n s_add_i32 %4, %4, 1\n"))
BENCH(branch_not_taken, 32, REP16("s_cmp_eq_u32 %5, 77\n s_cbranch_scc1 1\n"))
BENCH(store3, 48, REP4("global_store_dword %0, %1, %8\n global_store_dword %0, %2, %8 offset:256\n global_store_dword %0, %3, %8 offset:512\n" REP4("v_add_u32 %1, %1, 1\n") REP4("s_add_i32 %4, %4, 1\n") "s_nop 0\n"))
BENCH(vlshr64, 32, REP32("v_lshrrev_b64 v[10:11], %0, s[40:41]\n"))
BENCH(salu64_dep, 32, REP32("s_and_b64 s[40:41], s[40:41], exec\n"))
BENCH(vmul_lo, 32, REP16("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3\n"))
BENCH(vmul_hi, 32, REP16("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3\n"))
BENCH(vmad64, 32, REP32("v_mad_u64_u32 v[10:11], s[40:41], %0, %1, v[10:11]\n"))
BENCH(vfma_dep, 32, REP32("v_fma_f32 %0, %1, %2, %0\n"))
BENCH(vpkfma_dep, 32, REP32("v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[10:11]\n"))
BENCH(smul, 32, REP16("s_mul_i32 %4, %4, %5\n s_mul_hi_u32 %6, %6, %5\n"))
BENCH(vcmp_vcnd_sgpr, 32, REP16("v_cmp_lt_i32 s[40:41], %4, %0\n v_cndmask_b32 %0, %0, %1, s[40:41]\n"))
BENCH(vcmp3_vcnd3, 96, REP4(REP4("v_cmp_lt_i32 s[40:41], %4, %0\n v_cmp_lt_i32 s[42:43], %4, %1\n v_cmp_lt_i32 vcc, %4, %2\n v_cndmask_b32 %0, %0, %3, s[40:41]\n v_cndmask_b32 %1, %1, %3, s[42:43]\n v_cndmask_b32 %2, %2, %3, vcc\n")) )

struct B { const char *name; void (*fn)(long long *, int, float *); };

int main()
{
    long long *out;
    float *sink;
    hipMalloc(&out, 16);
    hipMalloc(&sink, 1 << 20);
    hipMemset(sink, 0, 1 << 20);
    B benches[] = {{"salu_dep", salu_dep}, {"salu_indep", salu_indep}, {"valu_dep", valu_dep}, {"valu_indep", valu_indep},
                   {"valu_indep4", valu_indep4}, {"salu_valu_alt_indep", salu_valu_alt_indep},
                   {"salu_to_valu", salu_to_valu}, {"vcmp_sand_vcnd", vcmp_sand_vcnd}, {"vcmp_vcnd_vcc", vcmp_vcnd_vcc},
                   {"readlane_chain(+s_nop 3)", readlane_chain}, {"readlane_const", readlane_const},
                   {"readlane_only", readlane_only}, {"s_nop 0", snop0}, {"scmp_cselect", scmp_cselect},
                   {"branch_taken", branch_taken}, {"branch_not_taken", branch_not_taken}, {"store3+8valu+4salu", store3},
                   {"v_lshrrev_b64", vlshr64}, {"salu64_dep", salu64_dep}, {"v_mul_lo_u32", vmul_lo}, {"v_mul_hi_u32", vmul_hi}, {"v_mad_u64_u32", vmad64}, {"v_fma_f32 dependent", vfma_dep}, {"v_pk_fma_f32 dependent", vpkfma_dep}, {"s_mul_i32/s_mul_hi_u32", smul}, {"vcmp->vcndmask via sgpr pair", vcmp_vcnd_sgpr}, {"3 vcmp then 3 vcndmask", vcmp3_vcnd3}};
    const int iters = 2000;
    for (auto &b : benches) {
        long long h[2];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(b.fn, dim3(1), dim3(64), 0, 0, out, iters, sink);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        printf("%-28s %8.2f cycles per instruction (%lld instr per iteration; loop overhead included)\n", b.name,
               (double)h[0] / ((double)iters * (double)h[1]), h[1]);
    }
    return 0;
}
