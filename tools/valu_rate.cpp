// valu_rate -- measured issue cost (shader cycles per wave64 instruction per SIMD, from s_memtime, and ns from HIP
// events) of the VALU ops the K2/K3 kernels are (or could be) built from, at 1, 2 and 4 waves per SIMD.
// Sets the VALU-side ceiling quoted in DESIGN.md.  Every kernel runs 8 independent dependency chains.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_rate.cpp -o tools/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
// T = asm template; %0 = the chain register (read-modify-write), %1 = b, %2 = c (loop-invariant VGPRs)
#define CH(T, r) asm volatile(T : "+v"(r) : "v"(b), "v"(c));
#define BODY(T) CH(T, a0) CH(T, a1) CH(T, a2) CH(T, a3) CH(T, a4) CH(T, a5) CH(T, a6) CH(T, a7)
#define OPS(name, T)                                                                                          \
    __global__ __launch_bounds__(256) void k_##name(uint32_t *out, unsigned long long *cyc, int iters)         \
    {                                                                                                         \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,            \
                 a6 = a0 + 6, a7 = a0 + 7, b = blockIdx.x | 1, c = 0x00030005;                                 \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
        for (int i = 0; i < iters; i++) {                                                                     \
            REP8(BODY(T))                                                                                     \
        }                                                                                                     \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                   \
        if ((threadIdx.x & 63) == 0)                                                                          \
            cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                                               \
    }

// ---- VOP2 (32-bit encodings)
OPS(add_u32, "v_add_u32 %0, %0, %1")
OPS(sub_u32, "v_sub_u32 %0, %0, %1")
OPS(and_b32, "v_and_b32 %0, %0, %1")
OPS(or_b32, "v_or_b32 %0, %0, %1")
OPS(lshlrev_b32, "v_lshlrev_b32 %0, 2, %0")
OPS(lshrrev_b32, "v_lshrrev_b32 %0, 2, %0")
OPS(max_u32, "v_max_u32 %0, %0, %1")
OPS(min_u32, "v_min_u32 %0, %0, %1")
OPS(mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
OPS(add_u16, "v_add_u16 %0, %0, %1")
OPS(sub_u16, "v_sub_u16 %0, %0, %1")
OPS(max_u16, "v_max_u16 %0, %0, %1")
OPS(mul_lo_u16, "v_mul_lo_u16 %0, %0, %1")
OPS(lshlrev_b16, "v_lshlrev_b16 %0, 2, %0")
OPS(add_f32, "v_add_f32 %0, %0, %1")
OPS(mov_b32, "v_mov_b32 %0, %1")
// ---- VOP3 (64-bit encodings)
OPS(fma_f32, "v_fma_f32 %0, %0, %1, %2")
OPS(add3_u32, "v_add3_u32 %0, %0, %1, %2")
OPS(lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
OPS(add_lshl_u32, "v_add_lshl_u32 %0, %0, %1, 2")
OPS(lshl_or_b32, "v_lshl_or_b32 %0, %0, 2, %1")
OPS(and_or_b32, "v_and_or_b32 %0, %0, %1, %2")
OPS(or3_b32, "v_or3_b32 %0, %0, %1, %2")
OPS(xad_u32, "v_xad_u32 %0, %0, %1, %2")
OPS(mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OPS(mad_u16, "v_mad_u16 %0, %0, %1, %2")
OPS(bfe_u32, "v_bfe_u32 %0, %0, 8, 8")
OPS(bfi_b32, "v_bfi_b32 %0, %0, %1, %2")
OPS(alignbit_b32, "v_alignbit_b32 %0, %0, %1, 16")
OPS(alignbyte_b32, "v_alignbyte_b32 %0, %0, %1, 1")
OPS(perm_b32, "v_perm_b32 %0, %0, %1, %2")
OPS(sad_u8, "v_sad_u8 %0, %0, %1, %2")
OPS(sad_u16, "v_sad_u16 %0, %0, %1, %2")
OPS(sad_u32, "v_sad_u32 %0, %0, %1, %2")
OPS(msad_u8, "v_msad_u8 %0, %0, %1, %2")
OPS(lerp_u8, "v_lerp_u8 %0, %0, %1, %2")
OPS(max3_u32, "v_max3_u32 %0, %0, %1, %2")
OPS(med3_u32, "v_med3_u32 %0, %0, %1, %2")
OPS(add_u32_e64, "v_add_u32_e64 %0, %0, %1")
OPS(sub_u16_clamp, "v_sub_u16_e64 %0, %0, %1 clamp")
OPS(add_u32_clamp, "v_add_u32_e64 %0, %0, %1 clamp")
OPS(sub_u32_clamp, "v_sub_u32_e64 %0, %0, %1 clamp")
OPS(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
OPS(mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
// ---- VOP3P
OPS(pk_add_u16, "v_pk_add_u16 %0, %0, %1")
OPS(pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
OPS(pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %2")
OPS(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
OPS(pk_max_u16, "v_pk_max_u16 %0, %0, %1")
OPS(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
OPS(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
OPS(pk_lshlrev_b16, "v_pk_lshlrev_b16 %0, 2, %0")
OPS(pk_lshrrev_b16, "v_pk_lshrrev_b16 %0, 8, %0")
OPS(dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %2")
OPS(dot2_u32_u16, "v_dot2_u32_u16 %0, %0, %1, %2")
OPS(dot8_u32_u4, "v_dot8_u32_u4 %0, %0, %1, %2")
// ---- SDWA / DPP
OPS(add_u32_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
OPS(sub_u16_sdwa, "v_sub_u16_sdwa %0, %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_3")
OPS(mov_b32_sdwa, "v_mov_b32_sdwa %0, %1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1")
OPS(or_b32_sdwa, "v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1")
OPS(mov_dpp_wave_shr, "v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")
OPS(mov_dpp_row_shr, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OPS(add_u32_dpp_row_shr, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
OPS(add_u32_dpp_wave_shr, "v_add_u32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf")

typedef void (*kern_t)(uint32_t *, unsigned long long *, int);

static int run(const char *name, kern_t kern, int waves_per_simd, uint32_t *out, unsigned long long *cyc, int cus)
{
    // 256-thread blocks = 4 waves = 1 wave per SIMD; waves_per_simd blocks per CU
    const int iters = 2000;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, cyc, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, cyc, iters);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    static unsigned long long h[8 * 256 * 4];
    CK(hipMemcpy(h, cyc, sizeof(unsigned long long) * grid.x * 4, hipMemcpyDeviceToHost));
    double sum = 0;
    for (unsigned i = 0; i < grid.x * 4; i++)
        sum += (double)h[i];
    const double wave_cycles = sum / (grid.x * 4);
    const double inst_per_wave = (double)iters * 64;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"cycles_per_inst_per_simd\": %.2f, \"ns_per_inst_per_simd\": %.3f}\n", name,
           waves_per_simd, wave_cycles / (inst_per_wave * waves_per_simd), ms * 1e6 / (inst_per_wave * waves_per_simd));
    CK(hipEventDestroy(a));
    CK(hipEventDestroy(b));
    return 0;
}

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    uint32_t *out;
    unsigned long long *cyc;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * 8));
    const char *only = argc > 1 ? argv[1] : nullptr;
#define R(n)                                                           \
    if (!only || strstr(#n, only))                                     \
        for (int w = 1; w <= 4; w *= 2)                                \
            if (run(#n, k_##n, w, out, cyc, cus))                      \
                return 1;
    R(add_u32) R(sub_u32) R(and_b32) R(or_b32) R(lshlrev_b32) R(lshrrev_b32) R(max_u32) R(min_u32) R(mul_u32_u24)
    R(add_u16) R(sub_u16) R(max_u16) R(mul_lo_u16) R(lshlrev_b16) R(add_f32) R(mov_b32)
    R(fma_f32) R(add3_u32) R(lshl_add_u32) R(add_lshl_u32) R(lshl_or_b32) R(and_or_b32) R(or3_b32) R(xad_u32)
    R(mad_u32_u24) R(mad_u16) R(bfe_u32) R(bfi_b32) R(alignbit_b32) R(alignbyte_b32) R(perm_b32) R(sad_u8) R(sad_u16)
    R(sad_u32) R(msad_u8) R(lerp_u8) R(max3_u32) R(med3_u32) R(add_u32_e64) R(sub_u16_clamp) R(add_u32_clamp)
    R(sub_u32_clamp) R(cvt_f32_ubyte0) R(mul_lo_u32)
    R(pk_add_u16) R(pk_sub_u16_clamp) R(pk_mad_u16) R(pk_mul_lo_u16) R(pk_max_u16) R(pk_max_i16) R(pk_sub_i16)
    R(pk_lshlrev_b16) R(pk_lshrrev_b16) R(dot4_u32_u8) R(dot2_u32_u16) R(dot8_u32_u4)
    R(add_u32_sdwa) R(sub_u16_sdwa) R(mov_b32_sdwa) R(or_b32_sdwa) R(mov_dpp_wave_shr) R(mov_dpp_row_shr)
    R(add_u32_dpp_row_shr) R(add_u32_dpp_wave_shr)
    return 0;
}
