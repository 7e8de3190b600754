// valu_rate -- measured issue cost (cycles per wave64 instruction per SIMD) of the VALU ops the K2/K3 kernels
// are built from, at 1..4 waves per SIMD.  Sets the VALU-side ceiling quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define REP8(x) x x x x x x x x
#define OPS(name, body)                                                                                     \
    __global__ __launch_bounds__(256) void k_##name(uint32_t *out, int iters)                                \
    {                                                                                                       \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                 a7 = a0 + 7, b = blockIdx.x | 1, c = 0x00030005;                                            \
        for (int i = 0; i < iters; i++) {                                                                   \
            REP8(body)                                                                                      \
        }                                                                                                   \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                 \
    }
// 8 independent chains, 8 instructions per body, REP8 -> 64 instructions per loop iteration
#define B1(op) asm volatile(op " %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a1) : "v"(b)); \
    asm volatile(op " %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a3) : "v"(b)); \
    asm volatile(op " %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a5) : "v"(b)); \
    asm volatile(op " %0, %0, %1" : "+v"(a6) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a7) : "v"(b));
#define B3(op, tail) asm volatile(op " %0, %0, %1, %2" tail : "+v"(a0) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a1) : "v"(b), "v"(c)); \
    asm volatile(op " %0, %0, %1, %2" tail : "+v"(a2) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a3) : "v"(b), "v"(c)); \
    asm volatile(op " %0, %0, %1, %2" tail : "+v"(a4) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a5) : "v"(b), "v"(c)); \
    asm volatile(op " %0, %0, %1, %2" tail : "+v"(a6) : "v"(b), "v"(c)); asm volatile(op " %0, %0, %1, %2" tail : "+v"(a7) : "v"(b), "v"(c));
#define BDPP asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a0) : "v"(a1)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a1) : "v"(a2)); \
    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a2) : "v"(a3)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a3) : "v"(a4)); \
    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a4) : "v"(a5)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a5) : "v"(a6)); \
    asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a6) : "v"(a7)); asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a7) : "v"(a0));

OPS(add_u32, B1("v_add_u32"))
OPS(pk_sub_u16_clamp, asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a0) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a1) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a2) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a3) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a4) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a5) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a6) : "v"(b)); asm volatile("v_pk_sub_u16 %0, %0, %1 clamp" : "+v"(a7) : "v"(b));)
OPS(pk_mad_u16, B3("v_pk_mad_u16", ""))
OPS(perm_b32, B3("v_perm_b32", ""))
OPS(alignbit_b32, asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a0) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a1) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a2) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a3) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a4) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a5) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a6) : "v"(b)); asm volatile("v_alignbit_b32 %0, %0, %1, 16" : "+v"(a7) : "v"(b));)
OPS(lshl_add_u32, asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a0) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a1) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a2) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a3) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a4) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a5) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a6) : "v"(b)); asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a7) : "v"(b));)
OPS(add3_u32, B3("v_add3_u32", ""))
OPS(mov_dpp_wave_shr, BDPP)

template <typename K>
int run(const char *name, K kern, int waves_per_simd, uint32_t *out, int cus)
{
    // 256-thread blocks = 4 waves = 1 wave per SIMD; waves_per_simd blocks per CU
    int iters = 4000;
    dim3 grid(cus * waves_per_simd), block(256);
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, iters);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    double inst_per_simd = (double)iters * 64 * waves_per_simd;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ns_per_inst_per_simd\": %.3f, \"cycles_at_2.4GHz\": %.2f}\n", name,
           waves_per_simd, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    int cus = p.multiProcessorCount;
    uint32_t *out;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    for (int w = 1; w <= 4; w++) {
        run("v_add_u32", k_add_u32, w, out, cus);
        run("v_pk_sub_u16_clamp", k_pk_sub_u16_clamp, w, out, cus);
        run("v_pk_mad_u16", k_pk_mad_u16, w, out, cus);
        run("v_perm_b32", k_perm_b32, w, out, cus);
        run("v_alignbit_b32", k_alignbit_b32, w, out, cus);
        run("v_lshl_add_u32", k_lshl_add_u32, w, out, cus);
        run("v_add3_u32", k_add3_u32, w, out, cus);
        run("v_mov_b32_dpp", k_mov_dpp_wave_shr, w, out, cus);
    }
    return 0;
}
