// valu_issue_bench.hip — what one SIMD of MI355X (gfx950) issues per cycle for the instruction kinds the raster kernel
// is made of, at the raster kernel's own occupancy (768-thread workgroups, 2 per CU = 6 waves per SIMD) and at 1..8
// waves per SIMD.  Settles the "2 or 4 cycles per wave64 VALU instruction" question the round-1 roofline left open.
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_issue_bench.hip -o gpurun_out/valu_issue_bench && gpurun_out/valu_issue_bench
//
// Every kernel runs ITER x UNROLL independent instructions of one kind per wave on private registers (eight accumulators,
// so no dependent-issue stalls), stamped with s_memtime on both sides; cycles per wave-instruction per SIMD =
// (cycles of the slowest wave) * waves_per_simd_resident / (waves per SIMD * instructions per wave) ... measured simply as
// total SIMD-cycles / total wave-instructions: wall cycles of a workgroup set that fills every SIMD evenly.
// Round 3: every wave also stamps s_memrealtime (a constant 100 MHz counter) beside s_memtime (shader cycles), so each figure comes
// with the clock the chip held WHILE it was measured: clock = delta s_memtime / delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md,
// "DVFS give-back" item 6).  A dense all-SIMD VALU load pulls the clock far below the 2.4 GHz of the data sheet, and a kernel that
// waits half of the time runs at a higher one: an issue peak in instructions per SECOND is only comparable at the same clock, so
// the comparable figure is shader CYCLES per wave-instruction, and the peak of another kernel = SIMDs x its own clock / that.
// Output: one JSON object on stdout (committed as profiles/r03_valu_issue.json).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

constexpr int ITER = 8192, UNROLL = 8;      // ~0.3-1 ms per launch: the launch itself is under 2 % of the wall time

enum Kind { K_FMA_F32 = 0, K_ADD_U32, K_MUL24, K_MUL_LO_U32, K_MAD_U64_U32, K_CNDMASK, K_RCP_F32, K_CVT_F32_I32, K_MIN3_I32, K_PK_FMA_F32, K_ADD_U64, K_MIX, N_KINDS };
static const char *kind_name[N_KINDS] = {"v_fma_f32", "v_add_u32", "v_mul_i32_i24", "v_mul_lo_u32", "v_mad_u64_u32", "v_cndmask_b32", "v_rcp_f32",
                                         "v_cvt_f32_i32", "v_min3_i32", "v_pk_fma_f32", "u64 add (2 x v_add_co)", "raster-like mix (fma, add, mul24, cndmask, min3, cvt; no transcendental)"};

template <int KIND>
__global__ void __launch_bounds__(1024, 8) issue_kernel(uint64_t *cycles, uint64_t *ticks, uint32_t *sink, uint32_t seed)
{
    uint32_t a[UNROLL];
    float f[UNROLL];
    uint64_t q[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; k++) { a[k] = seed + threadIdx.x * 7 + k; f[k] = (float)(a[k] & 1023) * 1.0e-3f + 1.0f; q[k] = ((uint64_t)a[k] << 20) | k; }
    const float fb = (float)(seed & 7) * 1.0e-7f + 1.0f, fc = (float)(seed & 3) * 1.0e-9f;
    const uint32_t ub = seed | 1u;
    const uint64_t sel = 0x5555AAAA3333CCCCull ^ seed;       // lane-select mask of the v_cndmask rows, in an SGPR pair (no VCC hazard nops)
    uint64_t carry = 0;
    __syncthreads();
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            if (KIND == K_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fb), "v"(fc));
            else if (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(ub));
            else if (KIND == K_MUL24) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[k]) : "v"(ub));
            else if (KIND == K_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[k]) : "v"(ub));
            else if (KIND == K_MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(q[k]), "=s"(carry) : "v"(a[k]), "v"(ub));
            else if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(ub), "s"(sel));
            else if (KIND == K_RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
            else if (KIND == K_CVT_F32_I32) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[k]) : "v"(a[k]));
            else if (KIND == K_MIN3_I32) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(ub), "v"(seed));
            else if (KIND == K_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(q[k]) : "v"(q[(k + 1) % UNROLL]));
            else if (KIND == K_ADD_U64) asm volatile("v_add_co_u32 %0, %2, %0, %3\n v_addc_co_u32 %1, %2, %1, 0, %2" : "+v"(a[k]), "+v"(a[(k + 4) % UNROLL]), "=&s"(carry) : "v"(ub));
            else {
                // eight instructions of the kinds the cull / set-up code is made of
                switch (k) {
                case 0: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[0]) : "v"(fb), "v"(fc)); break;
                case 1: asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[1]) : "v"(ub)); break;
                case 2: asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(a[2]) : "v"(ub)); break;
                case 3: asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[3]) : "v"(ub), "s"(sel)); break;
                case 4: asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a[4]) : "v"(ub), "v"(seed)); break;
                case 5: asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[5]) : "v"(a[5])); break;
                case 6: asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[6]) : "v"(ub)); break;
                default: asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[7]) : "v"(fb)); break;
                }
            }
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < UNROLL; k++) acc ^= a[k] ^ __float_as_uint(f[k]) ^ (uint32_t)q[k] ^ (uint32_t)(q[k] >> 32);
    if (acc == 0x12345678u) sink[0] = acc ^ (uint32_t)carry;                                   // keeps the registers alive
    if ((threadIdx.x & 63) == 0) {
        cycles[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
        ticks[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r1 - r0;          // 10 ns each
    }
}

template <int KIND>
static double run(int waves_per_simd, uint64_t *d_cycles, uint32_t *d_sink, std::vector<uint64_t> &h, double *wall_ms, double *clock_ghz)
{
    uint64_t *d_ticks = d_cycles + (1 << 16);
    // one workgroup per CU holding 4 * waves_per_simd waves, except the raster kernel's own shape for 6: 2 x 768 threads per CU
    const int n_cu = 256;
    int threads = 256 * waves_per_simd, blocks = n_cu;
    if (waves_per_simd == 6) { threads = 768; blocks = 2 * n_cu; }
    if (threads > 1024) { threads /= 2; blocks *= 2; }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // warm-up: long enough for the clock to settle under this load before the measured launch (the governor reacts in milliseconds)
    for (int w = 0; w < 40; w++) hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_ticks, d_sink, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(issue_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, d_cycles, d_ticks, d_sink, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    *wall_ms = ms;
    const int n_waves = blocks * (threads / 64);
    CHECK(hipMemcpy(h.data(), d_cycles, n_waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    std::vector<uint64_t> hr(n_waves);
    CHECK(hipMemcpy(hr.data(), d_ticks, n_waves * sizeof(uint64_t), hipMemcpyDeviceToHost));
    double mean = 0, real = 0;
    for (int i = 0; i < n_waves; i++) { mean += (double)h[i]; real += (double)hr[i]; }
    mean /= n_waves;
    real /= n_waves;
    *clock_ghz = real > 0 ? mean / (real * 10.0) : 0.0;                       // shader cycles per nanosecond while the loop ran
    // s_memtime counts at a fixed 100 MHz-derived rate?  No: on gfx950 it counts shader clock cycles (MI355X_MICROARCH.md).
    // A wave's stream of N instructions took `mean` cycles while waves_per_simd waves shared its SIMD:
    const double n_inst = (double)ITER * UNROLL * (KIND == K_ADD_U64 ? 2 : 1);
    return mean / (n_inst * waves_per_simd);                                  // SIMD cycles per wave-instruction
}

template <int KIND>
static void sweep(uint64_t *d_cycles, uint32_t *d_sink, std::vector<uint64_t> &h, bool last)
{
    printf("  {\"instruction\": \"%s\", \"wave_stamp_cycles_over_assumed_residency\": {", kind_name[KIND]);
    const int wps[] = {1, 2, 4, 6, 8};
    double wall[5], clk[5];
    for (int i = 0; i < 5; i++) {
        double ms;
        const double c = run<KIND>(wps[i], d_cycles, d_sink, h, &ms, &clk[i]);
        printf("\"%d\": %.3f%s", wps[i], c, i < 4 ? ", " : "");
        wall[i] = ms;
    }
    printf("}, \"clock_ghz\": {");
    for (int i = 0; i < 5; i++) printf("\"%d\": %.3f%s", wps[i], clk[i], i < 4 ? ", " : "");
    // THE figure: SIMD cycles per wave-instruction = wall time of the launch x the clock held inside it / instructions per SIMD.
    // (Round 2's column divided each wave's own stamp interval by the number of waves LAUNCHED per SIMD; where fewer were resident
    // at a time it read low by that factor.)
    printf("}, \"cycles_per_wave_instruction_per_simd\": {");
    for (int i = 0; i < 5; i++)
        printf("\"%d\": %.3f%s", wps[i], wall[i] * 1.0e6 / ((double)ITER * UNROLL * (KIND == K_ADD_U64 ? 2 : 1) * wps[i]) * clk[i], i < 4 ? ", " : "");
    // the same from the wall clock (HIP events around the launch): ns per wave-instruction per SIMD, launch overhead included
    printf("}, \"wall_ns_per_wave_instruction_per_simd\": {");
    for (int i = 0; i < 5; i++)
        printf("\"%d\": %.3f%s", wps[i], wall[i] * 1.0e6 / ((double)ITER * UNROLL * (KIND == K_ADD_U64 ? 2 : 1) * wps[i]), i < 4 ? ", " : "");
    printf("}}%s\n", last ? "" : ",");
}

int main()
{
    uint64_t *d_cycles;
    uint32_t *d_sink;
    CHECK(hipMalloc((void **)&d_cycles, 1 << 20));              // first half: s_memtime deltas, second half: s_memrealtime deltas
    CHECK(hipMalloc((void **)&d_sink, 64));
    std::vector<uint64_t> h(1 << 17);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iter\": %d, \"unroll\": %d,\n \"note\": \"SIMD cycles (s_memtime) per wave64 instruction with N waves resident per SIMD, "
           "independent instructions; 6 = two 768-thread workgroups per CU, the raster kernel's occupancy; clock_ghz = delta s_memtime / delta s_memrealtime "
           "(100 MHz) inside the measured loop, after 40 warm-up launches of the same load\",\n \"kinds\": [\n",
           prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, ITER, UNROLL);
    sweep<K_FMA_F32>(d_cycles, d_sink, h, false);
    sweep<K_ADD_U32>(d_cycles, d_sink, h, false);
    sweep<K_MUL24>(d_cycles, d_sink, h, false);
    sweep<K_MUL_LO_U32>(d_cycles, d_sink, h, false);
    sweep<K_MAD_U64_U32>(d_cycles, d_sink, h, false);
    sweep<K_CNDMASK>(d_cycles, d_sink, h, false);
    sweep<K_RCP_F32>(d_cycles, d_sink, h, false);
    sweep<K_CVT_F32_I32>(d_cycles, d_sink, h, false);
    sweep<K_MIN3_I32>(d_cycles, d_sink, h, false);
    sweep<K_PK_FMA_F32>(d_cycles, d_sink, h, false);
    sweep<K_ADD_U64>(d_cycles, d_sink, h, false);
    sweep<K_MIX>(d_cycles, d_sink, h, true);
    printf(" ]}\n");
    return 0;
}
