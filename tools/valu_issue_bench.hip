// valu_issue_bench.hip -- how many cycles does one SIMD of gfx950 need per wave64 VALU instruction, as a function of the
// number of waves resident on it?  The denominator of bench.py's "valu_issue" roofline.
//
//   hipcc --offload-arch=gfx950 -O2 -o valu_issue_bench tools/valu_issue_bench.hip && ./valu_issue_bench > profiles/r02_valu_issue.json
//
// Every wave runs ITERS x 128 independent v_fma_f32 (8 accumulator chains, 16 rounds per trip: a chain's next fma is 8 issues
// away, well past the 4-cycle dependent latency; the loop's scalar bookkeeping and taken branch are amortised over 128 issues).
// The number of waves per SIMD is set by the grid: k x (number of CUs) workgroups of 256 threads, which the dispatcher spreads
// evenly (k workgroups per CU => k waves on each of the 4 SIMDs; the p05 / p95 spread of the per-wave cycle counts shows how even).
// (A first version pinned the residency through the dynamic LDS size instead; from 3 workgroups per CU on that did not give the
// intended residency -- the per-wave cycle counts showed two resident workgroups and a second round -- so it was dropped.)  Reported per k:
//   cyc_per_inst_smemtime   median over waves of (s_memtime delta) / (k * instructions of one wave): SIMD cycles per instruction
//   cyc_per_inst_wall       kernel wall time (HIP events) * clock / (k * instructions): the same from outside, clock = the value
//                           returned by hipDeviceGetAttribute(ClockRate); under load the chip may clock lower (DVFS), which makes
//                           this figure read HIGH -- s_memtime counts shader cycles and is the one to use.
// Also a transcendental stream (v_rcp_f32), a stream with one independent SALU instruction after every VALU one (n_inst still counts the
// VALU instructions only: the excess over the plain stream is what a scalar instruction costs the wave / the SIMD), and a 7 : 1 fma : rcp mix.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void k_issue(float *out, int iters, unsigned long long *cyc) {
    extern __shared__ float pad[];
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float b = 0.999f, c = 1e-3f;
    unsigned s0 = blockIdx.x, s1 = blockIdx.x + 1u;
    if (iters < 0) pad[threadIdx.x] = a0;                                   // keeps the LDS allocation alive
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (OP == 0) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (OP == 1) {
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (OP == 2) {                                           // every v_fma followed by an independent scalar add: does a SALU instruction cost the wave a VALU issue slot?
                asm volatile("v_fma_f32 %0, %0, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %9, %9, 1\n"
                             "v_fma_f32 %2, %2, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %3, %3, %10, %11\n s_add_u32 %9, %9, 1\n"
                             "v_fma_f32 %4, %4, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %5, %5, %10, %11\n s_add_u32 %9, %9, 1\n"
                             "v_fma_f32 %6, %6, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %9, %9, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(b), "v"(c) : "scc");
            } else if (OP == 4) {                                           // 2 v_fma per s_add: separates "a scalar instruction takes an issue slot" (2 + 2/2 = 3 cycles per VALU)
                asm volatile("v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11\n s_add_u32 %9, %9, 1\n"
                             "v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n s_add_u32 %8, %8, 1\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %9, %9, 1\n"   // from "the CU's one scalar ALU is the limit" (max(2, 4/2) = 2)
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(b), "v"(c) : "scc");
            } else if (OP == 5) {                                           // 4 v_fma per s_add (2.5 against 2)
                asm volatile("v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n v_fma_f32 %3, %3, %10, %11\n s_add_u32 %8, %8, 1\n"
                             "v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n v_fma_f32 %6, %6, %10, %11\n v_fma_f32 %7, %7, %10, %11\n s_add_u32 %9, %9, 1\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : "v"(b), "v"(c) : "scc");
            } else {                                                        // the forward kernel's transcendental share: 1 v_rcp per 7 v_fma
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rcp_f32 %3, %3\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float) (s0 + s1);
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
    int dev = 0; CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount, clock_khz = prop.clockRate;
    const int iters = 1 << 13;                                              // 8192 x 128 = 1 048 576 instructions per wave
    const double n_inst = (double) iters * 128.0;
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"instructions_per_wave\": %.0f, \"results\": [", prop.name, prop.gcnArchName, cus, clock_khz, n_inst);
    bool first = true;
    for (int op = 0; op < 6; ++op)
        for (int k : { 1, 2, 3, 4, 5, 6, 8 }) {
            const int blocks = cus * k;
            const size_t lds = 0;
            float *out; unsigned long long *cyc;
            CHECK(hipMalloc((void **) &out, (size_t) blocks * 256 * 4)); CHECK(hipMalloc((void **) &cyc, (size_t) blocks * 4 * 8));
            auto kern = op == 0 ? k_issue<0> : (op == 1 ? k_issue<1> : (op == 2 ? k_issue<2> : (op == 3 ? k_issue<3> : (op == 4 ? k_issue<4> : k_issue<5>))));
            CHECK(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds));
            hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters, cyc);   // warm-up
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, out, iters, cyc);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h((size_t) blocks * 4);
            CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::sort(h.begin(), h.end());
            const double med = (double) h[h.size() / 2], lo = (double) h[h.size() / 20], hi = (double) h[h.size() - 1 - h.size() / 20];
            printf("%s\n  {\"op\": \"%s\", \"waves_per_simd\": %d, \"workgroups\": %d, \"kernel_ms\": %.4f, \"cyc_per_inst_smemtime\": %.3f, \"p05\": %.3f, \"p95\": %.3f, \"cyc_per_inst_wall\": %.3f}",
                   first ? "" : ",", op == 0 ? "v_fma_f32" : (op == 1 ? "v_rcp_f32" : (op == 2 ? "v_fma_f32 + s_add_u32 (per VALU instruction)" : (op == 3 ? "7 v_fma_f32 + 1 v_rcp_f32 (per instruction)" : (op == 4 ? "2 v_fma_f32 + 1 s_add_u32 (per VALU instruction)" : "4 v_fma_f32 + 1 s_add_u32 (per VALU instruction)")))), k, blocks, ms, med / (k * n_inst), lo / (k * n_inst), hi / (k * n_inst),
                   ms * 1e-3 * clock_khz * 1e3 / (k * n_inst));
            first = false;
            CHECK(hipFree(out)); CHECK(hipFree(cyc));
        }
    printf("\n]}\n");
    return 0;
}
