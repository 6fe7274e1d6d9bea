// Probe: per-CU ingest rate of LDS-DMA (global_load_lds_dwordx4) vs plain global_load_dwordx4 -> VGPR, from an
// L2-resident source, with and without concurrent ds_read_b128 traffic.  Build: hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define FW_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define FW_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define GLDS16(g, l) __builtin_amdgcn_global_load_lds(FW_GLB_PTR(g), FW_LDS_PTR(l), 16, 0, 0)

// MODE 0: DMA only; 1: DMA + ds_read (3 KiB read per KiB written, as the GEMM); 2: global_load -> VGPR only; 3: ds_read only
template <int MODE, int ROWB>
__global__ __launch_bounds__(512, 2) void probe(const char* __restrict__ src, size_t region, int iters, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 32768];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = src + (size_t)(blockIdx.x % 64) * region;      // 64 regions x `region` bytes: L2/MALL resident
    // lane -> source: rows of ROWB bytes at a 10240-byte row stride (like a K=5120 bf16 matrix)
    const int lanes_per_row = ROWB / 16;
    const unsigned loff = (unsigned)((lane / lanes_per_row) * 10240 + (lane % lanes_per_row) * 16);
    u32x4_t acc = {0, 0, 0, 0};
    int rs = 0;
    for (int it = 0; it < iters; ++it) {
        const char* g = base + (size_t)(it & 15) * ROWB + (size_t)wave * (64 / lanes_per_row) * 4 * 10240;
        if (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                GLDS16(g + (size_t)i * (64 / lanes_per_row) * 10240 + loff, smem + rs * 32768 + (wave * 4 + i) * 1024);
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                u32x4_t v = *(const u32x4_t*)(g + (size_t)i * (64 / lanes_per_row) * 10240 + loff);
                acc ^= v;
            }
        }
        if (MODE == 1 || MODE == 3) {
            const char* b = smem + ((rs + 2) % 5) * 32768;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                u32x4_t v = *(const u32x4_t*)(b + ((lane * 16 + i * 2048 + wave * 1024) & 32767));
                acc ^= v;
            }
        }
        if (MODE == 0 || MODE == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        rs = rs == 4 ? 0 : rs + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc[0] == 0x12345678u) sink[0] = acc[1] ^ acc[2] ^ acc[3];
}

template <int MODE, int ROWB>
void run(const char* name, const char* src, size_t region, unsigned* sink, double bytes_per_iter_dma, double bytes_per_iter_lds) {
    const int iters = 4000, grid = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((probe<MODE, ROWB>), dim3(grid), dim3(512), 0, 0, src, region, 200, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<MODE, ROWB>), dim3(grid), dim3(512), 0, 0, src, region, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double t = ms * 1e-3;
    printf("%-44s %8.3f ms  global->CU %7.1f GB/s/CU (%6.2f TB/s chip)   LDS reads %7.1f GB/s/CU\n", name, ms,
           bytes_per_iter_dma * iters / t / 1e9, bytes_per_iter_dma * iters * grid / t / 1e12, bytes_per_iter_lds * iters / t / 1e9);
}

int main() {
    const size_t region = 16 * 10240 * 64;   // 64 rows-groups... 10.5 MB per region? keep it modest
    const size_t total = 64 * region;
    char* src; hipMalloc(&src, total + (1 << 20)); hipMemset(src, 1, total + (1 << 20));
    unsigned* sink; hipMalloc(&sink, 64);
    // per iteration per CU (one 512-thread group per CU): 8 waves x 4 x 1 KiB = 32 KiB global; reads 8 x 12 KiB = 96 KiB
    run<0, 64>("DMA only, 64-B rows", src, region, sink, 32768, 0);
    run<0, 128>("DMA only, 128-B rows", src, region, sink, 32768, 0);
    run<0, 256>("DMA only, 256-B rows", src, region, sink, 32768, 0);
    run<1, 64>("DMA 64-B rows + ds_read_b128 (3x bytes)", src, region, sink, 32768, 98304);
    run<1, 128>("DMA 128-B rows + ds_read_b128 (3x bytes)", src, region, sink, 32768, 98304);
    run<2, 64>("global_load_dwordx4 -> VGPR, 64-B rows", src, region, sink, 32768, 0);
    run<2, 128>("global_load_dwordx4 -> VGPR, 128-B rows", src, region, sink, 32768, 0);
    run<3, 64>("ds_read_b128 only", src, region, sink, 0, 98304);
    return 0;
}
