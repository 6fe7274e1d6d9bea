// Does gfx950's raw-buffer range check include the SCALAR offset of a buffer_load?  (The attention and GEMM request paths put the
// tile / row-block offset there and rely on out-of-range -> zeros for ragged last tiles.)  Build + run on the box:
//   hipcc --offload-arch=gfx950 -O2 tools/soffset_probe.hip -o /tmp/soffset_probe && /tmp/soffset_probe
// Prints one line per case: what a dword load and a 16-byte LDS-DMA load return for an address whose voffset / soffset / sum lies past
// num_records.  The backing allocation is larger than num_records, so nothing faults either way.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint32_t* buf, int num_records, int voff, int soff, uint32_t* out) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, num_records, 0x00020000);
    __shared__ __attribute__((aligned(16))) uint32_t lds[4 * 64];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const int so = __builtin_amdgcn_readfirstlane(soff);
    out[0] = __builtin_amdgcn_raw_buffer_load_b32(rs, voff, so, 0);
    if (threadIdx.x == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, so, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) out[1] = lds[0];
}

int main() {
    const int N = 1 << 16;
    std::vector<uint32_t> h(N, 0x7f7f7f7fu);
    uint32_t *d, *o;
    hipMalloc(&d, N * 4); hipMalloc(&o, 8);
    hipMemcpy(d, h.data(), N * 4, hipMemcpyHostToDevice);
    struct { const char* what; int nr, vo, so; } cases[] = {
        {"in range (v 64, s 64, records 1024)", 1024, 64, 64},
        {"voffset past the end (v 2048, s 0)", 1024, 2048, 0},
        {"soffset past the end (v 0, s 2048)", 1024, 0, 2048},
        {"sum past the end (v 768, s 768)", 1024, 768, 768},
        {"16 B straddling the end (v 0, s 1016)", 1024, 0, 1016},
    };
    for (auto& c : cases) {
        uint32_t r[2] = {1, 1};
        hipMemset(o, 0xff, 8);
        probe<<<1, 64>>>(d, c.nr, c.vo, c.so, o);
        hipDeviceSynchronize();
        hipMemcpy(r, o, 8, hipMemcpyDeviceToHost);
        printf("%-44s dword %08x (%s)   lds-dma %08x (%s)\n", c.what, r[0], r[0] == 0 ? "ZERO: range-checked" : "data: not checked",
               r[1], r[1] == 0 ? "ZERO: range-checked" : r[1] == 0xdeadbeefu ? "not written" : "data: not checked");
    }
    return 0;
}
