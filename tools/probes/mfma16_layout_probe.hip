// Operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 on gfx950, found by experiment (one lane of ones at a time): which output
// (lane, register) positions see a given A lane / B lane, and which A lanes contract with which B lanes.
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma16_probe tools/probes/mfma16_layout_probe.hip && /tmp/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

template <int CBSZ>
__global__ void run(const int* a, const int* b, float* o) {      // a, b: [case][64 lanes][8 dwords]; o: [case][64][4]
    const int c = blockIdx.x, l = threadIdx.x;
    i32x8_t A, B;
    for (int i = 0; i < 8; ++i) { A[i] = a[(c * 64 + l) * 8 + i]; B[i] = b[(c * 64 + l) * 8 + i]; }
    f32x4_t z = {0.f, 0.f, 0.f, 0.f};
    f32x4_t d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, z, CBSZ, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 4; ++i) o[(c * 64 + l) * 4 + i] = d[i];
}

int main() {
    const int NC = 64 * 3;
    std::vector<int> a(NC * 64 * 8), b(NC * 64 * 8);
    const int ONE8 = 0x38383838;
    for (int c = 0; c < NC; ++c)
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 8; ++i) {
                const int kind = c / 64, L0 = c % 64;
                int av = ONE8, bv = ONE8;
                if (kind == 0) bv = (l == L0) ? ONE8 : 0;             // one B lane: which outputs see it
                if (kind == 1) av = (l == L0) ? ONE8 : 0;             // one A lane: which outputs see it
                if (kind == 2) { av = (l == L0) ? ONE8 : 0; bv = (l / 16 == 0) ? ONE8 : ((l / 16 == 1) ? 0x40404040 : ((l / 16 == 2) ? 0x48484848 : 0x50505050)); }   // B group g carries 2^g
                a[(c * 64 + l) * 8 + i] = av; b[(c * 64 + l) * 8 + i] = bv;
            }
    int *da, *db; float* dout;
    hipMalloc(&da, a.size() * 4); hipMalloc(&db, b.size() * 4); hipMalloc(&dout, NC * 64 * 4 * 4);
    hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(run<0>, dim3(NC), dim3(64), 0, 0, da, db, dout);
    std::vector<float> o(NC * 64 * 4);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    auto at = [&](int c, int l, int r) { return o[(c * 64 + l) * 4 + r]; };
    printf("one B lane L0 of ones (A all ones): output lanes that are non-zero (value)\n");
    for (int L0 = 0; L0 < 64; L0 += 1) {
        printf("  B lane %2d ->", L0);
        for (int l = 0; l < 64; ++l) if (at(L0, l, 0) != 0.f) printf(" %d", l);
        printf("   (= %g)\n", at(L0, L0 % 16, 0));
        if (L0 == 3) L0 = 14; if (L0 == 18) L0 = 30; if (L0 == 34) L0 = 46; if (L0 == 50) L0 = 61;
    }
    printf("one A lane L0 of ones (B all ones): output (lane, reg) that are non-zero, lanes 0..63 with reg list\n");
    for (int L0 = 0; L0 < 64; ++L0) {
        printf("  A lane %2d ->", L0);
        int first = -1, cnt = 0; 
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (at(64 + L0, l, r) != 0.f) { if (first < 0) first = l * 4 + r; ++cnt; }
        printf(" first (lane %d, reg %d), %d positions, value %g\n", first / 4, first % 4, cnt, first >= 0 ? o[((64 + L0) * 64) * 4 + first] : 0.f);
        if (L0 == 3) L0 = 14; if (L0 == 18) L0 = 30; if (L0 == 34) L0 = 46; if (L0 == 50) L0 = 61;
    }
    printf("one A lane L0 of ones, B lane group g = lane / 16 carries 2^g: value / 32 tells which B group that A lane contracts with\n");
    for (int L0 = 0; L0 < 64; L0 += 5) {
        float v = 0.f;
        for (int l = 0; l < 64 && v == 0.f; ++l) for (int r = 0; r < 4; ++r) if (at(128 + L0, l, r) != 0.f) { v = at(128 + L0, l, r); break; }
        printf("  A lane %2d (group %d): %g\n", L0, L0 / 16, v / 32);
    }
    return 0;
}
