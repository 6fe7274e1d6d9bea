// Probe: operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 (e4m3) A/B and unit E8M0 block scales.
// Hypothesis H1: lane l supplies A[row = l & 31][k = 32*(l >> 5) .. +31] as 32 consecutive bytes (8 VGPRs), B likewise with
// col = l & 31; C/D uses the standard 32x32 map (col = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)).
// H2: k interleaved in 16-byte halves: lane-half h holds k = 16*h..16*h+15 and 32+16*h..+15.
// Prints the max abs error of D against the exact product under each hypothesis (inputs are small exactly-representable values).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int hyp) {
    const int l = threadIdx.x, i = l & 31, h = l >> 5;
    v8i a, b;
    uint8_t ab[32], bb[32];
    for (int j = 0; j < 32; ++j) {
        int k = hyp == 1 ? 32 * h + j : (j < 16 ? 16 * h + j : 32 + 16 * h + (j - 16));
        ab[j] = A[i * 64 + k];
        bb[j] = B[i * 64 + k];      // B stored [col][k]
    }
    for (int j = 0; j < 8; ++j) {
        a[j] = ab[4 * j] | (ab[4 * j + 1] << 8) | (ab[4 * j + 2] << 16) | (ab[4 * j + 3] << 24);
        b[j] = bb[4 * j] | (bb[4 * j + 1] << 8) | (bb[4 * j + 2] << 16) | (bb[4 * j + 3] << 24);
    }
    v16f c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        D[row * 32 + i] = c[r];
    }
}

static float e4m3_to_f(uint8_t v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -x : x;
}

int main() {
    uint8_t hA[32 * 64], hB[32 * 64];
    uint32_t s = 12345;
    for (int i = 0; i < 32 * 64; ++i) {
        s = s * 1664525u + 1013904223u; hA[i] = (uint8_t)(((s >> 16) & 0x87) | 0x30 | ((s >> 8) & 0x08));   // exponents 6..7, 3-bit mantissa
        s = s * 1664525u + 1013904223u; hB[i] = (uint8_t)(((s >> 16) & 0x87) | 0x38);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    for (int hyp = 1; hyp <= 2; ++hyp) {
        float hD[32 * 32];
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
        hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < 64; ++k) ref += (double)e4m3_to_f(hA[i * 64 + k]) * e4m3_to_f(hB[j * 64 + k]);
            maxerr = fmax(maxerr, fabs(ref - hD[i * 32 + j])); maxref = fmax(maxref, fabs(ref));
        }
        printf("mfma_scale_probe hypothesis %d: max |D - exact| = %.4g (max |exact| = %.4g) %s\n", hyp, maxerr, maxref,
               maxerr < 1e-3 * maxref ? "MATCH" : "no");
    }
    return 0;
}
