// Probe: operand lane maps of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands (cdna_hip_programming.md section 3:
// "Other dtypes: check the map with exact integer data before relying on it").  Prints which hypothesis reproduces
// D = A * B exactly for small-integer matrices, and what the E8M0 scale operands do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int v8i;
typedef __attribute__((ext_vector_type(4))) float v4f;

__global__ void probe(const v8i* A, const v8i* B, v4f* D, int sa, int sb) {
    const int l = threadIdx.x;
    v4f c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A[l], B[l], c, 0, 0, 0, sa, 0, sb);
    D[l] = c;
}

static uint8_t e4m3(int v) {          // small integers -4..4 exactly
    static const uint8_t pos[5] = {0x00, 0x38, 0x40, 0x44, 0x48};   // 0,1,2,3,4
    return v >= 0 ? pos[v] : (uint8_t)(pos[-v] | 0x80);
}

int main() {
    int Am[16][128], Bm[128][16];
    srand(7);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) Am[i][k] = rand() % 7 - 3;
    for (int k = 0; k < 128; ++k) for (int j = 0; j < 16; ++j) Bm[k][j] = rand() % 7 - 3;
    float ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int s = 0; for (int k = 0; k < 128; ++k) s += Am[i][k] * Bm[k][j]; ref[i][j] = (float)s; }
    v8i *dA, *dB; v4f* dD;
    hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dD, 64 * 16);
    const char* names[3] = {"H1 k = 32*(l>>4) + p", "H2 k = 16*(l>>4) + (p&15) + 64*(p>>4)", "H3 k = 8*(l>>4) + (p&7) + 32*(p>>3)"};
    for (int h = 0; h < 3; ++h) {
        std::vector<uint8_t> a(64 * 32), b(64 * 32);
        for (int l = 0; l < 64; ++l) for (int p = 0; p < 32; ++p) {
            int k = h == 0 ? 32 * (l >> 4) + p : h == 1 ? 16 * (l >> 4) + (p & 15) + 64 * (p >> 4) : 8 * (l >> 4) + (p & 7) + 32 * (p >> 3);
            a[l * 32 + p] = e4m3(Am[l & 15][k]);
            b[l * 32 + p] = e4m3(Bm[k][l & 15]);
        }
        hipMemcpy(dA, a.data(), 64 * 32, hipMemcpyHostToDevice); hipMemcpy(dB, b.data(), 64 * 32, hipMemcpyHostToDevice);
        for (int sc = 0; sc < 3; ++sc) {
            int sa = sc == 0 ? 127 : (sc == 1 ? 128 : 127), sb = sc == 2 ? 126 : 127;
            // replicate the scale byte in all four bytes (opsel picks one)
            sa *= 0x01010101; sb *= 0x01010101;
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb);
            float out[64][4];
            hipMemcpy(out, dD, sizeof(out), hipMemcpyDeviceToHost);
            float expect = sc == 0 ? 1.f : (sc == 1 ? 2.f : 0.5f);
            int bad = 0;
            for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
                const int row = (l >> 4) * 4 + r, col = l & 15;
                if (out[l][r] != ref[row][col] * expect) ++bad;
            }
            printf("%s scale(a=%d,b=%d) expect x%.1f: %s (%d mismatches) sample D[0][0]=%g ref=%g\n", names[h], sa & 255, sb & 255, expect,
                   bad ? "NO" : "MATCH", bad, out[0][0], ref[0][0]);
        }
    }
    return 0;
}
