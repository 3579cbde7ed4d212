#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef double d4 __attribute__((ext_vector_type(4)));
// D[i][j] = sum_k A[i][k] * B[j][k], 16x16, K multiple of 4.  lane l: A[l&15][k = 4s + (l>>4)], B[l&15][k = 4s + (l>>4)]
__global__ void probe(const double* A, const double* B, double* D, int K)
{
    const int l = threadIdx.x;
    d4 acc = {0, 0, 0, 0};
    for (int s = 0; s < K / 4; s++) {
        const double a = A[(l & 15) * K + 4 * s + (l >> 4)];
        const double b = B[(l & 15) * K + 4 * s + (l >> 4)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    // C/D: col = lane & 15, row = (lane >> 4) + 4 * reg
    for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}
int main()
{
    const int K = 256;
    double *hA = new double[16 * K], *hB = new double[16 * K], hD[256];
    srand(3);
    for (int i = 0; i < 16 * K; i++) { hA[i] = (rand() / (double)RAND_MAX - 0.5) * exp((rand() % 40) - 20.0); hB[i] = (rand() / (double)RAND_MAX - 0.5); }
    double *dA, *dB, *dD;
    hipMalloc(&dA, 16 * K * 8); hipMalloc(&dB, 16 * K * 8); hipMalloc(&dD, 256 * 8);
    hipMemcpy(dA, hA, 16 * K * 8, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 16 * K * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(hD, dD, 256 * 8, hipMemcpyDeviceToHost);
    int bad_chain = 0, bad_pair = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) {
        double c = 0; for (int k = 0; k < K; k++) c = fma(hA[i * K + k], hB[j * K + k], c);
        // D[item i][user j] in my orientation: a = A rows (i), b = B rows (j)
        if (memcmp(&c, &hD[i * 16 + j], 8)) bad_chain++;
    }
    printf("f64 mfma vs k-ordered fma chain: %d of 256 differ\n", bad_chain);
    if (bad_chain) { // show a sample
        int i = 0, j = 0; double c = 0; for (int k = 0; k < K; k++) c = fma(hA[i * K + k], hB[j * K + k], c);
        printf("sample: chain %.17g mfma %.17g transposed? %.17g\n", c, hD[0], hD[0]);
    }
    return 0;
}
