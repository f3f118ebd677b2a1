// inst_rate.hip -- issue rate of the integer instructions the k2u kernels lean on (runs ON the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -o build/inst_rate tools/inst_rate.hip && ./build/inst_rate
// Every kernel runs kIter rounds of kChains independent chains of ONE operation per lane, at 8 waves per SIMD, so the
// figure printed is the steady-state cost of that operation in SIMD cycles per wavefront instruction (4 = full rate).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr int kIter = 4096, kChains = 8;

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(uint32_t *out, uint32_t a0, uint32_t b0) {
    uint32_t x[kChains];
    uint64_t y[kChains];
    for (int c = 0; c < kChains; ++c) { x[c] = a0 + threadIdx.x * 2654435761u + c; y[c] = ((uint64_t)x[c] << 32) | (b0 + c); }
    const uint32_t m = b0 | 1u;
    for (int i = 0; i < kIter; ++i) {
#pragma unroll
        for (int c = 0; c < kChains; ++c) {
            if (OP == 0) x[c] = x[c] ^ (x[c] >> 7);                                            // v_lshrrev_b32 + v_xor (2 instr)
            if (OP == 1) x[c] = x[c] * m;                                                       // v_mul_lo_u32
            if (OP == 2) x[c] = __umulhi(x[c], m);                                              // v_mul_hi_u32
            if (OP == 3) y[c] = (uint64_t)(uint32_t)y[c] * (uint64_t)m + (y[c] >> 32);          // v_mad_u64_u32 (+ mov)
            if (OP == 4) x[c] = __umul24(x[c], m) + 1;               // v_mad_u32_u24 / v_mul_u32_u24
            if (OP == 5) y[c] = (y[c] >> (m & 31)) + 1;                                         // v_lshrrev_b64 + add
            if (OP == 6) y[c] = y[c] + (uint64_t)m;                                             // 64-bit add
            if (OP == 7) x[c] = __builtin_amdgcn_alignbit(x[c], x[c], 13) + m;                  // rotate + add
            if (OP == 8) x[c] = (uint32_t)__popcll(y[c] + x[c]);                                // 64-bit add + 2 bcnt
            if (OP == 9) x[c] = __builtin_amdgcn_perm(x[c], m, 0x00010203u) + 1;                // v_perm_b32 + add
            if (OP == 10) x[c] = __builtin_amdgcn_sad_u16(x[c], m, x[c]);                       // v_sad_u16
            if (OP == 11) x[c] = x[c] * m + (x[c] >> 15);                                       // mul_lo + shift/add
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < kChains; ++c) s += x[c] + (uint32_t)y[c] + (uint32_t)(y[c] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int instr_per_op, uint32_t *out) {
    const int n_wg = 256 * 8; // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    rate_kernel<OP><<<n_wg, 256>>>(out, 12345u, 0x9E3779B1u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    rate_kernel<OP><<<n_wg, 256>>>(out, 12345u, 0x9E3779B1u);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double ops_per_simd = (double)n_wg * 4 / 1024.0 * kIter * kChains; // wave-level operations issued per SIMD
    const double cycles = ms * 1e-3 * 2.4e9;
    printf("%-28s %8.3f ms  %6.2f cycles per wave-op (%d instr per op as written)\n", name, ms, cycles / ops_per_simd, instr_per_op);
}

int main() {
    uint32_t *out;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("shr32 + xor", 2, out);
    run<1>("mul_lo_u32", 1, out);
    run<2>("mul_hi_u32", 1, out);
    run<3>("mad_u64_u32 (+mov)", 2, out);
    run<4>("mul_u32_u24 + add", 1, out);
    run<5>("shr64 + add64", 3, out);
    run<6>("add64", 2, out);
    run<7>("alignbit + add", 2, out);
    run<8>("add64 + popc64", 4, out);
    run<9>("perm_b32 + add", 2, out);
    run<10>("sad_u16", 1, out);
    run<11>("mul_lo + shr + add", 3, out);
    hipFree(out);
    return 0;
}
