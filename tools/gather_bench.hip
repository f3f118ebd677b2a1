// gather_bench.hip -- practical ceiling for the k2u path: random 64-byte-line gathers from a buffer
// much larger than the Infinity Cache.  Each lane issues ILP independent 8/16-byte loads at random
// line addresses per iteration.  Prints achieved lines/s and the line-granular traffic rate.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench && tools/gather_bench [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return x;
}

template <int ILP, int BYTES, bool DEP>
__global__ __launch_bounds__(256) void gather(const uint64_t *__restrict__ buf, uint64_t n_lines, int iters, uint64_t *out) {
    uint64_t tid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0, s = mix(tid + 1);
    for (int it = 0; it < iters; ++it) {
        uint64_t v[ILP];
#pragma unroll
        for (int j = 0; j < ILP; ++j) {
            s = mix(s + j + (DEP ? acc : 0));
            uint64_t line = (uint64_t)(((__uint128_t)s * n_lines) >> 64);
            const uint64_t *p = buf + line * 8 + ((s >> 60) & 6);
            if (BYTES == 8) v[j] = p[0];
            else if (BYTES == 16) { ulonglong2 t = *(const ulonglong2 *)p; v[j] = t.x ^ t.y; }
            else if (BYTES == 128) { // touch both 64-byte halves of one 128-byte aligned pair with TWO loads (counts as ONE gather)
                const uint64_t *q = buf + (line & ~1ull) * 8;
                v[j] = q[0] ^ q[8];
            } else if (BYTES == 1280) { // the line-scan pattern: eight lanes share one random 128-byte line, ONE 16-byte load each
                const uint64_t gs = mix(((tid >> 3) + 1) * 0x9E3779B97F4A7C15ull + (uint64_t)it * ILP + j);
                const uint64_t l2 = (uint64_t)(((__uint128_t)gs * (n_lines / 2)) >> 64);
                ulonglong2 t = *(const ulonglong2 *)(buf + l2 * 16 + 2 * (tid & 7));
                v[j] = t.x ^ t.y;
            } else { // BYTES == 64: the record pattern: every lane reads its own random 64-byte record with 2 x 8 + 2 x 16 byte loads
                const uint64_t *q = buf + line * 8;
                const unsigned o = (unsigned)(s >> 57) & 2u;
                ulonglong2 t0 = *(const ulonglong2 *)(q + 4), t1 = *(const ulonglong2 *)(q + 6);
                v[j] = q[o] ^ q[o + 1] ^ t0.x ^ t0.y ^ t1.x ^ t1.y;
            }
        }
#pragma unroll
        for (int j = 0; j < ILP; ++j) acc ^= v[j];
    }
    if (acc == 0x1234567) out[0] = acc;
}

template <int ILP, int BYTES, bool DEP>
static void run(const uint64_t *buf, uint64_t n_lines, uint64_t *out, int blocks, const char *name) {
    int iters = 64 / ILP * 4 * (BYTES == 1280 ? 4 : 1);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((gather<ILP, BYTES, DEP>), dim3(blocks), dim3(256), 0, 0, buf, n_lines, iters, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((gather<ILP, BYTES, DEP>), dim3(blocks), dim3(256), 0, 0, buf, n_lines, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    double loads = (double)blocks * 256 * iters * ILP / (BYTES == 1280 ? 8 : 1);
    printf("%-28s blocks=%6d  %7.2f G lines/s  = %6.2f TB/s of 64-B lines  (%.3f ms)\n", name, blocks, loads / ms / 1e6, loads * 64 / ms / 1e9, ms);
}

int main(int argc, char **argv) {
    double gb = argc > 1 ? atof(argv[1]) : 6.0;
    uint64_t n_lines = (uint64_t)(gb * 1e9 / 64);
    uint64_t *buf, *out;
    hipMalloc(&buf, n_lines * 64);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, n_lines * 64);
    printf("buffer %.1f GB (%llu lines)\n", gb, (unsigned long long)n_lines);
    for (int blocks : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
        run<1, 8, true>(buf, n_lines, out, blocks, "dependent chain, 8 B");
        run<1, 8, false>(buf, n_lines, out, blocks, "ILP1 8 B");
        run<4, 8, false>(buf, n_lines, out, blocks, "ILP4 8 B");
        run<8, 8, false>(buf, n_lines, out, blocks, "ILP8 8 B");
        run<4, 16, false>(buf, n_lines, out, blocks, "ILP4 16 B");
        run<4, 128, false>(buf, n_lines, out, blocks, "ILP4 2x64B adjacent (pairs/s)");
        run<8, 1280, false>(buf, n_lines, out, blocks, "ILP8 8 lanes x 16 B = one 128-B line");
        run<2, 64, false>(buf, n_lines, out, blocks, "ILP2 64-B record, 4 loads per lane");
    }
    return 0;
}
