// roofline_probe.hip -- dev tool (not product): what does a PURE streaming read of the S1 shape cost as one launch?
//   hipcc --offload-arch=gfx950 -O3 -o scripts/roofline_probe scripts/roofline_probe.hip && scripts/roofline_probe
// A one-shot grid of 4-wave workgroups, a wave = one step of 1024 rows of a u16 column + a u8 column (3 KB), nothing
// written unless a row matches a value that never occurs.  Kernel begin / end from the dispatch packet's own events,
// two table copies alternated (nothing cache-resident).  time(n) = a + n * 3 B / BW: the intercept `a` is the part of
// a 100 M-row launch that no hand-off design can remove; the filter's own intercept is measured with scripts/ab_scan.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

template <bool NT> __device__ __forceinline__ uint4 ldx4(const void *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (NT) { const u32x4 q = __builtin_nontemporal_load((const u32x4 *)p); return make_uint4(q.x, q.y, q.z, q.w); }
    else return *(const uint4 *)p;
}
template <bool NT> __device__ __forceinline__ uint2 ldx2(const void *p) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (NT) { const u32x2 q = __builtin_nontemporal_load((const u32x2 *)p); return make_uint2(q.x, q.y); }
    else return *(const uint2 *)p;
}

// S steps per wave (adjacent)
template <bool NT, int S, int WPC>
__global__ __launch_bounds__(256, WPC) void read_kernel(const uint16_t *c16, const uint8_t *c8, uint64_t steps, uint32_t *out) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t step0 = ((uint64_t)blockIdx.x * 4 + wv) * S;
    uint32_t acc = 0;
    uint4 a[S][2];
    uint2 b[S][2];
#pragma unroll
    for (int s = 0; s < S; s++) {
        if (step0 + s >= steps) break;
        const uint64_t row0 = (step0 + s) * 1024 + lane * 8;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            a[s][u] = ldx4<NT>(c16 + row0 + u * 512);
            b[s][u] = ldx2<NT>(c8 + row0 + u * 512);
        }
    }
#pragma unroll
    for (int s = 0; s < S; s++) {
        if (step0 + s >= steps) break;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            acc += (a[s][u].x == 0xFFFFFFFFu) + (a[s][u].y == 0xFFFFFFFFu) + (a[s][u].z == 0xFFFFFFFFu) + (a[s][u].w == 0xFFFFFFFFu);
            acc += (b[s][u].x == 0xFFFFFFFFu) + (b[s][u].y == 0xFFFFFFFFu);
        }
    }
    if (acc) atomicAdd(out, acc);
}

// The same read + the S1 predicate (u8 == k8 AND u16 == k16) in three forms, to see what the evaluation costs a
// one-shot scan: EV 1 = ballots into SGPR planes (the product's chain path), EV 2 = vector compares folded into a
// per-lane count (no scalar unit), EV 3 = EV 1 + the tile hand-off (LDS count words, barrier, one sc1 store per tile).
template <int EV>
__global__ __launch_bounds__(256, 8) void eval_kernel(const uint16_t *c16, const uint8_t *c8, uint64_t steps, uint32_t k16, uint32_t k8,
                                                      uint32_t *counts, uint32_t *out) {
    __shared__ uint32_t tile_cnt[4];
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t step = (uint64_t)blockIdx.x * 4 + wv;
    uint32_t cnt = 0;
    if (step < steps) {
        const uint64_t row0 = step * 1024 + lane * 8;
        uint4 a[2];
        uint2 b[2];
#pragma unroll
        for (int u = 0; u < 2; u++) { a[u] = ldx4<true>(c16 + row0 + u * 512); b[u] = ldx2<true>(c8 + row0 + u * 512); }
        uint32_t v16[16], v8[16];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const uint32_t w[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
#pragma unroll
            for (int i = 0; i < 4; i++) { v16[8 * u + 2 * i] = w[i] & 0xFFFFu; v16[8 * u + 2 * i + 1] = w[i] >> 16; }
            const uint32_t x[2] = {b[u].x, b[u].y};
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) v8[8 * u + 4 * i + j] = (x[i] >> (8 * j)) & 0xFFu;
        }
        if (EV == 2) {
            uint32_t c = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) c += (v16[r] == k16 && v8[r] == k8) ? 1u : 0u;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
            cnt = c;
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint64_t m = __ballot(v16[r] == k16) & __ballot(v8[r] == k8);
                cnt += (uint32_t)__popcll(m);
            }
        }
    }
    if (EV == 3) {
        if (lane == 0) tile_cnt[wv] = cnt;
        __syncthreads();
        if (wv == 0 && lane < 4) __hip_atomic_store(counts + (uint64_t)blockIdx.x * 4 + lane, tile_cnt[lane] | 0x10000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (cnt && lane == 0) atomicAdd(out, cnt);
}

template <int EV>
void run_eval(const char *name, uint64_t rows, uint16_t *const *c16, uint8_t *const *c8, uint32_t *counts, uint32_t *out, hipStream_t s) {
    const uint64_t steps = (rows + 1023) / 1024;
    const uint32_t grid = (uint32_t)((steps + 3) / 4);
    const int reps = 24;
    std::vector<hipEvent_t> e0(reps), e1(reps);
    for (int i = 0; i < reps; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    for (int i = 0; i < 4; i++) hipLaunchKernelGGL(eval_kernel<EV>, dim3(grid), dim3(256), 0, s, c16[i & 1], c8[i & 1], steps, 1030u, 0u, counts, out);
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < reps; i++) hipExtLaunchKernelGGL(eval_kernel<EV>, dim3(grid), dim3(256), 0, s, e0[i], e1[i], 0, c16[i & 1], c8[i & 1], steps, 1030u, 0u, counts, out);
    CK(hipStreamSynchronize(s));
    std::vector<float> t(reps);
    for (int i = 0; i < reps; i++) { CK(hipEventElapsedTime(&t[i], e0[i], e1[i])); CK(hipEventDestroy(e0[i])); CK(hipEventDestroy(e1[i])); }
    std::sort(t.begin(), t.end());
    printf("%-28s rows %11llu  median %7.1f us  best %7.1f us  -> %.2f TB/s (median)\n", name, (unsigned long long)rows, t[reps / 2] * 1e3, t[0] * 1e3,
           (double)rows * 3.0 / (t[reps / 2] * 1e-3) / 1e12);
    fflush(stdout);
}

__global__ void fill_kernel(uint32_t *p, uint64_t n, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        p[i] = (uint32_t)(i * 2654435761u + seed) & 0x07FF07FFu;
}

template <typename K>
void run(const char *name, K kern, int S, uint32_t lds, uint64_t rows, uint16_t *const *c16, uint8_t *const *c8, uint32_t *out, hipStream_t s) {
    const uint64_t steps = (rows + 1023) / 1024;
    const uint32_t grid = (uint32_t)((steps + 4 * S - 1) / (4 * S));
    const int reps = 24;
    std::vector<hipEvent_t> e0(reps), e1(reps);
    for (int i = 0; i < reps; i++) { CK(hipEventCreate(&e0[i])); CK(hipEventCreate(&e1[i])); }
    for (int i = 0; i < 4; i++) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, c16[i & 1], c8[i & 1], steps, out);
    CK(hipStreamSynchronize(s));
    for (int i = 0; i < reps; i++) hipExtLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, e0[i], e1[i], 0, c16[i & 1], c8[i & 1], steps, out);
    CK(hipStreamSynchronize(s));
    std::vector<float> t(reps);
    for (int i = 0; i < reps; i++) { CK(hipEventElapsedTime(&t[i], e0[i], e1[i])); CK(hipEventDestroy(e0[i])); CK(hipEventDestroy(e1[i])); }
    std::sort(t.begin(), t.end());
    const double med = t[reps / 2] * 1e3, best = t[0] * 1e3;
    printf("%-28s rows %11llu  median %7.1f us  best %7.1f us  -> %.2f TB/s (median)\n", name, (unsigned long long)rows, med, best,
           (double)rows * 3.0 / (med * 1e-6) / 1e12);
    fflush(stdout);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const uint64_t max_rows = 400000000ull;
    uint16_t *c16[2]; uint8_t *c8[2]; uint32_t *out;
    for (int k = 0; k < 2; k++) {
        CK(hipMalloc(&c16[k], max_rows * 2 + 8192)); CK(hipMalloc(&c8[k], max_rows + 8192));
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s, (uint32_t *)c16[k], max_rows / 2, 17u + k);
        hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, s, (uint32_t *)c8[k], max_rows / 4, 91u + k);
    }
    CK(hipMalloc(&out, 64)); CK(hipMemset(out, 0, 64));
    CK(hipStreamSynchronize(s));
    uint32_t *counts; CK(hipMalloc(&counts, (max_rows / 1024 + 8) * 4));
    for (uint64_t n : {100000000ull, 400000000ull}) {
        run("nt    S=1 8wg/cu", read_kernel<true, 1, 8>, 1, 0, n, c16, c8, out, s);
        run_eval<1>("nt + ballot eval", n, c16, c8, counts, out, s);
        run_eval<2>("nt + vector eval", n, c16, c8, counts, out, s);
        run_eval<3>("nt + ballot eval + hand-off", n, c16, c8, counts, out, s);
    }
    const uint64_t sizes[] = {25000000ull, 100000000ull};
    for (uint64_t n : sizes) {
        run("plain S=1 8wg/cu", read_kernel<false, 1, 8>, 1, 0, n, c16, c8, out, s);
        run("nt    S=1 8wg/cu", read_kernel<true, 1, 8>, 1, 0, n, c16, c8, out, s);
        run("plain S=2 8wg/cu", read_kernel<false, 2, 8>, 2, 0, n, c16, c8, out, s);
        run("nt    S=2 8wg/cu", read_kernel<true, 2, 8>, 2, 0, n, c16, c8, out, s);
        run("nt    S=4 8wg/cu", read_kernel<true, 4, 8>, 4, 0, n, c16, c8, out, s);
        run("nt    S=1 4wg/cu (40K lds)", read_kernel<true, 1, 8>, 1, 40960, n, c16, c8, out, s);
    }
    uint32_t h = 0; CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("(matches of the impossible value: %u)\n", h);
    return 0;
}
