// radix_sort.hpp -- stable LSD radix sort of (key, u32 value) pairs for gfx950, 8 bits per pass.
//
// Used by the index build: a B+-tree replacement is the rows sorted by (key asc, row desc),
// i.e. a STABLE ascending sort of the keys fed in descending row order (engine/bplus.c:282-358,471-517
// define that leaf order), and by the cross-shard merge of index-mode results.
//
// One pass = three steps over tiles of 4096 pairs (256 threads x 16 rounds, element r * 256 + t in
// round r, so that every global load is coalesced and "earlier in the tile" = "earlier round, then lower
// thread"):
//   histogram : per-tile digit counts (LDS atomics)           -> hist[digit][tile]
//   scan      : exclusive prefix over the digit-major counts  -> first output slot of (digit, tile)
//   scatter   : every element's stable rank inside its tile -- within a wave by eight ballots (the lanes
//               that share the digit), across waves and rounds through small LDS tables -- then the tile
//               is laid out SORTED in LDS and written out so that each digit's run is one contiguous,
//               coalesced store stream.
// Passes over bits that are the same in every key (above the highest bit in which the smallest and the
// largest key differ) are not run: a 5-valued i32 column sorts in one pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pqps_sort {

constexpr int kThreads = 256;
constexpr int kRounds = 16;
constexpr int kTile = kThreads * kRounds;          // 4096 pairs per workgroup
constexpr int kBins = 256;

// order-preserving unsigned image of a key (signed keys: flip the sign bit)
template <typename K, bool SIGNED>
__device__ __forceinline__ uint64_t ukey(K k) {
    if (SIGNED) return (uint64_t)((uint32_t)k ^ 0x80000000u);
    return (uint64_t)k;
}

template <typename K, bool SIGNED>
__global__ __launch_bounds__(kThreads) void histogram_kernel(const K *__restrict__ keys, uint64_t n, uint32_t shift,
                                                             uint32_t *__restrict__ hist, uint32_t tiles) {
    __shared__ uint32_t s_hist[kBins];
    const uint32_t t = threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        s_hist[t] = 0;
        __syncthreads();
        const uint64_t base = (uint64_t)tile * kTile;
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const uint64_t i = base + (uint64_t)r * kThreads + t;
            if (i < n) atomicAdd(&s_hist[(uint32_t)(ukey<K, SIGNED>(keys[i]) >> shift) & 0xFFu], 1u);
        }
        __syncthreads();
        hist[(uint64_t)t * tiles + tile] = s_hist[t];
        __syncthreads();
    }
}

// ---- exclusive scan of a u32 array (digit-major histogram), three small kernels -----------------
constexpr int kScanBlock = 4096;                    // elements per workgroup (16 per thread)

__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *s_wave, uint32_t &total) {
    // exclusive scan of one value per thread over 256 threads
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, off, 64);
        if (lane >= (uint32_t)off) incl += o;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) { const uint32_t x = s_wave[w]; if ((uint32_t)w < wave) before += x; all += x; }
    __syncthreads();
    total = all;
    return before + incl - v;
}

__global__ __launch_bounds__(kThreads) void scan_sums_kernel(const uint32_t *__restrict__ a, uint64_t n, uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * 16;
    uint32_t v = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) if (base + i < n) v += a[base + i];
    uint32_t total;
    (void)block_excl_scan_256(v, s_wave, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup: exclusive scan of the block sums in place (a few thousand values)
__global__ __launch_bounds__(kThreads) void scan_top_kernel(uint32_t *sums, uint32_t m) {
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < m; base += kThreads) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < m ? sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_excl_scan_256(v, s_wave, total);
        if (i < m) sums[i] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(kThreads) void scan_apply_kernel(uint32_t *__restrict__ a, uint64_t n, const uint32_t *__restrict__ sums) {
    __shared__ uint32_t s_wave[4];
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlock + (uint64_t)threadIdx.x * 16;
    uint32_t x[16], v = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { x[i] = base + i < n ? a[base + i] : 0u; v += x[i]; }
    uint32_t total;
    uint32_t run = sums[blockIdx.x] + block_excl_scan_256(v, s_wave, total);
#pragma unroll
    for (int i = 0; i < 16; i++) { if (base + i < n) a[base + i] = run; run += x[i]; }
}

// ---- scatter: stable ranks, tile sorted in LDS, coalesced runs out -------------------------------
template <typename K, bool SIGNED>
__global__ __launch_bounds__(kThreads) void scatter_kernel(const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                           K *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint64_t n,
                                                           uint32_t shift, const uint32_t *__restrict__ offsets, uint32_t tiles) {
    __shared__ K s_key[kTile];
    __shared__ uint32_t s_val[kTile];
    __shared__ uint32_t s_wc[4][kBins];             // (round + 1) << 16 | elements of that digit in the wave, this round
    __shared__ uint32_t s_base[kBins];              // elements of the digit in earlier rounds; later: start in the sorted tile
    __shared__ uint32_t s_goff[kBins];              // first global slot of (digit, tile)
    __shared__ uint32_t s_wave[4];
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const uint64_t base = (uint64_t)tile * kTile;
        s_base[t] = 0;
        s_wc[0][t] = 0; s_wc[1][t] = 0; s_wc[2][t] = 0; s_wc[3][t] = 0;
        s_goff[t] = offsets[(uint64_t)t * tiles + tile];
        __syncthreads();
        K key[kRounds];
        uint32_t val[kRounds], rank[kRounds];       // rank: position among the tile's elements of the same digit
#pragma unroll
        for (int r = 0; r < kRounds; r++) {                         // all 32 loads in flight before the first use
            const uint64_t i = base + (uint64_t)r * kThreads + t;
            key[r] = i < n ? keys_in[i] : (K)0;
            val[r] = i < n ? vals_in[i] : 0u;
        }
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const bool valid = base + (uint64_t)r * kThreads + t < n;
            const uint32_t d = (uint32_t)(ukey<K, SIGNED>(key[r]) >> shift) & 0xFFu;
            // lanes of this wave that hold the same digit (and are valid)
            uint64_t same = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const uint64_t bal = __ballot((d >> b) & 1u);
                same &= ((d >> b) & 1u) ? bal : ~bal;
            }
            const uint32_t in_wave = (uint32_t)__popcll(same & lt_mask);
            if (valid && in_wave == 0) s_wc[wave][d] = ((uint32_t)(r + 1) << 16) | (uint32_t)__popcll(same);
            __syncthreads();
            uint32_t before = s_base[d];
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t x = s_wc[w][d];
                if ((uint32_t)w < wave && (x >> 16) == (uint32_t)(r + 1)) before += x & 0xFFFFu;
            }
            rank[r] = valid ? before + in_wave : 0xFFFFFFFFu;
            __syncthreads();
            {   // thread t closes digit t's books for this round
                uint32_t add = 0;
#pragma unroll
                for (int w = 0; w < 4; w++) { const uint32_t x = s_wc[w][t]; if ((x >> 16) == (uint32_t)(r + 1)) add += x & 0xFFFFu; }
                s_base[t] += add;
            }
            __syncthreads();
        }
        // digit starts inside the sorted tile
        uint32_t total;
        const uint32_t cnt = s_base[t];
        const uint32_t start = block_excl_scan_256(cnt, s_wave, total);
        s_base[t] = start;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            if (rank[r] != 0xFFFFFFFFu) {
                const uint32_t d = (uint32_t)(ukey<K, SIGNED>(key[r]) >> shift) & 0xFFu;
                const uint32_t p = s_base[d] + rank[r];
                s_key[p] = key[r];
                s_val[p] = val[r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < kRounds; r++) {
            const uint32_t p = (uint32_t)r * kThreads + t;
            if (p < total) {
                const K k = s_key[p];
                const uint32_t d = (uint32_t)(ukey<K, SIGNED>(k) >> shift) & 0xFFu;
                const uint64_t g = (uint64_t)s_goff[d] + (p - s_base[d]);
                keys_out[g] = k;
                vals_out[g] = s_val[p];
            }
        }
        __syncthreads();
    }
}

// smallest and largest unsigned key image: the bits above the highest one in which they differ are the
// same in every key, and the passes over those bits would move nothing -- they are not run
template <typename K, bool SIGNED>
__global__ __launch_bounds__(kThreads) void key_range_kernel(const K *__restrict__ keys, uint64_t n, unsigned long long *out) {
    __shared__ unsigned long long s_min[4], s_max[4];
    unsigned long long lo = ~0ull, hi = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kThreads) {
        const unsigned long long u = ukey<K, SIGNED>(keys[i]);
        lo = u < lo ? u : lo;
        hi = u > hi ? u : hi;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned long long a = __shfl_xor(lo, off, 64), b = __shfl_xor(hi, off, 64);
        lo = a < lo ? a : lo;
        hi = b > hi ? b : hi;
    }
    if ((threadIdx.x & 63) == 0) { s_min[threadIdx.x >> 6] = lo; s_max[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) { lo = s_min[w] < lo ? s_min[w] : lo; hi = s_max[w] > hi ? s_max[w] : hi; }
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
    }
}

struct Workspace {
    uint32_t *hist = nullptr;      // [256 * tiles]
    uint32_t *sums = nullptr;      // [ceil(256 * tiles / 4096)]
    unsigned long long *key_range = nullptr;   // [0] min, [1] max of the unsigned key images
    uint64_t tiles = 0;
};

inline hipError_t workspace_alloc(Workspace &w, uint64_t n) {
    w.tiles = (n + kTile - 1) / kTile;
    if (w.tiles == 0) w.tiles = 1;
    const uint64_t m = (uint64_t)kBins * w.tiles;
    hipError_t e = hipMalloc((void **)&w.hist, m * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&w.sums, ((m + kScanBlock - 1) / kScanBlock) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&w.key_range, 64);
    return e;
}

inline void workspace_free(Workspace &w) {
    if (w.hist) (void)hipFree(w.hist);
    if (w.sums) (void)hipFree(w.sums);
    if (w.key_range) (void)hipFree(w.key_range);
    w = Workspace();
}

// Sorts n pairs ascending by key, stably.  (keys_a, vals_a) holds the input; (keys_b, vals_b) is a
// second buffer pair of the same size.  Returns in *result_in_a whether the sorted data ended up in
// the a-buffers (true) or the b-buffers (false).  `bits` = significant key bits (multiple of 8).
template <typename K, bool SIGNED>
hipError_t sort_pairs(Workspace &w, K *keys_a, uint32_t *vals_a, K *keys_b, uint32_t *vals_b, uint64_t n, uint32_t bits,
                      int compute_units, hipStream_t s, bool *result_in_a) {
    *result_in_a = true;
    if (n <= 1) return hipSuccess;
    const uint32_t tiles = (uint32_t)w.tiles;
    const uint64_t m = (uint64_t)kBins * tiles;
    const uint32_t scan_blocks = (uint32_t)((m + kScanBlock - 1) / kScanBlock);
    const uint32_t grid = tiles < (uint32_t)compute_units * 8u ? tiles : (uint32_t)compute_units * 8u;
    // bytes in which the keys can differ at all
    const unsigned long long init[2] = {~0ull, 0ull};
    hipError_t e = hipMemcpyAsync(w.key_range, init, sizeof init, hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((key_range_kernel<K, SIGNED>), dim3(grid), dim3(kThreads), 0, s, keys_a, n, w.key_range);
    unsigned long long range[2] = {0, 0};
    e = hipMemcpyAsync(range, w.key_range, sizeof range, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    const unsigned long long diff = range[0] ^ range[1];
    uint32_t used_bits = 0;
    while (used_bits < bits && (diff >> used_bits) != 0) used_bits += 8;
    K *kin = keys_a, *kout = keys_b;
    uint32_t *vin = vals_a, *vout = vals_b;
    bool in_a = true;
    for (uint32_t shift = 0; shift < used_bits; shift += 8) {
        hipLaunchKernelGGL((histogram_kernel<K, SIGNED>), dim3(grid), dim3(kThreads), 0, s, kin, n, shift, w.hist, tiles);
        hipLaunchKernelGGL(scan_sums_kernel, dim3(scan_blocks), dim3(kThreads), 0, s, w.hist, m, w.sums);
        hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(kThreads), 0, s, w.sums, scan_blocks);
        hipLaunchKernelGGL(scan_apply_kernel, dim3(scan_blocks), dim3(kThreads), 0, s, w.hist, m, w.sums);
        hipLaunchKernelGGL((scatter_kernel<K, SIGNED>), dim3(grid), dim3(kThreads), 0, s, kin, vin, kout, vout, n, shift, w.hist, tiles);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        K *tk = kin; kin = kout; kout = tk;
        uint32_t *tv = vin; vin = vout; vout = tv;
        in_a = !in_a;
    }
    *result_in_a = in_a;
    return hipSuccess;
}

}  // namespace pqps_sort
