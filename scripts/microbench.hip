// microbench.hip -- ad-hoc timing of the small kernels of the filter pipeline (dev tool, not product)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>
#include <hip/hip_runtime.h>
#include "pqps_hip.h"
#include "../parallel-query-processing-system_amd/csrc/filter_kernels.hpp"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__global__ void empty_kernel(int *p) { if (p && threadIdx.x == 12345) *p = 1; }

// variants of K2 to find what is slow
template <int V>
__global__ __launch_bounds__(kBlock) void k2_variant(const SumArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    for (uint64_t group = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); group < a.groups;
         group += (uint64_t)gridDim.x * kWaves) {
        const uint64_t step = group * kGroupSteps + lane;
        uint32_t c = 0;
        if (V != 1) c = step < a.steps ? (a.counts[step] & 0x0FFFFFFFu) : 0u;     // V1: no load
        uint32_t sum = c;
        if (V != 2) sum = wave_sum_u32(c);                                         // V2: no reduction
        if (lane == 0) {
            a.group_sum[group] = sum;
            if (V != 3 && sum) atomicAdd(&a.super_sum[(group / kSuperGroups) * kSuperStride], (unsigned long long)sum);   // V3: no atomic
        }
    }
}

__global__ void flush_kernel(uint4 *p, uint64_t n16) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        uint4 v = p[i]; v.x += 1; p[i] = v;
    }
}

// time f() alone, each repetition preceded by a cache-thrashing pass over 1 GiB
template <typename F>
float time_cold(hipStream_t s, int reps, uint4 *junk, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float sum = 0;
    for (int i = 0; i < reps; i++) {
        hipLaunchKernelGGL(flush_kernel, dim3(4096), dim3(256), 0, s, junk, (uint64_t)(1ull << 30) / 16);
        CK(hipEventRecord(a, s));
        f();
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        sum += ms;
    }
    return sum * 1000.f / reps;
}

template <typename F>
float time_loop(hipStream_t s, int reps, F f) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) f();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1000.f / reps;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const uint64_t rows = 100000000, steps = (rows + 1023) / 1024, groups = (steps + 63) / 64, supers = (groups + 63) / 64;
    uint32_t *counts, *group_sum, *ids; unsigned long long *super_sum; uint64_t *base_slot, *out_count; uint16_t *masks;
    CK(hipMalloc(&counts, steps * 4)); CK(hipMalloc(&group_sum, groups * 4)); CK(hipMalloc(&super_sum, supers * 8 * kSuperStride));
    CK(hipMalloc(&base_slot, 64)); CK(hipMalloc(&out_count, 64)); CK(hipMalloc(&masks, steps * 128)); CK(hipMalloc(&ids, rows / 4 * 4));
    std::vector<uint32_t> h(steps);
    for (int dens = 0; dens < 2; dens++) {
        srand(1);
        std::vector<uint16_t> hm(steps * 64, 0);
        if (!dens) {
            for (uint64_t i = 0; i < steps; i++) h[i] = ((rand() % 100) < 7 ? 1 : 0) | (3u << 28);
            for (uint64_t i = 0; i < steps; i++) { uint32_t c = h[i] & 0xFFFFFFF; for (uint32_t k = 0; k < c; k++) hm[i * 64 + k] |= 1; }
        } else {
            uint64_t x = 88172645463325252ull;
            for (uint64_t i = 0; i < steps; i++) {
                uint32_t c = 0;
                for (int l = 0; l < 64; l++) {
                    uint16_t m = 0;
                    for (int b = 0; b < 16; b++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; if ((x & 0xFFFF) < 2884) m |= 1u << b; }
                    hm[i * 64 + l] = m; c += __builtin_popcount(m);
                }
                h[i] = c | (2u << 28);
            }
        }
        CK(hipMemcpy(counts, h.data(), steps * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(masks, hm.data(), steps * 128, hipMemcpyHostToDevice));
        SumArgs sa; sa.counts = counts; sa.steps = steps; sa.groups = groups; sa.group_sum = group_sum; sa.super_sum = super_sum;
        sa.base_slot = base_slot; sa.out_count = out_count; sa.accumulate = 0;
        ExpandArgs ea; memset(&ea, 0, sizeof ea);
        ea.masks = masks; ea.counts = counts; ea.group_sum = group_sum; ea.super_sum = super_sum; ea.base_slot = base_slot;
        ea.out_count = out_count; ea.steps = steps; ea.groups = groups; ea.out_ids = ids; ea.out_cap = rows / 4;
        uint32_t sum_blocks = (groups + 3) / 4;
        printf("density %s: steps %llu groups %llu\n", dens ? "dense(45/step)" : "sparse(7%% steps)", (unsigned long long)steps, (unsigned long long)groups);
        printf("  empty kernel x1 (382 blocks)      : %.2f us\n", time_loop(s, 200, [&] { hipLaunchKernelGGL(empty_kernel, dim3(382), dim3(256), 0, s, (int *)nullptr); }));
        printf("  memset super + K2                 : %.2f us\n", time_loop(s, 200, [&] { hipMemsetAsync(super_sum, 0, supers * 8 * kSuperStride, s); hipLaunchKernelGGL(group_sum_kernel, dim3(sum_blocks), dim3(256), 0, s, sa); }));
        printf("  K2 alone (super not reset)        : %.2f us\n", time_loop(s, 200, [&] { hipLaunchKernelGGL(group_sum_kernel, dim3(sum_blocks), dim3(256), 0, s, sa); }));
        printf("  K2 V1 (no load)                   : %.2f us\n", time_loop(s, 200, [&] { hipLaunchKernelGGL(k2_variant<1>, dim3(sum_blocks), dim3(256), 0, s, sa); }));
        printf("  K2 V2 (no wave reduce)            : %.2f us\n", time_loop(s, 200, [&] { hipLaunchKernelGGL(k2_variant<2>, dim3(sum_blocks), dim3(256), 0, s, sa); }));
        printf("  K2 V3 (no atomic)                 : %.2f us\n", time_loop(s, 200, [&] { hipLaunchKernelGGL(k2_variant<3>, dim3(sum_blocks), dim3(256), 0, s, sa); }));
        printf("  K2 V0 (all, template copy)        : %.2f us\n", time_loop(s, 200, [&] { hipLaunchKernelGGL(k2_variant<0>, dim3(sum_blocks), dim3(256), 0, s, sa); }));
        CK(hipMemsetAsync(super_sum, 0, supers * 8 * kSuperStride, s));
        hipLaunchKernelGGL(group_sum_kernel, dim3(sum_blocks), dim3(256), 0, s, sa);
        printf("  K3 alone                          : %.2f us\n", time_loop(s, 200, [&] { hipLaunchKernelGGL(expand_kernel, dim3((uint32_t)groups), dim3(256), 0, s, ea); }));
        static uint4 *junk = nullptr; if (!junk) { CK(hipMalloc(&junk, 1ull << 30)); CK(hipMemset(junk, 0, 1ull << 30)); }
        printf("  K2 cold (after 1 GiB flush)       : %.2f us\n", time_cold(s, 20, junk, [&] { hipLaunchKernelGGL(group_sum_kernel, dim3(sum_blocks), dim3(256), 0, s, sa); }));
        CK(hipMemsetAsync(super_sum, 0, supers * 8 * kSuperStride, s));
        hipLaunchKernelGGL(group_sum_kernel, dim3(sum_blocks), dim3(256), 0, s, sa);
        printf("  K3 cold (after 1 GiB flush)       : %.2f us\n", time_cold(s, 20, junk, [&] { hipLaunchKernelGGL(expand_kernel, dim3((uint32_t)groups), dim3(256), 0, s, ea); }));
        printf("  empty cold                        : %.2f us\n", time_cold(s, 20, junk, [&] { hipLaunchKernelGGL(empty_kernel, dim3(382), dim3(256), 0, s, (int *)nullptr); }));
        uint64_t total; CK(hipMemcpy(&total, out_count, 8, hipMemcpyDeviceToHost));
        printf("  total matches reported by K3: %llu\n", (unsigned long long)total);
    }
    // ---- the real pipeline on a real column: risk_level > 3 (4.4 % hits), K1 -> K2 -> K3 timed separately
    {
        int32_t *risk; CK(hipMalloc(&risk, (rows + 4096) * 4));
        std::vector<int32_t> hr(rows);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (uint64_t i = 0; i < rows; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const uint32_t y = (uint32_t)(x >> 32); hr[i] = y < 2439541424u ? 1 : y < 3715147407u ? 2 : y < 4105988735u ? 3 : y < 4252017623u ? 4 : 5; }
        CK(hipMemcpy(risk, hr.data(), rows * 4, hipMemcpyHostToDevice));
        EvalArgs a; memset(&a, 0, sizeof a);
        a.col[0] = risk; a.width_log2[0] = 2; a.n_cols = 1; a.n_leaves = 1; a.leaf_begin[0] = 0; a.leaf_begin[1] = 1;
        a.lo[0] = 4; a.span[0] = 0x7FFFFFFFu - 4; a.truth = 2; a.n_rows = rows;
        a.masks = masks; a.counts = counts; a.super_sum = super_sum; a.n_super = (uint32_t)supers;
        uint64_t *partials; CK(hipMalloc(&partials, 4096 * 8)); a.partials = partials;
        SumArgs sa; sa.counts = counts; sa.steps = steps; sa.groups = groups; sa.group_sum = group_sum; sa.super_sum = super_sum;
        sa.base_slot = base_slot; sa.out_count = out_count; sa.accumulate = 0;
        ExpandArgs ea; memset(&ea, 0, sizeof ea);
        ea.masks = masks; ea.counts = counts; ea.group_sum = group_sum; ea.super_sum = super_sum; ea.base_slot = base_slot;
        ea.out_count = out_count; ea.steps = steps; ea.groups = groups; ea.out_ids = ids; ea.out_cap = rows / 4;
        hipEvent_t e0, e1, e2, e3; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
        float t1 = 0, t2 = 0, t3 = 0;
        const int reps = 30;
        for (int i = 0; i < reps + 3; i++) {
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL((eval_spec_kernel<MODE_IDS, 4, 0, 0, false>), dim3(4096), dim3(256), 0, s, a);
            CK(hipEventRecord(e1, s));
            hipLaunchKernelGGL(group_sum_kernel, dim3((uint32_t)((groups + 3) / 4)), dim3(256), 0, s, sa);
            CK(hipEventRecord(e2, s));
            hipLaunchKernelGGL(expand_kernel, dim3((uint32_t)groups), dim3(256), 0, s, ea);
            CK(hipEventRecord(e3, s));
            CK(hipEventSynchronize(e3));
            if (i >= 3) { float ms; CK(hipEventElapsedTime(&ms, e0, e1)); t1 += ms; CK(hipEventElapsedTime(&ms, e1, e2)); t2 += ms; CK(hipEventElapsedTime(&ms, e2, e3)); t3 += ms; }
        }
        uint64_t total; CK(hipMemcpy(&total, out_count, 8, hipMemcpyDeviceToHost));
        printf("real pipeline risk_level > 3: K1 %.1f us  K2 %.1f us  K3 %.1f us  matches %llu\n", t1 * 1000 / reps, t2 * 1000 / reps, t3 * 1000 / reps, (unsigned long long)total);
    }
    // ---- the same query through the shim (libpqps_hip.so), hipMalloc'd buffers, no torch
    {
        pqps_ctx *ctx; if (pqps_ctx_create(0, &ctx)) { printf("ctx: %s\n", pqps_last_error()); return 1; }
        int32_t *risk; CK(hipMalloc(&risk, (rows + 4096) * 4));
        std::vector<int32_t> hr(rows);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (uint64_t i = 0; i < rows; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; const uint32_t y = (uint32_t)(x >> 32); hr[i] = y < 2439541424u ? 1 : y < 3715147407u ? 2 : y < 4105988735u ? 3 : y < 4252017623u ? 4 : 5; }
        CK(hipMemcpy(risk, hr.data(), rows * 4, hipMemcpyHostToDevice));
        pqps_column col; col.data = risk; col.width = 4; col.reserved = 0;
        pqps_predicate pred; memset(&pred, 0, sizeof pred);
        pred.n_leaves = 1; pred.n_columns = 1; pred.truth = 2; pred.leaf[0].column = 0; pred.leaf[0].lo = 4; pred.leaf[0].span = 0x7FFFFFFFu - 4;
        pred.on_true[0] = PQPS_ACCEPT; pred.on_false[0] = PQPS_REJECT;
        uint32_t *oid; uint64_t *ocnt; CK(hipMalloc(&oid, rows * 4)); CK(hipMalloc(&ocnt, 64));
        for (int i = 0; i < 5; i++) if (pqps_filter_scan(ctx, &col, 1, rows, 0, &pred, oid, rows, ocnt, nullptr)) { printf("scan: %s\n", pqps_last_error()); return 1; }
        pqps_ctx_sync(ctx, nullptr);
        pqps_ctx_set_timing(ctx, 1);
        for (int i = 0; i < 30; i++) pqps_filter_scan(ctx, &col, 1, rows, 0, &pred, oid, rows, ocnt, nullptr);
        double ev, tot; int k; pqps_ctx_kernel_time(ctx, &ev, &tot, &k);
        uint64_t total; CK(hipMemcpy(&total, ocnt, 8, hipMemcpyDeviceToHost));
        printf("through the shim: K1 %.1f us  K1..K3 %.1f us (incl. 1 event record)  matches %llu\n", ev * 1000 / k, tot * 1000 / k, (unsigned long long)total);
        pqps_ctx_destroy(ctx);
    }
    return 0;
}
