// rt_order.hip — the last step of a large pass's emission order (see rt_order.h).
#include "rt_order.h"

#include <hip/hip_runtime.h>

namespace rtd
{

namespace
{

constexpr int kRefineThreads = 256;
constexpr int kRefineItems = kRefineRun / kRefineThreads;  // pairs a thread holds
static_assert(kRefineItems % 4 == 0 && kRefineRun % 4 == 0, "pairs are read four at a time");
constexpr uint32_t kRefineBins = 2048;
constexpr uint32_t kBinsPerThread = kRefineBins / kRefineThreads;

// One block per run of kRefineRun positions.  The pairs arrive in ascending order of key >> kRefineBits, so
// the run's keys lie between its first key without the low bits and its last key with all of them set; a
// pair's bin is its key's distance from that lower end, without as many low bits as it takes for the
// distance to fit kRefineBins bins — none in a large pass, where a run spans a few values of the sorted
// part (200 M photons: 1.3), more in a small one, where a cell holds a handful of photons and its last
// bits order nothing.  Counting the bins, an exclusive sum over them and a second pass over the pairs
// (held in registers) put the ids in bin order, which ascends with the key.
__global__ __launch_bounds__(kRefineThreads) void refineOrderKernel(const uint32_t* keys, const uint32_t* ids,
                                                                    unsigned long long n, uint32_t* order)
{
    __shared__ uint32_t bins[kRefineBins];
    __shared__ uint32_t waveTotals[kRefineThreads / 64];
    __shared__ uint32_t ordered[kRefineRun];  // the run's ids in their new order, written out in whole lines
    const unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * kRefineRun;
    const uint32_t count = n - base < kRefineRun ? static_cast<uint32_t>(n - base) : kRefineRun;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t lowest = (keys[base] >> kRefineBits) << kRefineBits;
    const uint32_t widest = (keys[base + count - 1] | ((1u << kRefineBits) - 1)) - lowest;
    int drop = 0;
    while ((widest >> drop) >= kRefineBins) ++drop;
#pragma unroll
    for (uint32_t k = 0; k < kBinsPerThread; ++k) bins[k * kRefineThreads + t] = 0;
    __syncthreads();
    // a thread reads four neighbouring pairs at a time (the lists start on a 256-byte boundary and a
    // run is a multiple of four pairs long: 16-byte loads, except in the last run of the pass)
    uint32_t bin[kRefineItems], id[kRefineItems];
#pragma unroll
    for (int q = 0; q < kRefineItems / 4; ++q)
    {
        const uint32_t i = (static_cast<uint32_t>(q) * kRefineThreads + t) * 4;
        uint32_t key[4] = {0, 0, 0, 0};
        if (i + 4 <= count)
        {
            const uint4 k4 = *reinterpret_cast<const uint4*>(keys + base + i);
            const uint4 i4 = *reinterpret_cast<const uint4*>(ids + base + i);
            key[0] = k4.x; key[1] = k4.y; key[2] = k4.z; key[3] = k4.w;
            id[4 * q + 0] = i4.x; id[4 * q + 1] = i4.y; id[4 * q + 2] = i4.z; id[4 * q + 3] = i4.w;
        }
        else
        {
#pragma unroll
            for (int e = 0; e < 4; ++e)
            {
                if (i + e < count)
                {
                    key[e] = keys[base + i + e];
                    id[4 * q + e] = ids[base + i + e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            if (i + e < count)
            {
                bin[4 * q + e] = (key[e] - lowest) >> drop;
                atomicAdd(&bins[bin[4 * q + e]], 1u);
            }
        }
    }
    __syncthreads();
    // exclusive sum over the bins: kBinsPerThread consecutive bins per thread, a scan across the wave, the
    // waves' totals in LDS
    uint32_t mine[kBinsPerThread];
    uint32_t sum = 0;
#pragma unroll
    for (uint32_t k = 0; k < kBinsPerThread; ++k)
    {
        mine[k] = bins[t * kBinsPerThread + k];
        sum += mine[k];
    }
    uint32_t inclusive = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const uint32_t up = __shfl_up(inclusive, d, 64);
        if (static_cast<int>(lane) >= d) inclusive += up;
    }
    if (lane == 63) waveTotals[wave] = inclusive;
    __syncthreads();
    uint32_t running = inclusive - sum;
    for (uint32_t w = 0; w < wave; ++w) running += waveTotals[w];
#pragma unroll
    for (uint32_t k = 0; k < kBinsPerThread; ++k)
    {
        bins[t * kBinsPerThread + k] = running;
        running += mine[k];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRefineItems / 4; ++q)
    {
        const uint32_t i = (static_cast<uint32_t>(q) * kRefineThreads + t) * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e)
        {
            if (i + e < count)
            {
                ordered[atomicAdd(&bins[bin[4 * q + e]], 1u)] = id[4 * q + e];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRefineItems / 4; ++q)
    {
        const uint32_t i = (static_cast<uint32_t>(q) * kRefineThreads + t) * 4;
        if (i + 4 <= count)
        {
            *reinterpret_cast<uint4*>(order + base + i) = *reinterpret_cast<const uint4*>(ordered + i);
        }
        else
        {
            for (uint32_t e = 0; e < 4 && i + e < count; ++e) order[base + i + e] = ordered[i + e];
        }
    }
}

}  // namespace

void launchRefineOrder(const uint32_t* keys, const uint32_t* ids, uint64_t n, uint32_t* order, hipStream_t stream)
{
    if (n == 0) return;
    const unsigned int blocks = static_cast<unsigned int>((n + kRefineRun - 1) / kRefineRun);
    hipLaunchKernelGGL(refineOrderKernel, dim3(blocks), dim3(kRefineThreads), 0, stream, keys, ids,
                       static_cast<unsigned long long>(n), order);
}

}  // namespace rtd
