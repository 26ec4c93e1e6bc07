// rt_order.hip — the first and the last step of a large pass's emission order (see rt_order.h).
#include "rt_order.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

namespace rtd
{

namespace
{

constexpr int kDigitThreads = 512;
constexpr int kDigitItems = kDigitTile / kDigitThreads;  // ids a thread takes
constexpr uint32_t kDigits = 1u << kDigitBits;
constexpr int kOrderXcds = 8;  // blocks go to the chip's eight XCDs in turn
static_assert(kDigits <= kDigitThreads, "a thread per digit in the tile's scan");
static_assert(kDigitTile <= 65536, "a pair's place in its tile is kept in 16 bits");

// Tile of a block: block b runs on XCD b % 8, and an XCD's blocks take neighbouring tiles, whose runs of
// one digit are neighbours in memory: the two parts of a line that two tiles share meet in one L2.
RT_DEV uint64_t digitTileOfBlock(uint64_t tilesPerXcd)
{
    return (blockIdx.x % kOrderXcds) * tilesPerXcd + blockIdx.x / kOrderXcds;
}


__global__ __launch_bounds__(kDigitThreads) void digitCountKernel(const TraceArgs a, uint64_t tiles, uint64_t tilesPerXcd, int shift,
                                                                  uint32_t* keys, uint32_t* counts)
{
    __shared__ uint32_t histogram[kDigits];
    const uint64_t tile = digitTileOfBlock(tilesPerXcd);
    if (tile >= tiles) return;
    if (threadIdx.x < kDigits) histogram[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long begin = tile * kDigitTile;
#pragma unroll 4
    for (int k = 0; k < kDigitItems; ++k)
    {
        const unsigned long long id = begin + static_cast<unsigned long long>(k) * kDigitThreads + threadIdx.x;
        if (id < a.totalWork)
        {
            int lightId;
            unsigned long long photonIndex;
            splitPhotonId(a, id, &lightId, &photonIndex);
            const uint32_t key =
                (static_cast<uint32_t>(lightId & 0xFF) << (2 * kGlobalAxisBits)) | emissionCell(a, lightId, photonIndex);
            keys[id] = key;
            atomicAdd(&histogram[(key >> shift) & (kDigits - 1)], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < kDigits) counts[threadIdx.x * tiles + tile] = histogram[threadIdx.x];
}

// ---- placing the pairs of a tile (steps 1 and 2) ----
// Given the exclusive sum of the [digit][tile] counts, a tile's pairs are gathered by digit in LDS and written
// out run by run.  A pair's place among its tile's pairs of the same digit is COUNTED, in arrival order: a wave
// takes 64 consecutive pairs at a time, the lanes with the same digit find each other with one ballot per bit
// of the digit, a pair's rank is the digit's count so far in its wave (kept in LDS, advanced by the first of
// the lanes that share it) plus the lanes before it among those; the waves' counts are then summed in wave
// order.  Step 2 needs that (pairs of equal digit must keep the order of step 1); step 1 does not, but ranks
// claimed with LDS atomics — 8192 pairs on 256 counters — cost it 0.9 ms where these cost 0.73, and the
// library's one-sweep pass, which this replaces in step 2 for keys of one light, moves the pairs at 2.3 TB/s
// where this moves them at 4.
constexpr int kTopWaves = kDigitThreads / 64;
constexpr int kTopSteps = kDigitTile / kDigitThreads;  // times a wave takes 64 pairs
static_assert(kDigitTile == kTopWaves * kTopSteps * 64, "a wave's pairs are consecutive");

__global__ __launch_bounds__(kDigitThreads) void digitHistogramKernel(uint64_t n, uint64_t tiles, uint64_t tilesPerXcd, int shift,
                                                                      const uint32_t* keys, uint32_t* counts)
{
    __shared__ uint32_t histogram[kDigits];
    const uint64_t tile = digitTileOfBlock(tilesPerXcd);
    if (tile >= tiles) return;
    if (threadIdx.x < kDigits) histogram[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long begin = tile * kDigitTile;
#pragma unroll 4
    for (int k = 0; k < kDigitItems; ++k)
    {
        const unsigned long long i = begin + static_cast<unsigned long long>(k) * kDigitThreads + threadIdx.x;
        if (i < n) atomicAdd(&histogram[(keys[i] >> shift) & (kDigits - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kDigits) counts[threadIdx.x * tiles + tile] = histogram[threadIdx.x];
}

// the digit: bits [shift, shift + 8) of the key; Implicit: a pair's id is its place in the list (ids is null)
template <bool Implicit>
__global__ __launch_bounds__(kDigitThreads) void stableScatterKernel(uint64_t n, uint64_t tiles, uint64_t tilesPerXcd, int shift,
                                                                     const uint32_t* keys, const uint32_t* ids,
                                                                     const uint32_t* offsets, uint32_t* keysOut, uint32_t* idsOut)
{
    auto digitHere = [shift](uint32_t key) { return (key >> shift) & (kDigits - 1); };
    __shared__ uint32_t waveCount[kTopWaves][kDigits];  // pairs of a digit in a wave's part of the tile, then those of the waves before it
    __shared__ uint32_t localStart[kDigits];            // where the digit's run starts in the gathered tile ...
    __shared__ uint32_t globalStart[kDigits];           // ... and in the list
    __shared__ uint32_t waveTotals[kTopWaves];
    __shared__ uint32_t gatheredKey[kDigitTile];
    __shared__ uint32_t gatheredId[kDigitTile];
    const uint64_t tile = digitTileOfBlock(tilesPerXcd);
    if (tile >= tiles) return;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (uint32_t k = t; k < kTopWaves * kDigits; k += kDigitThreads) (&waveCount[0][0])[k] = 0;
    __syncthreads();
    const unsigned long long begin = tile * kDigitTile;
    const uint32_t count = n - begin < kDigitTile ? static_cast<uint32_t>(n - begin) : kDigitTile;
    uint32_t key[kTopSteps], id[kTopSteps], rank[kTopSteps];
#pragma unroll
    for (int step = 0; step < kTopSteps; ++step)
    {
        const uint32_t i = (wave * kTopSteps + step) * 64 + lane;
        const bool there = i < count;
        key[step] = there ? keys[begin + i] : 0u;
        id[step] = there ? (Implicit ? static_cast<uint32_t>(begin) + i : ids[begin + i]) : 0u;
    }
#pragma unroll
    for (int step = 0; step < kTopSteps; ++step)
    {
        const uint32_t i = (wave * kTopSteps + step) * 64 + lane;
        const bool there = i < count;
        const uint32_t digit = digitHere(key[step]);
        unsigned long long peers = __ballot(there);
#pragma unroll
        for (int bit = 0; bit < kDigitBits; ++bit)
        {
            const bool set = ((digit >> bit) & 1u) != 0;
            const unsigned long long with = __ballot(set);
            peers &= set ? with : ~with;
        }
        // (a lane that holds no pair is among the peers of nobody: `there` is in every lane's mask)
        const int first = there ? __builtin_ctzll(peers) : static_cast<int>(lane);
        uint32_t before = 0;
        if (there && static_cast<int>(lane) == first)
        {
            before = waveCount[wave][digit];
            waveCount[wave][digit] = before + static_cast<uint32_t>(__builtin_popcountll(peers));
        }
        before = __shfl(before, first, 64);
        rank[step] = before + static_cast<uint32_t>(__builtin_popcountll(peers & ((1ull << lane) - 1ull)));
    }
    __syncthreads();
    // a thread per digit: the waves' counts become the counts of the waves before them, the digit's total
    // goes into the exclusive sum over the digits
    uint32_t mine = 0;
    if (t < kDigits)
    {
#pragma unroll
        for (int w = 0; w < kTopWaves; ++w)
        {
            const uint32_t c = waveCount[w][t];
            waveCount[w][t] = mine;
            mine += c;
        }
    }
    uint32_t inclusive = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const uint32_t up = __shfl_up(inclusive, d, 64);
        if (static_cast<int>(lane) >= d) inclusive += up;
    }
    if (lane == 63) waveTotals[wave] = inclusive;
    __syncthreads();
    if (t < kDigits)
    {
        uint32_t beforeWaves = 0;
        for (uint32_t w = 0; w < wave; ++w) beforeWaves += waveTotals[w];
        localStart[t] = beforeWaves + inclusive - mine;
        globalStart[t] = offsets[t * tiles + tile];
    }
    __syncthreads();
#pragma unroll
    for (int step = 0; step < kTopSteps; ++step)
    {
        const uint32_t i = (wave * kTopSteps + step) * 64 + lane;
        if (i < count)
        {
            const uint32_t digit = digitHere(key[step]);
            const uint32_t at = localStart[digit] + waveCount[wave][digit] + rank[step];
            gatheredKey[at] = key[step];
            gatheredId[at] = id[step];
        }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kDigitItems; ++k)
    {
        const uint32_t i = static_cast<uint32_t>(k) * kDigitThreads + t;
        if (i < count)
        {
            const uint32_t k32 = gatheredKey[i];
            const uint32_t digit = digitHere(k32);
            const uint32_t at = globalStart[digit] + (i - localStart[digit]);
            keysOut[at] = k32;
            idsOut[at] = gatheredId[i];
        }
    }
}

constexpr int kRefineThreads = 256;
constexpr int kRefineItems = kRefineRun / kRefineThreads;  // pairs a thread holds
static_assert(kRefineItems % 4 == 0 && kRefineRun % 4 == 0, "pairs are read four at a time");
constexpr uint32_t kRefineBins = 2048;
constexpr uint32_t kBinsPerThread = kRefineBins / kRefineThreads;

// One block per run of kRefineRun positions.  The pairs arrive in ascending order of key >> sortedShift, so
// the run's keys lie between its first key without the low bits and its last key with all of them set; a
// pair's bin is its key's distance from that lower end, without as many low bits as it takes for the
// distance to fit kRefineBins bins — none in a large pass, where a run spans a few values of the sorted
// part (200 M photons: 1.3), more in a small one, where a cell holds a handful of photons and its last
// bits order nothing.  Counting the bins, an exclusive sum over them and a second pass over the pairs
// (held in registers) put the ids in bin order, which ascends with the key.
__global__ __launch_bounds__(kRefineThreads) void refineOrderKernel(const uint32_t* keys, const uint32_t* ids,
                                                                    unsigned long long n, int sortedShift, uint32_t* order)
{
    __shared__ uint32_t bins[kRefineBins];
    __shared__ uint32_t waveTotals[kRefineThreads / 64];
    __shared__ uint32_t ordered[kRefineRun];  // the run's ids in their new order, written out in whole lines
    const unsigned long long base = static_cast<unsigned long long>(blockIdx.x) * kRefineRun;
    const uint32_t count = n - base < kRefineRun ? static_cast<uint32_t>(n - base) : kRefineRun;
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const uint32_t lowest = (keys[base] >> sortedShift) << sortedShift;
    const uint32_t widest = (keys[base + count - 1] | ((1u << sortedShift) - 1)) - lowest;
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

hipError_t digitOrder(const TraceArgs& args, int shift, void* scratch, size_t* scratchBytes, uint32_t* keys, uint32_t* counts,
                      uint32_t* offsets, uint32_t* keysOut, uint32_t* idsOut, hipStream_t stream)
{
    const uint64_t n = args.totalWork, tiles = digitTiles(n), entries = digitMatrixEntries(n);
    if (scratch == nullptr)
    {
        return hipcub::DeviceScan::ExclusiveSum(nullptr, *scratchBytes, counts, offsets, static_cast<int>(entries), stream);
    }
    if (n == 0) return hipSuccess;
    const uint64_t tilesPerXcd = (tiles + kOrderXcds - 1) / kOrderXcds;
    const dim3 grid(static_cast<unsigned int>(tilesPerXcd * kOrderXcds)), block(kDigitThreads);
    hipLaunchKernelGGL(digitCountKernel, grid, block, 0, stream, args, tiles, tilesPerXcd, shift, keys, counts);
    const hipError_t scanned =
        hipcub::DeviceScan::ExclusiveSum(scratch, *scratchBytes, counts, offsets, static_cast<int>(entries), stream);
    if (scanned != hipSuccess) return scanned;
    hipLaunchKernelGGL(stableScatterKernel<true>, grid, block, 0, stream, n, tiles, tilesPerXcd, shift, keys, nullptr, offsets,
                       keysOut, idsOut);
    return hipGetLastError();
}

hipError_t stableDigitPass(uint64_t n, int shift, void* scratch, size_t* scratchBytes, const uint32_t* keys, const uint32_t* ids,
                           uint32_t* counts, uint32_t* offsets, uint32_t* keysOut, uint32_t* idsOut, hipStream_t stream)
{
    const uint64_t tiles = digitTiles(n), entries = digitMatrixEntries(n);
    if (scratch == nullptr)
    {
        return hipcub::DeviceScan::ExclusiveSum(nullptr, *scratchBytes, counts, offsets, static_cast<int>(entries), stream);
    }
    if (n == 0) return hipSuccess;
    const uint64_t tilesPerXcd = (tiles + kOrderXcds - 1) / kOrderXcds;
    const dim3 grid(static_cast<unsigned int>(tilesPerXcd * kOrderXcds)), block(kDigitThreads);
    hipLaunchKernelGGL(digitHistogramKernel, grid, block, 0, stream, n, tiles, tilesPerXcd, shift, keys, counts);
    const hipError_t scanned =
        hipcub::DeviceScan::ExclusiveSum(scratch, *scratchBytes, counts, offsets, static_cast<int>(entries), stream);
    if (scanned != hipSuccess) return scanned;
    if (ids == nullptr)
    {
        hipLaunchKernelGGL(stableScatterKernel<true>, grid, block, 0, stream, n, tiles, tilesPerXcd, shift, keys, ids, offsets, keysOut,
                           idsOut);
    }
    else
    {
        hipLaunchKernelGGL(stableScatterKernel<false>, grid, block, 0, stream, n, tiles, tilesPerXcd, shift, keys, ids, offsets, keysOut,
                           idsOut);
    }
    return hipGetLastError();
}

void launchRefineOrder(const uint32_t* keys, const uint32_t* ids, uint64_t n, int sortedShift, uint32_t* order, hipStream_t stream)
{
    if (n == 0) return;
    const unsigned int blocks = static_cast<unsigned int>((n + kRefineRun - 1) / kRefineRun);
    hipLaunchKernelGGL(refineOrderKernel, dim3(blocks), dim3(kRefineThreads), 0, stream, keys, ids,
                       static_cast<unsigned long long>(n), sortedShift, order);
}

}  // namespace rtd
