// rt_store.hip — which records a bounce store that overflowed keeps.
//
// The reference's store takes records until it is full and drops the rest (reference
// src/BounceStore.cpp:11-33); its workers meet the photons in index order, light by light (reference
// src/Worker.cpp:429-505), so what a full store holds are the deposits of the FIRST photons — a fair
// sample of the emission directions.  The trace kernel meets the photons in emission order instead, which
// is anything but a fair sample of the directions at the moment the store fills up.  So while a pass runs
// the store takes records beyond its capacity too, each with its place in the reference's order as a
// key, and afterwards the kernels below move the `need` records with the smallest keys into the capacity:
// a radix selection of the need-th smallest key (one pass over the keys per digit, no sort), a sweep that
// lists the records to be replaced and the records to move in, and the move itself.

#include "rt_kernels.h"

namespace rtd
{

namespace
{

constexpr int kSelectThreads = 256;
constexpr int kSelectBins = 1 << kSelectDigitBits;

__global__ void storeSelectResetKernel(StoreSelectState* state, unsigned long long need)
{
    for (int bin = threadIdx.x; bin < kSelectBins; bin += blockDim.x)
    {
        state->histogram[bin] = 0;
    }
    if (threadIdx.x == 0)
    {
        state->prefix = 0;
        state->remaining = need;
        state->holes = 0;
        state->fillers = 0;
    }
}

// Histogram of one digit (bits shift .. shift + width) over the keys whose bits above the digit equal the
// prefix found so far: per block in LDS, then added to the state's.
__global__ __launch_bounds__(kSelectThreads) void storeSelectCountKernel(const unsigned long long* keys, uint64_t candidates,
                                                                         int shift, int width, StoreSelectState* state)
{
    __shared__ unsigned int local[kSelectBins];
    for (int bin = threadIdx.x; bin < kSelectBins; bin += kSelectThreads)
    {
        local[bin] = 0;
    }
    __syncthreads();
    const unsigned long long prefix = state->prefix;
    const int above = shift + width;  // bits from here up are settled
    const unsigned long long digitMask = (1ull << width) - 1ull;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kSelectThreads;
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * kSelectThreads + threadIdx.x; i < candidates; i += stride)
    {
        const unsigned long long key = keys[i];
        const bool matches = above >= 64 || (key >> above) == (prefix >> above);
        if (matches)
        {
            atomicAdd(&local[(key >> shift) & digitMask], 1u);
        }
    }
    __syncthreads();
    for (int bin = threadIdx.x; bin < kSelectBins; bin += kSelectThreads)
    {
        if (local[bin] != 0)
        {
            atomicAdd(&state->histogram[bin], local[bin]);
        }
    }
}

// One block: the digit in which the running count reaches `remaining` joins the prefix; the keys of the
// digits below it are all wanted and leave `remaining`; the histogram is cleared for the next digit.
__global__ __launch_bounds__(kSelectBins) void storeSelectPickKernel(int shift, StoreSelectState* state)
{
    __shared__ unsigned long long running[kSelectBins];
    const int bin = threadIdx.x;
    const unsigned int mine = state->histogram[bin];
    running[bin] = mine;
    __syncthreads();
    for (int step = 1; step < kSelectBins; step <<= 1)  // inclusive sums
    {
        const unsigned long long lower = bin >= step ? running[bin - step] : 0ull;
        __syncthreads();
        running[bin] += lower;
        __syncthreads();
    }
    const unsigned long long remaining = state->remaining;
    const unsigned long long before = running[bin] - mine;
    __syncthreads();
    if (before < remaining && remaining <= running[bin])
    {
        state->prefix |= static_cast<unsigned long long>(bin) << shift;
        state->remaining = remaining - before;
    }
    state->histogram[bin] = 0;
}

RT_DEV uint32_t lanesBelowMask(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}

// A wave takes a run of kClassifyRun consecutive slots at a time and walks it twice: first it counts the
// run's holes (inside the capacity, key above the threshold) and fillers (behind the capacity, key not
// above it) and claims room for both in the lists with ONE atomic each — a claim per 64 slots had all
// the waves of the chip queueing at two addresses, 5.7 ms for 32 M records —, then it reads the keys
// again (they are still in cache) and writes the slot numbers.
constexpr uint32_t kClassifyRun = 64 * 64;

__global__ __launch_bounds__(kSelectThreads) void storeSelectClassifyKernel(const unsigned long long* keys, uint64_t first,
                                                                            uint64_t candidates, uint64_t capacity,
                                                                            StoreSelectState* state, uint32_t* holes,
                                                                            uint32_t* fillers)
{
    const unsigned long long threshold = state->prefix;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (static_cast<uint64_t>(blockIdx.x) * kSelectThreads + threadIdx.x) >> 6;
    const uint64_t waves = (static_cast<uint64_t>(gridDim.x) * kSelectThreads) >> 6;
    const uint64_t runs = (candidates + kClassifyRun - 1) / kClassifyRun;
    for (uint64_t run = wave; run < runs; run += waves)
    {
        const uint64_t begin = run * kClassifyRun;
        const uint64_t end = begin + kClassifyRun < candidates ? begin + kClassifyRun : candidates;
        uint32_t holeCount = 0, fillerCount = 0;  // wave-uniform
        for (uint64_t i = begin + lane; i < begin + kClassifyRun; i += 64)
        {
            const bool valid = i < end;
            const uint64_t slot = first + i;
            const bool wanted = valid && keys[valid ? slot : first] <= threshold;
            const bool inside = slot < capacity;
            holeCount += static_cast<uint32_t>(__builtin_popcountll(__ballot(valid && inside && !wanted)));
            fillerCount += static_cast<uint32_t>(__builtin_popcountll(__ballot(valid && !inside && wanted)));
        }
        unsigned long long holeBase = 0, fillerBase = 0;
        if (lane == 0)
        {
            if (holeCount != 0) holeBase = atomicAdd(&state->holes, static_cast<unsigned long long>(holeCount));
            if (fillerCount != 0) fillerBase = atomicAdd(&state->fillers, static_cast<unsigned long long>(fillerCount));
        }
        holeBase = __shfl(holeBase, 0, 64);
        fillerBase = __shfl(fillerBase, 0, 64);
        if (holeCount == 0 && fillerCount == 0)
        {
            continue;
        }
        for (uint64_t i = begin + lane; i < begin + kClassifyRun; i += 64)
        {
            const bool valid = i < end;
            const uint64_t slot = first + i;
            const bool wanted = valid && keys[valid ? slot : first] <= threshold;
            const bool inside = slot < capacity;
            const bool hole = valid && inside && !wanted, filler = valid && !inside && wanted;
            const unsigned long long holeMask = __ballot(hole), fillerMask = __ballot(filler);
            if (hole) holes[holeBase + lanesBelowMask(holeMask)] = static_cast<uint32_t>(slot);
            if (filler) fillers[fillerBase + lanesBelowMask(fillerMask)] = static_cast<uint32_t>(slot);
            holeBase += static_cast<unsigned long long>(__builtin_popcountll(holeMask));
            fillerBase += static_cast<unsigned long long>(__builtin_popcountll(fillerMask));
        }
    }
}

__global__ __launch_bounds__(kSelectThreads) void storeSelectMoveKernel(float* records, unsigned long long* keys,
                                                                        const StoreSelectState* state, const uint32_t* holes,
                                                                        const uint32_t* fillers)
{
    const unsigned long long pairs = state->holes < state->fillers ? state->holes : state->fillers;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * kSelectThreads;
    // a thread per record: 52 bytes, 4-byte aligned, as three 16-byte accesses and a float
    typedef float UnalignedFloat4 __attribute__((ext_vector_type(4), aligned(4)));
    for (uint64_t pair = static_cast<uint64_t>(blockIdx.x) * kSelectThreads + threadIdx.x; pair < pairs; pair += stride)
    {
        const uint64_t to = holes[pair], from = fillers[pair];
        const float* source = records + 13 * from;
        float* target = records + 13 * to;
        const UnalignedFloat4 q0 = *reinterpret_cast<const UnalignedFloat4*>(source);
        const UnalignedFloat4 q1 = *reinterpret_cast<const UnalignedFloat4*>(source + 4);
        const UnalignedFloat4 q2 = *reinterpret_cast<const UnalignedFloat4*>(source + 8);
        const float last = source[12];
        *reinterpret_cast<UnalignedFloat4*>(target) = q0;
        *reinterpret_cast<UnalignedFloat4*>(target + 4) = q1;
        *reinterpret_cast<UnalignedFloat4*>(target + 8) = q2;
        target[12] = last;
        keys[to] = keys[from];
    }
}

int selectBlocks(uint64_t items)
{
    const uint64_t wanted = (items + kSelectThreads - 1) / kSelectThreads;
    return static_cast<int>(wanted < 1 ? 1 : (wanted > 4096 ? 4096 : wanted));
}

}  // namespace

void launchStoreSelectThreshold(const unsigned long long* keys, uint64_t candidates, uint64_t need, int bits,
                                StoreSelectState* state, hipStream_t stream)
{
    hipLaunchKernelGGL(storeSelectResetKernel, dim3(1), dim3(256), 0, stream, state, static_cast<unsigned long long>(need));
    int top = bits < 1 ? 1 : (bits > 64 ? 64 : bits);
    while (top > 0)
    {
        const int width = top < kSelectDigitBits ? top : kSelectDigitBits;
        const int shift = top - width;
        hipLaunchKernelGGL(storeSelectCountKernel, dim3(selectBlocks(candidates)), dim3(kSelectThreads), 0, stream, keys,
                           candidates, shift, width, state);
        hipLaunchKernelGGL(storeSelectPickKernel, dim3(1), dim3(kSelectBins), 0, stream, shift, state);
        top = shift;
    }
}

void launchStoreSelectClassify(const unsigned long long* keys, uint64_t first, uint64_t candidates, uint64_t capacity,
                               StoreSelectState* state, uint32_t* holes, uint32_t* fillers, hipStream_t stream)
{
    hipLaunchKernelGGL(storeSelectClassifyKernel, dim3(selectBlocks(candidates)), dim3(kSelectThreads), 0, stream, keys, first,
                       candidates, capacity, state, holes, fillers);
}

void launchStoreSelectMove(float* records, unsigned long long* keys, const StoreSelectState* state, const uint32_t* holes,
                           const uint32_t* fillers, uint64_t pairs, hipStream_t stream)
{
    hipLaunchKernelGGL(storeSelectMoveKernel, dim3(selectBlocks(pairs)), dim3(kSelectThreads), 0, stream, records, keys,
                       state, holes, fillers);
}

}  // namespace rtd
