// Radix sort of (24-bit key, id) pairs as the emission order does it, against the digit width of a pass.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fill(uint32_t* keys, uint32_t* ids, size_t n)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u;
    x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    keys[i] = x & 0xFFFFFF;
    ids[i] = (uint32_t)i;
}

template <class Config>
int run(const char* name, uint32_t* kin, uint32_t* kout, uint32_t* vin, uint32_t* vout, size_t n, unsigned bits)
{
    size_t bytes = 0;
    CHECK((rocprim::radix_sort_pairs<Config>(nullptr, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
    void* tmp = nullptr;
    CHECK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int it = 0; it < 4; ++it)
    {
        CHECK(hipEventRecord(a));
        CHECK((rocprim::radix_sort_pairs<Config>(tmp, bytes, kin, kout, vin, vout, n, 0u, bits, (hipStream_t)0)));
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    std::vector<uint32_t> h(1 << 16);
    CHECK(hipMemcpy(h.data(), kout + (n / 2), h.size() * 4, hipMemcpyDeviceToHost));
    bool sorted = true;
    for (size_t i = 1; i < h.size(); ++i) sorted = sorted && h[i - 1] <= h[i];
    printf("%-28s n=%zu bits=%u  %.3f ms  scratch %.1f MB  %s\n", name, n, bits, best, bytes / 1e6, sorted ? "sorted" : "NOT SORTED");
    CHECK(hipFree(tmp));
    return 0;
}

template <unsigned Bits, unsigned Block, unsigned Items, rocprim::block_radix_rank_algorithm Algo>
using Sweep = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config,
                                         rocprim::radix_sort_onesweep_config<rocprim::kernel_config<256, 12>,
                                                                             rocprim::kernel_config<Block, Items>, Bits, Algo>>;

int main(int argc, char** argv)
{
    size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 200000000ull;
    unsigned bits = argc > 2 ? (unsigned)atoi(argv[2]) : 24u;
    uint32_t *kin, *kout, *vin, *vout;
    CHECK(hipMalloc(&kin, n * 4)); CHECK(hipMalloc(&kout, n * 4)); CHECK(hipMalloc(&vin, n * 4)); CHECK(hipMalloc(&vout, n * 4));
    fill<<<(n + 255) / 256, 256>>>(kin, vin, n);
    CHECK(hipDeviceSynchronize());
    using rocprim::block_radix_rank_algorithm;
    if (run<rocprim::default_config>("default", kin, kout, vin, vout, n, bits)) return 1;
    if (run<Sweep<8, 1024, 8, block_radix_rank_algorithm::match>>("8 bits 1024x8 match", kin, kout, vin, vout, n, bits)) return 1;
    if (run<Sweep<8, 512, 16, block_radix_rank_algorithm::match>>("8 bits 512x16 match", kin, kout, vin, vout, n, bits)) return 1;
    if (run<Sweep<6, 1024, 8, block_radix_rank_algorithm::match>>("6 bits 1024x8 match", kin, kout, vin, vout, n, bits)) return 1;
#ifdef WIDE
    if (run<Sweep<WIDE, 1024, 8, block_radix_rank_algorithm::WIDE_ALGO>>("wide", kin, kout, vin, vout, n, bits)) return 1;
#endif
    return 0;
}
