// rt_order.h — the emission order of a pass: which key a photon id gets, and the in-LDS step that
// finishes the order of large passes (rt_order.hip).
#pragma once

#include "rt_kernels.h"

namespace rtd
{

// A pass numbers its photons light by light: id = light * photonCount + (index - firstPhoton).
// Most scenes have one light and most passes fit 32 bits; the 64-bit division is the last resort.
RT_DEV void splitPhotonId(const TraceArgs& a, unsigned long long id, int* lightId, unsigned long long* photonIndex)
{
    if (id < a.photonCount)
    {
        *lightId = 0;
        *photonIndex = a.firstPhoton + id;
    }
    else if (((id | a.photonCount) >> 32) == 0)
    {
        const uint32_t light = static_cast<uint32_t>(id) / static_cast<uint32_t>(a.photonCount);
        *lightId = static_cast<int>(light);
        *photonIndex = a.firstPhoton + (static_cast<uint32_t>(id) - light * static_cast<uint32_t>(a.photonCount));
    }
    else
    {
        *lightId = static_cast<int>(id / a.photonCount);
        *photonIndex = a.firstPhoton + id % a.photonCount;
    }
}

// Emission cell of a photon, 24 bits: photons with nearly the same initial ray get neighbouring numbers.
// A light that emits from a point is keyed by the Morton number of its cell in a 4096 x 4096 grid over
// the two draws that steer the direction.  An area light also draws the ray's origin on its face: rays
// of one direction that start a face apart share nothing, so its key interleaves four draws, 64 cells
// each: origin (2) and direction (2).
// Draw k of the emission stage is word pair (k & 1) of Philox block (k >> 1) and a draw's leading bits
// are those of its first word (rt_u53), so floor(draw * 2^b) is that word's top b bits: no double is made.
constexpr int kGlobalAxisBits = 12;

RT_DEV uint32_t spreadBits12(uint32_t v)
{
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

RT_DEV uint32_t emissionCell(const TraceArgs& a, int lightId, unsigned long long photonIndex)
{
    const uint32_t low = static_cast<uint32_t>(photonIndex), high = static_cast<uint32_t>(photonIndex >> 32);
    uint32_t first[4] = {low, high, RT_STAGE_EMIT, 0u};
    rt_philox4x32_10(first, a.scene.seed, static_cast<uint32_t>(lightId));
    if (a.scene.lights[lightId].type == RT_LIGHT_AREA)
    {
        // draws 0 and 1 place the origin on the face, 2 and 3 steer the direction
        uint32_t second[4] = {low, high, RT_STAGE_EMIT, 1u};
        rt_philox4x32_10(second, a.scene.seed, static_cast<uint32_t>(lightId));
        const uint32_t c[4] = {first[0] >> 26, first[2] >> 26, second[0] >> 26, second[2] >> 26};
        uint32_t cell = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
        {
#pragma unroll
            for (int bit = 0; bit < 6; ++bit)
            {
                cell |= ((c[k] >> bit) & 1u) << (4 * bit + (3 - k));
            }
        }
        return cell;
    }
    const int drop = 32 - kGlobalAxisBits;
    return spreadBits12(first[0] >> drop) | (spreadBits12(first[2] >> drop) << 1);
}

// Large passes are ordered in three steps, each of them one trip of the pairs through memory (a radix
// sort over all 24 bits of the cell is three passes that read and write every pair, after a pass that
// reads every key to count its digits):
//   1. the ids are put in the order of bits 8-15 of their keys while the keys are made: every block takes
//      a tile of consecutive ids, works out their keys (kept) and counts them by that digit in LDS, the
//      counts go to a [digit][tile] matrix, one exclusive sum over the matrix is where every (digit, tile)
//      starts, and the tiles then place their (key, id) pairs, gathered by digit in LDS so that whole runs
//      are written;
//   2. a stable pass over the bits above: one like step 1 whose ranks are counted in arrival order (keys of
//      one light: eight bits), else the library's radix pass;
//   3. every run of kRefineRun consecutive positions of that list is put in the order of the whole key
//      inside LDS.  The order is a schedule (any permutation of the ids is a correct one): a wave works
//      through a run of at most kRefineRun positions, and inside such a run the ids are then exactly as
//      finely ordered as after a sort over all the bits.
constexpr int kRefineBits = 8;
constexpr uint32_t kRefineRun = 4096;
constexpr int kDigitBits = 8;               // step 1 sorts bits [kRefineBits, kRefineBits + kDigitBits)
constexpr uint32_t kDigitTile = 8192;       // ids per tile of step 1
inline uint64_t digitTiles(uint64_t n) { return (n + kDigitTile - 1) / kDigitTile; }
inline uint64_t digitMatrixEntries(uint64_t n) { return digitTiles(n) << kDigitBits; }
// Step 1.  keys: n words (every id's key: the light above the emission cell), counts / offsets:
// digitMatrixEntries(n) words each, keysOut / idsOut: the pairs by digit.  With a null `scratch` nothing
// runs and *scratchBytes receives the exclusive sum's work space.
// (shift: the first of the eight key bits it sorts — kRefineBits in the three steps; passes of one to two million ids
// sort the TOP eight bits here, skip step 2 and leave the other sixteen to step 3: a run of step 3 then lies in
// one or two values of those bits and its 2048 bins order it by eleven bits more, which is as fine as a few
// million photons are spread, for four launches fewer; at four million the trace already loses what the sort saves)
hipError_t digitOrder(const TraceArgs& args, int shift, void* scratch, size_t* scratchBytes, uint32_t* keys, uint32_t* counts,
                      uint32_t* offsets, uint32_t* keysOut, uint32_t* idsOut, hipStream_t stream);
// One stable pass of a sort of (key, id) pairs by the eight key bits from `shift` up (step 2 for keys of one
// light: shift 16; the passes of the parked photons' sort): keys / ids go to keysOut / idsOut, pairs of equal
// bits in the order they arrive in.  A null `ids` stands for 0, 1, 2 ...  counts / offsets: digitMatrixEntries(n)
// words each.  With a null `scratch` nothing runs and *scratchBytes receives the exclusive sum's work space.
hipError_t stableDigitPass(uint64_t n, int shift, void* scratch, size_t* scratchBytes, const uint32_t* keys, const uint32_t* ids,
                           uint32_t* counts, uint32_t* offsets, uint32_t* keysOut, uint32_t* idsOut, hipStream_t stream);
// Step 3.  keys / ids: the pairs in ascending order of (key >> sortedShift); order: n words (the result)
void launchRefineOrder(const uint32_t* keys, const uint32_t* ids, uint64_t n, int sortedShift, uint32_t* order, hipStream_t stream);

}  // namespace rtd
