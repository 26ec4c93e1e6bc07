// rt_kernels.hip — the photon pass and the function-level kernels (gfx950).
//
// traceKernel is a persistent-thread tracer: the grid is sized to fill the chip
// once, every wave pulls chunks of photon ids from one global counter, and every
// LANE refills itself with a fresh photon as soon as its own photon terminates, so
// all 64 lanes of a wave execute the same stage — one bounce (cast, deposit,
// terminate-or-scatter) — on every trip round the loop whatever the individual
// path lengths are.  It replaces Worker::exec / processLights / processPhotons
// (reference src/Worker.cpp:116-174,429-680); the WorkQueue/LightQueue machinery
// that bounded in-flight photons on the CPU (reference src/WorkQueue.cpp,
// src/LightQueue.cpp) reduces to that counter, because photons are generated in
// registers and never stored.

#include "rt_kernels.h"
#include "rt_order.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include <hipcub/hipcub.hpp>
#include <hipcub/device/device_merge.hpp>
#include <rocprim/rocprim.hpp>

namespace rtd
{

namespace
{

constexpr int kTraceThreads = 256;
constexpr double kSelfHitThreshold = 1e-4;   // reference src/Worker.cpp:36

RT_DEV unsigned lanesBelow(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<unsigned>(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo(static_cast<unsigned>(mask), 0u));
}

// The value `v` holds in lane `lane` of the wave (the same lane for every caller: a scalar read, not a
// trip through the LDS crossbar).
RT_DEV float laneValue(float v, int lane)
{
    return __uint_as_float(static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(__float_as_uint(v)), lane)));
}

RT_DEV uint32_t laneValue(uint32_t v, int lane)
{
    return static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(v), lane));
}

RT_DEV unsigned long long waveSum(unsigned long long v)
{
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1)
    {
        v += __shfl_down(v, offset, 64);
    }
    return v;
}

RT_DEV double waveSumD(double v)
{
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1)
    {
        v += __shfl_down(v, offset, 64);
    }
    return v;
}

constexpr int64_t kPackLimit = 1 << 20;  // packed cell coordinates lie in [-2^20, 2^20)

RT_DEV uint64_t cellHash(int64_t kx, int64_t ky, int64_t kz)  // reference include/ProbeIndex.h:71-80
{
    uint64_t h = static_cast<uint64_t>(kx) * 73856093ull;
    h ^= static_cast<uint64_t>(ky) * 19349663ull;
    h ^= static_cast<uint64_t>(kz) * 83492791ull;
    return (h * 0x9E3779B97F4A7C15ull) >> 20;
}

// Is a probe of cell (kx, ky, kz) within the keep radius of p?
RT_DEV bool probeCellHas(const ProbeView& pv, int64_t kx, int64_t ky, int64_t kz, const Vec& p, double r2)
{
    uint64_t slot = cellHash(kx, ky, kz) & pv.slotMask;
    while (true)
    {
        const uint32_t count = pv.slotCount[slot];
        if (count == 0)
        {
            return false;
        }
        const int64_t* key = pv.slotKey + 3 * slot;
        if (key[0] == kx && key[1] == ky && key[2] == kz)
        {
            const double* pt = pv.points + 3ull * pv.slotStart[slot];
            for (uint32_t i = 0; i < count; ++i)
            {
                const double ddx = pt[3 * i] - p.x, ddy = pt[3 * i + 1] - p.y, ddz = pt[3 * i + 2] - p.z;
                if (ddx * ddx + ddy * ddy + ddz * ddz <= r2)
                {
                    return true;
                }
            }
            return false;
        }
        slot = (slot + 1) & pv.slotMask;
    }
}

RT_DEV unsigned long long packProbeCell(int64_t kx, int64_t ky, int64_t kz)
{
    return (static_cast<unsigned long long>(kx + kPackLimit) << 42) | (static_cast<unsigned long long>(ky + kPackLimit) << 21) |
           static_cast<unsigned long long>(kz + kPackLimit);
}

// The key the probes are sorted by when the grid is built in its brick form: cells ordered by
// (x >> 1, y >> 1, z, y & 1, x & 1), the order of the dense cell array (see ProbeView).
RT_DEV unsigned long long orderedProbeCell(int64_t kx, int64_t ky, int64_t kz)
{
    const unsigned long long x = static_cast<unsigned long long>(kx + kPackLimit);
    const unsigned long long y = static_cast<unsigned long long>(ky + kPackLimit);
    const unsigned long long z = static_cast<unsigned long long>(kz + kPackLimit);
    return ((x >> 1) << 43) | ((y >> 1) << 23) | (z << 2) | ((y & 1) << 1) | (x & 1);
}

// The keep test through the brick table.  Every lane walks its own chain of dependent reads (brick entry
// -> cell -> further probes of the cell -> next cell ...), and a wave pays a memory round trip and the
// instructions of a trip for every trip of its slowest lane, so a trip is kept lean: the brick entry is
// the only hashed lookup — it tells which of the 27 cells around p hold probes and where each of them
// lies in the dense cell array — and which of those cells are worth a look is settled once, before the
// walk (those whose box lies farther from p than the keep radius are not: gaps in cell units, in single
// precision with 1e-5 of slack, the bound only has to be safe), as a mask in the brick's own bit
// positions.  The walk then takes the deposit's own cell first (on a surface the camera sees it usually
// holds a probe in reach) and the others in the order of their bits.
RT_DEV bool anyProbeInNeighbourhood(const ProbeView& pv, const Vec& p, double r2, int64_t cx, int64_t cy, int64_t cz)
{
    typedef uint32_t Uint4At8 __attribute__((ext_vector_type(4), aligned(8)));
    // the brick that holds the 27 cells around (cx, cy, cz), and where in it the first of them lies (0 or 1)
    const int64_t mx = (cx - 1) >> 1, my = (cy - 1) >> 1, mz = (cz - 1) >> 1;
    const int ox = static_cast<int>(cx - 1 - 2 * mx), oy = static_cast<int>(cy - 1 - 2 * my), oz = static_cast<int>(cz - 1 - 2 * mz);
    const unsigned long long key = packProbeCell(mx, my, mz);
    uint64_t slot = cellHash(mx, my, mz) & pv.brickMask;
    unsigned long long occupied = 0;
    uint32_t base0 = 0, base1 = 0, base2 = 0, base3 = 0;
    while (true)
    {
        RT_DIAG_TRIP(12);
        const char* address = reinterpret_cast<const char*>(pv.bricks + slot);
        const Uint4At8 head = *reinterpret_cast<const Uint4At8*>(address);
        const Uint4At8 mid = *reinterpret_cast<const Uint4At8*>(address + 16);
        const unsigned long long first = (static_cast<unsigned long long>(head.y) << 32) | head.x;
        if (first == key)
        {
            occupied = ~((static_cast<unsigned long long>(head.w) << 32) | head.z);
            base0 = mid.x;
            base1 = mid.y;
            base2 = mid.z;
            base3 = mid.w;
            break;
        }
        if (first == kEmptyKey)
        {
            return false;  // no occupied cell around p
        }
        slot = (slot + 1) & pv.brickMask;
    }
    // the cells of the neighbourhood in reach of p, as bits of the brick
    const float fx = static_cast<float>(p.x * pv.invCellSize - static_cast<double>(cx));
    const float fy = static_cast<float>(p.y * pv.invCellSize - static_cast<double>(cy));
    const float fz = static_cast<float>(p.z * pv.invCellSize - static_cast<double>(cz));
    const float reach2 = static_cast<float>(r2 * pv.invCellSize * pv.invCellSize) + 1e-5f;
    const float lowX = fx * fx, highX = (1.0f - fx) * (1.0f - fx);
    const float gapY[3] = {fy * fy, 0.0f, (1.0f - fy) * (1.0f - fy)};
    const float gapZ[3] = {fz * fz, 0.0f, (1.0f - fz) * (1.0f - fz)};
    unsigned long long worth = 0;
#pragma unroll
    for (int iz = 0; iz < 3; ++iz)
    {
#pragma unroll
        for (int iy = 0; iy < 3; ++iy)
        {
            const float budget = reach2 - (gapY[iy] + gapZ[iz]);
            const uint32_t row = (lowX <= budget ? 1u : 0u) | (0.0f <= budget ? 2u : 0u) | (highX <= budget ? 4u : 0u);
            worth |= static_cast<unsigned long long>(row) << (ox + 4 * (oy + iy) + 16 * (oz + iz));
        }
    }
    unsigned long long candidates = occupied & worth;
    const int centre = (ox + 1) + 4 * (oy + 1) + 16 * (oz + 1);
    while (candidates != 0)
    {
        RT_DIAG_TRIP(12);
        const int bit = ((candidates >> centre) & 1ull) != 0 ? centre : __builtin_ctzll(candidates);
        candidates &= ~(1ull << bit);
        // the cell's place: its column's base + the occupied cells of the column below its bit
        // (column (x >> 1) + 2 (y >> 1): bits 1 and 3 of the cell's bit number)
        const unsigned long long inColumn = 0x0033003300330033ull << (bit & 10);
        const unsigned long long below = (1ull << bit) - 1ull;
        const int column = ((bit >> 1) & 1) + ((bit >> 2) & 2);
        const uint32_t base = column == 0 ? base0 : (column == 1 ? base1 : (column == 2 ? base2 : base3));
        const char* address =
            reinterpret_cast<const char*>(pv.cells + (base + static_cast<uint32_t>(__builtin_popcountll(occupied & inColumn & below))));
        const Uint4At8 head = *reinterpret_cast<const Uint4At8*>(address);
        const Uint4At8 mid = *reinterpret_cast<const Uint4At8*>(address + 16);
        const double qx = __builtin_bit_cast(double, (static_cast<unsigned long long>(head.y) << 32) | head.x);
        const double qy = __builtin_bit_cast(double, (static_cast<unsigned long long>(head.w) << 32) | head.z);
        const double qz = __builtin_bit_cast(double, (static_cast<unsigned long long>(mid.y) << 32) | mid.x);
        const double ddx = qx - p.x, ddy = qy - p.y, ddz = qz - p.z;
        if (ddx * ddx + ddy * ddy + ddz * ddz <= r2)
        {
            return true;
        }
        // the further probes of a cell that holds several follow its first in `points`: four a trip (a lane
        // that walks a crowded cell holds up its whole wave)
        const double* pt = pv.points + 3ull * mid.z;
        constexpr uint32_t kTogether = 4;
        for (uint32_t i = 1; i < mid.w; i += kTogether)
        {
            RT_DIAG_TRIP(12);
            double q[kTogether][3];
#pragma unroll
            for (uint32_t k = 0; k < kTogether; ++k)
            {
                const double* at = pt + 3 * (i + k < mid.w ? i + k : i);
                q[k][0] = at[0];
                q[k][1] = at[1];
                q[k][2] = at[2];
            }
            bool any = false;
#pragma unroll
            for (uint32_t k = 0; k < kTogether; ++k)
            {
                const double ex = q[k][0] - p.x, ey = q[k][1] - p.y, ez = q[k][2] - p.z;
                any = any || ex * ex + ey * ey + ez * ez <= r2;
            }
            if (any)
            {
                return true;
            }
        }
    }
    return false;
}

// ProbeIndex::anyWithin, reference src/ProbeIndex.cpp:27-68.  The result is a pure
// "exists", so the scan order is free; the squared-distance test is the reference's
// expression.  In the brick form of the grid one lookup tells which of the 27 cells
// around p hold probes at all: a deposit far from every probe costs that one lookup.
RT_DEV bool anyProbeWithin(const ProbeView& pv, const Vec& p)
{
    const double r = pv.keepRadius;
    if (r <= 0.0 || pv.probeCount == 0)
    {
        return false;
    }
    const double r2 = r * r;
    const int64_t cx = static_cast<int64_t>(__builtin_floor(p.x * pv.invCellSize));
    const int64_t cy = static_cast<int64_t>(__builtin_floor(p.y * pv.invCellSize));
    const int64_t cz = static_cast<int64_t>(__builtin_floor(p.z * pv.invCellSize));
    if (pv.brickMask != 0)
    {
        // probe cells lie strictly inside the packable range, so a cell on or beyond
        // its border has no occupied neighbour
        if (cx <= -kPackLimit || cx >= kPackLimit - 1 || cy <= -kPackLimit || cy >= kPackLimit - 1 ||
            cz <= -kPackLimit || cz >= kPackLimit - 1)
        {
            return false;
        }
        return anyProbeInNeighbourhood(pv, p, r2, cx, cy, cz);
    }
    const int64_t reach = pv.reach;
    for (int64_t dz = -reach; dz <= reach; ++dz)
    {
        for (int64_t dy = -reach; dy <= reach; ++dy)
        {
            for (int64_t dx = -reach; dx <= reach; ++dx)
            {
                if (probeCellHas(pv, cx + dx, cy + dy, cz + dz, p, r2))
                {
                    return true;
                }
            }
        }
    }
    return false;
}

// BounceStore::append, reference src/BounceStore.cpp:11-20, with the cursor claimed
// once per wave: the lanes that append in this step count themselves with a ballot,
// one lane adds the total, each lane takes base + its rank.
// A record's place in the order the reference's workers meet the photons in — light by light, index by
// index, bounce by bounce — as one number: what an overflowing store is put in order by.  The photon's
// number in the pass (light x photons per light + index - first index: 32 bits, larger passes are not
// ordered) sits above StoreView::keyBits bits of bounce.
struct StoreOrder
{
    uint32_t photonIndexLow, firstPhotonLow, photonsPerLightLow;  // low words: the pass holds fewer than 2^32 ids
    int lightId, bounces;
    RT_DEV unsigned long long key(const StoreView& sv) const
    {
        const uint32_t idInPass = static_cast<uint32_t>(lightId) * photonsPerLightLow + (photonIndexLow - firstPhotonLow);
        return (static_cast<unsigned long long>(idInPass) << sv.keyBits) | static_cast<unsigned long long>(bounces);
    }
};

RT_DEV void storeAppend(const StoreView& sv, bool keep, const Vec& position, const Vec& incoming, const Vec& normal,
                        float time, const Color& power, const StoreOrder& order)
{
    const unsigned long long mask = __ballot(keep);
    if (mask == 0)
    {
        return;
    }
    const int leader = __builtin_ctzll(mask);
    unsigned long long base = 0;
    if (static_cast<int>(threadIdx.x & 63) == leader)
    {
        base = atomicAdd(sv.cursor, static_cast<unsigned long long>(__builtin_popcountll(mask)));
    }
    base = __shfl(base, leader, 64);
    if (keep)
    {
        const unsigned long long slot = base + lanesBelow(mask);
        if (slot < sv.room)
        {
            // 52-byte records are only 4-byte aligned; the hardware takes unaligned 16-byte stores,
            // so a record is three of them and one float instead of thirteen floats
            typedef float UnalignedFloat4 __attribute__((ext_vector_type(4), aligned(4)));
            float* r = sv.records + 13ull * slot;
            UnalignedFloat4 q0, q1, q2;
            q0.x = static_cast<float>(position.x);
            q0.y = static_cast<float>(position.y);
            q0.z = static_cast<float>(position.z);
            q0.w = static_cast<float>(incoming.x);
            q1.x = static_cast<float>(incoming.y);
            q1.y = static_cast<float>(incoming.z);
            q1.z = static_cast<float>(normal.x);
            q1.w = static_cast<float>(normal.y);
            q2.x = static_cast<float>(normal.z);
            q2.y = time;
            q2.z = power.r;
            q2.w = power.g;
            *reinterpret_cast<UnalignedFloat4*>(r) = q0;
            *reinterpret_cast<UnalignedFloat4*>(r + 4) = q1;
            *reinterpret_cast<UnalignedFloat4*>(r + 8) = q2;
            r[12] = power.b;
            if (sv.keys) sv.keys[slot] = order.key(sv);
        }
        else if (sv.keys)
        {
            atomicMin(sv.keys - 1, order.key(sv));
        }
    }
}

// Tallies of one launch.  They live in LDS, not in registers (the trace kernels have none to
// spare): one set of event counters per WAVE — an event is counted for all the lanes it happens in
// at once, by a ballot and one addition of the lowest such lane — and the deposited power per
// lane.  Each wave adds its set to the pass counters when it ends.
enum TallySlot
{
    kTallyPhotons = 0, kTallyCasts, kTallyBounces, kTallyDeposits, kTallyKept, kTallyCulled,
    kTallySplatTotal, kTallySplatRadiusClamped, kTallySplatLuminanceClamped,
    kTallyDensityCells, kTallyDensityDeposits, kTallyDensityOverflow, kTallySlots
};

struct WaveTally
{
    unsigned int* slots;  // [kTallySlots] of this wave
    double* power;        // this lane's three sums, kTraceThreads apart
};

// Counts the lanes that execute this call with `happened` set.
RT_DEV void tally(const WaveTally& t, int slot, bool happened = true)
{
    const unsigned long long mask = __ballot(happened);
    if (happened && static_cast<int>(threadIdx.x & 63) == __builtin_ctzll(mask))
    {
        // different lanes of the wave add at different times: an LDS atomic, so that no lane works on a remembered value
        __hip_atomic_fetch_add(t.slots + slot, static_cast<unsigned int>(__builtin_popcountll(mask)), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// DensityGrid::add, reference src/DensityGrid.cpp:25-34.
RT_DEV void densityAdd(const DensityView& dv, const Vec& p, const Color& power, const WaveTally& lc)
{
    const int64_t cx = static_cast<int64_t>(__builtin_floor(p.x * dv.invCellSize));
    const int64_t cy = static_cast<int64_t>(__builtin_floor(p.y * dv.invCellSize));
    const int64_t cz = static_cast<int64_t>(__builtin_floor(p.z * dv.invCellSize));
    const int64_t lim = 1 << 20;
    if (cx < -lim || cx >= lim || cy < -lim || cy >= lim || cz < -lim || cz >= lim)
    {
        tally(lc, kTallyDensityOverflow);
        return;
    }
    const unsigned long long key = (static_cast<unsigned long long>(cx + lim) << 42) |
                                   (static_cast<unsigned long long>(cy + lim) << 21) |
                                   static_cast<unsigned long long>(cz + lim);
    uint64_t h = static_cast<uint64_t>(cx) * 73856093ull;
    h ^= static_cast<uint64_t>(cy) * 19349663ull;
    h ^= static_cast<uint64_t>(cz) * 83492791ull;
    uint64_t slot = (h * 0x9E3779B97F4A7C15ull) >> 20 & dv.slotMask;
    // a table that is (nearly) full must not turn every deposit into a walk over all of it: after
    // this many occupied slots in a row the deposit counts as overflow
    constexpr uint64_t kMostProbes = 512;
    for (uint64_t tries = 0; tries <= dv.slotMask && tries < kMostProbes; ++tries)
    {
        const unsigned long long seen = atomicCAS(dv.keys + slot, kEmptyKey, key);
        if (seen == kEmptyKey || seen == key)
        {
            tally(lc, kTallyDensityCells, seen == kEmptyKey);
            atomicAdd(dv.power + 3 * slot + 0, power.r);
            atomicAdd(dv.power + 3 * slot + 1, power.g);
            atomicAdd(dv.power + 3 * slot + 2, power.b);
            atomicAdd(dv.count + slot, 1ull);
            return;
        }
        slot = (slot + 1) & dv.slotMask;
    }
    tally(lc, kTallyDensityOverflow);
}

// Buffer::addColor (reference src/Buffer.cpp:22-37) for the lanes of a wave at once.  Photons that
// left the light together land in the same few pixels: the lanes that add to the same pixel are found
// with ballots (the lowest pending lane names its pixel, the lanes with that pixel answer), their
// contributions are added up in lane order, and one lane issues the three atomics for all of them.  A
// pixel only one lane adds to costs the ballots and nothing else.  Every lane of the wave that is
// active at the call takes part; `emit` says whether it has something to add.
RT_DEV void addToPixels(float* buffer, bool emit, uint32_t index, const Color& contribution, bool together)
{
    if (!together)
    {
        // every lane for itself (kept for measurements of what adding together saves)
        if (emit)
        {
            atomicAdd(buffer + index + 0, contribution.r);
            atomicAdd(buffer + index + 1, contribution.g);
            atomicAdd(buffer + index + 2, contribution.b);
        }
        return;
    }
    unsigned long long pending = __ballot(emit);
    // A wave whose lanes all add to different pixels (a large image, photons deep in their paths) would
    // go round once per lane for nothing: after a few rounds the lanes still pending add on their own.
    constexpr int kRounds = 6;
    for (int round = 0; pending != 0; ++round)
    {
        const int leader = __builtin_ctzll(pending);
        const uint32_t pixel = laneValue(index, leader);
        const bool same = emit && index == pixel;
        unsigned long long members = __ballot(same);
        const bool lastRound = round == kRounds - 1;
        const bool alone = (members & (members - 1)) == 0;
        if (lastRound || alone)
        {
            if (lastRound ? (emit && ((pending >> (threadIdx.x & 63)) & 1) != 0) : same)
            {
                atomicAdd(buffer + index + 0, contribution.r);
                atomicAdd(buffer + index + 1, contribution.g);
                atomicAdd(buffer + index + 2, contribution.b);
            }
            if (lastRound) break;
            pending &= ~members;
            continue;
        }
        pending &= ~members;
        float r = 0.0f, g = 0.0f, b = 0.0f;
        while (members != 0)
        {
            const int k = __builtin_ctzll(members);
            members &= members - 1;
            r += laneValue(contribution.r, k);
            g += laneValue(contribution.g, k);
            b += laneValue(contribution.b, k);
        }
        if (static_cast<int>(threadIdx.x & 63) == leader)
        {
            atomicAdd(buffer + pixel + 0, r);
            atomicAdd(buffer + pixel + 1, g);
            atomicAdd(buffer + pixel + 2, b);
        }
    }
}

// Worker::splatToCamera, reference src/Worker.cpp:221-427, with
// Camera::coordForPointSubPixel (src/Camera.cpp:239-310), Utility::flooredSplatRadius
// (src/Utility.cpp:43-50) and Buffer::addColor (src/Buffer.cpp:22-37).  Every lane that is active at
// the call goes through every camera (the pixel sums are formed across the wave); `splats` is false
// for a lane whose hit deposits nothing (a delta surface).
template <bool Animated, bool Grouped>
RT_DEV void splatToCamera(const TraceArgs& a, bool splats, const Vec& incoming, const Color& photonColor, int bounces,
                          int lightId, float time, const Hit& hit, const Material& material, const WaveTally& lc)
{
    const SceneView& s = a.scene;
    if (static_cast<double>(s.photonsPerLight) <= 0.0)
    {
        return;
    }
    const Vec wi = neg(incoming);
    for (int32_t c = 0; c < s.cameraCount; ++c)
    {
        const Camera& cam = s.cameras[c];
        if (cam.width <= 0 || cam.height <= 0 || cam.halfHeight <= 0.0) continue;  // the same for every lane
        const uint64_t w = static_cast<uint64_t>(cam.width), h = static_cast<uint64_t>(cam.height);
        bool emit = splats;
        if (cam.bounceFilter >= 0 && bounces != cam.bounceFilter) emit = false;
        if (cam.lightFilter >= 0 && lightId != cam.lightFilter) emit = false;
        uint32_t index = 0;
        Color clamped{0.0f, 0.0f, 0.0f};
        if (emit)
        {
            emit = false;
            do
            {
                const Vec eye = mk(cam.pos[0], cam.pos[1], cam.pos[2]);
                const Vec toPoint = hit.position - eye;
                const double depth = dot(toPoint, mk(cam.forward[0], cam.forward[1], cam.forward[2]));
                if (depth <= 0.0) break;
                const double dx = dot(toPoint, mk(cam.right[0], cam.right[1], cam.right[2]));
                const double dy = dot(toPoint, mk(cam.up[0], cam.up[1], cam.up[2]));
                const double sx = (dx / depth) / (cam.aspect * cam.halfHeight);
                const double sy = (dy / depth) / cam.halfHeight;
                if (sx < -1.0 || sx > 1.0 || sy < -1.0 || sy > 1.0) break;
                const double subX = ((sx + 1.0) / 2.0) * static_cast<double>(cam.width);
                const double subY = ((sy + 1.0) / 2.0) * static_cast<double>(cam.height);
                uint64_t px = static_cast<uint64_t>(dmax(0.0, __builtin_floor(subX)));
                uint64_t py = static_cast<uint64_t>(dmax(0.0, __builtin_floor(subY)));
                px = px < w - 1 ? px : w - 1;
                py = py < h - 1 ? py : h - 1;

                if (!(time >= cam.windowStart && time < cam.windowEnd)) break;

                const Vec toCamera = eye - hit.position;
                const double cameraDistance = magnitude(toCamera);
                if (cameraDistance < kSelfHitThreshold) break;
                const Vec toCameraDir = toCamera / cameraDistance;
                const double cosCamera = dot(toCameraDir, hit.normal);
                if (cosCamera <= 0.0) break;

                Ray shadow;
                shadow.o = hit.position;
                shadow.d = toCameraDir;
                Hit occluder;
                if (castScene<Animated, true, Grouped>(s, shadow, time, kSelfHitThreshold, false, &occluder) &&
                    occluder.distance < cameraDistance)
                {
                    break;
                }

                const double rawRadius = cameraDistance * cam.tanPixelHalfAngle;
                const double r = (a.minSplatRadius > 0.0 && rawRadius < a.minSplatRadius) ? a.minSplatRadius : rawRadius;
                tally(lc, kTallySplatTotal);
                tally(lc, kTallySplatRadiusClamped, r > rawRadius);
                if (r <= 0.0) break;

                const Color brdf = evaluateMaterial(material, wi, toCameraDir, hit.normal);
                const double area = kPi * r * r;
                const float scale = static_cast<float>(cosCamera / area);
                const Color contribution = brdf * photonColor * scale;
                if (brightness(contribution) <= 0.0f) break;

                clamped = contribution;
                if (s.splatLuminanceClamp > 0.0)
                {
                    const double b = static_cast<double>(brightness(contribution));
                    if (b > s.splatLuminanceClamp)
                    {
                        clamped = contribution * static_cast<float>(s.splatLuminanceClamp / b);
                        tally(lc, kTallySplatLuminanceClamped);
                    }
                }
                const uint64_t at = ((py * w) + px) * 3;
                if (at + 2 >= w * h * 3) break;
                index = static_cast<uint32_t>(at);
                emit = true;
            } while (false);
        }
        addToPixels(cam.buffer, emit, index, clamped, a.splatTogether != 0);
    }
}

// A pass is traced in EMISSION ORDER rather than id order (TraceArgs::emissionOrder): the 64
// photons a wave holds at any time leave the light in nearly the same direction, walk nearly
// the same nodes and finish their casts together.  A photon's path depends only on its id
// (the counter-based draws), so the order changes nothing but the schedule.

RT_DEV unsigned long long photonAt(const TraceArgs& a, unsigned long long position)
{
    return a.emissionOrder ? static_cast<unsigned long long>(a.emissionOrder[position]) : position;
}

// Whole-pass emission order: every id gets a key that puts photons with nearly the same initial ray
// next to each other (emissionCell), above it the light (so that lights do not interleave); a radix
// sort of (key, id) then orders the whole pass.
__global__ void emissionKeysKernel(const TraceArgs a, uint32_t* keys, uint32_t* ids)
{
    const unsigned long long id = static_cast<unsigned long long>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (id >= a.totalWork) return;
    int lightId;
    unsigned long long photonIndex;
    splitPhotonId(a, id, &lightId, &photonIndex);
    keys[id] = (static_cast<uint32_t>(lightId & 0xFF) << (2 * kGlobalAxisBits)) | emissionCell(a, lightId, photonIndex);
    ids[id] = static_cast<uint32_t>(id);
}

// Key of a parked photon's ray: the cell of its origin in a grid over the scene's box (Morton order, so
// that neighbouring keys are neighbouring cells) and the sector of its direction (one of six faces by
// the dominant axis and its sign, then the two other components relative to it in n x n cells).
// keyMode packs the layout: bits 0-3 the grid's bits per axis (0: no origin part), bits 4-7 n, bit 8
// set when the direction sector goes above the origin cell instead of below it.
RT_DEV uint32_t spreadBits3(uint32_t v)  // bit i of a 6-bit value -> bit 3 * i
{
    return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8) | ((v & 32u) << 10);
}

RT_DEV uint32_t rayKey(const TraceArgs& a, const Ray& ray)
{
    const int originBits = a.keyMode & 15, sectorCells = (a.keyMode >> 4) & 15;
    const bool directionAbove = (a.keyMode & 256) != 0;
    uint32_t morton = 0;
    if (originBits > 0)
    {
        const float cells = static_cast<float>(1 << originBits);
        uint32_t cell[3];
        const double o[3] = {ray.o.x, ray.o.y, ray.o.z};
#pragma unroll
        for (int k = 0; k < 3; ++k)
        {
            const float f = (static_cast<float>(o[k]) - a.keyOrigin[k]) * a.keyScale * cells;
            cell[k] = static_cast<uint32_t>(__builtin_fminf(__builtin_fmaxf(f, 0.0f), cells - 1.0f));
        }
        morton = spreadBits3(cell[0]) | (spreadBits3(cell[1]) << 1) | (spreadBits3(cell[2]) << 2);
    }
    if (sectorCells == 0)
    {
        return morton;
    }
    const float d[3] = {static_cast<float>(ray.d.x), static_cast<float>(ray.d.y), static_cast<float>(ray.d.z)};
    const float ax = __builtin_fabsf(d[0]), ay = __builtin_fabsf(d[1]), az = __builtin_fabsf(d[2]);
    const int major = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const float dm = major == 0 ? d[0] : (major == 1 ? d[1] : d[2]);
    const float du = major == 0 ? d[1] : d[0], dv = major == 2 ? d[1] : d[2];
    const uint32_t face = static_cast<uint32_t>(2 * major + (dm < 0.0f ? 1 : 0));  // 0..5
    const float inv = 1.0f / __builtin_fmaxf(__builtin_fabsf(dm), 1e-20f);
    const float n = static_cast<float>(sectorCells);
    const uint32_t cu = static_cast<uint32_t>(__builtin_fminf(__builtin_fmaxf((du * inv + 1.0f) * 0.5f * n, 0.0f), n - 1.0f));
    const uint32_t cv = static_cast<uint32_t>(__builtin_fminf(__builtin_fmaxf((dv * inv + 1.0f) * 0.5f * n, 0.0f), n - 1.0f));
    // snake order over the face's cells keeps neighbouring sector numbers neighbouring directions
    const uint32_t column = (cu & 1u) ? static_cast<uint32_t>(sectorCells) - 1u - cv : cv;
    const uint32_t sector = (face * static_cast<uint32_t>(sectorCells) + cu) * static_cast<uint32_t>(sectorCells) + column;
    int sectorBits = 0;
    while ((1u << sectorBits) < 6u * static_cast<uint32_t>(sectorCells * sectorCells)) ++sectorBits;
    return directionAbove ? (sector << (3 * originBits)) | morton : (morton << sectorBits) | sector;
}

// Sink selects the deposit path and Record the per-photon path dump at compile time, so
// that the probe-sink kernel carries neither the splat's shadow-ray cast nor the dump.
// Grouped: for scenes that cast several flat meshes of one pose at once (castScene); the kernels of the
// scenes without such casts are compiled without their handling.
template <int Sink, bool Record, bool Animated, bool Grouped>
__global__ __launch_bounds__(kTraceThreads) __attribute__((amdgpu_waves_per_eu((Animated && !Record) ? 2 : 1))) void traceKernel(const TraceArgs a)
{
    const SceneView& s = a.scene;
    const int lane = threadIdx.x & 63;
    RT_DIAG_WAVE_BEGIN;
    RT_DIAG_TIC(kernelStart);
    fillSceneCache(s);

    // work of this launch: the pass's emission-ordered id list, or the photons parked by the
    // launch of the previous bounce
    unsigned long long totalWork = a.totalWork;
    if (a.queueIn)
    {
        totalWork = a.queueInCount;
    }
    // this wave's run of positions in that list
    unsigned long long chunkBase = 0;     // wave-uniform
    uint32_t chunkNext = 0, chunkCount = 0;
    bool exhausted = false;               // the global id range is used up (wave-uniform)

    bool alive = false, finished = false;
    Ray ray;
    ray.o = mk(0, 0, 0);
    ray.d = mk(0, 0, 1);
    Color color{0.0f, 0.0f, 0.0f};
    int bounces = 0, lightId = 0;
    float time = 0.0f;
    unsigned long long photonIndex = 0;
    long long recordSlot = -1;
    int casts = 0, valid = 0;
    __shared__ unsigned int tallySlots[kTraceThreads / 64][kTallySlots];
    __shared__ double tallyPower[3][kTraceThreads];
    const WaveTally lc{tallySlots[threadIdx.x >> 6], &tallyPower[0][threadIdx.x]};
    if (lane < kTallySlots) lc.slots[lane] = 0;
    lc.power[0] = lc.power[kTraceThreads] = lc.power[2 * kTraceThreads] = 0.0;
    Rng rng;
    rng.tape = nullptr;
    rng.next = 0;
    rng.seed = s.seed;

    while (true)
    {
        // ---- refill idle lanes (Worker::processLights + Light::emit) ----
        // two passes: when the current chunk runs out in the middle of a refill, the rest of
        // the lanes are served from the next chunk
        RT_DIAG_TIC(refillStart);
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass)
        {
            const bool need = !alive && !finished;
            const unsigned long long needMask = __ballot(need);
            if (needMask == 0)
            {
                break;
            }
            if (chunkNext >= chunkCount && !exhausted)
            {
                unsigned long long fresh = 0;
                if (lane == 0)
                {
                    fresh = atomicAdd(a.workCounter, 1ull);  // the number of this claim
                }
                const unsigned long long claim = __shfl(fresh, 0, 64);
                const bool full = claim < a.fullClaims;
                const unsigned long long size = full ? a.chunk : a.tailChunk;
                // Runs of the emission-ordered list are taken in a scattered order: what is in the
                // store when it overflows is then spread over the whole emission domain, as in the
                // reference, whose workers meet the photons in index order (a run on its own stays
                // coherent, which is all a wave needs).
                const unsigned long long run = a.claimStride > 1 ? (claim * a.claimStride) % a.fullClaims : claim;
                chunkBase = full ? run * a.chunk : a.fullClaims * a.chunk + (claim - a.fullClaims) * a.tailChunk;
                chunkNext = 0;
                chunkCount = chunkBase < totalWork
                                 ? static_cast<uint32_t>(totalWork - chunkBase < size ? totalWork - chunkBase : size)
                                 : 0u;
                exhausted = chunkCount == 0;
            }
            const uint32_t rank = lanesBelow(needMask);
            const uint32_t avail = chunkCount - chunkNext;
            const uint32_t n = static_cast<uint32_t>(__builtin_popcountll(needMask));
            const bool served = need && rank < avail;
            const unsigned long long position = chunkBase + chunkNext + rank;
            const unsigned long long id = (served && !a.queueIn) ? photonAt(a, position) : 0ull;
            chunkNext += n < avail ? n : avail;
            if (need && !served && exhausted)
            {
                finished = true;
            }
            if (served && a.queueIn)
            {
                const PhotonState& parked = a.queueIn[a.queueInOrder ? a.queueInOrder[position] : position];
                ray.o = mk(parked.origin[0], parked.origin[1], parked.origin[2]);
                ray.d = mk(parked.direction[0], parked.direction[1], parked.direction[2]);
                color = Color{parked.color[0], parked.color[1], parked.color[2]};
                time = parked.time;
                photonIndex = parked.photonIndex;
                recordSlot = parked.recordSlot;
                lightId = parked.lightId;
                bounces = parked.bounces;
                casts = parked.casts;
                valid = parked.valid;
                alive = true;
            }
            else if (served)
            {
                {
                    splitPhotonId(a, id, &lightId, &photonIndex);
                    const Light& light = s.lights[lightId];
                    rng.begin(static_cast<uint32_t>(lightId), photonIndex, RT_STAGE_EMIT);
                    ray = emitRay(light, rng);
                    // LightQueue::registerLight bakes Phi / N (reference src/LightQueue.cpp:16)
                    const double photonFlux =
                        s.photonsPerLight > 0 ? light.flux / static_cast<double>(s.photonsPerLight) : 0.0;
                    color = Color{light.color[0], light.color[1], light.color[2]} * static_cast<float>(photonFlux);
                    bounces = 0;
                    time = 0.0f;
                    // time stamp, reference src/Worker.cpp:486-498
                    if (s.hasCamera && !__builtin_isinf(s.windowStart) && !__builtin_isnan(s.windowStart))
                    {
                        const bool finiteSpan = !__builtin_isinf(s.windowEnd) && !__builtin_isnan(s.windowEnd) &&
                                                s.windowEnd > s.windowStart;
                        if (finiteSpan)
                        {
                            const float span = s.windowEnd - s.windowStart;
                            rng.begin(static_cast<uint32_t>(lightId), photonIndex, RT_STAGE_TIME);
                            time = s.windowStart + static_cast<float>(rng.value(static_cast<double>(span)));
                        }
                        else
                        {
                            time = s.windowStart;
                        }
                    }
                    recordSlot = (Record && id < a.records.recordPhotons) ? static_cast<long long>(id) : -1;
                    if (recordSlot >= 0)
                    {
                        double* e = a.records.emitted + 6 * recordSlot;
                        e[0] = ray.o.x; e[1] = ray.o.y; e[2] = ray.o.z;
                        e[3] = ray.d.x; e[4] = ray.d.y; e[5] = ray.d.z;
                    }
                    casts = 0;
                    valid = 0;
                    alive = true;
                }
            }
        }
        RT_DIAG_TOC(refillStart, 4);
        if (__ballot(alive) == 0)
        {
            break;
        }

        // ---- one bounce (Worker::processPhotons loop body, src/Worker.cpp:530-673) ----
        bool goesOn = false;  // the photon left this bounce with a new ray
        if (alive)
        {
            RT_DIAG_TRIP(8);
            bool done = false;
            if (static_cast<double>(brightness(color)) < kEps)
            {
                done = true;
            }
            Hit hit;
            if (!done)
            {
                casts += 1;
                tally(lc, kTallyCasts);
                RT_DIAG_TIC(castStart);
                const bool hitSomething = castScene<Animated, true, Grouped>(s, ray, time, kSelfHitThreshold, true, &hit);
                RT_DIAG_TOC(castStart, 1);
                if (!hitSomething)
                {
                    done = true;  // escaped the scene
                }
            }
            if (!done)
            {
                const Material& material = s.materials[hit.material];
                const bool delta = materialIsDelta(material.type);
                if (Record && recordSlot >= 0 && valid < a.records.maxBounces)
                {
                    const long long slot = recordSlot * a.records.maxBounces + valid;
                    double* f = a.records.f64 + 10 * slot;
                    f[0] = hit.position.x; f[1] = hit.position.y; f[2] = hit.position.z;
                    f[3] = hit.normal.x; f[4] = hit.normal.y; f[5] = hit.normal.z;
                    f[6] = hit.distance;
                    f[7] = ray.d.x; f[8] = ray.d.y; f[9] = ray.d.z;
                    int32_t* r = a.records.i32 + 8 * slot;
                    r[0] = static_cast<int32_t>(photonIndex); r[1] = bounces; r[2] = hit.volume; r[3] = hit.triangle;
                    r[4] = hit.material; r[5] = 0; r[6] = delta ? 1 : 0; r[7] = lightId;
                    float* p = a.records.power + 3 * slot;
                    p[0] = color.r; p[1] = color.g; p[2] = color.b;
                }
                valid += 1;
                tally(lc, kTallyBounces);
                if (!delta)
                {
                    tally(lc, kTallyDeposits);
                    lc.power[0] += color.r;
                    lc.power[kTraceThreads] += color.g;
                    lc.power[2 * kTraceThreads] += color.b;
                }
                RT_DIAG_TIC(depositStart);
                if (Sink == kSinkProbe)
                {
                    bool keep = false;
                    if (!delta)
                    {
                        RT_DIAG_TRIP(10);
                        RT_DIAG_TIC(keepStart);
                        keep = anyProbeWithin(a.probes, hit.position);
                        RT_DIAG_TOC(keepStart, 11);
                        tally(lc, kTallyKept, keep);
                        tally(lc, kTallyCulled, !keep);
                    }
                    storeAppend(a.store, keep, hit.position, ray.d, hit.normal, time, color,
                                StoreOrder{static_cast<uint32_t>(photonIndex), static_cast<uint32_t>(a.firstPhoton),
                                           static_cast<uint32_t>(a.photonCount), lightId, bounces});
                }
                else if (Sink == kSinkSplat)
                {
                    if (a.density.enabled && !delta)
                    {
                        tally(lc, kTallyDensityDeposits);
                        densityAdd(a.density, hit.position, color, lc);
                    }
                    splatToCamera<Animated, Grouped>(a, !delta, ray.d, color, bounces, lightId, time, hit, material, lc);
                }

                RT_DIAG_TOC(depositStart, 5);
                RT_DIAG_TIC(scatterStart);

                // terminate: absolute decay floor or bounce cap (src/Worker.cpp:640-644,
                // include/Photon.h:55-72)
                const float mag = fmaxf(fmaxf(color.r, color.g), color.b);
                const bool decayAlive = static_cast<double>(mag) > s.terminationThreshold;
                if (!decayAlive || bounces >= s.bounceThreshold)
                {
                    done = true;
                }
                else
                {
                    // Material::generateDaughters with one daughter (src/Material.cpp:87-145)
                    rng.begin(static_cast<uint32_t>(lightId), photonIndex,
                              RT_STAGE_SCATTER + static_cast<uint32_t>(bounces));
                    const BSDFSample smp = sampleMaterial(material, ray.d, hit.normal, rng);
                    bounces += 1;
                    ray.o = hit.position;
                    if (!smp.valid)
                    {
                        ray.d = hit.normal;
                        color = Color{0.0f, 0.0f, 0.0f};
                    }
                    else
                    {
                        ray.d = smp.direction;
                        color = color * smp.weight;
                    }
                    goesOn = true;
                }
                RT_DIAG_TOC(scatterStart, 6);
            }
            else
            {
                // lanes that ended before the sink stage still take part in the wave-level
                // store append below with keep = false
            }
            if (done)
            {
                if (Record && recordSlot >= 0)
                {
                    a.records.castsValid[2 * recordSlot + 0] = casts;
                    a.records.castsValid[2 * recordSlot + 1] = valid;
                }
                tally(lc, kTallyPhotons);
                alive = false;
            }
        }
        // ---- park the photons that go on: their next bounce belongs to the next launch ----
        if (a.queueOut)
        {
            const unsigned long long parkMask = __ballot(goesOn);
            if (parkMask != 0)
            {
                const int leader = __builtin_ctzll(parkMask);
                unsigned long long base = 0;
                if (lane == leader)
                {
                    base = atomicAdd(a.queueOutCount, static_cast<unsigned long long>(__builtin_popcountll(parkMask)));
                }
                base = __shfl(base, leader, 64);
                const unsigned long long slot = base + lanesBelow(parkMask);
                if (goesOn && slot < a.queueOutCapacity)  // a full queue: the photon stays in its lane
                {
                    if (a.queueOutKeys)
                    {
                        a.queueOutKeys[slot] = rayKey(a, ray);
                    }
                    PhotonState& parked = a.queueOut[slot];
                    parked.origin[0] = ray.o.x; parked.origin[1] = ray.o.y; parked.origin[2] = ray.o.z;
                    parked.direction[0] = ray.d.x; parked.direction[1] = ray.d.y; parked.direction[2] = ray.d.z;
                    parked.color[0] = color.r; parked.color[1] = color.g; parked.color[2] = color.b;
                    parked.time = time;
                    parked.photonIndex = photonIndex;
                    parked.recordSlot = recordSlot;
                    parked.lightId = lightId;
                    parked.bounces = bounces;
                    parked.casts = casts;
                    parked.valid = valid;
                    alive = false;
                }
            }
        }
    }

    // ---- flush counters: one atomic per wave per counter ----
    RT_DIAG_TOC(kernelStart, 0);
    RT_DIAG_WAVE_END;
    static_assert(kTallySlots == 12 && offsetof(DeviceCounters, power) == 12 * sizeof(unsigned long long),
                  "tally slots follow the layout of DeviceCounters");
    if (lane < kTallySlots)
    {
        const unsigned int counted = __hip_atomic_load(lc.slots + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (counted != 0)
        {
            atomicAdd(&a.counters->photons + lane, static_cast<unsigned long long>(counted));
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const double total = waveSumD(lc.power[k * kTraceThreads]);
        if (lane == 0 && total != 0.0)
        {
            atomicAdd(&a.counters->power[k], total);
        }
    }
}

// ---------------------------------------------------------------- function-level kernels

__global__ __launch_bounds__(256) void castRaysKernel(const SceneView s, int64_t n, const double* rays, double selfHit, double* outF64,
                               int32_t* outI32)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ray ray;
    ray.o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
    ray.d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    Hit hit;
    const bool found = castScene<true>(s, ray, s.frameTime, selfHit, true, &hit);
    double* f = outF64 + 7 * i;
    int32_t* o = outI32 + 4 * i;
    if (found)
    {
        f[0] = hit.position.x; f[1] = hit.position.y; f[2] = hit.position.z;
        f[3] = hit.normal.x; f[4] = hit.normal.y; f[5] = hit.normal.z;
        f[6] = hit.distance;
        o[0] = 1; o[1] = hit.volume; o[2] = hit.triangle; o[3] = hit.material;
    }
    else
    {
        for (int k = 0; k < 7; ++k) f[k] = 0.0;
        o[0] = 0; o[1] = -1; o[2] = -1; o[3] = -1;
    }
}

__global__ void emitKernel(const SceneView s, int32_t lightIndex, int64_t n, const double* draws5,
                           uint64_t totalPhotons, double* rays, float* colors)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Light& light = s.lights[lightIndex];
    Rng rng;
    rng.tape = draws5 + 5 * i;
    rng.next = 0;
    rng.seed = 0;
    const Ray ray = emitRay(light, rng);
    const double photonFlux = light.flux / static_cast<double>(totalPhotons);
    const Color color = Color{light.color[0], light.color[1], light.color[2]} * static_cast<float>(photonFlux);
    double* o = rays + 6 * i;
    o[0] = ray.o.x; o[1] = ray.o.y; o[2] = ray.o.z;
    o[3] = ray.d.x; o[4] = ray.d.y; o[5] = ray.d.z;
    colors[3 * i] = color.r; colors[3 * i + 1] = color.g; colors[3 * i + 2] = color.b;
}

RT_DEV void writeHit(double* out, bool found, const Vec& position, const Vec& normal, double distance)
{
    out[0] = found ? 1.0 : 0.0;
    out[1] = found ? position.x : 0.0; out[2] = found ? position.y : 0.0; out[3] = found ? position.z : 0.0;
    out[4] = found ? normal.x : 0.0; out[5] = found ? normal.y : 0.0; out[6] = found ? normal.z : 0.0;
    out[7] = found ? distance : 0.0;
}

__global__ void intersectKernel(int kind, int64_t n, const double* in, void* outRaw)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Vec position = mk(0, 0, 0), normal = mk(0, 0, 0);
    double distance = 0.0;
    if (kind == 0)  // triangle: ray(6) a b c aN bN cN
    {
        const double* r = in + 24 * i;
        Ray ray;
        ray.o = mk(r[0], r[1], r[2]);
        ray.d = mk(r[3], r[4], r[5]);
        const RayPrep prep = prepareRay(ray);
        double bary[3];
        const bool found = rayIntersectsTriangle(ray, prep, r + 6, bary, &position, &distance);
        if (found) normal = triangleNormal(r + 15, bary);
        writeHit(static_cast<double*>(outRaw) + 8 * i, found, position, normal, distance);
    }
    else if (kind == 1)  // bounds: ray(6) corner corner (each axis sorted, reference src/Bounds.cpp:5-9)
    {
        const double* r = in + 12 * i;
        Ray ray;
        ray.o = mk(r[0], r[1], r[2]);
        ray.d = mk(r[3], r[4], r[5]);
        double lo[3], hi[3], entry;
        for (int k = 0; k < 3; ++k)
        {
            lo[k] = dmin(r[6 + k], r[9 + k]);
            hi[k] = dmax(r[6 + k], r[9 + k]);
        }
        const bool direct = rayIntersectsBounds(ray, lo, hi, &entry);
        const SlabPrep prep = prepareSlab(ray);
        // the hoisted test reads its box from memory: stage the sorted corners after the
        // row's own output slot is known to be free (rows are 12 doubles, the box is 6)
        double* box = const_cast<double*>(r) + 6;
        for (int k = 0; k < 3; ++k)
        {
            box[k] = lo[k];
            box[3 + k] = hi[k];
        }
        const bool hoisted = slabTest(ray, prep, box, &entry);
        // both forms must agree; report 2 if they ever do not
        static_cast<int32_t*>(outRaw)[i] = direct == hoisted ? (direct ? 1 : 0) : 2;
    }
    else if (kind == 2)  // sphere: ray(6) center radius
    {
        const double* r = in + 10 * i;
        Ray ray;
        ray.o = mk(r[0], r[1], r[2]);
        ray.d = mk(r[3], r[4], r[5]);
        const bool found = rayIntersectsSphere(ray, mk(r[6], r[7], r[8]), r[9], &position, &normal, &distance);
        writeHit(static_cast<double*>(outRaw) + 8 * i, found, position, normal, distance);
    }
    else if (kind == 3)  // quad: ray(6) origin edgeU edgeV
    {
        const double* r = in + 15 * i;
        Ray ray;
        ray.o = mk(r[0], r[1], r[2]);
        ray.d = mk(r[3], r[4], r[5]);
        double q[18];
        for (int k = 0; k < 9; ++k) q[k] = r[6 + k];
        quadUpdate(q);
        Vec uv = mk(0, 0, 0);
        const bool found = rayIntersectsQuad(ray, q, &position, &normal, &distance, &uv);
        double* o = static_cast<double*>(outRaw) + 20 * i;
        writeHit(o, found, position, normal, distance);
        o[8] = found ? uv.x : 0.0; o[9] = found ? uv.y : 0.0; o[10] = found ? uv.z : 0.0;
        for (int k = 0; k < 9; ++k) o[11 + k] = q[9 + k];
    }
    else  // plane: ray(6) normal dot
    {
        const double* r = in + 10 * i;
        Ray ray;
        ray.o = mk(r[0], r[1], r[2]);
        ray.d = mk(r[3], r[4], r[5]);
        const bool found = rayIntersectsPlane(ray, mk(r[6], r[7], r[8]), r[9], &position, &normal, &distance);
        writeHit(static_cast<double*>(outRaw) + 8 * i, found, position, normal, distance);
    }
}

RT_DEV Material materialFromRow(const double* r)
{
    Material m;
    m.type = static_cast<int32_t>(r[0]);
    m.pad = 0;
    m.color[0] = static_cast<float>(r[1]);
    m.color[1] = static_cast<float>(r[2]);
    m.color[2] = static_cast<float>(r[3]);
    m.pad2 = 0.0f;
    m.alpha = r[4];
    m.ior = 1.5;
    return m;
}

__global__ void bsdfSampleKernel(int64_t n, const double* in, double* out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = in + 14 * i;
    Material m = materialFromRow(r);
    m.ior = r[5];
    Rng rng;
    rng.tape = r + 12;
    rng.next = 0;
    rng.seed = 0;
    const BSDFSample s = sampleMaterial(m, mk(r[6], r[7], r[8]), mk(r[9], r[10], r[11]), rng);
    double* o = out + 10 * i;
    o[0] = s.valid ? 1.0 : 0.0;
    o[1] = s.direction.x; o[2] = s.direction.y; o[3] = s.direction.z;
    o[4] = s.weight.r; o[5] = s.weight.g; o[6] = s.weight.b;
    o[7] = s.pdf;
    o[8] = s.isDelta ? 1.0 : 0.0;
    o[9] = materialIsDelta(m.type) ? 1.0 : 0.0;
}

__global__ void bsdfEvaluateKernel(int64_t n, const double* in, double* out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* r = in + 14 * i;
    const Material m = materialFromRow(r);
    const Vec wi = mk(r[5], r[6], r[7]), wo = mk(r[8], r[9], r[10]), nn = mk(r[11], r[12], r[13]);
    const Color f = evaluateMaterial(m, wi, wo, nn);
    double* o = out + 4 * i;
    o[0] = f.r; o[1] = f.g; o[2] = f.b;
    o[3] = pdfMaterial(m, wi, wo, nn);
}


// ---------------------------------------------------------------- probe pass kernels
// ProbeGather::collectGatherPoints (reference src/ProbeGather.cpp:239-460), one thread
// per sampled pixel.  The reference walks pixels serially and appends records; here a
// first launch counts each pixel's surviving samples and a second one, given the
// exclusive scan of the counts, writes the records at their final positions — the same
// pixel-major order, with no atomics.
// Both kernels are held to two waves per SIMD: left alone the compiler takes 282 / 370 registers
// (one wave per SIMD), and the second wave is worth more than the registers it spills
// (MirrorKnot4k, 16.8 M pixels: 34.2 -> 19.3 ms for the pair).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gatherCountKernel(const SceneView s, int32_t camera, int32_t stride, uint32_t* counts,
                                  unsigned long long* stats)
{
    const Camera& cam = s.cameras[camera];
    const int32_t gw = (cam.width + stride - 1) / stride, gh = (cam.height + stride - 1) / stride;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned long long local[3] = {0, 0, 0};
    if (i < static_cast<int64_t>(gw) * gh)
    {
        const int32_t px = static_cast<int32_t>(i % gw) * stride, py = static_cast<int32_t>(i / gw) * stride;
        counts[i] = static_cast<uint32_t>(collectPixel(s, camera, px, py, local, [](const GatherPoint&, int) {}));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const unsigned long long total = waveSum(local[k]);
        if ((threadIdx.x & 63) == 0 && total != 0)
        {
            atomicAdd(stats + k, total);
        }
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gatherWriteKernel(const SceneView s, int32_t camera, int32_t stride, const uint32_t* counts,
                                  const uint64_t* offsets, GatherPoint* out)
{
    const Camera& cam = s.cameras[camera];
    const int32_t gw = (cam.width + stride - 1) / stride, gh = (cam.height + stride - 1) / stride;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= static_cast<int64_t>(gw) * gh || counts[i] == 0)
    {
        return;
    }
    const int32_t px = static_cast<int32_t>(i % gw) * stride, py = static_cast<int32_t>(i / gw) * stride;
    // sampleWeight = 1 / (surviving samples of the pixel), reference src/ProbeGather.cpp:445-456
    const float weight = 1.0f / static_cast<float>(counts[i]);
    GatherPoint* dst = out + offsets[i];
    unsigned long long ignored[3] = {0, 0, 0};
    collectPixel(s, camera, px, py, ignored, [&](const GatherPoint& gp, int k) {
        GatherPoint w = gp;
        w.sampleWeight = weight;
        dst[k] = w;
    });
}

// Where a pixel has one sample at most (no shutter samples, no lens samples, no dielectric in the scene:
// collectPixel's loops run once) a single pass of casts does: every pixel's record goes to the pixel's own
// slot, and the counts are scanned for the records' numbers (no second cast).  The pass is two launches: the
// first casts every pixel's camera ray and finishes the pixels that see a diffuse surface or nothing — a lean
// kernel, most pixels of most frames —, and lists the pixels that see a mirror; the second follows those
// rays on with the whole of collectPixel (the mirror chain, the neighbouring pixel's chain for the
// footprint: the kernel that does not fit its registers).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gatherPrimaryKernel(const SceneView s, int32_t camera, int32_t stride, uint32_t* counts,
                                    unsigned long long* stats, GatherPoint* slots, uint32_t* deferred,
                                    unsigned long long* deferredCount)
{
    const Camera& cam = s.cameras[camera];
    const int32_t gw = (cam.width + stride - 1) / stride, gh = (cam.height + stride - 1) / stride;
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned long long local[3] = {0, 0, 0};
    bool later = false;
    if (i < static_cast<int64_t>(gw) * gh)
    {
        const int32_t px = static_cast<int32_t>(i % gw) * stride, py = static_cast<int32_t>(i / gw) * stride;
        GatherPoint* dst = slots + i;
        const int survived = collectPixel<true>(s, camera, px, py, local, [&](const GatherPoint& gp, int k) {
            if (k == 0) *dst = gp;  // sampleWeight is 1 / 1 already
        });
        later = survived < 0;
        counts[i] = later ? 0u : static_cast<uint32_t>(survived);
    }
    // the pixels left for the second launch, listed with one atomic a wave
    const unsigned long long laterMask = __ballot(later);
    if (laterMask != 0)
    {
        const int leader = __builtin_ctzll(laterMask);
        unsigned long long base = 0;
        if (static_cast<int>(threadIdx.x & 63) == leader)
        {
            base = atomicAdd(deferredCount, static_cast<unsigned long long>(__builtin_popcountll(laterMask)));
        }
        base = __shfl(base, leader, 64);
        if (later) deferred[base + lanesBelow(laterMask)] = static_cast<uint32_t>(i);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const unsigned long long total = waveSum(local[k]);
        if ((threadIdx.x & 63) == 0 && total != 0)
        {
            atomicAdd(stats + k, total);
        }
    }
}

// deferred == null: every pixel; else the pixels the list names (launched for the most there can be)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void gatherCollectKernel(const SceneView s, int32_t camera, int32_t stride, uint32_t* counts,
                                    unsigned long long* stats, GatherPoint* slots, const uint32_t* deferred,
                                    const unsigned long long* deferredCount)
{
    const Camera& cam = s.cameras[camera];
    const int32_t gw = (cam.width + stride - 1) / stride, gh = (cam.height + stride - 1) / stride;
    const int64_t q = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t listed = deferred ? static_cast<int64_t>(*deferredCount) : static_cast<int64_t>(gw) * gh;
    if (static_cast<int64_t>(blockIdx.x) * blockDim.x >= listed)
    {
        return;  // the whole block lies beyond the list
    }
    unsigned long long local[3] = {0, 0, 0};
    if (q < listed)
    {
        const int64_t i = deferred ? static_cast<int64_t>(deferred[q]) : q;
        const int32_t px = static_cast<int32_t>(i % gw) * stride, py = static_cast<int32_t>(i / gw) * stride;
        GatherPoint* dst = slots + i;
        counts[i] = static_cast<uint32_t>(collectPixel(s, camera, px, py, local, [&](const GatherPoint& gp, int k) {
            if (k == 0) *dst = gp;  // sampleWeight is 1 / 1 already
        }));
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
        const unsigned long long total = waveSum(local[k]);
        if ((threadIdx.x & 63) == 0 && total != 0)
        {
            atomicAdd(stats + k, total);
        }
    }
}

// a record is 30 words: a thread per word
__global__ void gatherCompactKernel(const GatherPoint* slots, const uint32_t* counts, const uint64_t* offsets, int64_t pixels,
                                    GatherPoint* out)
{
    constexpr int kWords = static_cast<int>(sizeof(GatherPoint) / sizeof(uint32_t));
    const int64_t t = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t i = t / kWords;
    if (i >= pixels || counts[i] == 0) return;
    const int word = static_cast<int>(t - i * kWords);
    reinterpret_cast<uint32_t*>(out + offsets[i])[word] = reinterpret_cast<const uint32_t*>(slots + i)[word];
}

// ProbeIndex::anyWithinKeepRadius for host-supplied points (emitter deposits,
// reference src/ProbeGather.cpp:536-541).
__global__ void probeTestKernel(const ProbeView pv, int64_t n, const double* points, uint8_t* out)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = anyProbeWithin(pv, mk(points[3 * i], points[3 * i + 1], points[3 * i + 2])) ? 1 : 0;
}

// ---------------------------------------------------------------- the gather
// BounceStore::buildIndex (reference src/BounceStore.cpp:42-53) keys every record by the
// cell of its float position; here the key packs the three cell coordinates, z in the
// high bits, so that ascending key order IS the dz / dy / dx order in which
// BounceStore::radiusSearch (:55-98) walks the cells, and a stable sort keeps the records
// of one cell in store order.  The coordinates are taken relative to the box of the store's cells and
// packed without gaps (KeyLayout): the sort then runs over the bits the box needs, not over 63.  A record
// whose cell lies beyond +-2^20 cells gets kEmptyKey and is left out of the index.
RT_DEV unsigned long long cellKey(const KeyLayout& l, int64_t x, int64_t y, int64_t z)  // a cell of the box
{
    return (static_cast<unsigned long long>(z - l.low[2]) << (l.bitsX + l.bitsY)) |
           (static_cast<unsigned long long>(y - l.low[1]) << l.bitsX) | static_cast<unsigned long long>(x - l.low[0]);
}

RT_DEV int64_t keyCellX(const KeyLayout& l, unsigned long long key)
{
    return static_cast<int64_t>(key & ((1ull << l.bitsX) - 1)) + l.low[0];
}

RT_DEV bool recordCell(const float* r, double invCellSize, int64_t* x, int64_t* y, int64_t* z)
{
    *x = static_cast<int64_t>(__builtin_floor(static_cast<double>(r[0]) * invCellSize));
    *y = static_cast<int64_t>(__builtin_floor(static_cast<double>(r[1]) * invCellSize));
    *z = static_cast<int64_t>(__builtin_floor(static_cast<double>(r[2]) * invCellSize));
    return *x >= -kPackLimit && *x < kPackLimit && *y >= -kPackLimit && *y < kPackLimit && *z >= -kPackLimit && *z < kPackLimit;
}

// The box of the records' cells: every thread takes a few records, a block reduces in LDS and one
// thread asks for the atomics, and only where the block widens what it reads in the box (the box settles
// after a few blocks; unconditional atomics would queue the whole launch up at six words).
__global__ __launch_bounds__(256) void storeBoxKernel(const float* store, uint64_t n, double invCellSize, int32_t* cellBox)
{
    __shared__ int32_t blockLow[3], blockHigh[3];
    if (threadIdx.x < 3)
    {
        blockLow[threadIdx.x] = INT32_MAX;
        blockHigh[threadIdx.x] = INT32_MIN;
    }
    __syncthreads();
    int32_t low[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, high[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n;
         i += static_cast<uint64_t>(gridDim.x) * blockDim.x)
    {
        int64_t c[3];
        if (recordCell(store + 13 * i, invCellSize, &c[0], &c[1], &c[2]))
        {
#pragma unroll
            for (int k = 0; k < 3; ++k)
            {
                low[k] = min(low[k], static_cast<int32_t>(c[k]));
                high[k] = max(high[k], static_cast<int32_t>(c[k]));
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
#pragma unroll
        for (int step = 1; step < 64; step <<= 1)
        {
            low[k] = min(low[k], __shfl_xor(low[k], step));
            high[k] = max(high[k], __shfl_xor(high[k], step));
        }
        if ((threadIdx.x & 63) == 0 && low[k] <= high[k])
        {
            atomicMin(&blockLow[k], low[k]);
            atomicMax(&blockHigh[k], high[k]);
        }
    }
    __syncthreads();
    if (threadIdx.x < 3 && blockLow[threadIdx.x] <= blockHigh[threadIdx.x])
    {
        const int k = static_cast<int>(threadIdx.x);
        if (blockLow[k] < __hip_atomic_load(cellBox + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(cellBox + k, blockLow[k]);
        if (blockHigh[k] > __hip_atomic_load(cellBox + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        {
            atomicMax(cellBox + 3 + k, blockHigh[k]);
        }
    }
}

__global__ void storeKeysKernel(const float* store, uint64_t n, double invCellSize, KeyLayout layout, unsigned long long* keys,
                                uint32_t* order)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t x, y, z;
    const bool fits = recordCell(store + 13 * i, invCellSize, &x, &y, &z);
    keys[i] = fits ? cellKey(layout, x, y, z) : kEmptyKey;
    order[i] = static_cast<uint32_t>(i);
}

// Where a cell's records start in the sorted key list (GatherArgs::cellStart).
struct CellBox
{
    KeyLayout layout;
    int32_t dim[3];  // dim[0] == 1: a directory of rows, whatever their cells
};

RT_DEV uint32_t cellNumber(const CellBox& box, unsigned long long key)
{
    const KeyLayout& l = box.layout;
    const uint32_t x = box.dim[0] == 1 ? 0u : static_cast<uint32_t>(key & ((1ull << l.bitsX) - 1));
    const uint32_t y = static_cast<uint32_t>((key >> l.bitsX) & ((1ull << l.bitsY) - 1));
    const uint32_t z = static_cast<uint32_t>(key >> (l.bitsX + l.bitsY));
    return (z * static_cast<uint32_t>(box.dim[1]) + y) * static_cast<uint32_t>(box.dim[0]) + x;
}

// every cell's first record; cells without records keep the fill value (no record: UINT32_MAX)
__global__ void cellFirstsKernel(const unsigned long long* keys, uint64_t indexed, CellBox box, uint32_t* cellStart)
{
    const uint64_t j = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (j >= indexed) return;
    if (j == 0)
    {
        cellStart[box.dim[0] * box.dim[1] * box.dim[2]] = static_cast<uint32_t>(indexed);  // the end of the last cell
    }
    const uint32_t cell = cellNumber(box, keys[j]);
    if (j == 0 || cellNumber(box, keys[j - 1]) != cell)
    {
        cellStart[cell] = static_cast<uint32_t>(j);
    }
}

RT_DEV uint64_t lowerBound(const unsigned long long* keys, uint64_t n, unsigned long long key)
{
    uint64_t lo = 0, hi = n;
    while (lo < hi)
    {
        const uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__global__ void reorderStoreKernel(const float* store, const uint32_t* order, uint64_t floats, float* sorted)
{
    const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= floats) return;
    const uint64_t j = t / 13, k = t % 13;
    sorted[t] = store[13ull * order[j] + k];
}

__global__ void countIndexedKernel(const unsigned long long* keys, uint64_t n, unsigned long long* out)
{
    out[0] = lowerBound(keys, n, kEmptyKey);
}


// The directory entries that bound the records of cells x0..x1 of the rows ya..yb of slab z: the
// records lie in [cellStart[*from], cellStart[*to]) — exactly those records with a directory of cells
// and one row (ya == yb), otherwise a range that holds them.  *from >= *to: there are none.  Returns
// false without a directory: then the whole list is the range (*from < *to still says "maybe").
RT_DEV bool directoryEntries(const GatherArgs& a, int64_t x0, int64_t x1, int64_t ya, int64_t yb, int64_t z, uint32_t* from,
                             uint32_t* to)
{
    if (a.directoryKind == kDirectoryNone)
    {
        *from = 0;
        *to = 1;
        return false;
    }
    const int64_t xmin = a.cellMin[0], ymin = a.cellMin[1], zmin = a.cellMin[2];
    const int64_t nx = a.cellDim[0], ny = a.cellDim[1], nz = a.cellDim[2];  // nx is 1 in a directory of rows
    *from = *to = 0;
    if (z < zmin || z >= zmin + nz)
    {
        return true;
    }
    const int64_t ylo = ya < ymin ? ymin : ya, yhi = yb > ymin + ny - 1 ? ymin + ny - 1 : yb;
    if (ylo > yhi)
    {
        return true;
    }
    int64_t xa = 0, xb = 0;  // cells of the row, relative to its first
    if (a.directoryKind == kDirectoryCells && ya == yb)
    {
        xa = (x0 < xmin ? xmin : x0) - xmin;
        xb = (x1 > xmin + nx - 1 ? xmin + nx - 1 : x1) - xmin;
        if (xa > xb)
        {
            return true;
        }
    }
    else
    {
        xb = nx - 1;  // whole rows
    }
    *from = static_cast<uint32_t>(((z - zmin) * ny + (ylo - ymin)) * nx + xa);
    *to = static_cast<uint32_t>(((z - zmin) * ny + (yhi - ymin)) * nx + xb + 1);
    return true;
}

// testing::gatherRadiance, reference src/ProbeGather.cpp:596-728, for one gather point.
// (a wave a block: a block's place on the chip is free again when its last wave ends, and the pixels of a
// 256-thread block end far apart: MirrorKnot4k's gather 16.0 -> 14.4 ms)
constexpr int kGatherThreads = 64;   // threads of a gather block
constexpr int kGatherQueue = 16;     // records a lane of the one-lane gather queues before it settles them

// queue: the lane's column of the block's [kGatherQueue][kGatherThreads] words of LDS
// *heavy is set (and nothing else is done) when the point has more records to look at than
// a.heavyThreshold allows one lane.
RT_DEV Color gatherRadiance(const GatherArgs& a, const GatherPoint& gp, uint32_t* queue, unsigned long long* deposits,
                            bool* heavy)
{
    const Color black{0.0f, 0.0f, 0.0f};
    const bool isEmitter = gp.material < 0;
    const Vec wo = mk(gp.viewDir[0], gp.viewDir[1], gp.viewDir[2]);
    const Vec normal = mk(gp.normal[0], gp.normal[1], gp.normal[2]);
    const Vec position = mk(gp.position[0], gp.position[1], gp.position[2]);
    if (dot(wo, normal) <= 0.0)
    {
        return black;
    }
    const double r = (a.minGatherRadius > 0.0 && gp.footprintRadius < a.minGatherRadius) ? a.minGatherRadius
                                                                                         : gp.footprintRadius;
    if (r <= 0.0 || a.indexed == 0)
    {
        return black;
    }
    const double r2 = r * r;
    const double planeBand = 2.0 * r;
    const int64_t cx = static_cast<int64_t>(__builtin_floor(position.x * a.invCellSize));
    const int64_t cy = static_cast<int64_t>(__builtin_floor(position.y * a.invCellSize));
    const int64_t cz = static_cast<int64_t>(__builtin_floor(position.z * a.invCellSize));
    const int64_t reach = static_cast<int64_t>(__builtin_ceil(r * a.invCellSize));
    // the cube of cells, cut to the box of the cells that hold records (the keys are relative to it)
    const int64_t x0 = max(cx - reach, static_cast<int64_t>(a.layout.low[0])), x1 = min(cx + reach, static_cast<int64_t>(a.layout.high[0]));
    const int64_t y0 = max(cy - reach, static_cast<int64_t>(a.layout.low[1])), y1 = min(cy + reach, static_cast<int64_t>(a.layout.high[1]));
    const int64_t z0 = max(cz - reach, static_cast<int64_t>(a.layout.low[2])), z1 = min(cz + reach, static_cast<int64_t>(a.layout.high[2]));
    if (x0 > x1 || y0 > y1 || z0 > z1)
    {
        return black;
    }
    const Material* material = isEmitter ? nullptr : &a.scene.materials[gp.material];
    Color sum = black;
    unsigned long long kept = 0;
    // Rows of cells along x are contiguous in key order; when the cube is wide, whole z slabs are
    // scanned and filtered instead of searching every row.  Where the rows start in the key list is
    // looked up for three rows at a time (a slab of the neighbourhood of a footprint that fits a cell), the
    // three binary searches advancing in step: their reads, chains of dependent round trips each, are in
    // flight together, and so are the reads of every row's first key, which tell the empty rows (most of
    // them, on a thin store) from the ones worth a walk.  The rows are then walked in the reference's
    // order (z outside, y inside, store order within a cell).  (Nine rows in step and eight records a
    // read keep more reads in flight per lane but cost 212 registers, two waves per SIMD; three and four
    // fit three waves, which hides more: MirrorKnot4k's gather 14.4 -> 13.0 ms.)
    const bool bySlab = (y1 - y0) > 16;
    // (a footprint spans at most 2^21 cells along an axis and fewer than 17 rows per slab: 32-bit row numbers)
    const int rowsPerSlab = bySlab ? 1 : static_cast<int>(y1 - y0 + 1);
    const int rowCount = static_cast<int>(z1 - z0 + 1) * rowsPerSlab;
    constexpr int kRows = 3;
    // Records that lie within the footprint (and the shutter's window) wait in the lane's queue, in
    // their order, and go through the remaining tests and the BRDF together: were each taken where
    // it is found, the wave would run the BRDF for nearly every record any of its lanes looks at,
    // a few lanes at a time.  The sums are added in the order of the records either way.
    typedef float Float4At4 __attribute__((ext_vector_type(4), aligned(4)));
    int queued = 0;
    auto settle = [&]() {
        for (int q = 0; q < queued; ++q)
        {
            const float* rec = a.store + 13ull * queue[q * kGatherThreads];
            const Float4At4 q0 = *reinterpret_cast<const Float4At4*>(rec);
            const Float4At4 q1 = *reinterpret_cast<const Float4At4*>(rec + 4);
            const Float4At4 q2 = *reinterpret_cast<const Float4At4*>(rec + 8);
            const Vec dn = mk(q1.z, q1.w, q2.x);
            const Color power{q2.z, q2.w, rec[12]};
            if (isEmitter)
            {
                if (dot(dn, normal) < 0.5)
                {
                    continue;
                }
                sum.r += power.r;
                sum.g += power.g;
                sum.b += power.b;
                kept += 1;
                continue;
            }
            const double dnLen = magnitude(dn);
            if (dnLen > 1e-4)
            {
                if (dot(dn / dnLen, normal) < 0.5)
                {
                    continue;
                }
            }
            const double planeDist = dabs(dot(mk(q0.x, q0.y, q0.z) - position, normal));
            if (planeDist > planeBand)
            {
                continue;
            }
            const Vec wi = neg(mk(q0.w, q1.x, q1.y));
            const Color f = evaluateMaterial(*material, wi, wo, normal);
            const Color fp = f * power;
            sum.r += fp.r;
            sum.g += fp.g;
            sum.b += fp.b;
            kept += 1;
        }
        queued = 0;
    };
    for (int rowBase = 0; rowBase < rowCount; rowBase += kRows)
    {
        // the rows' first and last keys are recomputed where they are needed rather than kept (registers)
        auto firstKey = [&](int k) {
            const int row = rowBase + k;
            return row < rowCount ? cellKey(a.layout, x0, y0 + row % rowsPerSlab, z0 + row / rowsPerSlab) : kEmptyKey;
        };
        auto lastKey = [&](int k) {
            const int row = rowBase + k;
            const int64_t y = y0 + row % rowsPerSlab, z = z0 + row / rowsPerSlab;
            return bySlab ? cellKey(a.layout, x1, y1, z) : cellKey(a.layout, x1, y, z);
        };
        // Records [lo, hi) of every row.  With a directory of cells these are two of its entries; with a
        // directory of rows (or none) the entries (or the whole list) bound two binary searches, for the
        // row's first key and for the first key beyond its last.
        uint32_t lo[kRows], hi[kRows];
#pragma unroll
        for (int k = 0; k < kRows; ++k)
        {
            const int row = rowBase + k;
            uint32_t from = 0, to = 0;
            const bool listed = directoryEntries(a, x0, x1, y0 + row % rowsPerSlab, bySlab ? y1 : y0 + row % rowsPerSlab,
                                                 z0 + row / rowsPerSlab, &from, &to);
            const bool any = row < rowCount && from < to;
            lo[k] = any ? (listed ? a.cellStart[from] : 0u) : 0u;
            hi[k] = any ? (listed ? a.cellStart[to] : static_cast<uint32_t>(a.indexed)) : 0u;
        }
        if (a.directoryKind != kDirectoryCells)
        {
            uint32_t widest = 0;
#pragma unroll
            for (int k = 0; k < kRows; ++k)
            {
                widest = max(widest, hi[k] - lo[k]);
            }
            // Short ranges (a directory of rows over a thin store leaves a few records a row): their keys are
            // read in one go and counted against the row's first and last key, one round trip instead of the
            // two searches' four.
            constexpr uint32_t kShort = 4;
            if (widest != 0 && widest <= kShort)
            {
                unsigned long long near[kRows][kShort];
#pragma unroll
                for (int k = 0; k < kRows; ++k)
                {
#pragma unroll
                    for (uint32_t u = 0; u < kShort; ++u)
                    {
                        const uint32_t at = lo[k] + u < hi[k] ? lo[k] + u : lo[k];
                        near[k][u] = a.keys[at < a.indexed ? at : a.indexed - 1];
                    }
                }
#pragma unroll
                for (int k = 0; k < kRows; ++k)
                {
                    const unsigned long long first = firstKey(k), last = lastKey(k);
                    uint32_t below = 0, within = 0;  // keys ascend: those below the first, those up to the last
#pragma unroll
                    for (uint32_t u = 0; u < kShort; ++u)
                    {
                        const bool there = lo[k] + u < hi[k];
                        below += (there && near[k][u] < first) ? 1u : 0u;
                        within += (there && near[k][u] <= last) ? 1u : 0u;
                    }
                    hi[k] = lo[k] + within;
                    lo[k] = lo[k] + below;
                }
                widest = 0;  // (no search below)
            }
            int steps = 0;
            while (widest != 0 && (1u << steps) <= widest) ++steps;
            uint32_t end[kRows];
#pragma unroll
            for (int k = 0; k < kRows; ++k)
            {
                end[k] = hi[k];
            }
            // first key of the row or above: [lo, hi) closes in on it
            for (int step = 0; step < steps; ++step)
            {
                unsigned long long probe[kRows];
#pragma unroll
                for (int k = 0; k < kRows; ++k)
                {
                    const uint32_t mid = (lo[k] + hi[k]) >> 1;  // hi <= indexed < 2^31
                    probe[k] = a.keys[mid < a.indexed ? mid : a.indexed - 1];
                }
#pragma unroll
                for (int k = 0; k < kRows; ++k)
                {
                    const uint32_t mid = (lo[k] + hi[k]) >> 1;
                    const bool open = lo[k] < hi[k];
                    const bool below = probe[k] < firstKey(k);
                    lo[k] = (open && below) ? mid + 1 : lo[k];
                    hi[k] = (open && !below) ? mid : hi[k];
                }
            }
            // first key beyond the row's last, from there on: [hi, end) closes in on it
#pragma unroll
            for (int k = 0; k < kRows; ++k)
            {
                hi[k] = steps > 0 ? lo[k] : hi[k];
            }
            for (int step = 0; step < steps; ++step)
            {
                unsigned long long probe[kRows];
#pragma unroll
                for (int k = 0; k < kRows; ++k)
                {
                    const uint32_t mid = (hi[k] + end[k]) >> 1;
                    probe[k] = a.keys[mid < a.indexed ? mid : a.indexed - 1];
                }
#pragma unroll
                for (int k = 0; k < kRows; ++k)
                {
                    const uint32_t mid = (hi[k] + end[k]) >> 1;
                    const bool open = hi[k] < end[k];
                    const bool within = probe[k] <= lastKey(k);
                    hi[k] = (open && within) ? mid + 1 : hi[k];
                    end[k] = (open && !within) ? mid : end[k];
                }
            }
        }
        uint32_t worthAWalk = 0, records = 0;
#pragma unroll
        for (int k = 0; k < kRows; ++k)
        {
            if (lo[k] < hi[k])
            {
                worthAWalk |= 1u << k;
                records += hi[k] - lo[k];
            }
        }
        if (a.heavyThreshold != 0 && records > a.heavyThreshold)
        {
            *heavy = true;
            return black;
        }
#pragma unroll
        for (int k = 0; k < kRows; ++k)
        {
            if ((worthAWalk & (1u << k)) == 0)
            {
                continue;  // no such row, or nothing in it
            }
            // The walk is bound by the round trip of a record's read, so kChunk records are requested
            // together before any test looks at one.  Only what the first tests need is read here (the
            // key where slabs are filtered, position and time): few records of a
            // neighbourhood lie within the footprint, and those are queued for settle().
            constexpr uint32_t kChunk = 4;
            for (uint32_t base = lo[k]; base < hi[k]; base += kChunk)
            {
                unsigned long long key[kChunk];
                Float4At4 q0[kChunk];
                float time[kChunk];
#pragma unroll
                for (uint32_t u = 0; u < kChunk; ++u)
                {
                    const uint32_t j = base + u < hi[k] ? base + u : hi[k] - 1;
                    const float* row = a.store + 13ull * j;
                    key[u] = bySlab ? a.keys[j] : 0ull;
                    q0[u] = *reinterpret_cast<const Float4At4*>(row);
                    time[u] = row[9];
                }
#pragma unroll
                for (uint32_t u = 0; u < kChunk; ++u)
                {
                    asm volatile("" : "+v"(q0[u]), "+v"(time[u]));  // the reads stay ahead of the branches
                }
#pragma unroll
                for (uint32_t u = 0; u < kChunk; ++u)
                {
                    if (base + u >= hi[k])
                    {
                        continue;
                    }
                    if (bySlab)
                    {
                        const int64_t kx = keyCellX(a.layout, key[u]);
                        if (kx < x0 || kx > x1)
                        {
                            continue;
                        }
                    }
                    const double ddx = static_cast<double>(q0[u].x) - position.x;
                    const double ddy = static_cast<double>(q0[u].y) - position.y;
                    const double ddz = static_cast<double>(q0[u].z) - position.z;
                    if (!(ddx * ddx + ddy * ddy + ddz * ddz <= r2))
                    {
                        continue;
                    }
                    if (!__builtin_isinf(time[u]) || time[u] < 0.0f)  // +inf marks a timeless (emitter) deposit
                    {
                        if (__builtin_fabsf(time[u] - gp.sampleTime) > a.timeHalfWindow)
                        {
                            continue;
                        }
                    }
                    queue[queued * kGatherThreads] = base + u;
                    queued += 1;
                    if (queued == kGatherQueue)
                    {
                        settle();
                    }
                }
            }
        }
    }
    settle();
    *deposits = kept;
    if (kept == 0)
    {
        return black;
    }
    const double area = kPi * r * r;
    const float scale = static_cast<float>((4.0 / kPi) / area);
    return sum * scale;
}

// ProbeGather::run / gatherRecords, reference src/ProbeGather.cpp:865-994: one thread per
// sampled pixel adds that pixel's records in their order, as the reference's single-threaded
// loop does, then lands the sum in the camera buffer (Buffer::addColor, src/Buffer.cpp:22-37).
__global__ __launch_bounds__(kGatherThreads) void gatherKernel(const GatherArgs a)
{
    __shared__ uint32_t queues[kGatherQueue * kGatherThreads];
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    unsigned long long added = 0, depositsAccum = 0;
    bool heavy = false;
    if (i < a.pixels && a.counts[i] != 0)
    {
        const Camera& cam = a.scene.cameras[a.camera];
        const GatherPoint* points = a.points + (a.offsets ? a.offsets[i] : static_cast<uint64_t>(i));
        Color acc{0.0f, 0.0f, 0.0f};
        for (uint32_t k = 0; k < a.counts[i]; ++k)
        {
            const GatherPoint& gp = points[k];
            if (gp.material >= a.scene.materialCount)
            {
                continue;
            }
            unsigned long long deposits = 0;
            const Color radiance = gatherRadiance(a, gp, queues + threadIdx.x, &deposits, &heavy);
            if (heavy)
            {
                break;
            }
            const Color throughput{gp.throughput[0], gp.throughput[1], gp.throughput[2]};
            const Color contribution = throughput * radiance * gp.sampleWeight;
            depositsAccum += deposits;
            if (contribution.r == 0.0f && contribution.g == 0.0f && contribution.b == 0.0f)
            {
                continue;
            }
            acc.r += contribution.r;
            acc.g += contribution.g;
            acc.b += contribution.b;
            added += 1;
        }
        const uint64_t w = static_cast<uint64_t>(cam.width), h = static_cast<uint64_t>(cam.height);
        const uint64_t at = ((static_cast<uint64_t>(points[0].py) * w) + static_cast<uint64_t>(points[0].px)) * 3;
        if (heavy)
        {
            // the whole pixel is done again by a wave of its own: nothing of it is counted here
            added = depositsAccum = 0;
            a.heavyPixels[atomicAdd(a.heavyCount, 1ull)] = static_cast<uint32_t>(i);
        }
        else if (added != 0 && at + 2 < w * h * 3)
        {
            atomicAdd(cam.buffer + at + 0, acc.r);
            atomicAdd(cam.buffer + at + 1, acc.g);
            atomicAdd(cam.buffer + at + 2, acc.b);
        }
    }
    const unsigned long long totals[2] = {waveSum(added), waveSum(depositsAccum)};
    if ((threadIdx.x & 63) == 0)
    {
        if (totals[0] != 0) atomicAdd(a.stats + 0, totals[0]);
        if (totals[1] != 0) atomicAdd(a.stats + 1, totals[1]);
    }
}

// The same gather with G lanes (8 or 64) working on one pixel: where a gather point's footprint holds
// hundreds of records (few camera pixels looking at a densely lit surface) one lane per pixel leaves
// most of the chip idle and walks thousands of records in a row.  The G lanes of a group take G
// consecutive records of a row of cells, each lane runs the tests and the BRDF of its record, and
// the contributions that passed are then added in record order by every lane of the group alike —
// the additions, and with them the rounding, are exactly those of the one-lane walk.
// sum += the contributions of the lanes in passMask, one after the other in lane order (the rounding of
// a walk that adds record by record), for a whole wave at once.  The lanes that passed hand their
// contributions to lanes 0 .. passed - 1 in their order (one permutation through the LDS crossbar);
// lane 0 adds the sum so far to its own; then every lane adds its contribution to its lower
// neighbour's running sum (a DPP shift by one lane, three instructions a step for the three colours):
// after step k lanes 0 .. k hold the sums a serial walk would hold after their records, and steps
// beyond the last change none of them.  Lane passed - 1 ends with the new sum.  (Reading the lanes one
// by one into scalar registers costs eight instructions a record.)
RT_DEV void orderedAdd(unsigned long long passMask, uint32_t passed, const Color& contribution, Color* sum)
{
    const uint32_t lane = threadIdx.x & 63;
    const bool passes = ((passMask >> lane) & 1ull) != 0;
    const uint32_t below = lanesBelow(passMask);
    // those that passed go to the front in their order, the others behind them in theirs: a permutation
    const int target = static_cast<int>(passes ? below : passed + (lane - below)) * 4;
    float r = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(target, __builtin_bit_cast(int, contribution.r)));
    float g = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(target, __builtin_bit_cast(int, contribution.g)));
    float b = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(target, __builtin_bit_cast(int, contribution.b)));
    if (lane == 0)
    {
        r = sum->r + r;
        g = sum->g + g;
        b = sum->b + b;
    }
    float runR = r, runG = g, runB = b;
    // four steps a trip (a lane without a lower neighbour keeps what it has); the s_nop covers the two wait
    // states between a vector write and a DPP read of the same register, which later steps get from the
    // two instructions in between
    for (uint32_t step = 1; step < passed; step += 4)
    {
        asm volatile(
            "s_nop 1\n"
            "v_add_f32_dpp %0, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %0, %0, %3 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %1, %1, %4 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "v_add_f32_dpp %2, %2, %5 wave_shr:1 row_mask:0xf bank_mask:0xf\n"
            "s_nop 1\n"
            : "+v"(runR), "+v"(runG), "+v"(runB)
            : "v"(r), "v"(g), "v"(b));
    }
    sum->r = laneValue(runR, static_cast<int>(passed - 1));
    sum->g = laneValue(runG, static_cast<int>(passed - 1));
    sum->b = laneValue(runB, static_cast<int>(passed - 1));
}

template <int G>
RT_DEV uint32_t groupBits(unsigned long long waveMask, int groupBase)
{
    return G == 64 ? 0u : static_cast<uint32_t>((waveMask >> groupBase) & ((1ull << (G & 63)) - 1));
}

// Several waves on one pixel (gatherBlockKernel): what a wave's 64 records contribute waits here, in two
// sets used in turn, for the block's first wave to add it in record order.
template <int W>
struct GatherStage
{
    unsigned long long mask[2][W];
    float r[2][W][64], g[2][W][64], b[2][W][64];
};

// W > 1 (with G = 64): the block's W waves take the footprint's records 64 x W at a time, a wave 64 of them;
// every wave runs the tests and the BRDF of its records, and the first wave then adds what passed, wave by
// wave, in the order of the records.  The walk of a crowded pixel — thousands of records, each step a round
// trip to memory — is then W steps wide, and the additions, the one part that has to be serial, are all that
// stays so.  Every wave works out the rows' ranges for itself (they are the same numbers), so that the loops
// around the barriers run the same number of times in all of them; the result is the first wave's.
template <int G, int W = 1>
RT_DEV Color gatherRadianceGroup(const GatherArgs& a, const GatherPoint& gp, int member, int groupBase,
                                 unsigned long long* deposits, GatherStage<W>* stage = nullptr, int* stageTurn = nullptr)
{
    static_assert(W == 1 || G == 64, "several waves a pixel only with a whole wave a group");
    const int wave = W > 1 ? static_cast<int>(threadIdx.x >> 6) : 0;
    // which set of the stage this round writes: the sets alternate from round to round across gather points
    // and pixels alike (the caller keeps the count), so that a round never writes the set the first wave may
    // still be reading
    int turn = W > 1 ? *stageTurn : 0;
    struct KeepTurn
    {
        int* to;
        const int* from;
        RT_DEV ~KeepTurn()
        {
            if (to != nullptr) *to = *from;
        }
    } keepTurn{W > 1 ? stageTurn : nullptr, &turn};
    const Color black{0.0f, 0.0f, 0.0f};
    const bool isEmitter = gp.material < 0;
    const Vec wo = mk(gp.viewDir[0], gp.viewDir[1], gp.viewDir[2]);
    const Vec normal = mk(gp.normal[0], gp.normal[1], gp.normal[2]);
    const Vec position = mk(gp.position[0], gp.position[1], gp.position[2]);
    if (dot(wo, normal) <= 0.0)
    {
        return black;
    }
    const double r = (a.minGatherRadius > 0.0 && gp.footprintRadius < a.minGatherRadius) ? a.minGatherRadius
                                                                                         : gp.footprintRadius;
    if (r <= 0.0 || a.indexed == 0)
    {
        return black;
    }
    const double r2 = r * r;
    const double planeBand = 2.0 * r;
    const int64_t cx = static_cast<int64_t>(__builtin_floor(position.x * a.invCellSize));
    const int64_t cy = static_cast<int64_t>(__builtin_floor(position.y * a.invCellSize));
    const int64_t cz = static_cast<int64_t>(__builtin_floor(position.z * a.invCellSize));
    const int64_t reach = static_cast<int64_t>(__builtin_ceil(r * a.invCellSize));
    // the cube of cells, cut to the box of the cells that hold records (the keys are relative to it)
    const int64_t x0 = max(cx - reach, static_cast<int64_t>(a.layout.low[0])), x1 = min(cx + reach, static_cast<int64_t>(a.layout.high[0]));
    const int64_t y0 = max(cy - reach, static_cast<int64_t>(a.layout.low[1])), y1 = min(cy + reach, static_cast<int64_t>(a.layout.high[1]));
    const int64_t z0 = max(cz - reach, static_cast<int64_t>(a.layout.low[2])), z1 = min(cz + reach, static_cast<int64_t>(a.layout.high[2]));
    if (x0 > x1 || y0 > y1 || z0 > z1)
    {
        return black;
    }
    const Material* material = isEmitter ? nullptr : &a.scene.materials[gp.material];
    Color sum = black;
    unsigned long long kept = 0;
    const bool bySlab = (y1 - y0) > 16;  // as in gatherRadiance
    // The rows of cells are walked in the order of gatherRadiance (z outside, y inside), but where each
    // row starts in the key list is looked up G rows at a time, one row per lane of the group: the
    // binary searches, chains of dependent reads, run side by side instead of one after the other.
    const int64_t rowsPerSlab = bySlab ? 1 : (y1 - y0 + 1);
    const int64_t rowCount = (z1 - z0 + 1) * rowsPerSlab;
    for (int64_t rowBase = 0; rowBase < rowCount; rowBase += G)
    {
        // records [myStart, myEnd) of the row this lane looks up (see gatherRadiance)
        uint32_t myStart = 0, myEnd = 0;
        if (rowBase + member < rowCount)
        {
            const int64_t mine = rowBase + member;
            const int64_t y = y0 + mine % rowsPerSlab, z = z0 + mine / rowsPerSlab;
            uint32_t from = 0, to = 0;
            const bool listed = directoryEntries(a, x0, x1, y, bySlab ? y1 : y, z, &from, &to);
            if (from < to)
            {
                myStart = listed ? a.cellStart[from] : 0u;
                myEnd = listed ? a.cellStart[to] : static_cast<uint32_t>(a.indexed);
                if (a.directoryKind != kDirectoryCells)
                {
                    const unsigned long long first = cellKey(a.layout, x0, y, z);
                    const unsigned long long last = bySlab ? cellKey(a.layout, x1, y1, z) : cellKey(a.layout, x1, y, z);
                    myStart += static_cast<uint32_t>(lowerBound(a.keys + myStart, myEnd - myStart, first));
                    myEnd = myStart + static_cast<uint32_t>(lowerBound(a.keys + myStart, myEnd - myStart, last + 1));
                }
            }
        }
        // One step of the walk: lane `member` has read record `at` (fetch), runs the tests and the BRDF of its
        // record (inside: it has one), and the contributions that passed are added in lane order by every lane
        // alike (settle).
        typedef float Float4At4 __attribute__((ext_vector_type(4), aligned(4)));
        struct Fetched
        {
            unsigned long long key;
            Float4At4 q0, q1, q2;
            float q3;
        };
        auto fetch = [&](uint64_t at) {
            const float* row = a.store + 13ull * at;
            Fetched f;
            f.key = bySlab ? a.keys[at] : 0ull;
            f.q0 = *reinterpret_cast<const Float4At4*>(row);
            f.q1 = *reinterpret_cast<const Float4At4*>(row + 4);
            f.q2 = *reinterpret_cast<const Float4At4*>(row + 8);
            f.q3 = row[12];
            return f;
        };
        auto settle = [&](const Fetched& f, bool inside) {
            const unsigned long long key = f.key;
            const Float4At4 q0 = f.q0, q1 = f.q1, q2 = f.q2;
            const float q3 = f.q3;
            bool passes = inside;
            if (passes && bySlab)
            {
                const int64_t kx = keyCellX(a.layout, key);
                passes = kx >= x0 && kx <= x1;
            }
            Color contribution = black;
            if (passes)
            {
                const double ddx = static_cast<double>(q0.x) - position.x;
                const double ddy = static_cast<double>(q0.y) - position.y;
                const double ddz = static_cast<double>(q0.z) - position.z;
                passes = ddx * ddx + ddy * ddy + ddz * ddz <= r2;
            }
            if (passes)
            {
                const float time = q2.y;
                if (!__builtin_isinf(time) || time < 0.0f)  // +inf marks a timeless (emitter) deposit
                {
                    passes = !(__builtin_fabsf(time - gp.sampleTime) > a.timeHalfWindow);
                }
            }
            if (passes)
            {
                const Vec dn = mk(q1.z, q1.w, q2.x);
                const Color power{q2.z, q2.w, q3};
                if (isEmitter)
                {
                    passes = !(dot(dn, normal) < 0.5);
                    contribution = power;
                }
                else
                {
                    const double dnLen = magnitude(dn);
                    if (dnLen > 1e-4)
                    {
                        passes = !(dot(dn / dnLen, normal) < 0.5);
                    }
                    if (passes)
                    {
                        const double planeDist = dabs(dot(mk(q0.x, q0.y, q0.z) - position, normal));
                        passes = !(planeDist > planeBand);
                    }
                    if (passes)
                    {
                        const Vec wi = neg(mk(q0.w, q1.x, q1.y));
                        contribution = evaluateMaterial(*material, wi, wo, normal) * power;
                    }
                }
            }
            // the group's verdicts, then the additions in record order
            const unsigned long long passMask = __ballot(passes);
            if (G == 64 && W > 1)
            {
                if (member == 0) stage->mask[turn][wave] = passMask;
                stage->r[turn][wave][member] = contribution.r;
                stage->g[turn][wave][member] = contribution.g;
                stage->b[turn][wave][member] = contribution.b;
            }
            else if (G == 64)
            {
                const uint32_t passed = static_cast<uint32_t>(__builtin_popcountll(passMask));
                kept += passed;
                if (passed != 0)
                {
                    orderedAdd(passMask, passed, contribution, &sum);
                }
            }
            else
            {
                uint32_t todo = groupBits<G>(passMask, groupBase);
                kept += static_cast<unsigned long long>(__builtin_popcount(todo));
                while (todo != 0)
                {
                    const int k = __builtin_ctz(todo);
                    todo &= todo - 1;
                    sum.r += __shfl(contribution.r, k, G);
                    sum.g += __shfl(contribution.g, k, G);
                    sum.b += __shfl(contribution.b, k, G);
                }
            }
        };
        if (G == 64)
        {
            // A wave per pixel: the rows' records are laid end to end and taken 64 at a time, whichever rows
            // they belong to — a footprint's rows hold a few records each, and walked row by row every row
            // would cost a trip to memory for a handful of busy lanes.  The order of the records, and with it
            // of the additions, is the same: rows in their order, a row's records in theirs.
            const uint32_t length = myEnd > myStart ? myEnd - myStart : 0u;
            uint32_t upTo = length;  // records of this lane's row and the rows before it
#pragma unroll
            for (int d = 1; d < 64; d <<= 1)
            {
                const uint32_t before = __shfl_up(upTo, d, 64);
                if (member >= d) upTo += before;
            }
            const uint32_t total = laneValue(upTo, 63);
            // the record at `place` of the rows laid end to end (a lane beyond the end looks at the last one)
            auto recordAt = [&](uint32_t base) {
                const uint32_t mine = base + static_cast<uint32_t>(member);
                const uint32_t place = mine < total ? mine : total - 1;
                // the row that holds the place: the first whose running total exceeds it
                uint32_t holder = 0;
#pragma unroll
                for (uint32_t s = 32; s > 0; s >>= 1)
                {
                    const uint32_t edge = __shfl(upTo, static_cast<int>(holder + s - 1), 64);
                    if (edge <= place) holder += s;
                }
                const uint32_t holderUpTo = __shfl(upTo, static_cast<int>(holder), 64);
                const uint32_t holderEnd = __shfl(myEnd, static_cast<int>(holder), 64);
                return static_cast<uint64_t>(holderEnd - (holderUpTo - place));
            };
            if (W > 1)
            {
                // (total is the same in every wave: the barrier is reached by all of them or none)
                for (uint32_t round = 0; round < total; round += 64u * W)
                {
                    const uint32_t base = round + 64u * static_cast<uint32_t>(wave);
                    settle(fetch(recordAt(base)), base + static_cast<uint32_t>(member) < total);
                    __syncthreads();
                    if (wave == 0)
                    {
#pragma unroll
                        for (int w = 0; w < W; ++w)
                        {
                            const unsigned long long passMask = stage->mask[turn][w];
                            const uint32_t passed = static_cast<uint32_t>(__builtin_popcountll(passMask));
                            kept += passed;
                            if (passed != 0)
                            {
                                const Color staged{stage->r[turn][w][member], stage->g[turn][w][member], stage->b[turn][w][member]};
                                orderedAdd(passMask, passed, staged, &sum);
                            }
                        }
                    }
                    // the next round writes the other set: nobody writes this one again before the first wave,
                    // done with it, has passed the next barrier
                    turn ^= 1;
                }
            }
            else
            {
                for (uint32_t base = 0; base < total; base += 64)
                {
                    settle(fetch(recordAt(base)), base + static_cast<uint32_t>(member) < total);
                }
            }
        }
        else
        {
            const int rowsHere = static_cast<int>(rowCount - rowBase < G ? rowCount - rowBase : G);
            for (int t = 0; t < rowsHere; ++t)
            {
                const uint32_t start = __shfl(myStart, t, G);
                const uint32_t end = __shfl(myEnd, t, G);
                if (start >= end)
                {
                    continue;  // the same for every lane of the group
                }
                for (uint64_t base = start;; base += G)
                {
                    const uint64_t j = base + static_cast<uint64_t>(member);
                    const bool inside = j < end;
                    settle(fetch(inside ? j : end - 1), inside);
                    if (groupBits<G>(__ballot(!inside), groupBase) != 0) break;
                }
            }
        }
    }
    *deposits = kept;
    if (kept == 0)
    {
        return black;
    }
    const double area = kPi * r * r;
    const float scale = static_cast<float>((4.0 / kPi) / area);
    return sum * scale;
}

template <int G>
__global__ __launch_bounds__(256) void gatherGroupKernel(const GatherArgs a)
{
    const int64_t thread = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const int member = static_cast<int>(thread % G);
    const int groupBase = static_cast<int>(threadIdx.x & 63) - member;
    unsigned long long added = 0, depositsAccum = 0;
    // all the pixels, a group each, or (second launch of the one-lane gather) the listed ones, which the
    // grid's groups share out among themselves
    const int64_t groups = static_cast<int64_t>(gridDim.x) * (blockDim.x / G);
    const int64_t listed = a.pixelList != nullptr ? static_cast<int64_t>(*a.pixelListCount) : 0;
    for (int64_t turn = thread / G; a.pixelList != nullptr ? turn < listed : turn == thread / G; turn += groups)
    {
        const int64_t i = a.pixelList != nullptr ? static_cast<int64_t>(a.pixelList[turn]) : turn;
        if (!(i < a.pixels && a.counts[i] != 0))
        {
            continue;
        }
        const Camera& cam = a.scene.cameras[a.camera];
        const GatherPoint* points = a.points + (a.offsets ? a.offsets[i] : static_cast<uint64_t>(i));
        Color acc{0.0f, 0.0f, 0.0f};
        unsigned long long addedHere = 0;
        for (uint32_t k = 0; k < a.counts[i]; ++k)
        {
            const GatherPoint& gp = points[k];
            if (gp.material >= a.scene.materialCount)
            {
                continue;
            }
            unsigned long long deposits = 0;
            const Color radiance = gatherRadianceGroup<G>(a, gp, member, groupBase, &deposits);
            const Color throughput{gp.throughput[0], gp.throughput[1], gp.throughput[2]};
            const Color contribution = throughput * radiance * gp.sampleWeight;
            depositsAccum += deposits;
            if (contribution.r == 0.0f && contribution.g == 0.0f && contribution.b == 0.0f)
            {
                continue;
            }
            acc.r += contribution.r;
            acc.g += contribution.g;
            acc.b += contribution.b;
            addedHere += 1;
        }
        added += addedHere;
        const uint64_t w = static_cast<uint64_t>(cam.width), h = static_cast<uint64_t>(cam.height);
        const uint64_t at = ((static_cast<uint64_t>(points[0].py) * w) + static_cast<uint64_t>(points[0].px)) * 3;
        if (member == 0 && addedHere != 0 && at + 2 < w * h * 3)
        {
            atomicAdd(cam.buffer + at + 0, acc.r);
            atomicAdd(cam.buffer + at + 1, acc.g);
            atomicAdd(cam.buffer + at + 2, acc.b);
        }
    }
    // every lane of a group carries the group's tallies: count them once
    const unsigned long long totals[2] = {waveSum(member == 0 ? added : 0ull), waveSum(member == 0 ? depositsAccum : 0ull)};
    if ((threadIdx.x & 63) == 0)
    {
        if (totals[0] != 0) atomicAdd(a.stats + 0, totals[0]);
        if (totals[1] != 0) atomicAdd(a.stats + 1, totals[1]);
    }
}

// The pixels of a list, W waves each (see gatherRadianceGroup): a block takes the list's pixels in turns.
template <int W>
__global__ __launch_bounds__(64 * W) void gatherBlockKernel(const GatherArgs a)
{
    __shared__ GatherStage<W> stage;
    int stageTurn = 0;
    const int member = static_cast<int>(threadIdx.x & 63);
    const bool first = threadIdx.x < 64;  // the wave that holds the sums
    unsigned long long added = 0, depositsAccum = 0;
    const int64_t listed = static_cast<int64_t>(*a.pixelListCount);
    for (int64_t turn = blockIdx.x; turn < listed; turn += gridDim.x)
    {
        const int64_t i = static_cast<int64_t>(a.pixelList[turn]);
        if (!(i < a.pixels && a.counts[i] != 0))
        {
            continue;  // (the same for the whole block)
        }
        const Camera& cam = a.scene.cameras[a.camera];
        const GatherPoint* points = a.points + (a.offsets ? a.offsets[i] : static_cast<uint64_t>(i));
        Color acc{0.0f, 0.0f, 0.0f};
        unsigned long long addedHere = 0;
        for (uint32_t k = 0; k < a.counts[i]; ++k)
        {
            const GatherPoint& gp = points[k];
            if (gp.material >= a.scene.materialCount)
            {
                continue;
            }
            unsigned long long deposits = 0;
            const Color radiance = gatherRadianceGroup<64, W>(a, gp, member, 0, &deposits, &stage, &stageTurn);
            if (!first)
            {
                continue;  // (no barrier below this line)
            }
            const Color throughput{gp.throughput[0], gp.throughput[1], gp.throughput[2]};
            const Color contribution = throughput * radiance * gp.sampleWeight;
            depositsAccum += deposits;
            if (contribution.r == 0.0f && contribution.g == 0.0f && contribution.b == 0.0f)
            {
                continue;
            }
            acc.r += contribution.r;
            acc.g += contribution.g;
            acc.b += contribution.b;
            addedHere += 1;
        }
        added += addedHere;
        const uint64_t w = static_cast<uint64_t>(cam.width), h = static_cast<uint64_t>(cam.height);
        const uint64_t at = ((static_cast<uint64_t>(points[0].py) * w) + static_cast<uint64_t>(points[0].px)) * 3;
        if (threadIdx.x == 0 && addedHere != 0 && at + 2 < w * h * 3)
        {
            atomicAdd(cam.buffer + at + 0, acc.r);
            atomicAdd(cam.buffer + at + 1, acc.g);
            atomicAdd(cam.buffer + at + 2, acc.b);
        }
    }
    if (threadIdx.x == 0)
    {
        if (added != 0) atomicAdd(a.stats + 0, added);
        if (depositsAccum != 0) atomicAdd(a.stats + 1, depositsAccum);
    }
}

// ---------------------------------------------------------------- emitter deposits
// ProbeGather::depositEmitters, reference src/ProbeGather.cpp:464-550: sample (iu, iv) of a
// patch's grid, or no sample when it falls outside a disc.  The arithmetic is the reference's.
RT_DEV bool emitterSample(const EmitterGrid& g, long long index, Vec* position)
{
    const int iu = static_cast<int>(index / g.countV), iv = static_cast<int>(index % g.countV);
    const double u = -g.extentU + (iu + 0.5) * g.stepU;
    const double v = -g.extentV + (iv + 0.5) * g.stepV;
    if (g.isDisc && (u * u + v * v) > (g.radius * g.radius))
    {
        return false;
    }
    const Vec center = mk(g.center[0], g.center[1], g.center[2]);
    const Vec right = mk(g.right[0], g.right[1], g.right[2]), up = mk(g.up[0], g.up[1], g.up[2]);
    *position = center + right * u + up * v;
    return true;
}

__global__ void emitterCountKernel(const EmitterGrid g, unsigned long long* count)
{
    const long long total = static_cast<long long>(g.countU) * g.countV;
    unsigned long long mine = 0;
    for (long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; i < total;
         i += static_cast<long long>(gridDim.x) * blockDim.x)
    {
        Vec p;
        if (emitterSample(g, i, &p)) mine += 1;
    }
    mine = waveSum(mine);
    if ((threadIdx.x & 63) == 0 && mine != 0) atomicAdd(count, mine);
}

__global__ void emitterDepositKernel(const EmitterGrid g, const ProbeView probes, const StoreView store,
                                     unsigned long long* kept)
{
    const long long total = static_cast<long long>(g.countU) * g.countV;
    const long long stride = static_cast<long long>(gridDim.x) * blockDim.x;
    const long long rounds = (total + stride - 1) / stride;
    unsigned long long mine = 0;
    for (long long round = 0; round < rounds; ++round)  // whole waves stay together for the append
    {
        const long long i = round * stride + static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
        Vec p = mk(0, 0, 0);
        const bool keep = i < total && emitterSample(g, i, &p) && anyProbeWithin(probes, p);
        const Vec n = mk(g.normal[0], g.normal[1], g.normal[2]);
        // incoming direction = normal, time = +inf: a timeless deposit (reference :528-541)
        storeAppend(store, keep, p, n, n, __builtin_inff(), Color{g.power[0], g.power[1], g.power[2]}, StoreOrder{0, 0, 0, 0, 0});
        if (keep) mine += 1;
    }
    mine = waveSum(mine);
    if ((threadIdx.x & 63) == 0 && mine != 0) atomicAdd(kept, mine);
}

// positions of a camera's gather points, 3 doubles each: the probes of the keep test — from records that lie
// at their pixels' places (counts[i] is 0 or 1), written together, and from records that lie together
__global__ void pixelProbePositionsKernel(const GatherPoint* slots, const uint32_t* counts, const uint64_t* offsets, int64_t pixels,
                                          double* positions)
{
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= pixels || counts[i] == 0) return;
    double* out = positions + 3 * offsets[i];
    out[0] = slots[i].position[0];
    out[1] = slots[i].position[1];
    out[2] = slots[i].position[2];
}

__global__ void probePositionsKernel(const GatherPoint* points, uint64_t n, double* positions)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    positions[3 * i + 0] = points[i].position[0];
    positions[3 * i + 1] = points[i].position[1];
    positions[3 * i + 2] = points[i].position[2];
}

// ---------------------------------------------------------------- probe grid build
// ProbeIndex's constructor (reference src/ProbeIndex.cpp:5-25) hashes every probe into an
// unordered_map of cells.  Here the probes are keyed by cell (orderedProbeCell), sorted, and the runs of
// equal keys become the dense array of cells; the bricks (see ProbeView) are entered into their table
// by the groups of 2 x 2 x 2 cells, each of which lies in eight bricks.
__global__ void probeKeysKernel(const ProbeBuild b, uint32_t* unpackable)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= b.n) return;
    const int64_t x = static_cast<int64_t>(__builtin_floor(b.points[3 * i + 0] * b.invCellSize));
    const int64_t y = static_cast<int64_t>(__builtin_floor(b.points[3 * i + 1] * b.invCellSize));
    const int64_t z = static_cast<int64_t>(__builtin_floor(b.points[3 * i + 2] * b.invCellSize));
    const int64_t lim = kPackLimit - 4;  // leaves room for the bricks around a border cell
    const bool fits = x > -lim && x < lim && y > -lim && y < lim && z > -lim && z < lim;
    if (!fits)
    {
        atomicAdd(unpackable, 1u);
    }
    b.keys[i] = fits ? orderedProbeCell(x, y, z) : kEmptyKey;
    b.order[i] = static_cast<uint32_t>(i);
}

// runs[0] += number of runs of equal (key >> shift); a block counts 16 keys per
// thread and adds once, so that the one counter sees a few thousand atomics, not millions
constexpr int kRunsPerThread = 16;
__global__ void countRunsKernel(const unsigned long long* keys, uint64_t n, int shift, unsigned long long* runs)
{
    __shared__ unsigned int blockHeads;
    if (threadIdx.x == 0) blockHeads = 0;
    __syncthreads();
    const uint64_t blockBase = static_cast<uint64_t>(blockIdx.x) * blockDim.x * kRunsPerThread;
    unsigned int heads = 0;
    for (int k = 0; k < kRunsPerThread; ++k)
    {
        const uint64_t i = blockBase + static_cast<uint64_t>(k) * blockDim.x + threadIdx.x;
        if (i < n && (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift))) heads += 1;
    }
#pragma unroll
    for (int offset = 32; offset > 0; offset >>= 1)
    {
        heads += __shfl_down(heads, offset, 64);
    }
    if ((threadIdx.x & 63) == 0 && heads != 0) atomicAdd(&blockHeads, heads);
    __syncthreads();
    if (threadIdx.x == 0 && blockHeads != 0) atomicAdd(runs, static_cast<unsigned long long>(blockHeads));
}

__global__ void runHeadsKernel(const unsigned long long* keys, uint64_t n, int shift, uint32_t* heads)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    heads[i] = (i == 0 || (keys[i] >> shift) != (keys[i - 1] >> shift)) ? 1u : 0u;
}

// every probe is copied to its place in cell order; the first probe of every cell writes the cell
__global__ void denseCellsKernel(const unsigned long long* keys, const uint32_t* order, const uint64_t* ordinal,
                                 const double* points, uint64_t n, DenseCell* cells, double* grouped)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t src = order[i];
    DenseCell cell;
    cell.first[0] = points[3 * static_cast<uint64_t>(src) + 0];
    cell.first[1] = points[3 * static_cast<uint64_t>(src) + 1];
    cell.first[2] = points[3 * static_cast<uint64_t>(src) + 2];
    grouped[3 * i + 0] = cell.first[0];
    grouped[3 * i + 1] = cell.first[1];
    grouped[3 * i + 2] = cell.first[2];
    const unsigned long long key = keys[i];
    if (i != 0 && keys[i - 1] == key)
    {
        return;
    }
    uint32_t count = 1;
    while (i + count < n && keys[i + count] == key)
    {
        count += 1;
    }
    cell.start = static_cast<uint32_t>(i);
    cell.count = count;
    cells[ordinal[i]] = cell;
}

// A group is the run of sorted keys that agree above their low three bits: the cells of one block of
// 2 x 2 x 2 (x >> 1, y >> 1, z >> 1).  The group at (hx, hy, hz) lies in the eight bricks at (hx - sx, hy - sy,
// hz - sz), s = 0 or 1 per axis (brick coordinates are in units of two cells).  No atomic operation
// gathers a brick's eight groups: the groups are in sorted order, so the list of the bricks they lie in
// at one s is sorted too (a packed key minus a constant), the eight lists are merged, and the entries of
// one brick then follow each other.
constexpr int kGroupIndexBits = 29;
__global__ void cellGroupsKernel(const unsigned long long* keys, const uint64_t* cellOrdinal, const uint64_t* groupOrdinal,
                                 uint64_t n, unsigned long long* groupKey, uint32_t* groupOwn, uint32_t* groupFirst)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long head = keys[i];
    if (i != 0 && (keys[i - 1] >> 3) == (head >> 3)) return;
    // the group's cells as bits of a brick whose origin is the group's: bit x + 4 y + 16 z, x, y, z in 0..1
    uint32_t own = 0;
    for (uint64_t j = i; j < n && (keys[j] >> 3) == (head >> 3); ++j)
    {
        const unsigned long long k = keys[j];
        own |= 1u << ((k & 1) + 4 * ((k >> 1) & 1) + 16 * ((k >> 2) & 1));
    }
    const int64_t hx = static_cast<int64_t>(head >> 43) - kPackLimit / 2;
    const int64_t hy = static_cast<int64_t>((head >> 23) & ((1ull << 20) - 1)) - kPackLimit / 2;
    const int64_t hz = static_cast<int64_t>((head >> 3) & ((1ull << 20) - 1)) - kPackLimit / 2;
    const uint64_t g = groupOrdinal[i];
    groupKey[g] = packProbeCell(hx, hy, hz);
    groupOwn[g] = own;
    groupFirst[g] = static_cast<uint32_t>(cellOrdinal[i]);
}

// the merged lists: the first entry of every brick gathers the brick and enters it into the table
__global__ void bricksKernel(const unsigned long long* keys, const uint32_t* entries, uint64_t total, const uint32_t* groupOwn,
                             const uint32_t* groupFirst, ProbeBrick* bricks, uint64_t brickMask)
{
    const uint64_t t = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const unsigned long long key = keys[t];
    if (t != 0 && keys[t - 1] == key) return;
    unsigned long long vacant = ~0ull;
    uint32_t base0 = ~0u, base1 = ~0u, base2 = ~0u, base3 = ~0u;
    // a brick has eight entries at most (one per group that lies in it): all of them are fetched before
    // any is looked at, and the groups they name likewise — two round trips, not one per entry
    constexpr int kMost = 8;
    uint32_t entry[kMost], own[kMost], first[kMost];
    bool mine[kMost];
    bool sofar = true;
#pragma unroll
    for (int k = 0; k < kMost; ++k)
    {
        const uint64_t u = t + k < total ? t + k : total - 1;
        mine[k] = sofar && t + k < total && keys[u] == key;
        sofar = mine[k];
        entry[k] = entries[u];
    }
#pragma unroll
    for (int k = 0; k < kMost; ++k)
    {
        const uint32_t g = entry[k] & ((1u << kGroupIndexBits) - 1u);
        own[k] = groupOwn[g];
        first[k] = groupFirst[g];
    }
#pragma unroll
    for (int k = 0; k < kMost; ++k)
    {
        if (!mine[k]) continue;
        const int sx = (entry[k] >> kGroupIndexBits) & 1, sy = (entry[k] >> (kGroupIndexBits + 1)) & 1, sz = entry[k] >> (kGroupIndexBits + 2);
        vacant &= ~(static_cast<unsigned long long>(own[k]) << (2 * sx + 8 * sy + 32 * sz));
        const int column = sx + 2 * sy;
        base0 = column == 0 ? min(base0, first[k]) : base0;
        base1 = column == 1 ? min(base1, first[k]) : base1;
        base2 = column == 2 ? min(base2, first[k]) : base2;
        base3 = column == 3 ? min(base3, first[k]) : base3;
    }
    const int64_t mx = static_cast<int64_t>(key >> 42) - kPackLimit;
    const int64_t my = static_cast<int64_t>((key >> 21) & ((1ull << 21) - 1)) - kPackLimit;
    const int64_t mz = static_cast<int64_t>(key & ((1ull << 21) - 1)) - kPackLimit;
    uint64_t slot = cellHash(mx, my, mz) & brickMask;
    while (atomicCAS(&bricks[slot].key, kEmptyKey, key) != kEmptyKey)
    {
        slot = (slot + 1) & brickMask;
    }
    bricks[slot].vacant = vacant;
    bricks[slot].base[0] = base0;
    bricks[slot].base[1] = base1;
    bricks[slot].base[2] = base2;
    bricks[slot].base[3] = base3;
}

int gridFor(int64_t n) { return static_cast<int>((n + 255) / 256); }

// dynamic LDS of a kernel that casts rays: one traversal stack per lane
size_t stackBytes(const SceneView& scene, int threads)
{
    return static_cast<size_t>(scene.stackDepth > 0 ? scene.stackDepth : 1) * static_cast<size_t>(threads) *
           sizeof(unsigned long long);
}

}  // namespace

#ifdef RT_DIAG
void printDiagnostics()
{
    unsigned long long host[32] = {};
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_rtDiag), sizeof(host)) != hipSuccess) return;
#if RT_DIAG == 2
    const char* names[] = {"kernel",       "casts",         "node loops",     "triangle loops", "refill",      "deposit",
                           "scatter",      "cast: box",     "cast: pose",     "cast: normal",   "cast: world", "deposit: keep"};
    for (int k = 0; k < 12; ++k)
    {
        fprintf(stderr, "[rt diag] wave cycles  %-15s %16llu  (%.1f %% of the kernel)\n", names[k], host[k],
                host[0] ? 100.0 * static_cast<double>(host[k]) / static_cast<double>(host[0]) : 0.0);
    }
#else
    const char* names[] = {"mesh casts", "outer rounds", "node steps", "triangle tests", "main loop", "keep tests",
                           "keep-test loads"};
    for (int k = 0; k < 7; ++k)
    {
        fprintf(stderr, "[rt diag] %-15s wave trips %12llu  active lanes %14llu  (%.1f / 64)\n", names[k], host[2 * k],
                host[2 * k + 1], host[2 * k] ? static_cast<double>(host[2 * k + 1]) / host[2 * k] : 0.0);
    }
#endif
    unsigned long long zero[32] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rtDiag), zero, sizeof(zero));
}
#endif

namespace
{

// The kernel variant a pass runs: sink, per-photon records, keyframed volumes.
typedef void (*TraceKernel)(const TraceArgs);

TraceKernel traceVariant(const TraceArgs& args)
{
    const bool record = args.records.recordPhotons > 0;
    const bool animated = args.scene.animationCount > 0;
    const bool grouped = args.scene.groupCount > 0;
    const int variant = (record ? 1 : 0) | (animated ? 2 : 0) | (grouped ? 4 : 0);
#define RT_TRACE_VARIANTS(sink)                                                  \
    switch (variant)                                                             \
    {                                                                            \
    case 0: return traceKernel<sink, false, false, false>;                       \
    case 1: return traceKernel<sink, true, false, false>;                        \
    case 2: return traceKernel<sink, false, true, false>;                        \
    case 3: return traceKernel<sink, true, true, false>;                         \
    case 4: return traceKernel<sink, false, false, true>;                        \
    case 5: return traceKernel<sink, true, false, true>;                         \
    case 6: return traceKernel<sink, false, true, true>;                         \
    default: return traceKernel<sink, true, true, true>;                         \
    }
    switch (args.sinkMode)
    {
    case kSinkProbe: RT_TRACE_VARIANTS(kSinkProbe)
    case kSinkSplat: RT_TRACE_VARIANTS(kSinkSplat)
    default: RT_TRACE_VARIANTS(kSinkNone)
    }
#undef RT_TRACE_VARIANTS
}

size_t traceLdsBytes(const SceneView& scene)
{
    return stackBytes(scene, kTraceThreads) + scene.cacheBytes;
}

}  // namespace

void launchEmissionKeys(const TraceArgs& args, uint32_t* keys, uint32_t* ids, hipStream_t stream)
{
    hipLaunchKernelGGL(emissionKeysKernel, dim3(gridFor(static_cast<int64_t>(args.totalWork))), dim3(256), 0, stream, args, keys,
                       ids);
}

hipError_t sortEmissionPairs(void* scratch, size_t* scratchBytes, const uint32_t* keysIn, uint32_t* keysOut,
                             const uint32_t* idsIn, uint32_t* idsOut, uint64_t n, int lightCount, int skipBits,
                             hipStream_t stream)
{
    int lightBits = 0;
    while ((1 << lightBits) < lightCount && lightBits < 8) ++lightBits;
    return hipcub::DeviceRadixSort::SortPairs(scratch, *scratchBytes, keysIn, keysOut, idsIn, idsOut, static_cast<int>(n),
                                              skipBits, 2 * kGlobalAxisBits + lightBits, stream);
}

namespace
{
__global__ void iotaKernel(uint32_t* values, uint64_t n)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) values[i] = static_cast<uint32_t>(i);
}
}  // namespace

void launchIota(uint32_t* values, uint64_t n, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(iotaKernel, dim3(static_cast<unsigned int>((n + 255) / 256)), dim3(256), 0, stream, values, n);
}

hipError_t sortPairs32(void* scratch, size_t* scratchBytes, const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valuesIn,
                       uint32_t* valuesOut, uint64_t n, int bits, hipStream_t stream)
{
    return hipcub::DeviceRadixSort::SortPairs(scratch, *scratchBytes, keysIn, keysOut, valuesIn, valuesOut, static_cast<int>(n), 0,
                                              bits, stream);
}

int traceBlockThreads() { return kTraceThreads; }

size_t traceCacheBudget(int stackDepth)
{
    // all of a CU's LDS but the stacks and a margin (the census builds keep a few words of their own)
    const size_t total = 160 * 1024 / (512 / kTraceThreads), margin = 8192;
    const size_t stacks = static_cast<size_t>(stackDepth > 0 ? stackDepth : 1) * kTraceThreads * sizeof(unsigned long long);
    return stacks + margin < total ? (total - stacks - margin) & ~static_cast<size_t>(63) : 0;
}

hipError_t launchTrace(const TraceArgs& args, int blocks, hipStream_t stream)
{
    const TraceKernel kernel = traceVariant(args);
    const size_t lds = traceLdsBytes(args.scene);
    if (lds > 64 * 1024)
    {
        // more dynamic LDS than a kernel gets by default has to be asked for, once per kernel
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 static_cast<int>(lds));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kTraceThreads), lds, stream, args);
    return hipGetLastError();
}

// Resident blocks per CU of the variant this pass runs (registers and LDS differ between them;
// the persistent grid is sized to exactly fill the chip).
int traceBlocksPerCu(const TraceArgs& args)
{
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, traceVariant(args), kTraceThreads,
                                                     traceLdsBytes(args.scene)) != hipSuccess ||
        blocks < 1)
    {
        blocks = 1;
    }
    return blocks;
}

void launchCastRays(const SceneView& scene, int64_t n, const double* rays, double selfHit, double* outF64,
                    int32_t* outI32, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(castRaysKernel, dim3(gridFor(n)), dim3(256), stackBytes(scene, 256), stream, scene, n, rays, selfHit, outF64, outI32);
}

void launchEmit(const SceneView& scene, int32_t light, int64_t n, const double* draws5, uint64_t totalPhotons,
                double* rays, float* colors, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(emitKernel, dim3(gridFor(n)), dim3(256), 0, stream, scene, light, n, draws5, totalPhotons, rays,
                       colors);
}

void launchIntersect(int kind, int64_t n, const double* in, void* out, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(intersectKernel, dim3(gridFor(n)), dim3(256), 0, stream, kind, n, in, out);
}

void launchBsdfSample(int64_t n, const double* in14, double* out10, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(bsdfSampleKernel, dim3(gridFor(n)), dim3(256), 0, stream, n, in14, out10);
}

void launchGatherCount(const SceneView& scene, int32_t camera, int32_t stride, int64_t pixels, uint32_t* counts,
                       unsigned long long* stats, hipStream_t stream)
{
    if (pixels <= 0) return;
    // a wave a block: pixels of a block end far apart (some rays leave the scene at once, others pass glass
    // and mirrors), and a block's place is free again only when its last wave ends (MirrorKnot4k's probe
    // pass 19.2 -> 12.0 ms)
    hipLaunchKernelGGL(gatherCountKernel, dim3(static_cast<uint32_t>((pixels + 63) / 64)), dim3(64), stackBytes(scene, 64), stream, scene,
                       camera, stride, counts, stats);
}

void launchGatherWrite(const SceneView& scene, int32_t camera, int32_t stride, int64_t pixels, const uint32_t* counts,
                       const uint64_t* offsets, GatherPoint* out, hipStream_t stream)
{
    if (pixels <= 0) return;
    hipLaunchKernelGGL(gatherWriteKernel, dim3(static_cast<uint32_t>((pixels + 63) / 64)), dim3(64), stackBytes(scene, 64), stream, scene,
                       camera, stride, counts, offsets, out);
}

void launchGatherCollect(const SceneView& scene, int32_t camera, int32_t stride, int64_t pixels, uint32_t* counts,
                         unsigned long long* stats, GatherPoint* slots, uint32_t* deferred, unsigned long long* deferredCount,
                         hipStream_t stream)
{
    if (pixels <= 0) return;
    const dim3 grid(static_cast<uint32_t>((pixels + 63) / 64)), block(64);
    hipLaunchKernelGGL(gatherPrimaryKernel, grid, block, stackBytes(scene, 64), stream, scene, camera, stride, counts, stats, slots,
                       deferred, deferredCount);
    hipLaunchKernelGGL(gatherCollectKernel, grid, block, stackBytes(scene, 64), stream, scene, camera, stride, counts, stats, slots,
                       deferred, deferredCount);
}

void launchGatherCompact(const GatherPoint* slots, const uint32_t* counts, const uint64_t* offsets, int64_t pixels, GatherPoint* out,
                         hipStream_t stream)
{
    if (pixels <= 0) return;
    const int64_t threads = pixels * static_cast<int64_t>(sizeof(GatherPoint) / sizeof(uint32_t));
    hipLaunchKernelGGL(gatherCompactKernel, dim3(static_cast<uint32_t>((threads + 255) / 256)), dim3(256), 0, stream, slots, counts,
                       offsets, pixels, out);
}

hipError_t buildCellDirectory(void* scratch, size_t* scratchBytes, const unsigned long long* sortedKeys, uint64_t indexed,
                              const KeyLayout& layout, const int32_t cellMin[3], const int32_t cellDim[3], uint32_t* cellStart,
                              hipStream_t stream)
{
    (void)cellMin;  // the keys are relative to the box already
    const size_t cells = static_cast<size_t>(cellDim[0]) * static_cast<size_t>(cellDim[1]) * static_cast<size_t>(cellDim[2]);
    // an empty cell starts where the next cell with records does: a running minimum from the end
    rocprim::reverse_iterator<uint32_t*> fromEnd(cellStart + cells + 1);
    if (scratch == nullptr)
    {
        return rocprim::inclusive_scan(nullptr, *scratchBytes, fromEnd, fromEnd, cells + 1, rocprim::minimum<uint32_t>(), stream);
    }
    if (indexed == 0) return hipErrorInvalidValue;
    const hipError_t status = hipMemsetAsync(cellStart, 0xFF, (cells + 1) * sizeof(uint32_t), stream);
    if (status != hipSuccess) return status;
    CellBox box;
    box.layout = layout;
    for (int k = 0; k < 3; ++k)
    {
        box.dim[k] = cellDim[k];
    }
    hipLaunchKernelGGL(cellFirstsKernel, dim3(gridFor(static_cast<int64_t>(indexed))), dim3(256), 0, stream, sortedKeys, indexed,
                       box, cellStart);
    return rocprim::inclusive_scan(scratch, *scratchBytes, fromEnd, fromEnd, cells + 1, rocprim::minimum<uint32_t>(), stream);
}

void launchStoreBox(const float* store, uint64_t n, double invCellSize, int32_t* cellBox, hipStream_t stream)
{
    if (n == 0) return;
    const int64_t blocks = std::min<int64_t>(2048, static_cast<int64_t>((n + 255) / 256));
    hipLaunchKernelGGL(storeBoxKernel, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, store, n, invCellSize, cellBox);
}

KeyLayout keyLayoutFor(const int32_t box[6])
{
    KeyLayout l{};
    int bits[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
    {
        l.low[k] = box[k];
        l.high[k] = box[3 + k];
        if (box[k] > box[3 + k])
        {
            l.low[k] = l.high[k] = 0;  // no record packs: nothing is keyed
        }
        // the greatest coordinate of the part, and for z one more (its all-ones value stays unused)
        const uint64_t most = static_cast<uint64_t>(static_cast<int64_t>(l.high[k]) - l.low[k]) + (k == 2 ? 1 : 0);
        while ((1ull << bits[k]) <= most) ++bits[k];
    }
    l.bitsX = bits[0];
    l.bitsY = bits[1];
    l.bits = bits[0] + bits[1] + bits[2];
    return l;
}

void launchStoreKeys(const float* store, uint64_t n, double invCellSize, const KeyLayout& layout, unsigned long long* keys,
                     uint32_t* order, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(storeKeysKernel, dim3(gridFor(static_cast<int64_t>(n))), dim3(256), 0, stream, store, n,
                       invCellSize, layout, keys, order);
}

hipError_t sortStoreKeys(void* scratch, size_t* scratchBytes, const unsigned long long* keysIn,
                         unsigned long long* keysOut, const uint32_t* orderIn, uint32_t* orderOut, uint64_t n, int bits,
                         hipStream_t stream)
{
    // radix sort is stable: records of one cell keep their store order.  Up to a million pairs the library
    // sorts by merging (a block sort and two launches for every doubling: two dozen launches for a store of a
    // million records); the keys here are 20 to 40 bits wide, three to five radix passes, which are the
    // faster way from a few ten thousand records on (RT_AMD_ONESWEEP_FROM=<records> moves the line).
    using RadixFromSmall = rocprim::radix_sort_config<rocprim::default_config, rocprim::default_config, rocprim::default_config, 4096>;
    static const uint64_t radixFrom = [] {
        const char* v = std::getenv("RT_AMD_ONESWEEP_FROM");
        return v != nullptr ? std::strtoull(v, nullptr, 10) : 65536ull;
    }();
    const unsigned int endBit = static_cast<unsigned int>(std::max(1, std::min(64, bits)));
    if (n >= radixFrom)
    {
        return rocprim::radix_sort_pairs<RadixFromSmall>(scratch, *scratchBytes, keysIn, keysOut, orderIn, orderOut,
                                                         static_cast<unsigned int>(n), 0u, endBit, stream);
    }
    return hipcub::DeviceRadixSort::SortPairs(scratch, *scratchBytes, keysIn, keysOut, orderIn, orderOut,
                                              static_cast<int>(n), 0, static_cast<int>(endBit), stream);
}

namespace
{
struct WidenCount
{
    __host__ __device__ uint64_t operator()(uint32_t v) const { return static_cast<uint64_t>(v); }
};
}  // namespace

hipError_t scanCounts(void* scratch, size_t* scratchBytes, const uint32_t* counts, uint64_t* offsets, int64_t n,
                      hipStream_t stream)
{
    hipcub::TransformInputIterator<uint64_t, WidenCount, const uint32_t*> wide(counts, WidenCount());
    return hipcub::DeviceScan::ExclusiveSum(scratch, *scratchBytes, wide, offsets, static_cast<int>(n), stream);
}

void launchCountIndexed(const unsigned long long* sortedKeys, uint64_t n, unsigned long long* out, hipStream_t stream)
{
    hipLaunchKernelGGL(countIndexedKernel, dim3(1), dim3(1), 0, stream, sortedKeys, n, out);
}

void launchReorderStore(const float* store, const uint32_t* order, uint64_t indexed, float* sorted, hipStream_t stream)
{
    if (indexed == 0) return;
    const uint64_t floats = 13ull * indexed;
    hipLaunchKernelGGL(reorderStoreKernel, dim3(static_cast<unsigned int>((floats + 255) / 256)), dim3(256), 0, stream, store,
                       order, floats, sorted);
}

void launchGather(const GatherArgs& args, int lanesPerPixel, int computeUnits, hipStream_t stream)
{
    if (args.pixels <= 0) return;
    if (lanesPerPixel >= 64)
    {
        // (a wave a block, as for the one-lane kernel)
        hipLaunchKernelGGL(gatherGroupKernel<64>, dim3(static_cast<uint32_t>(args.pixels)), dim3(64), 0, stream, args);
    }
    else if (lanesPerPixel >= 8)
    {
        hipLaunchKernelGGL(gatherGroupKernel<8>, dim3(gridFor(args.pixels * 8)), dim3(256), 0, stream, args);
    }
    else
    {
        hipLaunchKernelGGL(gatherKernel, dim3(static_cast<uint32_t>((args.pixels + kGatherThreads - 1) / kGatherThreads)), dim3(kGatherThreads), 0,
                           stream, args);
        if (args.heavyThreshold != 0)
        {
            // the pixels left over, a wave each: a grid that fills the chip takes them in turns
            GatherArgs rest = args;
            rest.heavyThreshold = 0;
            rest.pixelList = args.heavyPixels;
            rest.pixelListCount = args.heavyCount;
            // the pixels left over, two waves each (see gatherRadianceGroup): a grid that fills the chip takes them
            // in turns
            const uint32_t places = static_cast<uint32_t>(std::max(1, computeUnits) * 32);  // waves the grid holds
            if (args.heavyWaves >= 8)
            {
                hipLaunchKernelGGL(gatherBlockKernel<8>, dim3(places / 8), dim3(512), 0, stream, rest);
            }
            else if (args.heavyWaves >= 4)
            {
                hipLaunchKernelGGL(gatherBlockKernel<4>, dim3(places / 4), dim3(256), 0, stream, rest);
            }
            else if (args.heavyWaves >= 2)
            {
                hipLaunchKernelGGL(gatherBlockKernel<2>, dim3(places / 2), dim3(128), 0, stream, rest);
            }
            else
            {
                hipLaunchKernelGGL(gatherGroupKernel<64>, dim3(places), dim3(64), 0, stream, rest);
            }
        }
    }
}

void launchEmitterCount(const EmitterGrid& grid, unsigned long long* count, hipStream_t stream)
{
    const long long total = static_cast<long long>(grid.countU) * grid.countV;
    const int blocks = static_cast<int>(std::min<long long>(4096, (total + 255) / 256));
    hipLaunchKernelGGL(emitterCountKernel, dim3(blocks), dim3(256), 0, stream, grid, count);
}

void launchEmitterDeposit(const EmitterGrid& grid, const ProbeView& probes, const StoreView& store,
                          unsigned long long* kept, hipStream_t stream)
{
    const long long total = static_cast<long long>(grid.countU) * grid.countV;
    const int blocks = static_cast<int>(std::min<long long>(4096, (total + 255) / 256));
    hipLaunchKernelGGL(emitterDepositKernel, dim3(blocks), dim3(256), 0, stream, grid, probes, store, kept);
}

void launchProbePositions(const GatherPoint* points, uint64_t n, double* positions, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(probePositionsKernel, dim3(gridFor(static_cast<int64_t>(n))), dim3(256), 0, stream, points, n, positions);
}

void launchPixelProbePositions(const GatherPoint* slots, const uint32_t* counts, const uint64_t* offsets, int64_t pixels,
                               double* positions, hipStream_t stream)
{
    if (pixels <= 0) return;
    hipLaunchKernelGGL(pixelProbePositionsKernel, dim3(gridFor(pixels)), dim3(256), 0, stream, slots, counts, offsets, pixels,
                       positions);
}

void launchProbeKeys(const ProbeBuild& b, uint32_t* unpackable, hipStream_t stream)
{
    if (b.n == 0) return;
    hipLaunchKernelGGL(probeKeysKernel, dim3(gridFor(static_cast<int64_t>(b.n))), dim3(256), 0, stream, b, unpackable);
}

hipError_t sortKeyPairs(void* scratch, size_t* scratchBytes, const unsigned long long* keysIn, unsigned long long* keysOut,
                        const uint32_t* orderIn, uint32_t* orderOut, uint64_t n, hipStream_t stream)
{
    return hipcub::DeviceRadixSort::SortPairs(scratch, *scratchBytes, keysIn, keysOut, orderIn, orderOut,
                                              static_cast<int>(n), 0, 64, stream);
}

void launchCountRuns(const unsigned long long* sortedKeys, uint64_t n, int shift, unsigned long long* runs, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(countRunsKernel, dim3(gridFor(static_cast<int64_t>((n + kRunsPerThread - 1) / kRunsPerThread))), dim3(256),
                       0, stream, sortedKeys, n, shift, runs);
}

void launchRunHeads(const unsigned long long* sortedKeys, uint64_t n, int shift, uint32_t* heads, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(runHeadsKernel, dim3(gridFor(static_cast<int64_t>(n))), dim3(256), 0, stream, sortedKeys, n, shift, heads);
}

void launchDenseCells(const unsigned long long* sortedKeys, const uint32_t* sortedOrder, const uint64_t* ordinal,
                      const double* points, uint64_t n, DenseCell* cells, double* groupedPoints, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(denseCellsKernel, dim3(gridFor(static_cast<int64_t>(n))), dim3(256), 0, stream, sortedKeys, sortedOrder,
                       ordinal, points, n, cells, groupedPoints);
}

void launchCellGroups(const unsigned long long* sortedKeys, const uint64_t* cellOrdinal, const uint64_t* groupOrdinal, uint64_t n,
                      unsigned long long* groupKey, uint32_t* groupOwn, uint32_t* groupFirst, hipStream_t stream)
{
    if (n == 0) return;
    hipLaunchKernelGGL(cellGroupsKernel, dim3(gridFor(static_cast<int64_t>(n))), dim3(256), 0, stream, sortedKeys, cellOrdinal,
                       groupOrdinal, n, groupKey, groupOwn, groupFirst);
}

namespace
{
struct LessBy
{
    unsigned long long by;
    __host__ __device__ unsigned long long operator()(unsigned long long key) const { return key - by; }
};
}  // namespace

// the first merges read the lists straight from the groups' records: list s is "the group's key less the
// packed offset s" beside "the group's number with s above it"
hipError_t mergeBrickLists(void* scratch, size_t* scratchBytes, const unsigned long long* groupKey, uint64_t groups, int listA,
                           int listB, unsigned long long* keysOut, uint32_t* entriesOut, hipStream_t stream)
{
    auto offsetOf = [](int s) {
        return (static_cast<unsigned long long>(s & 1) << 42) + (static_cast<unsigned long long>((s >> 1) & 1) << 21) +
               static_cast<unsigned long long>(s >> 2);
    };
    typedef hipcub::TransformInputIterator<unsigned long long, LessBy, const unsigned long long*> Keys;
    typedef hipcub::CountingInputIterator<uint32_t> Entries;
    return hipcub::DeviceMerge::MergePairs(scratch, *scratchBytes, Keys(groupKey, LessBy{offsetOf(listA)}),
                                           Entries(static_cast<uint32_t>(listA) << kGroupIndexBits), static_cast<int>(groups),
                                           Keys(groupKey, LessBy{offsetOf(listB)}),
                                           Entries(static_cast<uint32_t>(listB) << kGroupIndexBits), static_cast<int>(groups),
                                           keysOut, entriesOut, ::rocprim::less<unsigned long long>(), stream);
}

hipError_t mergeKeyPairs(void* scratch, size_t* scratchBytes, const unsigned long long* keysA, const uint32_t* valuesA, uint64_t nA,
                         const unsigned long long* keysB, const uint32_t* valuesB, uint64_t nB, unsigned long long* keysOut,
                         uint32_t* valuesOut, hipStream_t stream)
{
    return hipcub::DeviceMerge::MergePairs(scratch, *scratchBytes, keysA, valuesA, static_cast<int>(nA), keysB, valuesB,
                                           static_cast<int>(nB), keysOut, valuesOut, ::rocprim::less<unsigned long long>(), stream);
}

void launchBricks(const unsigned long long* mergedKeys, const uint32_t* mergedEntries, uint64_t total, const uint32_t* groupOwn,
                  const uint32_t* groupFirst, ProbeBrick* bricks, uint64_t brickMask, hipStream_t stream)
{
    if (total == 0) return;
    hipLaunchKernelGGL(bricksKernel, dim3(gridFor(static_cast<int64_t>(total))), dim3(256), 0, stream, mergedKeys, mergedEntries,
                       total, groupOwn, groupFirst, bricks, brickMask);
}

void launchProbeTest(const ProbeView& probes, int64_t n, const double* points, uint8_t* out, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(probeTestKernel, dim3(gridFor(n)), dim3(256), 0, stream, probes, n, points, out);
}

void launchBsdfEvaluate(int64_t n, const double* in14, double* out4, hipStream_t stream)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(bsdfEvaluateKernel, dim3(gridFor(n)), dim3(256), 0, stream, n, in14, out4);
}

}  // namespace rtd
