// rt_kernels.h — kernel argument blocks and launch declarations shared by the
// kernels (rt_kernels.hip) and the C-ABI (rt_api.hip).
#pragma once

#include "rt_device.h"

namespace rtd
{

// Sink A: the probe keep-test grid (reference src/ProbeIndex.cpp) in HBM, and the raw bounce store
// (reference src/BounceStore.cpp).  The grid has two forms.
//
// The general form is an open-addressed table of cells (slotKey / slotStart / slotCount): the keep
// test looks up every cell within reach.
//
// The usual case — a keep radius of one cell, cell coordinates that pack into 3 x 21 bits — is built
// on the device in a form that costs one table lookup per test: the occupied cells lie in one dense
// array (`cells`, each with its first probe), ordered by (x >> 1, y >> 1, z, y & 1, x & 1), and a table of
// BRICKS tells where.  A brick is a block of 4 x 4 x 4 cells that starts at even coordinates, so the
// bricks overlap and the 27 cells around any cell c lie inside the one brick that starts at
// ((c - 1) >> 1) * 2.  Its entry holds one bit per cell (bit x + 4 y + 16 z in brick coordinates, CLEAR
// where the cell holds probes) and, for each of its four columns of 2 x 2 x 4 cells, the place in
// `cells` of the column's first occupied cell: the cells of a column follow each other in `cells` in
// the order of their bits, so a cell's place is its column's base plus the number of occupied cells
// of the column below its bit.  Only bricks that hold a probe have an entry.
struct ProbeBrick;
struct DenseCell;

struct ProbeView
{
    const int64_t* slotKey;      // [slots][3] cell coordinates
    const uint32_t* slotStart;   // [slots] first probe of the cell in `points`
    const uint32_t* slotCount;   // [slots] 0 = empty slot
    const double* points;        // [probes][3], grouped by cell (one more double may be read behind the last)
    uint64_t slotMask;           // slots - 1 (power of two)
    uint64_t probeCount;
    double invCellSize;
    double keepRadius;
    int64_t reach;               // ceil(keepRadius * invCellSize)
    uint64_t brickMask;          // brick slots - 1 (power of two); 0: the general form
    const ProbeBrick* bricks;    // [brickMask + 1], a free slot holds kEmptyKey
    const DenseCell* cells;
};

struct alignas(16) ProbeBrick  // 32 bytes
{
    unsigned long long key;     // packed brick coordinates (in units of two cells)
    unsigned long long vacant;  // bit x + 4 y + 16 z: clear where the cell holds probes
    uint32_t base[4];           // column (x >> 1) + 2 (y >> 1): the place of its first occupied cell in `cells`
};

struct alignas(16) DenseCell  // 32 bytes
{
    double first[3];        // the first of the cell's probes (most cells hold one probe: no second read)
    uint32_t start, count;  // all of them in `points`
};

// The store drops what arrives beyond its capacity (reference src/BounceStore.cpp:21-33), and which
// records those are is decided by the order the reference's workers meet the photons in: index order,
// light by light (reference src/Worker.cpp:429-505).  The device traces photons in an order of its own, so
// a store that overflows keeps taking records — into room behind its capacity, with each record's place
// in the reference's order beside it (`keys`) — and when the pass is over the records that come first in
// that order are moved into the capacity (launchStoreSelect*).  `room` is the capacity and `keys` null
// where no pass can overflow the store: then a record beyond the capacity is dropped on arrival.
struct StoreView
{
    float* records;                  // [room][13]
    unsigned long long* cursor;      // claimed slots, may exceed room
    uint64_t room;                   // records there is room for while a pass runs: the capacity, or more (see above)
    // null, or [-1 .. room): the key of the record in each slot — the photon's number in the pass (light x
    // photons per light + index - first index, below 2^32) above `keyBits` bits of bounce — and, at [-1], the
    // least key among the records that found no room at all
    unsigned long long* keys;
    int32_t keyBits;
};

// Sink B density grid (reference src/DensityGrid.cpp): one 64-bit packed cell key per
// slot claimed by compare-and-swap, power and count accumulated with atomics.
struct DensityView
{
    unsigned long long* keys;  // [slots], kEmptyKey when free
    float* power;              // [slots][3]
    unsigned long long* count; // [slots]
    uint64_t slotMask;
    double invCellSize;
    int32_t enabled;
};
constexpr unsigned long long kEmptyKey = ~0ull;
constexpr int32_t kSinkNone = 0, kSinkProbe = 1, kSinkSplat = 2;  // rt_sink_mode of include/rt_amd.h

struct DeviceCounters
{
    unsigned long long photons, casts, bounces, deposits, kept, culled;
    unsigned long long splatTotal, splatRadiusClamped, splatLuminanceClamped;
    unsigned long long densityCells, densityDeposits, densityOverflow;
    double power[3];
};

struct RecordView
{
    uint64_t recordPhotons;
    int32_t maxBounces;
    double* emitted;      // [M][6]
    int32_t* castsValid;  // [M][2]
    double* f64;          // [M][B][10]
    int32_t* i32;         // [M][B][8]
    float* power;         // [M][B][3]
};

// A photon between two bounces, parked in a queue so that the next bounce of all photons of a
// pass runs as a launch of its own (photons of different depth do not share waves).
struct alignas(16) PhotonState
{
    double origin[3], direction[3];
    float color[3], time;
    unsigned long long photonIndex;
    long long recordSlot;
    int32_t lightId, bounces, casts, valid;
};
static_assert(sizeof(PhotonState) == 96, "PhotonState layout");

struct TraceArgs
{
    SceneView scene;
    ProbeView probes;
    StoreView store;
    DensityView density;
    RecordView records;
    DeviceCounters* counters;
    unsigned long long* workCounter;
    uint32_t chunk;                 // photon ids a wave claims at a time (<= 4096, multiple of 64) ...
    uint32_t tailChunk;             // ... for the first `fullClaims` claims of the launch, then this many: the
    uint64_t fullClaims;            // end of a launch is handed out in small pieces so that the waves end together
    uint64_t claimStride;           // full claim c takes run (c * claimStride) % fullClaims of the list (1: in list order)
    const uint32_t* emissionOrder;  // optional: position in the pass -> photon id, sorted by emission cell
    uint64_t firstPhoton;
    uint64_t photonCount;  // per light
    uint64_t totalWork;    // lights * photonCount
    double minSplatRadius;
    int32_t sinkMode;
    // bounce-level queues (all null: photons are emitted from ids and followed to their end)
    const PhotonState* queueIn;               // work of this launch: parked photons instead of ids
    uint64_t queueInCount;                    // how many
    PhotonState* queueOut;                    // photons that go on after this launch's bounce are parked here
    unsigned long long* queueOutCount;        // slots claimed (may exceed queueOutCapacity)
    uint64_t queueOutCapacity;                // beyond it a photon is followed on in its lane
    // Parked photons are scattered over the scene in every direction.  Each one leaves a key of its
    // new ray behind (queueOutKeys[slot]: origin cell, then direction sector); the host sorts the
    // (key, slot) pairs and the next launch walks the queue through that order (queueInOrder), so
    // that the 64 photons of a wave start near each other and head the same way.
    uint32_t* queueOutKeys;                   // optional
    const uint32_t* queueInOrder;             // optional: position in the launch's work -> slot of queueIn
    float keyOrigin[3];                       // corner of the scene's box and 1 / (its longest side)
    float keyScale;
    int32_t keyMode;                          // how the key is laid out (see rayKey)
    int32_t splatTogether;                    // splat sink: lanes of a wave that hit one pixel add their sum once
};

hipError_t launchTrace(const TraceArgs& args, int blocks, hipStream_t stream);
int traceBlockThreads();                   // threads of a trace block (one block per CU)
size_t traceCacheBudget(int stackDepth);   // LDS bytes a trace block has for the scene cache behind its stacks
// Whole-pass emission order: keys and ids of every photon of the pass, then a radix sort of the pairs.
// The sorts and the scan below follow hipCUB's convention: with a null `scratch` nothing runs and
// *scratchBytes receives the work space the call needs; the library's status is returned.
void launchEmissionKeys(const TraceArgs& args, uint32_t* keys, uint32_t* ids, hipStream_t stream);

// (skipBits: the sort leaves out that many of the keys' lowest bits)
hipError_t sortEmissionPairs(void* scratch, size_t* scratchBytes, const uint32_t* keysIn, uint32_t* keysOut,
                             const uint32_t* idsIn, uint32_t* idsOut, uint64_t n, int lightCount, int skipBits,
                             hipStream_t stream);
void launchCastRays(const SceneView& scene, int64_t n, const double* rays, double selfHit, double* outF64,
                    int32_t* outI32, hipStream_t stream);
void launchEmit(const SceneView& scene, int32_t light, int64_t n, const double* draws5, uint64_t totalPhotons,
                double* rays, float* colors, hipStream_t stream);
void launchIntersect(int kind, int64_t n, const double* in, void* out, hipStream_t stream);
void launchBsdfSample(int64_t n, const double* in14, double* out10, hipStream_t stream);
void launchBsdfEvaluate(int64_t n, const double* in14, double* out4, hipStream_t stream);
int traceBlocksPerCu(const TraceArgs& args);
// values[i] = i
void launchIota(uint32_t* values, uint64_t n, hipStream_t stream);
// stable ascending sort of 32-bit (key, value) pairs over the low `bits` bits of the key
hipError_t sortPairs32(void* scratch, size_t* scratchBytes, const uint32_t* keysIn, uint32_t* keysOut, const uint32_t* valuesIn,
                       uint32_t* valuesOut, uint64_t n, int bits, hipStream_t stream);
#ifdef RT_DIAG
void printDiagnostics();
#endif
// Probe pass, two launches per camera: count survivors per pixel, then (with the
// exclusive scan of the counts in `offsets`) write the records in pixel-major order.
void launchGatherCount(const SceneView& scene, int32_t camera, int32_t stride, int64_t pixels, uint32_t* counts,
                       unsigned long long* stats, hipStream_t stream);
void launchGatherWrite(const SceneView& scene, int32_t camera, int32_t stride, int64_t pixels, const uint32_t* counts,
                       const uint64_t* offsets, GatherPoint* out, hipStream_t stream);
// Probe pass of a camera whose pixels yield one record at most, in one launch of casts: every pixel's
// record into slots[pixel], then (with the scan of the counts) the records moved into pixel-major order.
// (deferred: `pixels` words, deferredCount: one, zero before the call — the pixels whose camera ray meets a mirror
// are finished by a second launch)
void launchGatherCollect(const SceneView& scene, int32_t camera, int32_t stride, int64_t pixels, uint32_t* counts,
                         unsigned long long* stats, GatherPoint* slots, uint32_t* deferred, unsigned long long* deferredCount,
                         hipStream_t stream);
void launchGatherCompact(const GatherPoint* slots, const uint32_t* counts, const uint64_t* offsets, int64_t pixels, GatherPoint* out,
                         hipStream_t stream);
// The gather (ProbeGather::run) over the bounce store of the last pass.
enum : int32_t
{
    kDirectoryNone = 0,
    kDirectoryCells = 1,
    kDirectoryRows = 2
};

// How a record's cell becomes its key: the cell relative to the least cell of the store's box, x in the
// lowest bitsX bits, y in the next bitsY, z above (as many bits as the box needs: the radix sort runs a
// pass per 8 of them).  No valid key has all bits of its z part set, so kEmptyKey's low bits sort last.
struct KeyLayout
{
    int32_t low[3], high[3];  // least and greatest cell of the box
    int32_t bitsX, bitsY, bits;  // bits = all three parts together
};

struct GatherArgs
{
    SceneView scene;
    const float* store;                 // [records][13], IN KEY ORDER (launchReorderStore): row j belongs to keys[j]
    const unsigned long long* keys;     // key of each record (see KeyLayout), ascending
    KeyLayout layout;
    uint64_t indexed;                   // records that carry a valid key
    double invCellSize;
    double minGatherRadius;
    float timeHalfWindow;
    int32_t camera;
    int64_t pixels;                     // sampled pixels of the camera's probe pass
    const uint32_t* counts;             // gather points per sampled pixel
    const uint64_t* offsets;            // first gather point of each sampled pixel; null: pixel i's one record lies at points[i]
    const GatherPoint* points;
    unsigned long long* stats;          // [0] contributions added, [1] deposits summed
    // Directory of the key list over the box of cells that hold records.  kDirectoryCells:
    // cellStart[c] = first record of cell c or, when the cell is empty, of the next cell that is not;
    // c = ((z - cellMin[2]) * cellDim[1] + (y - cellMin[1])) * cellDim[0] + (x - cellMin[0]), which
    // ascends with the keys; cellStart[number of cells] = indexed.  kDirectoryRows (the box has too many
    // cells): the same over whole rows of cells along x (cellDim[0] = 1, cellMin[0] = INT32_MIN); a
    // row's records are then searched for the cells wanted.  kDirectoryNone: the whole list is searched.
    int32_t directoryKind;
    const uint32_t* cellStart;
    int32_t cellMin[3], cellDim[3];
    // One lane per pixel, with the directory: a pixel one of whose gather points has more than
    // heavyThreshold records in its rows of cells is left to a second launch, heavyWaves waves per pixel, and
    // appended to heavyPixels (heavyCount: how many so far).  0: every pixel is done where it is.
    uint32_t heavyThreshold;
    uint32_t heavyWaves;  // waves that work together on a pixel of the second launch (1, 2, 4 or 8)
    uint32_t* heavyPixels;
    unsigned long long* heavyCount;
    // Second launch: the pixels to work on (null: all of them, thread by thread)
    const uint32_t* pixelList;
    const unsigned long long* pixelListCount;
};
// cellBox: [0..2] least and [3..5] greatest x, y, z of the records' cells that pack, to be preset to
// INT32_MAX / INT32_MIN
void launchStoreBox(const float* store, uint64_t n, double invCellSize, int32_t* cellBox, hipStream_t stream);
// The layout of the keys for a box (empty box: any layout will do, there is no key to make)
KeyLayout keyLayoutFor(const int32_t box[6]);
void launchStoreKeys(const float* store, uint64_t n, double invCellSize, const KeyLayout& layout, unsigned long long* keys,
                     uint32_t* order, hipStream_t stream);
// The directory GatherArgs::cellStart describes, from the sorted keys: `cells` + 1 entries.
hipError_t buildCellDirectory(void* scratch, size_t* scratchBytes, const unsigned long long* sortedKeys, uint64_t indexed,
                              const KeyLayout& layout, const int32_t cellMin[3], const int32_t cellDim[3], uint32_t* cellStart,
                              hipStream_t stream);
// Stable ascending sort of (keys, order) over the lowest `bits` bits of the keys.
hipError_t sortStoreKeys(void* scratch, size_t* scratchBytes, const unsigned long long* keysIn,
                         unsigned long long* keysOut, const uint32_t* orderIn, uint32_t* orderOut, uint64_t n, int bits,
                         hipStream_t stream);
// out[0] = number of leading keys below the "no cell" key (records of unpackable cells sort to the end)
void launchCountIndexed(const unsigned long long* sortedKeys, uint64_t n, unsigned long long* out, hipStream_t stream);
// sorted[j] = store[order[j]] for the first `indexed` positions: the gather then reads a cell's
// records from consecutive rows instead of chasing the sort's permutation record by record.
void launchReorderStore(const float* store, const uint32_t* order, uint64_t indexed, float* sorted, hipStream_t stream);
// lanesPerPixel: 1, 8 or 64 lanes work on one sampled pixel's records (same additions in the same order)
// (with one lane and GatherArgs::heavyThreshold set, a second launch on a grid sized for computeUnits
// takes the pixels the first one left over)
void launchGather(const GatherArgs& args, int lanesPerPixel, int computeUnits, hipStream_t stream);
// Device-side build of the probe grid (rt_set_probes): see the kernels in rt_kernels.hip.
struct ProbeBuild
{
    const double* points;          // [n][3] as given by the caller
    uint64_t n;
    double invCellSize;
    unsigned long long* keys;      // [n] packed cell of each probe (kEmptyKey: does not pack)
    uint32_t* order;               // [n] probe index
};
void launchProbeKeys(const ProbeBuild& b, uint32_t* unpackable, hipStream_t stream);
// offsets[i] = counts[0] + ... + counts[i - 1]
hipError_t scanCounts(void* scratch, size_t* scratchBytes, const uint32_t* counts, uint64_t* offsets, int64_t n,
                      hipStream_t stream);
// ProbeGather::depositEmitters for one emitter patch: a countU x countV grid of cell-centred
// samples over the patch (inside the disc for a disc).
struct EmitterGrid
{
    double center[3], right[3], up[3];
    float normal[3], power[3];
    double extentU, extentV, stepU, stepV, radius;
    int32_t countU, countV, isDisc;
};
// number of samples of the grid that lie on the patch (adds to *count)
void launchEmitterCount(const EmitterGrid& grid, unsigned long long* count, hipStream_t stream);
// appends the samples that pass the keep test to the store (adds their number to *kept)
void launchEmitterDeposit(const EmitterGrid& grid, const ProbeView& probes, const StoreView& store,
                          unsigned long long* kept, hipStream_t stream);
void launchProbePositions(const GatherPoint* points, uint64_t n, double* positions, hipStream_t stream);
void launchPixelProbePositions(const GatherPoint* slots, const uint32_t* counts, const uint64_t* offsets, int64_t pixels,
                               double* positions, hipStream_t stream);
// stable ascending sort of (keys, order) pairs of any length below 2^31
hipError_t sortKeyPairs(void* scratch, size_t* scratchBytes, const unsigned long long* keysIn, unsigned long long* keysOut,
                        const uint32_t* orderIn, uint32_t* orderOut, uint64_t n, hipStream_t stream);
// runs[0] += the number of runs of equal (key >> shift) in the sorted keys
void launchCountRuns(const unsigned long long* sortedKeys, uint64_t n, int shift, unsigned long long* runs, hipStream_t stream);
// heads[i] = 1 where sortedKeys[i] starts a run of equal (key >> shift), else 0
void launchRunHeads(const unsigned long long* sortedKeys, uint64_t n, int shift, uint32_t* heads, hipStream_t stream);
// the probes in cell order and the dense array of cells (ordinal[i]: the number of cells before probe i's)
void launchDenseCells(const unsigned long long* sortedKeys, const uint32_t* sortedOrder, const uint64_t* ordinal,
                      const double* points, uint64_t n, DenseCell* cells, double* groupedPoints, hipStream_t stream);
// The bricks (see ProbeView) are gathered from the groups of 2 x 2 x 2 cells: a record per group (fewer than
// 2^29 of them); eight sorted lists of (brick, group) — list s: every group's key less the packed offset s,
// beside the group's number with s above it — that the caller merges pair by pair: the lists themselves
// with mergeBrickLists (they are never written out), their merges with mergeKeyPairs; and a table entry per
// run of equal bricks in the merged list.
void launchCellGroups(const unsigned long long* sortedKeys, const uint64_t* cellOrdinal, const uint64_t* groupOrdinal, uint64_t n,
                      unsigned long long* groupKey, uint32_t* groupOwn, uint32_t* groupFirst, hipStream_t stream);
hipError_t mergeBrickLists(void* scratch, size_t* scratchBytes, const unsigned long long* groupKey, uint64_t groups, int listA,
                           int listB, unsigned long long* keysOut, uint32_t* entriesOut, hipStream_t stream);
hipError_t mergeKeyPairs(void* scratch, size_t* scratchBytes, const unsigned long long* keysA, const uint32_t* valuesA, uint64_t nA,
                         const unsigned long long* keysB, const uint32_t* valuesB, uint64_t nB, unsigned long long* keysOut,
                         uint32_t* valuesOut, hipStream_t stream);
// fills the brick table (every byte 0xFF before) from the merged list
void launchBricks(const unsigned long long* mergedKeys, const uint32_t* mergedEntries, uint64_t total, const uint32_t* groupOwn,
                  const uint32_t* groupFirst, ProbeBrick* bricks, uint64_t brickMask, hipStream_t stream);
void launchProbeTest(const ProbeView& probes, int64_t n, const double* points, uint8_t* out, hipStream_t stream);

// ---- which records an overflowing store keeps (rt_store.hip) ----
// A pass left `candidates` records with keys[0 .. candidates) (all different), of which the `need` with
// the smallest keys are wanted.  StoreSelectState is the device-side state of that selection.
constexpr int kSelectDigitBits = 10;
struct StoreSelectState
{
    unsigned long long prefix;     // the leading bits of the need-th smallest key found so far
    unsigned long long remaining;  // how many keys with that prefix, counted from the smallest, are still wanted
    unsigned long long holes, fillers;  // records to replace inside the capacity / to move in from behind it
    unsigned int histogram[1 << kSelectDigitBits];
};
// Finds the need-th smallest of keys[0 .. candidates) (1 <= need <= candidates), digit by digit from bit
// `bits` - 1 down: state->prefix holds it afterwards.  Resets the state first.
void launchStoreSelectThreshold(const unsigned long long* keys, uint64_t candidates, uint64_t need, int bits,
                                StoreSelectState* state, hipStream_t stream);
// Records are at slots first .. first + candidates; those at slots below `capacity` whose key exceeds the
// threshold are listed in holes, those at slots from `capacity` on whose key does not in fillers (as many).
void launchStoreSelectClassify(const unsigned long long* keys, uint64_t first, uint64_t candidates, uint64_t capacity,
                               StoreSelectState* state, uint32_t* holes, uint32_t* fillers, hipStream_t stream);
// records[holes[i]] = records[fillers[i]] for every pair (keys likewise); `pairs` is an upper bound of their number
void launchStoreSelectMove(float* records, unsigned long long* keys, const StoreSelectState* state, const uint32_t* holes,
                           const uint32_t* fillers, uint64_t pairs, hipStream_t stream);

}  // namespace rtd
