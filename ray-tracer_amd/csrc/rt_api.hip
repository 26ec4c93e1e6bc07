// rt_api.hip — the C-ABI of include/rt_amd.h: scene flattening and upload, sink
// set-up, launches and read-back.  Host code only; every kernel lives in
// rt_kernels.hip.

#include "rt_amd.h"

#include "rt_bvh.h"
#include "rt_kernels.h"
#include "rt_order.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstddef>
#include <cstring>
#include <dlfcn.h>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <unordered_map>
#include <utility>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace
{

thread_local std::string g_error;

struct ApiError : std::runtime_error
{
    int code;
    ApiError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

void check(hipError_t e, const char* what)
{
    if (e != hipSuccess)
    {
        throw ApiError(RT_ERR_DEVICE, std::string(what) + ": " + hipGetErrorString(e));
    }
}

void require(bool ok, const char* what)
{
    if (!ok)
    {
        throw ApiError(RT_ERR_ARGUMENT, what);
    }
}

// Device allocations and releases this library has made since it was loaded (every one goes through
// DeviceArray): what rt_debug_device_allocations reports.
std::atomic<uint64_t> g_deviceAllocations{0}, g_deviceReleases{0};

template <typename T>
struct DeviceArray
{
    T* ptr = nullptr;
    size_t count = 0;
    DeviceArray() = default;
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;
    ~DeviceArray() { release(); }
    void release()
    {
        if (ptr)
        {
            (void)hipFree(ptr);
            g_deviceReleases.fetch_add(1, std::memory_order_relaxed);
            ptr = nullptr;
            count = 0;
        }
    }
    void allocate(size_t n)
    {
        release();
        if (n > 0)
        {
            check(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)), "hipMalloc");
            g_deviceAllocations.fetch_add(1, std::memory_order_relaxed);
            count = n;
        }
    }
    // room for n elements at least: an allocation that is large enough already is kept (set-up that runs
    // once per frame would otherwise hand gigabytes back to the driver and ask for them again)
    void reserve(size_t n)
    {
        if (ptr == nullptr || count < n)
        {
            allocate(n);
        }
    }
    void upload(const std::vector<T>& host)
    {
        allocate(host.size());
        if (!host.empty())
        {
            check(hipMemcpy(ptr, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D");
        }
    }
    void zero(hipStream_t stream)
    {
        if (ptr)
        {
            check(hipMemsetAsync(ptr, 0, count * sizeof(T), stream), "hipMemsetAsync");
        }
    }
};

rtd::Quat quatInverse(const rtd::Quat& q)  // reference src/Quaternion.cpp:39-52
{
    const double m = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    const double s = 1.0 / m;
    return rtd::Quat{-q.x * s, -q.y * s, -q.z * s, q.w * s};
}

// Host copy of the device quaternion product (reference src/Quaternion.cpp:88-109);
// used once per light / camera to resolve forward / right / up.
void hostRotate(const double* q, const double* v, double* out)
{
    const double px = v[0], py = v[1], pz = v[2], pw = 0.0;
    const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    const double ax = qw * px + qx * pw + qy * pz - qz * py;
    const double ay = qw * py + qy * pw + qz * px - qx * pz;
    const double az = qw * pz + qz * pw + qx * py - qy * px;
    const double aw = qw * pw - qx * px - qy * py - qz * pz;
    const double cx = -qx, cy = -qy, cz = -qz, cw = qw;
    out[0] = aw * cx + ax * cw + ay * cz - az * cy;
    out[1] = aw * cy + ay * cw + az * cx - ax * cz;
    out[2] = aw * cz + az * cw + ax * cy - ay * cx;
}

double radians(double degrees) { return M_PI * (degrees / 180.0); }  // reference src/Utility.cpp:35-38

}  // namespace

// ---- scratch of rt_trace_static: a pass's emission-order list and the radix sort's work space in
// memory that belongs to the library's code object, so that the call allocates nothing ----
constexpr uint64_t kStaticMaxIds = 1ull << 20;             // photon ids of one rt_trace_static call at most
constexpr size_t kStaticSortBytes = 16ull << 20;           // work space of the radix sort over that many pairs
constexpr size_t kStaticHeaderBytes = 256;                 // [0] the pass's work counter
constexpr size_t kStaticListBytes = kStaticMaxIds * sizeof(uint32_t);
constexpr size_t kStaticScratchBytes = kStaticHeaderBytes + 4 * kStaticListBytes + kStaticSortBytes;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace
{

// Every GPU holds its own copy of `scratch` at its own address; in host code the array's name is
// only a handle for hipGetSymbolAddress.  The address is looked up once per device (by rt_create,
// with that device current) and everything after that uses the pointer.
constexpr int kMaxDevices = 64;
unsigned char* g_staticScratch[kMaxDevices] = {};
std::mutex g_staticScratchMutex;

unsigned char* staticScratchOf(int device)
{
    require(device >= 0 && device < kMaxDevices, "device index beyond the static scratch table");
    std::lock_guard<std::mutex> lock(g_staticScratchMutex);
    if (g_staticScratch[device] == nullptr)
    {
        void* address = nullptr;
        check(hipGetSymbolAddress(&address, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
        require(address != nullptr, "static scratch has no device address");
        g_staticScratch[device] = static_cast<unsigned char*>(address);
    }
    return g_staticScratch[device];
}

}  // namespace

// bounce-level photon queues (see rt_trace)
constexpr int kQueueLevels = 12;
constexpr uint64_t kQueueCapacity = 48ull << 20;  // photons parked per level at most (4.5 GB per queue)
constexpr uint64_t kQueueMinimumWork = 4ull << 20;  // smaller passes are traced in one launch
constexpr uint64_t kOwnLevelSortMinimum = 1ull << 20;  // parked photons from which the sort's passes are our own
constexpr uint64_t kThreeStepMinimum = 2ull << 20;     // smaller passes of one light skip the second of the three steps
constexpr uint64_t kRefineOrderMinimum = 1ull << 20;  // smaller passes are radix-sorted over all the bits of their keys (fewer launches)

struct rt_tracer
{
    int device = 0;
    hipStream_t stream = nullptr;
    rt_settings settings{};
    std::vector<rt_camera> cameraDesc;
    int64_t lightCount = 0;

    DeviceArray<rtd::Volume> volumes;
    DeviceArray<unsigned char> cacheImage;
    DeviceArray<rtd::Node> nodes;
    DeviceArray<double> triRec, triNrm;
    DeviceArray<int32_t> triId, triOwner;
    DeviceArray<rtd::Material> materials;
    bool hasDielectric = false;
    DeviceArray<rtd::Light> lights;
    DeviceArray<rtd::Camera> cameras;
    DeviceArray<rt_animation> animations;
    DeviceArray<rt_keyframe> keyframes;
    std::vector<rtd::Camera> cameraHost;
    std::vector<std::unique_ptr<DeviceArray<float>>> cameraBuffers;
    std::vector<float*> boundBuffers;

    // sink A
    struct ProbeBuildSpace  // work space of the probe index build
    {
        DeviceArray<double> points;
        DeviceArray<unsigned long long> keysIn, keysOut, scalars, groupKey, listKeys[2];
        DeviceArray<uint64_t> cellOrdinal, groupOrdinal;
        DeviceArray<uint32_t> orderIn, orderOut, flag, heads, groupOwn, groupFirst, listEntries[2];
        DeviceArray<unsigned char> scratch, mergeScratch;
    } probeBuildSpace;
    bool probesSet = false;
    uint64_t probeVersion = 0;  // bumped whenever the probes are set anew
    // The emitters' deposits of a frame depend on the probes, the patches and the spacing only: the
    // records of one run are kept and copied into an empty store the next time.
    struct EmitterDeposits
    {
        DeviceArray<float> records;
        uint64_t kept = 0, samples = 0, probeVersion = 0;
        double spacing = -1.0;
    } emitterDeposits;
    DeviceArray<int64_t> slotKey;
    DeviceArray<uint32_t> slotStart, slotCount;
    DeviceArray<double> probePoints;
    DeviceArray<rtd::ProbeBrick> bricks;
    DeviceArray<rtd::DenseCell> denseCells;
    rtd::ProbeView probeView{};
    DeviceArray<float> storeRecords;
    uint64_t storeCapacity = 0;    // records the store takes (rt_set_store_limit may hold it below what was configured)
    uint64_t storeConfigured = 0;  // the capacity rt_set_probes* was given
    uint64_t storeAllocated = 0;   // records there is room for while a pass runs: the configured capacity and, where a
                                   // pass can overflow it, room behind it (rtd::StoreView)
    DeviceArray<unsigned long long> storeKeys;         // [1 + storeAllocated] rtd::StoreView::keys from [-1] on, or empty
    DeviceArray<rtd::StoreSelectState> storeSelectState;
    DeviceArray<uint32_t> storeSelectLists;            // holes, then fillers
    // A pass that may have overflowed the store and has not been looked at yet (settleStore): the cursor
    // before it, and how many bits its keys have.
    bool storePending = false;
    uint64_t storePassStart = 0;
    int storePassKeyBits = 1;
    bool storePassOrdered = false;
    bool storeCursorKnown = false;  // storeCursorHost is the device's cursor
    uint64_t storeCursorHost = 0;
    bool storeOrderWarned = false;

    // sink B
    bool splatSet = false;
    double minSplatRadius = 0.0;
    DeviceArray<unsigned long long> densityKeys, densityCount;
    DeviceArray<float> densityPower;
    rtd::DensityView densityView{};

    // shared device scalars: [0] work counter, [1] store cursor
    DeviceArray<unsigned long long> scalars;
    DeviceArray<rtd::DeviceCounters> counters;
    DeviceArray<unsigned char> countersStaged;  // an rt_counters on the device, for rt_allreduce
    // emission-order list of a pass's photon ids
    DeviceArray<uint32_t> sortIds, sortOrder, sortKeysIn, sortKeysOut;
    const uint32_t* orderList = nullptr;          // the finished emission order: sortOrder or (large passes) sortIds
    DeviceArray<uint32_t> orderCounts, orderOffsets;  // large passes: ids per digit and tile, and where each such run starts
    uint64_t orderedFirst = 0, orderedCount = 0;  // the range of photons per light sortOrder holds the order of (0: none)
    uint64_t orderedLength = 0;                   // ids in the emission order of the last rt_trace (0: traced unsorted)
    DeviceArray<rtd::PhotonState> photonQueue[2];
    // keys of the parked photons' rays, their sorted copy, the slot numbers in key order, 0..n-1
    DeviceArray<uint32_t> queueKeys, queueKeysSorted, queueOrder, queueIota, queueSpare;
    float keyOrigin[3] = {0.0f, 0.0f, 0.0f};
    float keyScale = 1.0f;

    DeviceArray<unsigned long long> queueCounts;
    DeviceArray<unsigned char> sortScratch;
    // events around the stages of the last rt_trace: before the sort, after it, after the trace kernel
    hipEvent_t stageEvents[3] = {nullptr, nullptr, nullptr};
    bool timed = false;

    // records of the last trace
    uint64_t recordPhotons = 0;
    DeviceArray<double> recEmitted, recF64;
    DeviceArray<int32_t> recCasts, recI32;
    DeviceArray<float> recPower;

    // probe pass
    DeviceArray<rtd::EmitterPatch> patches;
    std::vector<rtd::EmitterPatch> patchHost;
    struct GatherCache  // one camera's probe pass, kept on the device for the gather
    {
        bool valid = false;
        int64_t pixels = 0;
        DeviceArray<uint32_t> counts;
        DeviceArray<uint64_t> offsets;
        DeviceArray<rtd::GatherPoint> devicePoints;
        DeviceArray<rtd::GatherPoint> slots;    // a record per pixel (one-pass form)
        DeviceArray<uint32_t> deferred;         // one-pass form: the pixels its second launch finishes
        DeviceArray<unsigned long long> deferredCount;
        // one-pass form: the records lie in `slots` at their pixels' places; devicePoints holds them together
        // only once somebody wants them so (gatheredTogether)
        bool byPixel = false, together = false;
        uint64_t total = 0;                     // records on the device
        std::vector<rtd::GatherPoint> points;   // host copy, made when first asked for
        bool hostCopied = false;
        uint64_t stats[3] = {0, 0, 0};
    };
    std::vector<std::unique_ptr<GatherCache>> gatherCache;
    struct GatherSpace  // work space of rt_gather, kept between calls
    {
        DeviceArray<unsigned long long> keysIn, keysOut, stats;
        DeviceArray<uint32_t> orderIn, orderOut;
        DeviceArray<unsigned char> scratch;
        DeviceArray<float> sortedStore;
        DeviceArray<int32_t> cellBox;     // [0..2] least, [3..5] greatest cell x, y, z of the records that pack
        DeviceArray<uint32_t> cellStart;  // directory of the sorted keys over that box (GatherArgs::cellStart)
        DeviceArray<unsigned char> directoryScratch;
        DeviceArray<uint32_t> heavyPixels;  // pixels the one-lane gather leaves to a wave each
        int32_t cellMin[3] = {0, 0, 0}, cellDim[3] = {0, 0, 0};
        int32_t box[6] = {0, 0, 0, -1, -1, -1};  // least and greatest cell of the records that pack
        rtd::KeyLayout layout{};
        int32_t directoryKind = 0;  // rtd::kDirectory...
        // the index these arrays hold: valid while the store has not changed since (storeVersion)
        uint64_t indexedVersion = 0, indexedRecords = 0, indexed = 0;
        double indexedCellSize = 0.0;
    } gatherSpace;
    uint64_t storeVersion = 1;  // bumped by everything that writes the store or its cursor

    // Device time of the frame's camera-side set-up, as pairs of events on the tracer's stream: [0] the probe
    // pass of every camera, [1] the probe index (keys, sort, dense cells), [2] its table of bricks
    struct TimedSpan
    {
        hipEvent_t begin = nullptr, end = nullptr;
    };
    std::vector<TimedSpan> setupSpans[3];

    rtd::SceneView view{};
    int computeUnits = 0;
    unsigned char* staticScratch = nullptr;  // this device's copy of `scratch` (rt_trace_static)
    size_t staticSortBytes = 0;              // what the radix sort asks for at kStaticMaxIds pairs
};

namespace
{

void activate(const rt_tracer* t) { check(hipSetDevice(t->device), "hipSetDevice"); }

void buildScene(rt_tracer* t, const rt_scene* scene)
{
    require(scene != nullptr, "scene is null");
    require(scene->volume_count >= 0 && scene->mesh_count >= 0 && scene->material_count > 0, "bad scene counts");
    t->settings = scene->settings;
    t->lightCount = scene->light_count;

    // ---- meshes: reference tree, flattened; triangles re-ordered by page ----
    std::vector<rtd::Node> nodes;
    std::vector<double> triRec, triNrm;
    std::vector<int32_t> triId;
    // Triangles per leaf: a triangle test costs about two node steps, and with two a leaf a cast of
    // the configurations' meshes tests a third fewer triangles for a step or two more (measured best on
    // MirrorBombardment-200M and MirrorKnot4k; 1, 3, 4 and 6 were tried).
    int leafSize = 2;
    if (const char* leaf = std::getenv("RT_AMD_LEAF"))
    {
        leafSize = std::atoi(leaf);
        require(leafSize >= 1 && leafSize <= 16, "RT_AMD_LEAF must be between 1 and 16");
    }
    // coordinate scale of the scene: a bound on the coordinates a ray origin or a hit can have,
    // in world space and (volume positions included) in any volume's object space
    double sceneScale = 0.0, reach = 0.0;
    for (int64_t i = 0; i < scene->volume_count; ++i)
    {
        for (int k = 0; k < 3; ++k) sceneScale = std::max(sceneScale, std::fabs(scene->volumes[i].position[k]));
        if (scene->volumes[i].type != RT_VOLUME_MESH && scene->volumes[i].type != RT_VOLUME_PLANE)
        {
            for (double p : scene->volumes[i].param) reach = std::max(reach, std::fabs(p));
        }
    }
    for (int64_t i = 0; i < scene->keyframe_count; ++i)
    {
        for (int k = 0; k < 3; ++k) sceneScale = std::max(sceneScale, std::fabs(scene->keyframes[i].value[k]));
    }
    for (int64_t i = 0; i < scene->animation_count; ++i)
    {
        for (int k = 0; k < 3; ++k)
        {
            sceneScale = std::max(sceneScale, std::fabs(scene->animations[i].constant_position[k]));
        }
    }
    for (int64_t i = 0; i < scene->light_count; ++i)
    {
        for (int k = 0; k < 3; ++k) sceneScale = std::max(sceneScale, std::fabs(scene->lights[i].position[k]));
    }
    for (int64_t i = 0; i < scene->camera_count; ++i)
    {
        for (int k = 0; k < 3; ++k) sceneScale = std::max(sceneScale, std::fabs(scene->cameras[i].position[k]));
    }
    for (int64_t i = 0; i < 9 * scene->triangle_count; ++i)
    {
        reach = std::max(reach, std::fabs(scene->tri_pos[i]));
    }
    sceneScale = 2.0 * (sceneScale + reach);
    struct MeshSpan
    {
        int32_t nodeBase, nodeCount;
        int64_t triBase, triCount;
        int32_t ldsNodeCount = 0, ldsNodeOffset = 0, ldsTriOffset = -1;
        double extent;
        double lo[3], hi[3];  // object-space bounds
    };
    std::vector<MeshSpan> spans;
    std::vector<int32_t> triOwner;  // per slot: the volume a grouped triangle belongs to (-1: the volume that is cast)
    int maxDepth = 1;
    const double pad = 4e-6 * sceneScale;
    // One mesh as the device walks it: `count` triangles (pos / nrm: 9 doubles each) with the reference's
    // tree over them (ranks and node boxes); ownerOf / idOf give, per triangle, the volume it reports and
    // its index there (null: the volume that is cast, the triangle's own index).
    auto addSpan = [&](const double* pos, const double* nrm, int64_t count, const rtbvh::FlatMesh& reference,
                       const int32_t* ownerOf, const int32_t* idOf, const char* label, long long number) {
        // the reference's tree gives every triangle its visiting rank and node box; the tree the
        // device walks is built over the same triangles.  Its boxes are tested in single precision
        // against ray origins anywhere in the scene: padding of a few millionths of the scene's
        // scale is an order of magnitude above what that loses (6e-8 of the origin's size in the
        // origin, 2e-7 of a plane's distance in the products), and costs no traversal steps.
        double extent = 0.0;
        for (int64_t i = 0; i < 9 * count; ++i)
        {
            extent = std::max(extent, std::fabs(pos[i]));
        }
        const rtbvh::FlatMesh flat = rtbvh::buildSurfaceArea(pos, count, reference, pad, leafSize);
        const rtbvh::WideMesh wide = rtbvh::breadthFirst(rtbvh::widen(flat));
        maxDepth = std::max(maxDepth, wide.pushDepth);
        if (std::getenv("RT_AMD_VERBOSE"))
        {
            std::fprintf(stderr, "[rt_amd] %s %lld: %lld triangles, %zu inner nodes, %d levels of them\n", label, number,
                         static_cast<long long>(count), wide.nodes.size(), wide.pushDepth);
        }
        MeshSpan span;
        span.nodeBase = static_cast<int32_t>(nodes.size());
        span.nodeCount = static_cast<int32_t>(wide.nodes.size());
        span.triBase = static_cast<int64_t>(triId.size());
        span.triCount = count;
        span.extent = extent;
        for (int k = 0; k < 3; ++k)
        {
            span.lo[k] = span.hi[k] = count > 0 ? pos[k] : 0.0;
        }
        for (int64_t i = 0; i < 3 * count; ++i)
        {
            for (int k = 0; k < 3; ++k)
            {
                span.lo[k] = std::min(span.lo[k], pos[3 * i + k]);
                span.hi[k] = std::max(span.hi[k], pos[3 * i + k]);
            }
        }
        for (const rtbvh::WideNode& wn : wide.nodes)
        {
            rtd::Node n;
            static_assert(sizeof(n) == sizeof(wn), "host and device node layouts differ");
            std::memcpy(&n, &wn, sizeof(n));
            nodes.push_back(n);
        }
        for (size_t slot = 0; slot < flat.order.size(); ++slot)
        {
            const int32_t original = flat.order[slot];
            const double* p = pos + 9 * original;
            // One 128-byte record per triangle: vertices, one word, the box of its reference node.  The
            // word holds its reference rank (low half) and, as a float, the smallest barycentric
            // weight from which a hit is certain to pass the reference's slab test on that box
            // without running it (high half).  A hit point with weights >= w lies at least
            // w x (the triangle's extent along k) inside the box along every axis k the triangle
            // extends in, and that has to exceed the few-ulp relative rounding of the slab
            // parameters, i.e. ~1e-15 x the scene's coordinate scale; the constant below leaves four
            // orders of magnitude.
            float certain;
            {
                double narrowest = std::numeric_limits<double>::infinity();
                for (int k = 0; k < 3; ++k)
                {
                    const double extent = std::max({p[k], p[3 + k], p[6 + k]}) - std::min({p[k], p[3 + k], p[6 + k]});
                    if (extent > 0.0) narrowest = std::min(narrowest, extent);
                }
                const double weight = 1e-10 * sceneScale / narrowest;  // 0 for a triangle without extent: it is never hit
                certain = static_cast<float>(std::min(weight, 2.0));
                certain = std::nextafter(certain, 3.0f);  // the conversion may have rounded down
            }
            uint32_t certainBits;
            std::memcpy(&certainBits, &certain, sizeof(certainBits));
            const uint64_t word = (static_cast<uint64_t>(certainBits) << 32) | static_cast<uint32_t>(flat.rank[slot]);
            double wordAsDouble;
            std::memcpy(&wordAsDouble, &word, sizeof(word));
            triRec.insert(triRec.end(), p, p + 9);
            triRec.push_back(wordAsDouble);
            triRec.insert(triRec.end(), flat.referenceBox.begin() + 6 * slot, flat.referenceBox.begin() + 6 * slot + 6);
            triNrm.insert(triNrm.end(), nrm + 9 * original, nrm + 9 * original + 9);
            triId.push_back(idOf ? idOf[original] : original);
            triOwner.push_back(ownerOf ? ownerOf[original] : -1);
        }
        spans.push_back(span);
    };
    for (int64_t m = 0; m < scene->mesh_count; ++m)
    {
        const rt_mesh& mesh = scene->meshes[m];
        require(mesh.first_triangle >= 0 && mesh.triangle_count >= 0 &&
                    mesh.first_triangle + mesh.triangle_count <= scene->triangle_count,
                "mesh triangle range out of bounds");
        require(mesh.triangle_count < (1ll << 26), "mesh too large");
        const double* pos = scene->tri_pos + 9 * mesh.first_triangle;
        const double* nrm = scene->tri_nrm + 9 * mesh.first_triangle;
        addSpan(pos, nrm, mesh.triangle_count, rtbvh::build(pos, mesh.triangle_count), nullptr, nullptr, "mesh", static_cast<long long>(m));
    }

    // ---- flat meshes of one pose, cast together ----
    // A room's walls are five meshes of two triangles in one pose.  Cast one by one (as the reference's loop
    // over the volumes does, src/Worker.cpp:539-567), each costs every wave a box test, a walk and two triangle
    // tests for the few lanes whose ray goes its way.  Static mesh volumes that share a pose and whose
    // triangles all lie in one plane are therefore ALSO put into one tree, which is cast once in their stead:
    // a ray meets a plane once, so such a volume's closest hit is its only hit, and "per volume the closest hit,
    // dropped when it is not beyond the self-hit floor, then the nearest of the volumes, the first of them on
    // a tie" is "the nearest hit beyond the floor over all their triangles, on a tie the first volume's, then
    // the first in that volume's visiting order" — which is how the joint tree ranks its triangles.  (Two
    // triangles of one volume that overlap at the hit point give two hits of the same point; should their
    // distances straddle the floor by a rounding error the reference drops the volume and the joint cast keeps
    // the farther one.  The floor is 1e-4 and a distance would have to equal it to the last bit.)
    struct Group
    {
        std::vector<int64_t> members;  // volume indices, ascending
        size_t span = 0;
    };
    std::vector<Group> groups;
    std::vector<int32_t> groupOf(static_cast<size_t>(std::max<int64_t>(0, scene->volume_count)), -1);
    {
        const char* wanted = std::getenv("RT_AMD_GROUP_FLAT_MESHES");
        const bool enabled = wanted == nullptr || std::atoi(wanted) != 0;
        auto flat = [&](const rt_mesh& mesh) {
            // every vertex in the plane of the first triangle that has an area, exactly (axis-parallel walls are;
            // anything else may fail by a rounding error and is then simply cast on its own)
            const double* pos = scene->tri_pos + 9 * mesh.first_triangle;
            for (int64_t t = 0; t < mesh.triangle_count; ++t)
            {
                const double* p = pos + 9 * t;
                const double e1[3] = {p[3] - p[0], p[4] - p[1], p[5] - p[2]}, e2[3] = {p[6] - p[0], p[7] - p[1], p[8] - p[2]};
                const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                if (n[0] == 0.0 && n[1] == 0.0 && n[2] == 0.0) continue;
                for (int64_t i = 0; i < 3 * mesh.triangle_count; ++i)
                {
                    const double* v = pos + 3 * i;
                    if ((v[0] - p[0]) * n[0] + (v[1] - p[1]) * n[1] + (v[2] - p[2]) * n[2] != 0.0) return false;
                }
                return true;
            }
            return false;  // no triangle with an area
        };
        for (int64_t i = 0; enabled && i < scene->volume_count; ++i)
        {
            const rt_volume& rv = scene->volumes[i];
            if (rv.type != RT_VOLUME_MESH || rv.animation != 0 || rv.mesh < 0 || rv.mesh >= scene->mesh_count) continue;
            const rt_mesh& mesh = scene->meshes[rv.mesh];
            if (mesh.triangle_count <= 0 || mesh.triangle_count > 4096 || !flat(mesh)) continue;
            int32_t found = -1;
            for (size_t g = 0; g < groups.size() && found < 0; ++g)
            {
                const rt_volume& first = scene->volumes[groups[g].members.front()];
                if (std::memcmp(first.position, rv.position, sizeof(rv.position)) == 0 &&
                    std::memcmp(first.rotation, rv.rotation, sizeof(rv.rotation)) == 0)
                {
                    found = static_cast<int32_t>(g);
                }
            }
            if (found < 0)
            {
                groups.push_back(Group{});
                found = static_cast<int32_t>(groups.size() - 1);
            }
            groups[static_cast<size_t>(found)].members.push_back(i);
            groupOf[static_cast<size_t>(i)] = found;
        }
        for (size_t g = 0; g < groups.size(); ++g)
        {
            Group& group = groups[g];
            if (group.members.size() < 2)
            {
                for (int64_t member : group.members) groupOf[static_cast<size_t>(member)] = -1;
                group.members.clear();
                continue;
            }
            // the members' triangles one after the other, and the reference's trees likewise: a triangle's
            // rank is its member's place in the group, then its rank in that member's own tree
            std::vector<double> pos, nrm;
            std::vector<int32_t> ownerOf, idOf;
            rtbvh::FlatMesh reference;
            for (int64_t member : group.members)
            {
                const rt_mesh& mesh = scene->meshes[scene->volumes[member].mesh];
                const double* mpos = scene->tri_pos + 9 * mesh.first_triangle;
                const double* mnrm = scene->tri_nrm + 9 * mesh.first_triangle;
                const int32_t base = static_cast<int32_t>(idOf.size());
                const rtbvh::FlatMesh own = rtbvh::build(mpos, mesh.triangle_count);
                for (rtbvh::FlatNode node : own.nodes)
                {
                    node.triStart += base;
                    reference.nodes.push_back(node);
                }
                for (int32_t original : own.order) reference.order.push_back(base + original);
                pos.insert(pos.end(), mpos, mpos + 9 * mesh.triangle_count);
                nrm.insert(nrm.end(), mnrm, mnrm + 9 * mesh.triangle_count);
                for (int64_t t = 0; t < mesh.triangle_count; ++t)
                {
                    ownerOf.push_back(static_cast<int32_t>(member));
                    idOf.push_back(static_cast<int32_t>(t));
                }
            }
            group.span = spans.size();
            addSpan(pos.data(), nrm.data(), static_cast<int64_t>(idOf.size()), reference, ownerOf.data(), idOf.data(), "group of flat meshes",
                    static_cast<long long>(g));
        }
    }

    // ---- the scene cache of the trace kernel: what goes into LDS behind the traversal stacks ----
    // The top of every mesh's tree (a prefix of its breadth-first node order; all of it when it
    // fits) and then, smallest mesh first, the triangles of the meshes that still fit, as 80-byte
    // records (nine vertex coordinates and the rank / admission word; the reference box, read only
    // for hits near an edge, stays in global memory).  RT_AMD_LDS_CACHE=0 leaves the cache empty.
    require(maxDepth <= 48, "mesh tree too deep for the traversal stack");
    std::vector<unsigned char> cacheImage;
    {
        size_t budget = rtd::traceCacheBudget(std::max(1, maxDepth));
        if (const char* v = std::getenv("RT_AMD_LDS_CACHE"))
        {
            if (std::atoi(v) == 0) budget = 0;
        }
        size_t wanted = 0, smallNodes = 0, largeNodes = 0;
        for (const MeshSpan& span : spans)
        {
            wanted += 64 * static_cast<size_t>(span.nodeCount);
            (span.nodeCount <= 32 ? smallNodes : largeNodes) += static_cast<size_t>(span.nodeCount);
        }
        const size_t nodeBudget = budget / 64;
        const size_t forLarge = nodeBudget > smallNodes ? nodeBudget - smallNodes : 0;
        for (MeshSpan& span : spans)
        {
            size_t share = static_cast<size_t>(span.nodeCount);
            if (wanted > budget)
            {
                if (span.nodeCount <= 32)
                {
                    share = nodeBudget >= smallNodes ? share : 0;
                }
                else
                {
                    share = std::min(share, forLarge * static_cast<size_t>(span.nodeCount) / std::max<size_t>(1, largeNodes));
                }
            }
            span.ldsNodeCount = static_cast<int32_t>(share);
            span.ldsNodeOffset = static_cast<int32_t>(cacheImage.size());
            const unsigned char* first = reinterpret_cast<const unsigned char*>(nodes.data() + span.nodeBase);
            cacheImage.insert(cacheImage.end(), first, first + 64 * share);
        }
        std::vector<size_t> bySize(spans.size());
        for (size_t i = 0; i < spans.size(); ++i) bySize[i] = i;
        std::sort(bySize.begin(), bySize.end(), [&](size_t a, size_t b) { return spans[a].triCount < spans[b].triCount; });
        for (size_t i : bySize)
        {
            MeshSpan& span = spans[i];
            const size_t bytes = 80 * static_cast<size_t>(span.triCount);
            if (span.triCount == 0 || cacheImage.size() + bytes > budget) continue;
            span.ldsTriOffset = static_cast<int32_t>(cacheImage.size());
            for (int64_t slot = 0; slot < span.triCount; ++slot)
            {
                const unsigned char* record = reinterpret_cast<const unsigned char*>(triRec.data() + 16 * (span.triBase + slot));
                cacheImage.insert(cacheImage.end(), record, record + 80);
            }
        }
        cacheImage.resize((cacheImage.size() + 15) & ~static_cast<size_t>(15), 0);
        if (std::getenv("RT_AMD_VERBOSE"))
        {
            std::fprintf(stderr, "[rt_amd] scene cache: %zu of %zu bytes of LDS\n", cacheImage.size(), budget);
            for (size_t i = 0; i < spans.size(); ++i)
            {
                std::fprintf(stderr, "[rt_amd]   mesh %zu: %d of %d nodes, triangles %s\n", i, spans[i].ldsNodeCount,
                             spans[i].nodeCount, spans[i].ldsTriOffset >= 0 ? "cached" : "in global memory");
            }
        }
    }

    // ---- volumes ----
    std::vector<rtd::Volume> volumes;
    for (int64_t i = 0; i < scene->volume_count; ++i)
    {
        const rt_volume& rv = scene->volumes[i];
        require(rv.material >= 0 && rv.material < scene->material_count, "volume material index out of range");
        rtd::Volume v{};
        v.type = rv.type;
        v.material = rv.material;
        v.ldsTriOffset = -1;
        std::memcpy(v.pos, rv.position, sizeof(v.pos));
        v.rot = rtd::Quat{rv.rotation[0], rv.rotation[1], rv.rotation[2], rv.rotation[3]};
        v.inv = quatInverse(v.rot);
        std::memcpy(v.param, rv.param, sizeof(v.param));
        require(rv.animation >= 0 && rv.animation <= scene->animation_count, "volume animation index out of range");
        v.animation = rv.animation - 1;  // the description counts from 1, 0 meaning static
        // the first member of a group of flat meshes is cast for all of them, the others are passed over
        const int32_t groupIndex = groupOf[static_cast<size_t>(i)];
        const bool leads = groupIndex >= 0 && groups[static_cast<size_t>(groupIndex)].members.front() == i;
        const MeshSpan* castSpan = nullptr;
        if (rv.type == RT_VOLUME_MESH)
        {
            require(rv.mesh >= 0 && rv.mesh < scene->mesh_count, "volume mesh index out of range");
            castSpan = &spans[groupIndex >= 0 ? groups[static_cast<size_t>(groupIndex)].span : static_cast<size_t>(rv.mesh)];
            v.nodeBase = castSpan->nodeBase;
            v.nodeCount = castSpan->nodeCount;
            v.triBase = castSpan->triBase;
            v.extent = castSpan->extent;
            v.ldsNodeCount = castSpan->ldsNodeCount;
            v.ldsNodeOffset = castSpan->ldsNodeOffset;
            v.ldsTriOffset = castSpan->ldsTriOffset;
            if (groupIndex >= 0) v.type = leads ? rtd::kVolumeMeshGroup : rtd::kVolumePassedOver;
        }
        else
        {
            require(rv.type == RT_VOLUME_SPHERE || rv.type == RT_VOLUME_QUAD || rv.type == RT_VOLUME_PLANE,
                    "unknown volume type");
        }
        // pose number: the first static volume with the same position and rotation bits
        v.poseId = static_cast<int32_t>(i);
        if (v.animation < 0)
        {
            for (int64_t j = 0; j < i; ++j)
            {
                if (volumes[static_cast<size_t>(j)].animation < 0 &&
                    std::memcmp(scene->volumes[j].position, rv.position, sizeof(rv.position)) == 0 &&
                    std::memcmp(scene->volumes[j].rotation, rv.rotation, sizeof(rv.rotation)) == 0)
                {
                    v.poseId = volumes[static_cast<size_t>(j)].poseId;
                    break;
                }
            }
        }
        // world-space box of a static volume: its object-space corners through the volume's
        // transform, grown by a few millionths of the scene's scale (the padding of the trees' boxes:
        // the test is the same single-precision one) and rounded outward to single precision
        std::vector<std::array<double, 3>> corners;
        if (v.animation < 0)
        {
            if (rv.type == RT_VOLUME_MESH && scene->meshes[rv.mesh].triangle_count > 0)
            {
                const MeshSpan& span = *castSpan;
                for (int c = 0; c < 8; ++c)
                {
                    corners.push_back({(c & 1) ? span.hi[0] : span.lo[0], (c & 2) ? span.hi[1] : span.lo[1],
                                       (c & 4) ? span.hi[2] : span.lo[2]});
                }
            }
            else if (rv.type == RT_VOLUME_SPHERE)
            {
                corners.push_back({rv.param[0], rv.param[1], rv.param[2]});  // the centre; the radius is added below
            }
            else if (rv.type == RT_VOLUME_QUAD)
            {
                for (int c = 0; c < 4; ++c)
                {
                    std::array<double, 3> corner{rv.param[0], rv.param[1], rv.param[2]};
                    for (int k = 0; k < 3; ++k)
                    {
                        corner[k] += ((c & 1) ? rv.param[3 + k] : 0.0) + ((c & 2) ? rv.param[6 + k] : 0.0);
                    }
                    corners.push_back(corner);
                }
            }
        }
        if (!corners.empty())
        {
            double lo[3], hi[3];
            bool finite = true;
            for (size_t c = 0; c < corners.size(); ++c)
            {
                double world[3];
                hostRotate(rv.rotation, corners[c].data(), world);
                for (int k = 0; k < 3; ++k)
                {
                    world[k] += rv.position[k];
                    finite = finite && std::isfinite(world[k]);
                    lo[k] = c == 0 ? world[k] : std::min(lo[k], world[k]);
                    hi[k] = c == 0 ? world[k] : std::max(hi[k], world[k]);
                }
            }
            const double grow =
                4e-6 * sceneScale + (rv.type == RT_VOLUME_SPHERE ? std::fabs(rv.param[3]) * (1.0 + 1e-6) : 0.0);
            if (finite)
            {
                v.hasWorldBox = 1;
                for (int k = 0; k < 3; ++k)
                {
                    v.worldBox[k] = std::nextafter(static_cast<float>(lo[k] - grow), -std::numeric_limits<float>::infinity());
                    v.worldBox[3 + k] = std::nextafter(static_cast<float>(hi[k] + grow), std::numeric_limits<float>::infinity());
                }
            }
        }
        volumes.push_back(v);
    }

    // ---- materials ----
    std::vector<rtd::Material> materials;
    for (int64_t i = 0; i < scene->material_count; ++i)
    {
        const rt_material& rm = scene->materials[i];
        require(rm.type >= RT_MATERIAL_LAMBERTIAN && rm.type <= RT_MATERIAL_MICROFACET, "unknown material type");
        rtd::Material m{};
        m.type = rm.type;
        std::memcpy(m.color, rm.color, sizeof(m.color));
        m.alpha = rm.alpha;
        m.ior = rm.ior;
        materials.push_back(m);
    }

    // ---- lights ----
    std::vector<rtd::Light> lights;
    for (int64_t i = 0; i < scene->light_count; ++i)
    {
        const rt_light& rl = scene->lights[i];
        require(rl.type >= RT_LIGHT_OMNI && rl.type <= RT_LIGHT_AREA, "unknown light type");
        require(rl.type != RT_LIGHT_SPOT || rl.linearity == 1.0,
                "spot light linearity other than 1 is not supported on the device");
        rtd::Light l{};
        l.type = rl.type;
        l.shape = rl.shape;
        std::memcpy(l.pos, rl.position, sizeof(l.pos));
        l.rot = rtd::Quat{rl.rotation[0], rl.rotation[1], rl.rotation[2], rl.rotation[3]};
        const double unitZ[3] = {0, 0, 1};
        hostRotate(rl.rotation, unitZ, l.forward);  // Object::forward, reference src/Object.cpp:45-48
        std::memcpy(l.color, rl.color, sizeof(l.color));
        l.flux = rl.luminous_flux;
        l.innerRadius = rl.inner_radius;
        l.maxAngle = rl.max_angle;
        l.radius = rl.radius;
        l.width = rl.width;
        l.height = rl.height;
        lights.push_back(l);
    }

    // ---- cameras (projection constants of reference src/Camera.cpp:239-275 and the
    //      pixel half angle of src/Worker.cpp:344-363, resolved once on the host) ----
    t->cameraDesc.assign(scene->cameras, scene->cameras + scene->camera_count);
    t->cameraHost.clear();
    t->cameraBuffers.clear();
    for (int64_t i = 0; i < scene->camera_count; ++i)
    {
        t->cameraBuffers.emplace_back(new DeviceArray<float>());
    }
    t->boundBuffers.assign(static_cast<size_t>(scene->camera_count), nullptr);
    const float start = static_cast<float>(scene->settings.frame_time);
    const float shutter = static_cast<float>(scene->settings.shutter_time);
    const float windowStart = start;  // reference src/Renderer.cpp:145-179
    const float windowEnd = shutter > 0.0f ? start + shutter : std::numeric_limits<float>::infinity();
    for (int64_t i = 0; i < scene->camera_count; ++i)
    {
        const rt_camera& rc = scene->cameras[i];
        require(rc.width > 0 && rc.height > 0, "camera resolution must be positive");
        require(static_cast<uint64_t>(rc.width) * static_cast<uint64_t>(rc.height) * 3 < (1ull << 32),
                "camera buffer too large (its float index must fit 32 bits)");
        rtd::Camera c{};
        std::memcpy(c.pos, rc.position, sizeof(c.pos));
        const double ux[3] = {1, 0, 0}, uy[3] = {0, 1, 0}, uz[3] = {0, 0, 1};
        hostRotate(rc.rotation, ux, c.right);
        hostRotate(rc.rotation, uy, c.up);
        hostRotate(rc.rotation, uz, c.forward);
        c.aspect = static_cast<double>(rc.width) / static_cast<double>(rc.height);
        c.halfHeight = std::tan(radians(rc.vertical_fov_degrees) / 2.0);
        const double pixelHalfAngle = 0.5 * radians(rc.vertical_fov_degrees) / static_cast<double>(rc.height);
        c.tanPixelHalfAngle = std::tan(pixelHalfAngle);
        c.width = rc.width;
        c.height = rc.height;
        c.bounceFilter = rc.bounce_filter;
        c.lightFilter = rc.light_filter;
        c.projection = rc.projection;
        c.orthographicHeight = rc.orthographic_height;
        c.apertureRadius = rc.aperture_radius;
        c.focusDistance = rc.focus_distance;
        require(rc.animation >= 0 && rc.animation <= scene->animation_count, "camera animation index out of range");
        c.animation = rc.animation - 1;
        c.windowStart = windowStart;
        c.windowEnd = windowEnd;
        t->cameraBuffers[i]->allocate(static_cast<size_t>(rc.width) * rc.height * 3);
        c.buffer = t->cameraBuffers[i]->ptr;
        t->cameraHost.push_back(c);
    }

    // ---- emitter patches: the lit face of every area light with positive radiance
    //      (reference src/EmitterPatch.cpp:15-53, AreaLight::surfaceRadiance src/AreaLight.cpp:80-104) ----
    std::vector<rtd::EmitterPatch> patches;
    for (int64_t i = 0; i < scene->light_count; ++i)
    {
        const rt_light& rl = scene->lights[i];
        if (rl.type != RT_LIGHT_AREA)
        {
            continue;
        }
        const double area = rl.shape == 1 ? M_PI * rl.radius * rl.radius : rl.width * rl.height;
        float radiance[3] = {0.0f, 0.0f, 0.0f};
        if (area > 0.0)
        {
            const double exitance = rl.luminous_flux / area;
            const float rad = static_cast<float>(exitance / M_PI);
            for (int k = 0; k < 3; ++k)
            {
                radiance[k] = rl.color[k] * rad;
            }
        }
        if (radiance[0] <= 0.0f && radiance[1] <= 0.0f && radiance[2] <= 0.0f)
        {
            continue;
        }
        rtd::EmitterPatch patch{};
        std::memcpy(patch.center, rl.position, sizeof(patch.center));
        const double axes[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        double* outs[3] = {patch.right, patch.up, patch.normal};
        for (int a = 0; a < 3; ++a)
        {
            double r[3];
            hostRotate(rl.rotation, axes[a], r);
            const double m = std::sqrt((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
            for (int k = 0; k < 3; ++k)
            {
                outs[a][k] = r[k] / m;
            }
        }
        std::memcpy(patch.radiance, radiance, sizeof(radiance));
        if (rl.shape == 1)
        {
            patch.isDisc = 1;
            patch.radius = rl.radius;
        }
        else
        {
            patch.halfWidth = rl.width * 0.5;
            patch.halfHeight = rl.height * 0.5;
        }
        patches.push_back(patch);
    }
    t->patches.upload(patches);
    t->patchHost = patches;
    t->gatherCache.clear();
    for (int64_t i = 0; i < scene->camera_count; ++i)
    {
        t->gatherCache.emplace_back(new rt_tracer::GatherCache());
    }

    // ---- keyframed poses: every track's key range is checked here, the kernels index without checks ----
    require(scene->animation_count >= 0 && scene->keyframe_count >= 0, "bad animation counts");
    std::vector<rt_animation> animations(scene->animations, scene->animations + scene->animation_count);
    std::vector<rt_keyframe> keyframes(scene->keyframes, scene->keyframes + scene->keyframe_count);
    for (const rt_animation& a : animations)
    {
        require(a.position_key_count >= 0 && a.first_position_key >= 0 &&
                    a.first_position_key + a.position_key_count <= scene->keyframe_count,
                "animation position keys out of range");
        require(a.rotation_key_count >= 0 && a.first_rotation_key >= 0 &&
                    a.first_rotation_key + a.rotation_key_count <= scene->keyframe_count,
                "animation rotation keys out of range");
    }
    t->animations.upload(animations);
    t->keyframes.upload(keyframes);

    // box of the ray keys: the world boxes of the static volumes (a volume without one counts as the scene's scale)
    {
        float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
        bool any = false, open = false;
        for (const rtd::Volume& v : volumes)
        {
            if (!v.hasWorldBox)
            {
                open = true;
                continue;
            }
            any = true;
            for (int k = 0; k < 3; ++k)
            {
                lo[k] = std::min(lo[k], v.worldBox[k]);
                hi[k] = std::max(hi[k], v.worldBox[3 + k]);
            }
        }
        if (!any || open)
        {
            for (int k = 0; k < 3; ++k)
            {
                lo[k] = any ? std::min(lo[k], -static_cast<float>(sceneScale)) : -static_cast<float>(sceneScale);
                hi[k] = any ? std::max(hi[k], static_cast<float>(sceneScale)) : static_cast<float>(sceneScale);
            }
        }
        const float side = std::max({hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2], 1e-6f});
        for (int k = 0; k < 3; ++k) t->keyOrigin[k] = lo[k];
        t->keyScale = 1.0f / side;
    }
    t->volumes.upload(volumes);
    t->cacheImage.upload(cacheImage);
    t->nodes.upload(nodes);
    t->triRec.upload(triRec);
    t->triNrm.upload(triNrm);
    t->triId.upload(triId);
    t->triOwner.upload(triOwner);
    t->materials.upload(materials);
    t->hasDielectric = false;
    for (const rtd::Material& m : materials)
    {
        if (m.type == RT_MATERIAL_DIELECTRIC) t->hasDielectric = true;
    }
    t->lights.upload(lights);
    t->cameras.upload(t->cameraHost);
    t->scalars.allocate(4);
    t->counters.allocate(1);

    rtd::SceneView& view = t->view;
    view.volumes = t->volumes.ptr;
    view.nodes = t->nodes.ptr;
    view.triRec = t->triRec.ptr;
    view.triNrm = t->triNrm.ptr;
    view.triId = t->triId.ptr;
    view.triOwner = t->triOwner.ptr;
    view.materials = t->materials.ptr;
    view.lights = t->lights.ptr;
    view.cameras = t->cameras.ptr;
    view.volumeCount = static_cast<int32_t>(volumes.size());
    view.groupCount = 0;
    for (const Group& group : groups) view.groupCount += group.members.empty() ? 0 : 1;
    view.lightCount = static_cast<int32_t>(lights.size());
    view.cameraCount = static_cast<int32_t>(t->cameraHost.size());
    view.materialCount = static_cast<int32_t>(materials.size());
    view.bounceThreshold = scene->settings.bounce_threshold;
    view.terminationThreshold = std::max(0.0, scene->settings.termination_threshold);  // reference src/Worker.cpp:194-199
    view.splatLuminanceClamp = std::max(0.0, scene->settings.splat_luminance_clamp);
    view.photonsPerLight = scene->settings.photons_per_light;
    view.seed = scene->settings.seeded ? scene->settings.seed : 0u;
    view.windowStart = windowStart;
    view.windowEnd = windowEnd;
    view.hasCamera = scene->camera_count > 0 ? 1 : 0;
    view.stackDepth = std::max(1, maxDepth);
    view.cacheBytes = static_cast<uint32_t>(cacheImage.size());
    view.cacheImage = t->cacheImage.ptr;
    view.patches = t->patches.ptr;
    view.patchCount = static_cast<int32_t>(patches.size());
    view.frameTime = start;
    view.shutterSpan = std::max(0.0f, shutter);
    view.cameraTimeSamples = scene->settings.camera_time_samples;
    // probeSeed: base seed + 0x9E3779B9, truncated to 32 bits (reference src/Renderer.cpp:92-100)
    view.probeSeed = static_cast<uint32_t>(static_cast<long long>(view.seed) + 0x9E3779B9LL);
    view.animations = t->animations.ptr;
    view.keyframes = t->keyframes.ptr;
    view.animationCount = static_cast<int32_t>(animations.size());
}

void resetState(rt_tracer* t)
{
    t->storePending = false;
    t->storeCursorKnown = true;
    t->storeCursorHost = 0;
    t->scalars.zero(t->stream);
    t->counters.zero(t->stream);
    for (size_t c = 0; c < t->cameraHost.size(); ++c)
    {
        const size_t bytes = static_cast<size_t>(t->cameraHost[c].width) * t->cameraHost[c].height * 3 * sizeof(float);
        check(hipMemsetAsync(t->cameraHost[c].buffer, 0, bytes, t->stream), "hipMemsetAsync camera buffer");
    }
    if (t->densityKeys.ptr)
    {
        check(hipMemsetAsync(t->densityKeys.ptr, 0xFF, t->densityKeys.count * sizeof(unsigned long long), t->stream),
              "hipMemsetAsync density keys");
        t->densityPower.zero(t->stream);
        t->densityCount.zero(t->stream);
    }
}

// bits that hold every value below `count`
int bitsFor(uint64_t count)
{
    int bits = 0;
    while (bits < 63 && (1ull << bits) < count) ++bits;
    return bits;
}

// The store's cursor as the device holds it (waits for the stream unless the host already knows it).
uint64_t storeCursor(rt_tracer* tracer)
{
    if (!tracer->storeCursorKnown)
    {
        unsigned long long cursor = 0;
        check(hipMemcpyAsync(&cursor, tracer->scalars.ptr + 1, sizeof(cursor), hipMemcpyDeviceToHost, tracer->stream),
              "hipMemcpyAsync store cursor");
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        tracer->storeCursorHost = cursor;
        tracer->storeCursorKnown = true;
    }
    return tracer->storeCursorHost;
}

// After a pass that overflowed the store: the records the reference would hold — the first in the order
// its workers meet the photons in (rtd::StoreView) — are moved into the capacity.  Everything that reads the
// store, or appends to it after a pass, comes through here first; a store that did not overflow costs the
// look at its cursor.  Waits for the stream.
void settleStore(rt_tracer* tracer)
{
    if (!tracer->storePending)
    {
        return;
    }
    tracer->storePending = false;
    tracer->storeCursorKnown = false;
    const uint64_t cursor = storeCursor(tracer);
    const uint64_t capacity = tracer->storeCapacity, first = tracer->storePassStart;
    if (cursor <= capacity || first >= capacity)
    {
        return;  // nothing was dropped, or the pass found the store full
    }
    const uint64_t held = std::min<uint64_t>(cursor, tracer->storePassOrdered ? tracer->storeAllocated : capacity);
    const uint64_t candidates = held - first, need = capacity - first;
    const bool verbose = std::getenv("RT_AMD_VERBOSE") != nullptr;
    if (!tracer->storePassOrdered || candidates <= need)
    {
        // no room behind the capacity: the store holds what arrived first, in the device's own order
        if (!tracer->storeOrderWarned)
        {
            std::fprintf(stderr, "[rt_amd] the bounce store overflowed in a pass it had no room to order: it holds the records "
                                 "that arrived first on the device, not the first in photon order\n");
            tracer->storeOrderWarned = true;
        }
        return;
    }
    hipStream_t stream = tracer->stream;
    if (tracer->storeSelectState.count == 0) tracer->storeSelectState.allocate(1);
    if (tracer->storeSelectLists.count < 2 * need) tracer->storeSelectLists.allocate(static_cast<size_t>(2 * need));
    rtd::StoreSelectState* state = tracer->storeSelectState.ptr;
    uint32_t* holes = tracer->storeSelectLists.ptr;
    uint32_t* fillers = tracer->storeSelectLists.ptr + need;
    rtd::launchStoreSelectThreshold(tracer->storeKeys.ptr + 1 + first, candidates, need, tracer->storePassKeyBits, state, stream);
    rtd::launchStoreSelectClassify(tracer->storeKeys.ptr + 1, first, candidates, capacity, state, holes, fillers, stream);
    rtd::launchStoreSelectMove(tracer->storeRecords.ptr, tracer->storeKeys.ptr + 1, state, holes, fillers, need, stream);
    check(hipGetLastError(), "store selection");
    unsigned long long found[4] = {0, 0, 0, 0};  // prefix, remaining, holes, fillers
    unsigned long long leastDropped = ~0ull;
    check(hipMemcpyAsync(found, state, sizeof(found), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync selection");
    check(hipMemcpyAsync(&leastDropped, tracer->storeKeys.ptr, sizeof(leastDropped), hipMemcpyDeviceToHost, stream),
          "hipMemcpyAsync selection");
    check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (verbose)
    {
        std::fprintf(stderr, "[rt_amd] store overflow: %llu records of the pass for %llu places, %llu moved in\n",
                     static_cast<unsigned long long>(candidates), static_cast<unsigned long long>(need), found[3]);
    }
    if (found[2] != found[3])
    {
        throw ApiError(RT_ERR_STATE, "store selection: " + std::to_string(found[2]) + " records to replace, " +
                                         std::to_string(found[3]) + " to move in");
    }
    if (cursor > tracer->storeAllocated && leastDropped <= found[0] && !tracer->storeOrderWarned)
    {
        std::fprintf(stderr, "[rt_amd] the bounce store overflowed its room as well (%llu records for %llu): some of the first "
                             "records in photon order were dropped\n",
                     static_cast<unsigned long long>(cursor), static_cast<unsigned long long>(tracer->storeAllocated));
        tracer->storeOrderWarned = true;
    }
}

// What a pass needs to know about the store before it appends: the cursor it starts from (an earlier pass
// is settled first), the order keys' width.
void beginStorePass(rt_tracer* tracer, const rtd::TraceArgs& args)
{
    settleStore(tracer);
    tracer->storePassStart = storeCursor(tracer);
    tracer->storePassOrdered = args.store.keys != nullptr;
    tracer->storePassKeyBits = args.store.keyBits + std::max(1, bitsFor(args.totalWork));
    if (tracer->storePassOrdered)
    {
        const unsigned long long none = ~0ull;
        check(hipMemcpyAsync(tracer->storeKeys.ptr, &none, sizeof(none), hipMemcpyHostToDevice, tracer->stream), "hipMemcpyAsync");
    }
    tracer->storePending = true;
    tracer->storeCursorKnown = false;
}

struct CellKey
{
    int64_t x, y, z;
    bool operator<(const CellKey& o) const { return std::tie(x, y, z) < std::tie(o.x, o.y, o.z); }
};

uint64_t cellSlot(const CellKey& k, uint64_t mask)
{
    uint64_t h = static_cast<uint64_t>(k.x) * 73856093ull;  // reference include/ProbeIndex.h:71-80
    h ^= static_cast<uint64_t>(k.y) * 19349663ull;
    h ^= static_cast<uint64_t>(k.z) * 83492791ull;
    return ((h * 0x9E3779B97F4A7C15ull) >> 20) & mask;
}

// A span of device work of kind `kind` (rt_tracer::setupSpans) starts / ends here on the tracer's stream.
size_t beginSetupSpan(rt_tracer* tracer, int kind)
{
    rt_tracer::TimedSpan span;
    check(hipEventCreate(&span.begin), "hipEventCreate");
    check(hipEventCreate(&span.end), "hipEventCreate");
    check(hipEventRecord(span.begin, tracer->stream), "hipEventRecord");
    tracer->setupSpans[kind].push_back(span);
    return tracer->setupSpans[kind].size() - 1;
}

void endSetupSpan(rt_tracer* tracer, int kind, size_t index)
{
    check(hipEventRecord(tracer->setupSpans[kind][index].end, tracer->stream), "hipEventRecord");
}

void clearSetupSpans(rt_tracer* tracer, int kind)
{
    for (rt_tracer::TimedSpan& span : tracer->setupSpans[kind])
    {
        if (span.begin) (void)hipEventDestroy(span.begin);
        if (span.end) (void)hipEventDestroy(span.end);
    }
    tracer->setupSpans[kind].clear();
}

// Runs the probe pass of one camera once and keeps its records on the device (for the
// gather) and on the host (for rt_collect_gather_points).
rt_tracer::GatherCache& ensureGatherPoints(rt_tracer* tracer, int32_t camera)
{
    rt_tracer::GatherCache& cache = *tracer->gatherCache[static_cast<size_t>(camera)];
    if (cache.valid)
    {
        return cache;
    }
    const rtd::Camera& cam = tracer->cameraHost[camera];
    const int32_t stride = static_cast<int32_t>(std::max<uint64_t>(1, tracer->settings.probe_sub_sample));
    const int64_t gw = (cam.width + stride - 1) / stride, gh = (cam.height + stride - 1) / stride;
    const int64_t pixels = gw * gh;
    DeviceArray<unsigned long long> stats;
    cache.pixels = pixels;
    cache.counts.reserve(static_cast<size_t>(pixels));
    stats.allocate(3);
    const size_t span = beginSetupSpan(tracer, 0);
    cache.counts.zero(tracer->stream);
    stats.zero(tracer->stream);
    // one record a pixel at most (collectPixel's sample loops run once): one pass of casts does
    const bool lensSamples = cam.projection == RT_PROJECTION_REALLENS && cam.apertureRadius > 0.0;
    const bool shutterSamples = tracer->view.shutterSpan > 0.0f && tracer->view.cameraTimeSamples > 1;
    // (RT_AMD_PROBE_PASS=two takes such a camera through the count-then-write form as well: for tests)
    const char* form = std::getenv("RT_AMD_PROBE_PASS");
    const bool onePass = !lensSamples && !shutterSamples && !tracer->hasDielectric && !(form != nullptr && std::string(form) == "two");
    if (onePass)
    {
        cache.slots.reserve(static_cast<size_t>(std::max<int64_t>(pixels, 1)));
        cache.deferred.reserve(static_cast<size_t>(std::max<int64_t>(pixels, 1)));
        cache.deferredCount.reserve(1);
        cache.deferredCount.zero(tracer->stream);
        rtd::launchGatherCollect(tracer->view, camera, stride, pixels, cache.counts.ptr, stats.ptr, cache.slots.ptr,
                                 cache.deferred.ptr, cache.deferredCount.ptr, tracer->stream);
    }
    else
    {
        rtd::launchGatherCount(tracer->view, camera, stride, pixels, cache.counts.ptr, stats.ptr, tracer->stream);
    }
    check(hipGetLastError(), "gather count launch");
    // offsets of every pixel's records: exclusive scan of the counts, on the device
    cache.offsets.reserve(static_cast<size_t>(pixels));
    size_t scanBytes = 0;
    check(rtd::scanCounts(nullptr, &scanBytes, cache.counts.ptr, cache.offsets.ptr, pixels, tracer->stream), "scan size query");
    DeviceArray<unsigned char> scanScratch;
    scanScratch.allocate(std::max<size_t>(scanBytes, 1));
    check(rtd::scanCounts(scanScratch.ptr, &scanBytes, cache.counts.ptr, cache.offsets.ptr, pixels, tracer->stream),
          "gather offsets scan");
    check(hipGetLastError(), "gather offsets scan");
    unsigned long long hostStats[3] = {0, 0, 0};
    check(hipMemcpyAsync(hostStats, stats.ptr, sizeof(hostStats), hipMemcpyDeviceToHost, tracer->stream),
          "hipMemcpyAsync stats");
    uint64_t lastOffset = 0;
    uint32_t lastCount = 0;
    if (pixels > 0)
    {
        check(hipMemcpyAsync(&lastOffset, cache.offsets.ptr + (pixels - 1), sizeof(lastOffset), hipMemcpyDeviceToHost,
                             tracer->stream),
              "hipMemcpyAsync last offset");
        check(hipMemcpyAsync(&lastCount, cache.counts.ptr + (pixels - 1), sizeof(lastCount), hipMemcpyDeviceToHost,
                             tracer->stream),
              "hipMemcpyAsync last count");
    }
    check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
    const uint64_t total = lastOffset + lastCount;
    cache.total = total;
    cache.points.clear();
    cache.hostCopied = false;
    if (!onePass) cache.devicePoints.reserve(static_cast<size_t>(std::max<uint64_t>(total, 1)));
    cache.byPixel = onePass;
    cache.together = !onePass;
    if (total > 0 && !onePass)
    {
        rtd::launchGatherWrite(tracer->view, camera, stride, pixels, cache.counts.ptr, cache.offsets.ptr,
                               cache.devicePoints.ptr, tracer->stream);
        check(hipGetLastError(), "gather write launch");
    }
    endSetupSpan(tracer, 0, span);
    for (int k = 0; k < 3; ++k)
    {
        cache.stats[k] = hostStats[k];
    }
    cache.valid = true;
    return cache;
}

// The probe pass's records on the host (copied once, when a caller wants to read them).
const std::vector<rtd::GatherPoint>& hostGatherPoints(rt_tracer* tracer, rt_tracer::GatherCache& cache)
{
    if (!cache.together)
    {
        // the one-pass form keeps the records at their pixels: moved together when they are first asked for
        cache.devicePoints.reserve(static_cast<size_t>(std::max<uint64_t>(cache.total, 1)));
        rtd::launchGatherCompact(cache.slots.ptr, cache.counts.ptr, cache.offsets.ptr, cache.pixels, cache.devicePoints.ptr,
                                 tracer->stream);
        check(hipGetLastError(), "gather compact launch");
        cache.together = true;
    }
    if (!cache.hostCopied)
    {
        cache.points.assign(static_cast<size_t>(cache.total), rtd::GatherPoint{});
        if (cache.total > 0)
        {
            check(hipMemcpyAsync(cache.points.data(), cache.devicePoints.ptr, cache.total * sizeof(rtd::GatherPoint),
                                 hipMemcpyDeviceToHost, tracer->stream),
                  "hipMemcpyAsync gather points");
            check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        }
        cache.hostCopied = true;
    }
    return cache.points;
}

// Device-side build of the probe grid.  Returns false (nothing changed) when the general host
// build must be used: a keep radius that spans more than one cell, or cell coordinates beyond
// the packable +-2^20 range.
bool buildProbesOnDevice(rt_tracer* tracer, const double* probes, bool probesOnDevice, uint64_t n, double invCellSize,
                         double keepRadius)
{
    const int64_t reach = static_cast<int64_t>(std::ceil(keepRadius * invCellSize));
    if (n == 0 || reach != 1 || n >= (1ull << 29))  // a group's number and its place in a brick share 32 bits
    {
        return false;
    }
    hipStream_t stream = tracer->stream;
    clearSetupSpans(tracer, 1);
    clearSetupSpans(tracer, 2);
    const size_t indexSpan = beginSetupSpan(tracer, 1);
    // work space kept between builds (a frame of a sequence builds the index of much the same probes again)
    rt_tracer::ProbeBuildSpace& space = tracer->probeBuildSpace;
    DeviceArray<double>& points = space.points;
    DeviceArray<unsigned long long>&keysIn = space.keysIn, &keysOut = space.keysOut, &scalars = space.scalars;
    DeviceArray<uint32_t>&orderIn = space.orderIn, &orderOut = space.orderOut, &flag = space.flag;
    DeviceArray<unsigned char>& scratch = space.scratch;
    points.reserve(3 * n);
    keysIn.reserve(n);
    keysOut.reserve(n);
    orderIn.reserve(n);
    orderOut.reserve(n);
    flag.reserve(1);
    scalars.reserve(3);
    flag.zero(stream);
    scalars.zero(stream);
    check(hipMemcpyAsync(points.ptr, probes, 3 * n * sizeof(double),
                         probesOnDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream),
          "hipMemcpyAsync probes");
    rtd::ProbeBuild build{points.ptr, n, invCellSize, keysIn.ptr, orderIn.ptr};
    rtd::launchProbeKeys(build, flag.ptr, stream);
    size_t bytes = 0;
    check(rtd::sortKeyPairs(nullptr, &bytes, keysIn.ptr, keysOut.ptr, orderIn.ptr, orderOut.ptr, n, stream), "sort size query");
    space.heads.reserve(n);
    space.cellOrdinal.reserve(n);
    space.groupOrdinal.reserve(n);
    size_t scanBytes = 0;
    check(rtd::scanCounts(nullptr, &scanBytes, space.heads.ptr, space.cellOrdinal.ptr, static_cast<int64_t>(n), stream),
          "scan size query");
    scratch.reserve(std::max<size_t>(std::max(bytes, scanBytes), 1));
    check(rtd::sortKeyPairs(scratch.ptr, &bytes, keysIn.ptr, keysOut.ptr, orderIn.ptr, orderOut.ptr, n, stream), "probe cell sort");
    // scalars: [0] cells, [1] groups of 2 x 2 x 2 cells, [2] bricks
    rtd::launchCountRuns(keysOut.ptr, n, 0, scalars.ptr, stream);
    rtd::launchCountRuns(keysOut.ptr, n, 3, scalars.ptr + 1, stream);
    // every probe's cell and group by number
    rtd::launchRunHeads(keysOut.ptr, n, 0, space.heads.ptr, stream);
    check(rtd::scanCounts(scratch.ptr, &scanBytes, space.heads.ptr, space.cellOrdinal.ptr, static_cast<int64_t>(n), stream),
          "cell ordinal scan");
    rtd::launchRunHeads(keysOut.ptr, n, 3, space.heads.ptr, stream);
    check(rtd::scanCounts(scratch.ptr, &scanBytes, space.heads.ptr, space.groupOrdinal.ptr, static_cast<int64_t>(n), stream),
          "group ordinal scan");
    unsigned long long counts[2] = {0, 0};
    uint32_t unpackable = 0;
    check(hipMemcpyAsync(counts, scalars.ptr, sizeof(counts), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync runs");
    check(hipMemcpyAsync(&unpackable, flag.ptr, sizeof(unpackable), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync flag");
    check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    if (unpackable != 0)
    {
        return false;
    }
    const uint64_t cellCount = counts[0], groups = counts[1];

    tracer->probePoints.reserve(3 * n + 2);  // the keep test reads 32 bytes at a probe position
    tracer->denseCells.reserve(cellCount);
    rtd::launchDenseCells(keysOut.ptr, orderOut.ptr, space.cellOrdinal.ptr, points.ptr, n, tracer->denseCells.ptr,
                          tracer->probePoints.ptr, stream);
    endSetupSpan(tracer, 1, indexSpan);

    const size_t filterSpan = beginSetupSpan(tracer, 2);
    // the bricks: a record per group, the eight lists of the bricks the groups lie in, merged pair by pair
    space.groupKey.reserve(groups);
    space.groupOwn.reserve(groups);
    space.groupFirst.reserve(groups);
    rtd::launchCellGroups(keysOut.ptr, space.cellOrdinal.ptr, space.groupOrdinal.ptr, n, space.groupKey.ptr, space.groupOwn.ptr,
                          space.groupFirst.ptr, stream);
    const uint64_t total = 8 * groups;
    space.listKeys[0].reserve(total);
    space.listKeys[1].reserve(total);
    space.listEntries[0].reserve(total);
    space.listEntries[1].reserve(total);
    size_t mergeBytes = 0;
    check(rtd::mergeBrickLists(nullptr, &mergeBytes, space.groupKey.ptr, groups, 0, 1, space.listKeys[1].ptr, space.listEntries[1].ptr,
                               stream),
          "merge size query");
    for (uint64_t width = 2 * groups; width < total; width *= 2)
    {
        size_t need = 0;
        check(rtd::mergeKeyPairs(nullptr, &need, space.listKeys[0].ptr, space.listEntries[0].ptr, width, space.listKeys[0].ptr,
                                 space.listEntries[0].ptr, width, space.listKeys[1].ptr, space.listEntries[1].ptr, stream),
              "merge size query");
        mergeBytes = std::max(mergeBytes, need);
    }
    space.mergeScratch.reserve(std::max<size_t>(mergeBytes, 1));
    // lists 0 + 1, 2 + 3, 4 + 5, 6 + 7 (straight from the groups), then the merges pair by pair
    int from = 1;
    for (int pairOfLists = 0; pairOfLists < 4; ++pairOfLists)
    {
        size_t need = mergeBytes;
        check(rtd::mergeBrickLists(space.mergeScratch.ptr, &need, space.groupKey.ptr, groups, 2 * pairOfLists, 2 * pairOfLists + 1,
                                   space.listKeys[from].ptr + 2 * groups * pairOfLists,
                                   space.listEntries[from].ptr + 2 * groups * pairOfLists, stream),
              "brick list merge");
    }
    for (uint64_t width = 2 * groups; width < total; width *= 2, from ^= 1)
    {
        for (uint64_t at = 0; at < total; at += 2 * width)
        {
            size_t need = mergeBytes;
            check(rtd::mergeKeyPairs(space.mergeScratch.ptr, &need, space.listKeys[from].ptr + at, space.listEntries[from].ptr + at,
                                     width, space.listKeys[from].ptr + at + width, space.listEntries[from].ptr + at + width, width,
                                     space.listKeys[from ^ 1].ptr + at, space.listEntries[from ^ 1].ptr + at, stream),
                  "brick list merge");
        }
    }
    rtd::launchCountRuns(space.listKeys[from].ptr, total, 0, scalars.ptr + 2, stream);
    unsigned long long brickCount = 0;
    check(hipMemcpyAsync(&brickCount, scalars.ptr + 2, sizeof(brickCount), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync bricks");
    check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    uint64_t brickSlots = 16;
    while (brickSlots < 3 * brickCount + 1)
    {
        brickSlots <<= 1;
    }
    tracer->bricks.reserve(brickSlots);
    check(hipMemsetAsync(tracer->bricks.ptr, 0xFF, brickSlots * sizeof(rtd::ProbeBrick), stream), "hipMemsetAsync");
    rtd::launchBricks(space.listKeys[from].ptr, space.listEntries[from].ptr, total, space.groupOwn.ptr, space.groupFirst.ptr,
                      tracer->bricks.ptr, brickSlots - 1, stream);
    check(hipGetLastError(), "probe grid build");
    endSetupSpan(tracer, 2, filterSpan);
    check(hipStreamSynchronize(stream), "hipStreamSynchronize");

    rtd::ProbeView& pv = tracer->probeView;
    pv.slotKey = nullptr;
    pv.slotStart = nullptr;
    pv.slotCount = nullptr;
    pv.points = tracer->probePoints.ptr;
    pv.slotMask = 0;
    pv.probeCount = n;
    pv.invCellSize = invCellSize;
    pv.keepRadius = keepRadius;
    pv.reach = reach;
    pv.brickMask = brickSlots - 1;
    pv.bricks = tracer->bricks.ptr;
    pv.cells = tracer->denseCells.ptr;
    tracer->slotKey.release();
    tracer->slotStart.release();
    tracer->slotCount.release();
    return true;
}

template <typename F>
int32_t guarded(F&& body)
{
    try
    {
        body();
        return RT_OK;
    }
    catch (const ApiError& e)
    {
        g_error = e.what();
        return e.code;
    }
    catch (const std::exception& e)
    {
        g_error = e.what();
        return RT_ERR_ARGUMENT;
    }
}

// Run a function-level kernel over host rows: copy in, launch, copy out.
template <typename Launch>
void runRows(int32_t device, int64_t n, const void* in, size_t inBytes, void* out, size_t outBytes, Launch&& launch)
{
    require(n >= 0, "negative row count");
    if (n == 0)
    {
        return;
    }
    require(in != nullptr && out != nullptr, "null buffer");
    check(hipSetDevice(device), "hipSetDevice");
    DeviceArray<unsigned char> dIn, dOut;
    dIn.allocate(inBytes);
    dOut.allocate(outBytes);
    check(hipMemcpy(dIn.ptr, in, inBytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
    launch(dIn.ptr, dOut.ptr);
    check(hipGetLastError(), "kernel launch");
    check(hipDeviceSynchronize(), "hipDeviceSynchronize");
    check(hipMemcpy(out, dOut.ptr, outBytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
}

}  // namespace

extern "C" {

const char* rt_last_error(void) { return g_error.c_str(); }

uint64_t rt_abi_struct_size(int32_t which)
{
    switch (which)
    {
        case 0: return sizeof(rt_scene);
        case 1: return sizeof(rt_volume);
        case 2: return sizeof(rt_mesh);
        case 3: return sizeof(rt_material);
        case 4: return sizeof(rt_light);
        case 5: return sizeof(rt_camera);
        case 6: return sizeof(rt_settings);
        case 7: return sizeof(rt_counters);
        case 8: return sizeof(rt_trace_params);
        case 9: return sizeof(rt_animation);
        case 10: return sizeof(rt_keyframe);
        default: return 0;
    }
}

int32_t rt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
    {
        return 0;
    }
    return n;
}

int32_t rt_check_tree(const double* tri_pos, int64_t count, int32_t leaf_size, double pad, int64_t stats[6])
{
    return guarded([&]() {
        require(count >= 0 && (tri_pos != nullptr || count == 0) && stats != nullptr, "null argument");
        require(leaf_size >= 1 && leaf_size <= 16 && pad >= 0.0, "bad leaf size or padding");
        const rtbvh::FlatMesh reference = rtbvh::build(tri_pos, count);
        const rtbvh::FlatMesh tree = rtbvh::buildSurfaceArea(tri_pos, count, reference, pad, leaf_size);
        const size_t n = static_cast<size_t>(count);
        auto triangleInside = [&](int32_t original, const double* lo, const double* hi, double margin) {
            for (int v = 0; v < 3; ++v)
            {
                for (int k = 0; k < 3; ++k)
                {
                    const double c = tri_pos[9 * static_cast<size_t>(original) + 3 * v + k];
                    if (!(c - margin >= lo[k] && c + margin <= hi[k])) return false;
                }
            }
            return true;
        };
        // both trees: every triangle once; links consistent
        int64_t leaves = 0, largestLeaf = 0;
        for (const rtbvh::FlatMesh* mesh : {&reference, &tree})
        {
            require(mesh->order.size() == n && mesh->rank.size() == n && mesh->referenceBox.size() == 6 * n,
                    "tree does not hold one slot per triangle");
            std::vector<char> seen(n, 0);
            for (int32_t original : mesh->order)
            {
                require(original >= 0 && static_cast<size_t>(original) < n && !seen[static_cast<size_t>(original)],
                        "triangle missing or repeated");
                seen[static_cast<size_t>(original)] = 1;
            }
            const int32_t nodeCount = static_cast<int32_t>(mesh->nodes.size());
            require(nodeCount >= 1, "tree has no root");
            int64_t slots = 0;
            for (int32_t i = 0; i < nodeCount; ++i)
            {
                const rtbvh::FlatNode& node = mesh->nodes[static_cast<size_t>(i)];
                require(node.skip > i && node.skip <= nodeCount, "skip link out of range");
                require(node.triCount >= 0 && node.triStart == slots, "pages are not contiguous in node order");
                slots += node.triCount;
                const bool hasLeft = node.skip > i + 1 && node.right != i + 1;
                require(node.right == -1 || (node.right > i && node.right < node.skip), "right child out of range");
                if (hasLeft)
                {
                    const int32_t afterLeft = mesh->nodes[static_cast<size_t>(i + 1)].skip;
                    require(node.right == -1 ? afterLeft == node.skip : afterLeft == node.right,
                            "left subtree does not end where the right one starts");
                }
                if (node.right >= 0)
                {
                    require(mesh->nodes[static_cast<size_t>(node.right)].skip == node.skip,
                            "right subtree does not end with its parent");
                }
            }
            require(slots == count, "pages do not add up to the triangle count");
        }
        // surface-area tree: boxes contain their triangles with the padding to spare, and their children
        std::vector<int32_t> parent(tree.nodes.size(), -1);
        for (size_t i = 0; i < tree.nodes.size(); ++i)
        {
            const rtbvh::FlatNode& node = tree.nodes[i];
            const bool leaf = node.skip == static_cast<int32_t>(i) + 1;
            if (leaf)
            {
                leaves += 1;
                largestLeaf = std::max<int64_t>(largestLeaf, node.triCount);
                require(count == 0 || node.triCount >= 1, "empty leaf");
            }
            else
            {
                require(node.triCount == 0 && node.right > static_cast<int32_t>(i) + 1, "inner node holds triangles or lacks a child");
                parent[i + 1] = static_cast<int32_t>(i);
                parent[static_cast<size_t>(node.right)] = static_cast<int32_t>(i);
            }
            for (int32_t slot = node.triStart; slot < node.triStart + node.triCount; ++slot)
            {
                // checked against the node and, through the chain of parents, every ancestor
                for (int32_t a = static_cast<int32_t>(i); a >= 0; a = parent[static_cast<size_t>(a)])
                {
                    const rtbvh::FlatNode& box = tree.nodes[static_cast<size_t>(a)];
                    require(triangleInside(tree.order[static_cast<size_t>(slot)], box.bmin, box.bmax, 0.999 * pad),
                            "padded box does not contain its triangle");
                }
            }
        }
        require(largestLeaf <= leaf_size, "leaf larger than asked for");
        // ranks and reference boxes
        std::vector<char> rankSeen(n, 0);
        for (size_t slot = 0; slot < n; ++slot)
        {
            const int32_t rank = tree.rank[slot];
            require(rank >= 0 && static_cast<size_t>(rank) < n && !rankSeen[static_cast<size_t>(rank)], "ranks are not a permutation");
            rankSeen[static_cast<size_t>(rank)] = 1;
            require(reference.order[static_cast<size_t>(rank)] == tree.order[slot], "rank points at another triangle");
            require(std::equal(tree.referenceBox.begin() + 6 * slot, tree.referenceBox.begin() + 6 * slot + 6,
                               reference.referenceBox.begin() + 6 * rank),
                    "reference box differs from the reference tree's");
            require(triangleInside(tree.order[slot], &tree.referenceBox[6 * slot], &tree.referenceBox[6 * slot + 3], 0.0),
                    "reference box does not contain its triangle");
        }
        // the walk's nodes: every child box contains the tree node's (padded) box; following the
        // references from the root reaches every leaf's slots exactly once
        const rtbvh::WideMesh wide = rtbvh::widen(tree);
        require(!wide.nodes.empty(), "walk has no root");
        // the device walks the breadth-first numbering: same nodes, every child behind its parent
        const rtbvh::WideMesh walked = rtbvh::breadthFirst(wide);
        require(walked.nodes.size() == wide.nodes.size(), "breadth-first numbering lost or repeated a node");
        for (size_t i = 0; i < walked.nodes.size(); ++i)
        {
            for (int32_t child : {walked.nodes[i].left, walked.nodes[i].right})
            {
                require(child < 0 || static_cast<size_t>(child) > i, "breadth-first numbering puts a child before its parent");
            }
        }
        {
            std::vector<char> slotSeen(n, 0);
            std::vector<std::pair<int32_t, int>> pending{{0, 1}};  // (reference, inner levels so far)
            std::vector<char> nodeSeen(walked.nodes.size(), 0);
            int deepest = 0;
            while (!pending.empty())
            {
                const auto [ref, level] = pending.back();
                pending.pop_back();
                if (ref == rtbvh::kNoChild) continue;
                if (ref < 0)
                {
                    const int32_t packed = -1 - ref;
                    const int32_t start = packed >> rtbvh::kLeafCountBits;
                    const int32_t countHere = (packed & ((1 << rtbvh::kLeafCountBits) - 1)) + 1;
                    for (int32_t slot = start; slot < start + countHere; ++slot)
                    {
                        require(slot >= 0 && static_cast<size_t>(slot) < n && !slotSeen[static_cast<size_t>(slot)],
                                "leaf reference out of range or repeated");
                        slotSeen[static_cast<size_t>(slot)] = 1;
                    }
                    continue;
                }
                require(static_cast<size_t>(ref) < walked.nodes.size() && !nodeSeen[static_cast<size_t>(ref)],
                        "node reference out of range or repeated");
                nodeSeen[static_cast<size_t>(ref)] = 1;
                deepest = std::max(deepest, level);
                const rtbvh::WideNode& node = walked.nodes[static_cast<size_t>(ref)];
                pending.push_back({node.left, level + 1});
                pending.push_back({node.right, level + 1});
            }
            for (size_t slot = 0; slot < n; ++slot) require(slotSeen[slot], "a triangle is not reachable");
            require(deepest <= std::max(1, wide.pushDepth), "stack depth understated");
            // boxes: walk both trees in step (inner tree nodes in preorder are the wide nodes in order)
            size_t wideAt = 0;
            for (size_t i = 0; i < tree.nodes.size(); ++i)
            {
                const rtbvh::FlatNode& node = tree.nodes[i];
                if (node.skip == static_cast<int32_t>(i) + 1) continue;
                const rtbvh::WideNode& w = wide.nodes[wideAt++];
                const rtbvh::FlatNode* kids[2] = {&tree.nodes[i + 1], &tree.nodes[static_cast<size_t>(node.right)]};
                const float* boxes[2] = {w.leftBox, w.rightBox};
                for (int c = 0; c < 2; ++c)
                {
                    for (int k = 0; k < 3; ++k)
                    {
                        require(static_cast<double>(boxes[c][k]) <= kids[c]->bmin[k] &&
                                    static_cast<double>(boxes[c][3 + k]) >= kids[c]->bmax[k],
                                "single-precision box does not contain the tree's box");
                    }
                }
            }
        }
        stats[0] = static_cast<int64_t>(reference.nodes.size());
        stats[1] = reference.depth;
        stats[2] = static_cast<int64_t>(tree.nodes.size());
        stats[3] = tree.depth;
        stats[4] = leaves;
        stats[5] = largestLeaf;
    });
}

int32_t rt_create(const rt_scene* scene, int32_t device, rt_tracer** out)
{
    if (out == nullptr)
    {
        g_error = "out is null";
        return RT_ERR_ARGUMENT;
    }
    *out = nullptr;
    return guarded([&]() {
        int n = 0;
        check(hipGetDeviceCount(&n), "hipGetDeviceCount");
        if (device < 0 || device >= n)
        {
            throw ApiError(RT_ERR_DEVICE, "no such GPU (device " + std::to_string(device) + " of " + std::to_string(n) + ")");
        }
        check(hipSetDevice(device), "hipSetDevice");
        std::unique_ptr<rt_tracer> t(new rt_tracer());
        t->device = device;
        buildScene(t.get(), scene);
        hipDeviceProp_t prop{};
        check(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
        t->computeUnits = prop.multiProcessorCount;
        t->staticScratch = staticScratchOf(device);
        t->countersStaged.allocate(sizeof(rt_counters));
        check(rtd::sortEmissionPairs(nullptr, &t->staticSortBytes, nullptr, nullptr, nullptr, nullptr, kStaticMaxIds,
                                     static_cast<int>(std::max<int64_t>(1, t->lightCount)), 0, nullptr),
              "sort size query");
        resetState(t.get());
        check(hipStreamSynchronize(t->stream), "hipStreamSynchronize");
        *out = t.release();
    });
}

void rt_destroy(rt_tracer* tracer)
{
    if (tracer)
    {
        (void)hipSetDevice(tracer->device);
        (void)hipDeviceSynchronize();
        for (hipEvent_t e : tracer->stageEvents)
        {
            if (e != nullptr) (void)hipEventDestroy(e);
        }
        for (int kind = 0; kind < 3; ++kind) clearSetupSpans(tracer, kind);
#ifdef RT_DIAG
        rtd::printDiagnostics();
#endif
        delete tracer;
    }
}

int32_t rt_set_stream(rt_tracer* tracer, void* hip_stream)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        tracer->stream = static_cast<hipStream_t>(hip_stream);
    });
}

namespace
{

// Deposits the scene's photon budget can make at most: photons x lights x bounces a photon can make.
uint64_t depositBound(const rt_tracer* tracer)
{
    const long double bound = static_cast<long double>(tracer->settings.photons_per_light) *
                              static_cast<long double>(std::max<int64_t>(1, tracer->lightCount)) *
                              static_cast<long double>(std::max(0, tracer->settings.bounce_threshold) + 1);
    return bound >= 1.8e19L ? ~0ull : static_cast<uint64_t>(bound);
}

uint64_t overflowRoom(const rt_tracer* tracer, uint64_t capacity)
{
    if (const char* v = std::getenv("RT_AMD_STORE_ROOM")) return std::strtoull(v, nullptr, 10);  // tests: a given room
    const uint64_t bound = depositBound(tracer);
    if (bound < capacity / 4) return 0;  // (the emitters' deposits share the store: a margin for them)
    size_t freeBytes = 0, totalBytes = 0;
    if (hipMemGetInfo(&freeBytes, &totalBytes) != hipSuccess) return 0;
    const uint64_t perRecord = 13 * sizeof(float) + sizeof(unsigned long long);
    const uint64_t affordable = static_cast<uint64_t>(freeBytes) / 4 / perRecord;
    uint64_t room = std::min<uint64_t>(bound, std::max<uint64_t>(8 * capacity, 64ull << 20));
    room = std::min(room, affordable);
    // slots are 32-bit numbers wherever they are listed
    const uint64_t slots = (1ull << 32) - 2;
    return capacity >= slots ? 0 : std::min(room, slots - capacity);
}

// ProbeIndex's constructor and the BounceStore behind it (reference src/ProbeIndex.cpp:5-25,
// src/Renderer.cpp:372-392) from `n` positions in host or device memory.
void setProbes(rt_tracer* tracer, const double* probes, bool probesOnDevice, uint64_t n, double cell_size, double keep_radius,
               uint64_t store_capacity)
{
    // ProbeIndex constructor, reference src/ProbeIndex.cpp:5-16
    const double cellSize = cell_size > 0.0 ? cell_size : 1.0;
    const double invCellSize = 1.0 / cellSize;
    const double keepRadius = keep_radius > 0.0 ? keep_radius : 0.0;
    tracer->storeCapacity = store_capacity;
    tracer->storeConfigured = store_capacity;
    tracer->storeVersion += 1;
    tracer->storePending = false;
    tracer->storeCursorKnown = false;
    // Room behind the capacity for the passes that can overflow it (see rtd::StoreView): as many records as
    // the scene's photon budget can deposit, within eight capacities (64 Mi records at least) and a quarter
    // of the memory that is free; none when the budget cannot come near the capacity.
    const uint64_t room = overflowRoom(tracer, store_capacity);
    tracer->storeAllocated = store_capacity + room;
    tracer->storeRecords.reserve(static_cast<size_t>(tracer->storeAllocated) * 13);
    if (room > 0) tracer->storeKeys.reserve(static_cast<size_t>(tracer->storeAllocated) + 1);
    else tracer->storeKeys.release();
    // the usual case (one cell of reach, coordinates that pack) is built on the device
    if (buildProbesOnDevice(tracer, probes, probesOnDevice, n, invCellSize, keepRadius))
    {
        tracer->probesSet = true;
        tracer->probeVersion += 1;
        return;
    }
    // the general build works on the host
    std::vector<double> fetched;
    if (probesOnDevice && n > 0)
    {
        fetched.resize(static_cast<size_t>(3 * n));
        check(hipMemcpy(fetched.data(), probes, fetched.size() * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy probes");
        probes = fetched.data();
    }
    std::map<CellKey, std::vector<uint32_t>> cells;
    for (uint64_t i = 0; i < n; ++i)
    {
        const CellKey key{static_cast<int64_t>(std::floor(probes[3 * i] * invCellSize)),
                          static_cast<int64_t>(std::floor(probes[3 * i + 1] * invCellSize)),
                          static_cast<int64_t>(std::floor(probes[3 * i + 2] * invCellSize))};
        cells[key].push_back(static_cast<uint32_t>(i));
    }
    uint64_t slots = 16;
    while (slots < 2 * cells.size() + 1)
    {
        slots <<= 1;
    }
    std::vector<int64_t> slotKey(3 * slots, 0);
    std::vector<uint32_t> slotStart(slots, 0), slotCount(slots, 0);
    std::vector<double> points;
    points.reserve(3 * n);
    for (const auto& cell : cells)
    {
        uint64_t slot = cellSlot(cell.first, slots - 1);
        while (slotCount[slot] != 0)
        {
            slot = (slot + 1) & (slots - 1);
        }
        slotKey[3 * slot + 0] = cell.first.x;
        slotKey[3 * slot + 1] = cell.first.y;
        slotKey[3 * slot + 2] = cell.first.z;
        slotStart[slot] = static_cast<uint32_t>(points.size() / 3);
        slotCount[slot] = static_cast<uint32_t>(cell.second.size());
        for (uint32_t index : cell.second)
        {
            points.insert(points.end(), probes + 3 * index, probes + 3 * index + 3);
        }
    }
    tracer->slotKey.upload(slotKey);
    tracer->slotStart.upload(slotStart);
    tracer->slotCount.upload(slotCount);
    tracer->probePoints.upload(points);
    rtd::ProbeView& pv = tracer->probeView;
    pv.slotKey = tracer->slotKey.ptr;
    pv.slotStart = tracer->slotStart.ptr;
    pv.slotCount = tracer->slotCount.ptr;
    pv.points = tracer->probePoints.ptr;
    pv.slotMask = slots - 1;
    pv.probeCount = n;
    pv.invCellSize = invCellSize;
    pv.keepRadius = keepRadius;
    pv.reach = static_cast<int64_t>(std::ceil(keepRadius * invCellSize));
    require(pv.reach <= 64, "keep radius spans too many probe cells");

    // the general form of the grid (the device build above makes the brick form whenever it applies):
    // the keep test scans the cells within reach
    pv.brickMask = 0;
    pv.bricks = nullptr;
    pv.cells = nullptr;
    tracer->bricks.release();
    tracer->denseCells.release();
    tracer->probesSet = true;
    tracer->probeVersion += 1;
}

}  // namespace

int32_t rt_set_probes(rt_tracer* tracer, const double* probes, uint64_t n, double cell_size, double keep_radius,
                      uint64_t store_capacity)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        require(n == 0 || probes != nullptr, "probes is null");
        require(n < (1ull << 32), "too many probes");
        activate(tracer);
        setProbes(tracer, probes, false, n, cell_size, keep_radius, store_capacity);
    });
}

int32_t rt_set_probes_from_cameras(rt_tracer* tracer, double cell_size, double keep_radius, uint64_t records_per_probe,
                                   uint64_t max_store_capacity, uint64_t* probe_count, uint64_t* store_capacity)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        activate(tracer);
        // the probe pass of every camera that takes part in the gather (debug cameras, which filter
        // by bounce or light, do not: reference src/Renderer.cpp:297-371)
        std::vector<rt_tracer::GatherCache*> caches;
        uint64_t total = 0;
        for (size_t c = 0; c < tracer->cameraDesc.size(); ++c)
        {
            if (tracer->cameraDesc[c].bounce_filter >= 0 || tracer->cameraDesc[c].light_filter >= 0) continue;
            rt_tracer::GatherCache& cache = ensureGatherPoints(tracer, static_cast<int32_t>(c));
            caches.push_back(&cache);
            total += cache.total;
        }
        require(total < (1ull << 32), "too many probes");
        DeviceArray<double> positions;
        positions.allocate(static_cast<size_t>(3 * total));
        uint64_t at = 0;
        for (rt_tracer::GatherCache* cache : caches)
        {
            if (cache->byPixel)
            {
                rtd::launchPixelProbePositions(cache->slots.ptr, cache->counts.ptr, cache->offsets.ptr, cache->pixels,
                                               positions.ptr + 3 * at, tracer->stream);
            }
            else
            {
                rtd::launchProbePositions(cache->devicePoints.ptr, cache->total, positions.ptr + 3 * at, tracer->stream);
            }
            at += cache->total;
        }
        check(hipGetLastError(), "probe positions launch");
        // BounceStore sizing, reference src/Renderer.cpp:382-390
        const uint64_t sized = total == 0 ? max_store_capacity : total * records_per_probe;
        const uint64_t capacity = std::min<uint64_t>(max_store_capacity, std::max<uint64_t>(1, sized));
        setProbes(tracer, positions.ptr, true, total, cell_size, keep_radius, capacity);
        if (probe_count != nullptr) *probe_count = total;
        if (store_capacity != nullptr) *store_capacity = capacity;
    });
}

int32_t rt_set_store_limit(rt_tracer* tracer, uint64_t max_records)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        if (!tracer->probesSet)
        {
            throw ApiError(RT_ERR_STATE, "rt_set_probes must be called before rt_set_store_limit");
        }
        activate(tracer);
        const uint64_t limit = (max_records == 0 || max_records > tracer->storeConfigured) ? tracer->storeConfigured : max_records;
        // the cursor counts the appends that were tried: a store that was over its old limit holds that limit
        settleStore(tracer);
        const uint64_t cursor = storeCursor(tracer);
        const uint64_t held = std::min<uint64_t>(cursor, tracer->storeCapacity);
        if (held > limit)
        {
            throw ApiError(RT_ERR_STATE, "rt_set_store_limit: the store holds " + std::to_string(held) + " records, more than the limit of " +
                                             std::to_string(limit));
        }
        tracer->storeCapacity = limit;
        tracer->storeVersion += 1;
        // a share of a store that could not overflow as a whole may: what lies behind the limit is its room
        if (tracer->storeKeys.ptr == nullptr && limit < tracer->storeAllocated && depositBound(tracer) >= limit / 4)
        {
            tracer->storeKeys.allocate(static_cast<size_t>(tracer->storeAllocated) + 1);
        }
    });
}

int32_t rt_deposit_emitters(rt_tracer* tracer, double deposit_spacing, uint64_t* samples, uint64_t* kept)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        tracer->storeVersion += 1;
        if (!tracer->probesSet)
        {
            throw ApiError(RT_ERR_STATE, "rt_deposit_emitters needs the probes (rt_set_probes) first");
        }
        activate(tracer);
        hipStream_t stream = tracer->stream;
        DeviceArray<unsigned long long> scalars;  // [0] samples on the patch, [1] kept
        scalars.allocate(2);
        const double spacing = std::max(deposit_spacing, 1e-6);
        settleStore(tracer);
        // (emitter deposits are laid down before the photons' so that no overflow drops them: they are
        // ordered among themselves by arrival, with no room behind the capacity)
        rtd::StoreView store{tracer->storeRecords.ptr, tracer->scalars.ptr + 1, tracer->storeCapacity, nullptr, 0};
        tracer->storeCursorKnown = false;
        uint64_t sampleTotal = 0, keptTotal = 0;
        // Into an empty store, with the probes and the spacing of the last run: the same records again
        // (their order in the store is the schedule's, then as now).  (RT_AMD_EMITTER_CACHE=0: always anew)
        rt_tracer::EmitterDeposits& last = tracer->emitterDeposits;
        const char* cacheWanted = std::getenv("RT_AMD_EMITTER_CACHE");
        const bool mayReuse = cacheWanted == nullptr || std::atoi(cacheWanted) != 0;
        unsigned long long cursorBefore = 0;
        check(hipMemcpyAsync(&cursorBefore, tracer->scalars.ptr + 1, sizeof(cursorBefore), hipMemcpyDeviceToHost, stream),
              "hipMemcpyAsync cursor");
        check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        if (mayReuse && cursorBefore == 0 && last.spacing == spacing && last.probeVersion == tracer->probeVersion &&
            last.kept <= tracer->storeCapacity)
        {
            if (last.kept > 0)
            {
                check(hipMemcpyAsync(tracer->storeRecords.ptr, last.records.ptr, last.kept * 13 * sizeof(float),
                                     hipMemcpyDeviceToDevice, stream),
                      "hipMemcpyAsync emitter deposits");
                const unsigned long long cursor = last.kept;
                check(hipMemcpyAsync(tracer->scalars.ptr + 1, &cursor, sizeof(cursor), hipMemcpyHostToDevice, stream),
                      "hipMemcpyAsync cursor");
                check(hipStreamSynchronize(stream), "hipStreamSynchronize");
            }
            tracer->storeCursorHost = last.kept;
            tracer->storeCursorKnown = true;
            if (samples != nullptr) *samples = last.samples;
            if (kept != nullptr) *kept = last.kept;
            return;
        }
        for (const rtd::EmitterPatch& patch : tracer->patchHost)
        {
            const double extentU = patch.isDisc ? patch.radius : patch.halfWidth;
            const double extentV = patch.isDisc ? patch.radius : patch.halfHeight;
            if (extentU <= 0.0 || extentV <= 0.0)
            {
                continue;
            }
            rtd::EmitterGrid grid{};
            const double cellsU = std::ceil(2.0 * extentU / spacing), cellsV = std::ceil(2.0 * extentV / spacing);
            require(cellsU < 2147483647.0 && cellsV < 2147483647.0, "emitter deposit grid too fine");
            grid.countU = std::max(1, static_cast<int>(cellsU));
            grid.countV = std::max(1, static_cast<int>(cellsV));
            grid.stepU = (2.0 * extentU) / grid.countU;
            grid.stepV = (2.0 * extentV) / grid.countV;
            grid.extentU = extentU;
            grid.extentV = extentV;
            grid.radius = patch.radius;
            grid.isDisc = patch.isDisc;
            std::memcpy(grid.center, patch.center, sizeof(grid.center));
            std::memcpy(grid.right, patch.right, sizeof(grid.right));
            std::memcpy(grid.up, patch.up, sizeof(grid.up));
            for (int k = 0; k < 3; ++k)
            {
                grid.normal[k] = static_cast<float>(patch.normal[k]);
            }
            const double area = patch.isDisc ? (M_PI * patch.radius * patch.radius)
                                             : (2.0 * patch.halfWidth) * (2.0 * patch.halfHeight);
            // the deposits share the patch's power: their number comes first
            uint64_t count = static_cast<uint64_t>(grid.countU) * static_cast<uint64_t>(grid.countV);
            if (patch.isDisc)
            {
                scalars.zero(stream);
                rtd::launchEmitterCount(grid, scalars.ptr, stream);
                unsigned long long inside = 0;
                check(hipMemcpyAsync(&inside, scalars.ptr, sizeof(inside), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
                check(hipStreamSynchronize(stream), "hipStreamSynchronize");
                count = inside;
            }
            if (count == 0)
            {
                continue;
            }
            sampleTotal += count;
            const double perDepositScale = (M_PI * area) / (4.0 * static_cast<double>(count));
            for (int k = 0; k < 3; ++k)
            {
                grid.power[k] = patch.radiance[k] * static_cast<float>(perDepositScale);
            }
            scalars.zero(stream);
            rtd::launchEmitterDeposit(grid, tracer->probeView, store, scalars.ptr + 1, stream);
            check(hipGetLastError(), "emitter deposit launch");
            unsigned long long keptHere = 0;
            check(hipMemcpyAsync(&keptHere, scalars.ptr + 1, sizeof(keptHere), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync");
            check(hipStreamSynchronize(stream), "hipStreamSynchronize");
            keptTotal += keptHere;
        }
        last.spacing = -1.0;
        if (mayReuse && cursorBefore == 0 && keptTotal <= tracer->storeCapacity)
        {
            if (last.records.count < keptTotal * 13) last.records.allocate(std::max<uint64_t>(1, keptTotal * 13));
            if (keptTotal > 0)
            {
                check(hipMemcpyAsync(last.records.ptr, tracer->storeRecords.ptr, keptTotal * 13 * sizeof(float),
                                     hipMemcpyDeviceToDevice, stream),
                      "hipMemcpyAsync emitter deposits");
            }
            last.kept = keptTotal;
            last.samples = sampleTotal;
            last.probeVersion = tracer->probeVersion;
            last.spacing = spacing;
        }
        tracer->storeCursorHost = cursorBefore + keptTotal;
        tracer->storeCursorKnown = true;
        if (samples != nullptr) *samples = sampleTotal;
        if (kept != nullptr) *kept = keptTotal;
    });
}

int32_t rt_set_splat(rt_tracer* tracer, double min_splat_radius, double density_cell_size, uint64_t density_capacity)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        activate(tracer);
        tracer->minSplatRadius = std::max(0.0, min_splat_radius);  // reference src/Worker.cpp:211-214
        rtd::DensityView& dv = tracer->densityView;
        dv = rtd::DensityView{};
        tracer->densityKeys.release();
        tracer->densityPower.release();
        tracer->densityCount.release();
        if (density_cell_size > 0.0)
        {
            uint64_t slots = 1024;
            while (slots < 2 * density_capacity)
            {
                slots <<= 1;
            }
            tracer->densityKeys.allocate(slots);
            tracer->densityPower.allocate(3 * slots);
            tracer->densityCount.allocate(slots);
            dv.keys = tracer->densityKeys.ptr;
            dv.power = tracer->densityPower.ptr;
            dv.count = tracer->densityCount.ptr;
            dv.slotMask = slots - 1;
            dv.invCellSize = 1.0 / density_cell_size;  // reference src/DensityGrid.cpp:5-9
            dv.enabled = 1;
            check(hipMemsetAsync(dv.keys, 0xFF, slots * sizeof(unsigned long long), tracer->stream), "hipMemsetAsync");
            tracer->densityPower.zero(tracer->stream);
            tracer->densityCount.zero(tracer->stream);
        }
        tracer->splatSet = true;
    });
}

int32_t rt_bind_camera_buffer(rt_tracer* tracer, int32_t camera, float* device_rgb)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        require(camera >= 0 && static_cast<size_t>(camera) < tracer->cameraHost.size(), "camera index out of range");
        activate(tracer);
        tracer->boundBuffers[camera] = device_rgb;
        tracer->cameraHost[camera].buffer = device_rgb ? device_rgb : tracer->cameraBuffers[camera]->ptr;
        check(hipMemcpyAsync(tracer->cameras.ptr + camera, &tracer->cameraHost[camera], sizeof(rtd::Camera),
                             hipMemcpyHostToDevice, tracer->stream),
              "hipMemcpyAsync camera");
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
    });
}

int32_t rt_reset(rt_tracer* tracer)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        tracer->storeVersion += 1;
        activate(tracer);
        resetState(tracer);
    });
}

namespace
{

// The photon range of a pass: indices [first, first + count) of every light, cut to the budget.
// photon_count == 0 asks for everything from first_photon on (rt_amd.h).
void resolveRange(const rt_tracer* tracer, const rt_trace_params* params, uint64_t* first, uint64_t* count)
{
    const uint64_t total = tracer->settings.photons_per_light;
    *first = std::min<uint64_t>(params->first_photon, total);
    *count = params->photon_count;
    if (*count == 0 || *count > total - *first)
    {
        *count = total - *first;
    }
    const uint64_t lights = static_cast<uint64_t>(std::max<int64_t>(1, tracer->lightCount));
    require(*count <= std::numeric_limits<uint64_t>::max() / lights, "photon range too large");
}

void requireSink(const rt_tracer* tracer, const rt_trace_params* params)
{
    if (params->sink_mode == RT_SINK_PROBE && !tracer->probesSet)
    {
        throw ApiError(RT_ERR_STATE, "rt_set_probes must be called before tracing with RT_SINK_PROBE");
    }
    if (params->sink_mode == RT_SINK_SPLAT && !tracer->splatSet)
    {
        throw ApiError(RT_ERR_STATE, "rt_set_splat must be called before tracing with RT_SINK_SPLAT");
    }
    require(params->sink_mode >= RT_SINK_NONE && params->sink_mode <= RT_SINK_SPLAT, "unknown sink mode");
}

// Everything of a pass's kernel arguments that does not depend on how its work is scheduled.
rtd::TraceArgs passArguments(const rt_tracer* tracer, const rt_trace_params* params, uint64_t first, uint64_t count)
{
    rtd::TraceArgs args{};
    args.scene = tracer->view;
    args.probes = tracer->probeView;
    args.store.records = tracer->storeRecords.ptr;
    args.store.cursor = tracer->scalars.ptr + 1;
    // the store takes records into the room behind its capacity when it has order keys to settle them by
    const int bounceBits = bitsFor(static_cast<uint64_t>(std::max(0, tracer->settings.bounce_threshold)) + 1);
    const uint64_t ids = count * static_cast<uint64_t>(std::max<int64_t>(1, tracer->lightCount));
    const bool ordered = tracer->storeKeys.ptr != nullptr && tracer->storeAllocated > tracer->storeCapacity && ids < (1ull << 32);
    args.store.room = ordered ? tracer->storeAllocated : tracer->storeCapacity;
    args.store.keys = ordered ? tracer->storeKeys.ptr + 1 : nullptr;
    args.store.keyBits = bounceBits;
    args.density = tracer->densityView;
    args.counters = tracer->counters.ptr;
    args.workCounter = tracer->scalars.ptr;
    args.firstPhoton = first;
    args.photonCount = count;
    args.totalWork = count * static_cast<uint64_t>(tracer->lightCount);
    args.minSplatRadius = tracer->minSplatRadius;
    args.sinkMode = params->sink_mode;
    // (RT_AMD_SPLAT_TOGETHER=0: every lane adds its own splat, for measurements)
    const char* together = std::getenv("RT_AMD_SPLAT_TOGETHER");
    args.splatTogether = together == nullptr || std::atoi(together) != 0;
    return args;
}

// Run lengths of one launch over `work` items: every resident wave should see `runsPerWaveWanted` runs (so
// small launches use short runs; at most 4096 positions a run, 256 for an unordered list), and the last
// sixteenth of a launch goes out in pieces a quarter as long so that the waves end together.  Returns the
// number of blocks to launch.
int scheduleLaunch(rtd::TraceArgs& launch, uint64_t work, int residentBlocks, bool ordered, int runsPerWaveWanted)
{
    const uint64_t wavesPerBlock = static_cast<uint64_t>(rtd::traceBlockThreads()) / 64;
    const uint64_t waves = static_cast<uint64_t>(residentBlocks) * wavesPerBlock;
    uint64_t runsPerWave = static_cast<uint64_t>(std::max(1, runsPerWaveWanted));
    if (const char* v = std::getenv("RT_AMD_RUNS_PER_WAVE")) runsPerWave = std::max<uint64_t>(1, std::strtoull(v, nullptr, 10));
    uint64_t chunk = work / (waves * runsPerWave);
    // parked photons are in no particular order: short runs, for balance
    chunk = std::min<uint64_t>(ordered ? 4096 : 256, std::max<uint64_t>(64, chunk)) & ~63ull;
    launch.chunk = static_cast<uint32_t>(chunk);
    launch.tailChunk = static_cast<uint32_t>(std::max<uint64_t>(64, (chunk / 4) & ~63ull));
    launch.fullClaims = (work - work / 16) / chunk;
    // emission-ordered ids: the runs go out in a scattered order (stride ~ 0.618 of their number, coprime to it)
    launch.claimStride = 1;
    if (ordered && launch.fullClaims > 2)
    {
        uint64_t stride = std::max<uint64_t>(2, static_cast<uint64_t>(0.6180339887 * static_cast<double>(launch.fullClaims)));
        while (std::gcd(stride, launch.fullClaims) != 1) ++stride;
        launch.claimStride = stride % launch.fullClaims;
    }
    const uint64_t wavesNeeded = (work + 63) / 64;
    return std::max(1, static_cast<int>(std::min<uint64_t>(static_cast<uint64_t>(residentBlocks),
                                                          (wavesNeeded + wavesPerBlock - 1) / wavesPerBlock)));
}

}  // namespace

int32_t rt_trace(rt_tracer* tracer, const rt_trace_params* params)
{
    return guarded([&]() {
        require(tracer != nullptr && params != nullptr, "null argument");
        activate(tracer);
        tracer->storeVersion += 1;
        uint64_t first = 0, count = 0;
        resolveRange(tracer, params, &first, &count);
        requireSink(tracer, params);
        rtd::TraceArgs args = passArguments(tracer, params, first, count);
        if (params->sink_mode == RT_SINK_PROBE) beginStorePass(tracer, args);

        const int32_t maxBounces = tracer->settings.bounce_threshold + 1;
        const uint64_t record = std::min<uint64_t>(params->record_photons, args.totalWork);
        tracer->recordPhotons = record;
        tracer->recEmitted.allocate(record * 6);
        tracer->recCasts.allocate(record * 2);
        tracer->recF64.allocate(record * maxBounces * 10);
        tracer->recI32.allocate(record * maxBounces * 8);
        tracer->recPower.allocate(record * maxBounces * 3);
        tracer->recEmitted.zero(tracer->stream);
        tracer->recCasts.zero(tracer->stream);
        tracer->recF64.zero(tracer->stream);
        if (tracer->recI32.ptr)
        {
            check(hipMemsetAsync(tracer->recI32.ptr, 0xFF, tracer->recI32.count * sizeof(int32_t), tracer->stream),
                  "hipMemsetAsync records");
        }
        tracer->recPower.zero(tracer->stream);
        args.records.recordPhotons = record;
        args.records.maxBounces = maxBounces;
        args.records.emitted = tracer->recEmitted.ptr;
        args.records.castsValid = tracer->recCasts.ptr;
        args.records.f64 = tracer->recF64.ptr;
        args.records.i32 = tracer->recI32.ptr;
        args.records.power = tracer->recPower.ptr;

        // a fresh range of photon ids for this launch
        check(hipMemsetAsync(tracer->scalars.ptr, 0, sizeof(unsigned long long), tracer->stream), "hipMemsetAsync");
        tracer->orderedLength = 0;
        if (args.totalWork == 0)
        {
            return;
        }
        for (hipEvent_t& e : tracer->stageEvents)
        {
            if (e == nullptr) check(hipEventCreate(&e), "hipEventCreate");
        }
        check(hipEventRecord(tracer->stageEvents[0], tracer->stream), "hipEventRecord");
        // Passes are traced in emission order (a schedule only; see emissionKeysKernel): every id is
        // keyed by its emission cell and the (key, id) pairs are radix-sorted.
        args.emissionOrder = nullptr;
        if (args.totalWork < (1ull << 31))
        {
            tracer->orderedLength = args.totalWork;
            const uint64_t n = args.totalWork;
            const int lights = static_cast<int>(tracer->lightCount);
            // Large passes are ordered in three steps of which only one is a pass of the radix sort
            // (rt_order.h); RT_AMD_EMISSION_SORT=radix|refine overrides the choice.
            // (beyond 2^28 ids a value of the sorted bits holds more ids than a run, and a run would then take
            // its ids from all over that value's cells: such passes are radix-sorted over all the bits)
            bool refine = n >= kRefineOrderMinimum && n <= (1ull << 28);
            if (const char* v = std::getenv("RT_AMD_EMISSION_SORT"))
            {
                refine = std::strcmp(v, "radix") != 0;
            }
            if (tracer->sortOrder.count < n)
            {
                tracer->sortOrder.allocate(n);
                tracer->sortIds.allocate(n);
                tracer->sortKeysIn.allocate(n);
                tracer->sortKeysOut.allocate(n);
                tracer->orderedCount = 0;
            }
            // The order depends on the seed, the lights' kinds and the ids only, all fixed for a tracer: a
            // caller that traces the same range again (the frames of a sequence) may ask for the list the
            // last pass left (RT_TRACE_REUSE_ORDER).
            const bool reuse = (params->flags & RT_TRACE_REUSE_ORDER) != 0 && tracer->orderedCount == args.photonCount &&
                               tracer->orderedFirst == args.firstPhoton && tracer->orderList != nullptr;
            if (!reuse && refine)
            {
                // bits 8-15 of the keys while the keys are made, the bits above by a radix pass, the low eight
                // run by run in LDS (rt_order.h)
                const uint64_t entries = rtd::digitMatrixEntries(n);
                if (tracer->orderCounts.count < entries)
                {
                    tracer->orderCounts.allocate(entries);
                    tracer->orderOffsets.allocate(entries);
                }
                // A pass of one light below kThreeStepMinimum ids that picked this order by itself sorts its keys'
                // top eight bits in the first step and goes straight to the last (rt_order.h).
                const bool twoSteps = lights == 1 && n < kThreeStepMinimum && std::getenv("RT_AMD_EMISSION_SORT") == nullptr;
                const int firstShift = twoSteps ? rtd::kRefineBits + rtd::kDigitBits : rtd::kRefineBits;
                size_t bytes = 0;
                check(rtd::digitOrder(args, firstShift, nullptr, &bytes, tracer->sortKeysIn.ptr, tracer->orderCounts.ptr,
                                      tracer->orderOffsets.ptr, tracer->sortKeysOut.ptr, tracer->sortIds.ptr, tracer->stream),
                      "scan size query");
                if (tracer->sortScratch.count < bytes) tracer->sortScratch.allocate(bytes);
                check(rtd::digitOrder(args, firstShift, tracer->sortScratch.ptr, &bytes, tracer->sortKeysIn.ptr, tracer->orderCounts.ptr,
                                      tracer->orderOffsets.ptr, tracer->sortKeysOut.ptr, tracer->sortIds.ptr, tracer->stream),
                      "emission order, first digit");
                // (the list of keys by id has been read: it takes the sorted keys)
                const int skipBits = rtd::kRefineBits + rtd::kDigitBits;
                const char* second = std::getenv("RT_AMD_EMISSION_SECOND");  // "library": the radix pass for one light too
                if (twoSteps)
                {
                    rtd::launchRefineOrder(tracer->sortKeysOut.ptr, tracer->sortIds.ptr, n, firstShift, tracer->sortOrder.ptr, tracer->stream);
                }
                else if (lights == 1 && !(second != nullptr && std::strcmp(second, "library") == 0))
                {
                    // one light: the keys end eight bits above the first digit, and a pass of our own places them
                    // (the matrix of counts and the scan's work space are those of the first digit)
                    check(rtd::stableDigitPass(n, skipBits, tracer->sortScratch.ptr, &bytes, tracer->sortKeysOut.ptr, tracer->sortIds.ptr,
                                               tracer->orderCounts.ptr, tracer->orderOffsets.ptr, tracer->sortKeysIn.ptr,
                                               tracer->sortOrder.ptr, tracer->stream),
                          "emission order, second digit");
                }
                else
                {
                    check(rtd::sortEmissionPairs(nullptr, &bytes, tracer->sortKeysOut.ptr, tracer->sortKeysIn.ptr, tracer->sortIds.ptr,
                                                 tracer->sortOrder.ptr, n, lights, skipBits, tracer->stream),
                          "sort size query");
                    if (tracer->sortScratch.count < bytes) tracer->sortScratch.allocate(bytes);
                    check(rtd::sortEmissionPairs(tracer->sortScratch.ptr, &bytes, tracer->sortKeysOut.ptr, tracer->sortKeysIn.ptr,
                                                 tracer->sortIds.ptr, tracer->sortOrder.ptr, n, lights, skipBits, tracer->stream),
                          "emission sort");
                }
                if (!twoSteps)
                {
                    rtd::launchRefineOrder(tracer->sortKeysIn.ptr, tracer->sortOrder.ptr, n, rtd::kRefineBits, tracer->sortIds.ptr, tracer->stream);
                }
                tracer->orderList = twoSteps ? tracer->sortOrder.ptr : tracer->sortIds.ptr;
                check(hipGetLastError(), "emission sort");
                tracer->orderedCount = args.photonCount;
                tracer->orderedFirst = args.firstPhoton;
            }
            else if (!reuse)
            {
                rtd::launchEmissionKeys(args, tracer->sortKeysIn.ptr, tracer->sortIds.ptr, tracer->stream);
                size_t bytes = 0;
                check(rtd::sortEmissionPairs(nullptr, &bytes, tracer->sortKeysIn.ptr, tracer->sortKeysOut.ptr, tracer->sortIds.ptr,
                                             tracer->sortOrder.ptr, n, lights, 0, tracer->stream),
                      "sort size query");
                if (tracer->sortScratch.count < bytes) tracer->sortScratch.allocate(bytes);
                check(rtd::sortEmissionPairs(tracer->sortScratch.ptr, &bytes, tracer->sortKeysIn.ptr, tracer->sortKeysOut.ptr,
                                             tracer->sortIds.ptr, tracer->sortOrder.ptr, n, lights, 0, tracer->stream),
                      "emission sort");
                tracer->orderList = tracer->sortOrder.ptr;
                check(hipGetLastError(), "emission sort");
                tracer->orderedCount = args.photonCount;
                tracer->orderedFirst = args.firstPhoton;
            }
            args.emissionOrder = tracer->orderList;
        }
        check(hipEventRecord(tracer->stageEvents[1], tracer->stream), "hipEventRecord");
        // the persistent grid fills the chip with the variant this pass runs
        const int residentBlocks = tracer->computeUnits * rtd::traceBlocksPerCu(args);
        // Bounce by bounce (large passes): the first launch emits the photons and parks those that
        // go on; each further launch takes the parked photons of the one before.  Photons of
        // different depth then never share a wave, so the first bounce keeps its emission-order
        // coherence (sorting the parked photons by origin cell and direction octant was tried as
        // well: it cost what it gained).  Photons beyond the queues' capacity, or deeper than
        // kQueueLevels, are followed to their end in their lanes as in a single launch — which is
        // also what small passes do, where the extra launches cost more than they save.  The count
        // of parked photons is read back between launches: such a pass is synchronous up to its
        // last launch.
        // (RT_AMD_QUEUE_MIN and RT_AMD_QUEUE_CAPACITY override the two limits: tests use them to
        // take small passes through the queues and through their overflow)
        uint64_t minimumWork = kQueueMinimumWork, capacityLimit = kQueueCapacity;
        if (const char* v = std::getenv("RT_AMD_QUEUE_MIN")) minimumWork = std::strtoull(v, nullptr, 10);
        if (const char* v = std::getenv("RT_AMD_QUEUE_CAPACITY")) capacityLimit = std::max<uint64_t>(1, std::strtoull(v, nullptr, 10));
        const bool bounceLevels = args.totalWork >= minimumWork;
        const int levels = bounceLevels ? std::min(kQueueLevels, std::max(0, tracer->settings.bounce_threshold)) : 0;
        const uint64_t capacity = levels > 0 ? std::min<uint64_t>(args.totalWork, capacityLimit) : 0;
        if (capacity > 0 && tracer->photonQueue[0].count < capacity)
        {
            tracer->photonQueue[0].allocate(capacity);
            tracer->photonQueue[1].allocate(capacity);
        }
        // Parked photons are re-ordered by a key of their ray before the next launch (RT_AMD_LEVEL_SORT=0
        // turns that off, any other value is the key's layout as rayKey reads it; RT_AMD_LEVEL_SORT_MIN=<photons>
        // is the smallest level worth a sort).
        int keyMode = 5 | (7 << 4);  // 32 cells per axis above 7 x 7 sectors per face (measured best on MirrorKnot4k)
        uint64_t sortMinimum = 1ull << 20;
        if (const char* v = std::getenv("RT_AMD_LEVEL_SORT")) keyMode = static_cast<int>(std::strtol(v, nullptr, 0));
        if (const char* v = std::getenv("RT_AMD_LEVEL_SORT_MIN")) sortMinimum = std::strtoull(v, nullptr, 10);
        const int originBits = keyMode & 15, sectorCells = (keyMode >> 4) & 15;
        int keyBits = 3 * originBits;
        while (sectorCells > 0 && (1 << (keyBits - 3 * originBits)) < 6 * sectorCells * sectorCells) ++keyBits;
        const bool levelSort = levels > 0 && keyMode > 0 && originBits <= 6 && keyBits >= 1 && keyBits <= 32 && capacity < (1ull << 31);
        if (levelSort && tracer->queueKeys.count < capacity)
        {
            tracer->queueKeys.allocate(capacity);
            tracer->queueKeysSorted.allocate(capacity);
            tracer->queueOrder.allocate(capacity);
            tracer->queueIota.allocate(capacity);
            tracer->queueSpare.allocate(capacity);
            rtd::launchIota(tracer->queueIota.ptr, capacity, tracer->stream);
        }
        if (levels > 0 && tracer->queueCounts.count == 0)
        {
            tracer->queueCounts.allocate(static_cast<size_t>(2 * (kQueueLevels + 1)));
        }
        if (levels > 0)
        {
            tracer->queueCounts.zero(tracer->stream);
        }
        uint64_t parked = 0;  // photons in the queue the next launch reads
        bool sortedLevel = false;  // ... in the order of queueOrder
        const bool verbose = std::getenv("RT_AMD_VERBOSE") != nullptr;
        for (int level = 0; level <= levels; ++level)
        {
            rtd::TraceArgs launch = args;
            // per level: [2 * level] = slots claimed in the queue this level fills, [2 * level + 1] = its work counter
            unsigned long long* counts = tracer->queueCounts.ptr;
            uint64_t work = args.totalWork;
            if (level > 0)
            {
                launch.queueIn = tracer->photonQueue[(level - 1) & 1].ptr;
                launch.queueInCount = parked;
                launch.workCounter = counts + 2 * level + 1;
                launch.emissionOrder = nullptr;
                work = parked;
            }
            if (level < levels)
            {
                launch.queueOut = tracer->photonQueue[level & 1].ptr;
                launch.queueOutCount = counts + 2 * level;
                launch.queueOutCapacity = capacity;
                if (levelSort)
                {
                    launch.queueOutKeys = tracer->queueKeys.ptr;
                    launch.keyMode = keyMode;
                    launch.keyScale = tracer->keyScale;
                    std::memcpy(launch.keyOrigin, tracer->keyOrigin, sizeof(launch.keyOrigin));
                }
            }
            if (level > 0 && sortedLevel)
            {
                launch.queueInOrder = tracer->queueOrder.ptr;
            }
            // Runs per wave.  Where photons stay in their lanes for several bounces (no bounce levels, or
            // the launch after the last level), lanes refill one by one and a wave that changes runs often is
            // a wave whose lanes stand in different corners of the scene: four long runs per wave.  A launch
            // whose photons all make one bounce and leave can take shorter runs, so that the waves end
            // together: sixteen per wave for parked photons (their bounces are the expensive, divergent
            // ones: MirrorKnot4k 24.8 -> 20.7 ms), eight for the emission of fewer than 16 Mi photons (short runs
            // cost a cheap scene more than balance gains it: CornellBox with its 10 M prefers four), sixteen
            // for larger ones (MirrorKnot4k's 20 M: 21.1 -> 20.7 ms, a 25 M share of Bombardment 3.18 -> 3.06).
            // (RT_AMD_RUNS_PER_WAVE overrides.)
            const bool oneTrip = level < levels || (levels > 0 && levels >= tracer->settings.bounce_threshold);
            const int runsPerWave = !oneTrip ? 4 : ((level == 0 && work < (16ull << 20)) ? 8 : 16);
            const int blocks = scheduleLaunch(launch, work, residentBlocks, level == 0 || sortedLevel, runsPerWave);
            if (level > 0) launch.claimStride = 1;
            const auto levelStart = std::chrono::steady_clock::now();
            check(rtd::launchTrace(launch, blocks, tracer->stream), "trace launch");
            if (level == levels)
            {
                break;
            }
            unsigned long long claimed = 0;
            check(hipMemcpyAsync(&claimed, counts + 2 * level, sizeof(claimed), hipMemcpyDeviceToHost, tracer->stream),
                  "hipMemcpyAsync queue count");
            check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
            if (verbose)
            {
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - levelStart).count();
                std::fprintf(stderr, "[rt_amd] bounce level %d: %llu photons in, %llu go on, %.3f ms\n", level,
                             static_cast<unsigned long long>(work), claimed, ms);
            }
            parked = std::min<uint64_t>(claimed, capacity);
            if (parked == 0)
            {
                break;
            }
            sortedLevel = levelSort && parked >= sortMinimum;
            // (RT_AMD_LEVEL_SORTER=library: the library's radix sort whatever the size)
            const char* levelSorter = std::getenv("RT_AMD_LEVEL_SORTER");
            if (sortedLevel && parked >= kOwnLevelSortMinimum && !(levelSorter != nullptr && std::strcmp(levelSorter, "library") == 0))
            {
                // Stable passes of our own over the key's bits, eight at a time (rt_order.h): they move the pairs
                // at twice the library's rate.  The pairs go back and forth between two sets of lists; the first
                // pass takes the photons' places in the queue as its ids and is aimed so that the last one ends
                // in queueOrder.
                const uint64_t entries = rtd::digitMatrixEntries(parked);
                if (tracer->orderCounts.count < entries)
                {
                    tracer->orderCounts.allocate(entries);
                    tracer->orderOffsets.allocate(entries);
                }
                size_t bytes = 0;
                check(rtd::stableDigitPass(parked, 0, nullptr, &bytes, nullptr, nullptr, tracer->orderCounts.ptr, tracer->orderOffsets.ptr,
                                           nullptr, nullptr, tracer->stream),
                      "scan size query");
                if (tracer->sortScratch.count < bytes) tracer->sortScratch.allocate(bytes);
                const int passes = (keyBits + 7) / 8;
                uint32_t* keyLists[2] = {tracer->queueKeys.ptr, tracer->queueKeysSorted.ptr};
                uint32_t* idLists[2] = {tracer->queueOrder.ptr, tracer->queueSpare.ptr};
                int idsAt = passes % 2 == 1 ? 1 : 0;  // the list the NEXT pass reads its ids from (none for the first)
                for (int pass = 0; pass < passes; ++pass)
                {
                    check(rtd::stableDigitPass(parked, 8 * pass, tracer->sortScratch.ptr, &bytes, keyLists[pass & 1],
                                               pass == 0 ? nullptr : idLists[idsAt], tracer->orderCounts.ptr, tracer->orderOffsets.ptr,
                                               keyLists[(pass & 1) ^ 1], idLists[idsAt ^ 1], tracer->stream),
                          "parked photon sort");
                    idsAt ^= 1;
                }
            }
            else if (sortedLevel)
            {
                size_t bytes = 0;
                check(rtd::sortPairs32(nullptr, &bytes, tracer->queueKeys.ptr, tracer->queueKeysSorted.ptr, tracer->queueIota.ptr,
                                       tracer->queueOrder.ptr, parked, keyBits, tracer->stream),
                      "sort size query");
                if (tracer->sortScratch.count < bytes) tracer->sortScratch.allocate(bytes);
                check(rtd::sortPairs32(tracer->sortScratch.ptr, &bytes, tracer->queueKeys.ptr, tracer->queueKeysSorted.ptr,
                                       tracer->queueIota.ptr, tracer->queueOrder.ptr, parked, keyBits, tracer->stream),
                      "parked photon sort");
            }
        }
        check(hipEventRecord(tracer->stageEvents[2], tracer->stream), "hipEventRecord");
        tracer->timed = true;
    });
}

uint64_t rt_trace_static_max_photons(void) { return kStaticMaxIds; }

void rt_debug_device_allocations(uint64_t* allocations, uint64_t* releases)
{
    if (allocations != nullptr) *allocations = g_deviceAllocations.load(std::memory_order_relaxed);
    if (releases != nullptr) *releases = g_deviceReleases.load(std::memory_order_relaxed);
}

int32_t rt_debug_device_memory(int32_t device, uint64_t* free_bytes, uint64_t* total_bytes)
{
    return guarded([&]() {
        check(hipSetDevice(device), "hipSetDevice");
        size_t freeNow = 0, total = 0;
        check(hipMemGetInfo(&freeNow, &total), "hipMemGetInfo");
        if (free_bytes != nullptr) *free_bytes = freeNow;
        if (total_bytes != nullptr) *total_bytes = total;
    });
}

int32_t rt_trace_static(rt_tracer* tracer, const rt_trace_params* params)
{
    return guarded([&]() {
        require(tracer != nullptr && params != nullptr, "null argument");
        activate(tracer);
        tracer->storeVersion += 1;
        uint64_t first = 0, count = 0;
        resolveRange(tracer, params, &first, &count);
        requireSink(tracer, params);
        require(params->record_photons == 0, "rt_trace_static keeps no per-photon records (record_photons must be 0)");
        rtd::TraceArgs args = passArguments(tracer, params, first, count);
        if (args.totalWork > kStaticMaxIds)
        {
            throw ApiError(RT_ERR_ARGUMENT, "rt_trace_static: " + std::to_string(args.totalWork) +
                                                " photons (photon_count x lights) exceed the static scratch's " +
                                                std::to_string(kStaticMaxIds) + "; split the range or use rt_trace");
        }
        // The sort picks its algorithm (and with it its work space) by the number of pairs and does not check
        // the size it is handed: ask what THIS pass needs (a host-side query, nothing is launched) as well as
        // what the largest one would.
        size_t needed = 0;
        if (args.totalWork > 0)
        {
            check(rtd::sortEmissionPairs(nullptr, &needed, nullptr, nullptr, nullptr, nullptr, args.totalWork,
                                         static_cast<int>(std::max<int64_t>(1, tracer->lightCount)), 0, nullptr),
                  "sort size query");
        }
        needed = std::max(needed, tracer->staticSortBytes);
        if (needed > kStaticSortBytes)
        {
            throw ApiError(RT_ERR_STATE, "rt_trace_static: the radix sort asks for " + std::to_string(needed) +
                                             " bytes of work space, the static scratch reserves " +
                                             std::to_string(kStaticSortBytes));
        }
        if (params->sink_mode == RT_SINK_PROBE) beginStorePass(tracer, args);  // (allocates nothing)
        tracer->recordPhotons = 0;  // the buffers of an earlier rt_trace stay allocated and are not read
        tracer->timed = false;
        tracer->orderedLength = 0;  // this pass's order lives in the static array
        if (args.totalWork == 0)
        {
            return;
        }
        // the device's copy of `scratch`: work counter, four id / key lists, the sort's work space
        unsigned char* base = tracer->staticScratch;
        unsigned long long* workCounter = reinterpret_cast<unsigned long long*>(base);
        uint32_t* keysIn = reinterpret_cast<uint32_t*>(base + kStaticHeaderBytes);
        uint32_t* keysOut = reinterpret_cast<uint32_t*>(base + kStaticHeaderBytes + kStaticListBytes);
        uint32_t* idsIn = reinterpret_cast<uint32_t*>(base + kStaticHeaderBytes + 2 * kStaticListBytes);
        uint32_t* idsOut = reinterpret_cast<uint32_t*>(base + kStaticHeaderBytes + 3 * kStaticListBytes);
        unsigned char* sortSpace = base + kStaticHeaderBytes + 4 * kStaticListBytes;
        // the array keeps the last call's contents: the counter has to start at zero, the lists are
        // written before they are read
        check(hipMemsetAsync(workCounter, 0, kStaticHeaderBytes, tracer->stream), "hipMemsetAsync");
        args.workCounter = workCounter;
        rtd::launchEmissionKeys(args, keysIn, idsIn, tracer->stream);
        size_t bytes = kStaticSortBytes;
        check(rtd::sortEmissionPairs(sortSpace, &bytes, keysIn, keysOut, idsIn, idsOut, args.totalWork,
                                     static_cast<int>(tracer->lightCount), 0, tracer->stream),
              "emission sort");
        check(hipGetLastError(), "emission sort");
        args.emissionOrder = idsOut;
        const int residentBlocks = tracer->computeUnits * rtd::traceBlocksPerCu(args);
        const int blocks = scheduleLaunch(args, args.totalWork, residentBlocks, true, 4);
        check(rtd::launchTrace(args, blocks, tracer->stream), "trace launch");
    });
}

int32_t rt_read_emission_order(rt_tracer* tracer, uint32_t* ids, uint64_t capacity, uint64_t* count)
{
    return guarded([&]() {
        require(tracer != nullptr && count != nullptr && (ids != nullptr || capacity == 0), "null argument");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        *count = tracer->orderedLength;
        const uint64_t n = std::min<uint64_t>(capacity, tracer->orderedLength);
        if (n > 0)
        {
            check(hipMemcpy(ids, tracer->orderList, n * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy D2H");
        }
    });
}

int32_t rt_read_timings(rt_tracer* tracer, float* sort_ms, float* trace_ms)
{
    return guarded([&]() {
        require(tracer != nullptr && sort_ms != nullptr && trace_ms != nullptr, "null argument");
        if (!tracer->timed)
        {
            throw ApiError(RT_ERR_STATE, "rt_read_timings needs a completed rt_trace");
        }
        activate(tracer);
        check(hipEventSynchronize(tracer->stageEvents[2]), "hipEventSynchronize");
        check(hipEventElapsedTime(sort_ms, tracer->stageEvents[0], tracer->stageEvents[1]), "hipEventElapsedTime");
        check(hipEventElapsedTime(trace_ms, tracer->stageEvents[1], tracer->stageEvents[2]), "hipEventElapsedTime");
    });
}

int32_t rt_read_setup_timings(rt_tracer* tracer, float* probe_pass_ms, float* index_build_ms, float* filter_build_ms)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        float* out[3] = {probe_pass_ms, index_build_ms, filter_build_ms};
        for (int kind = 0; kind < 3; ++kind)
        {
            float total = 0.0f;
            for (const rt_tracer::TimedSpan& span : tracer->setupSpans[kind])
            {
                float ms = 0.0f;
                check(hipEventElapsedTime(&ms, span.begin, span.end), "hipEventElapsedTime");
                total += ms;
            }
            if (out[kind] != nullptr) *out[kind] = total;
        }
    });
}

int32_t rt_forget_gather_points(rt_tracer* tracer)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        for (auto& cache : tracer->gatherCache)
        {
            cache->valid = false;
        }
        clearSetupSpans(tracer, 0);
    });
}

int32_t rt_synchronize(rt_tracer* tracer)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
    });
}

int32_t rt_read_counters(rt_tracer* tracer, rt_counters* out)
{
    return guarded([&]() {
        require(tracer != nullptr && out != nullptr, "null argument");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        rtd::DeviceCounters dc{};
        unsigned long long scalars[4] = {0, 0, 0, 0};
        check(hipMemcpy(&dc, tracer->counters.ptr, sizeof(dc), hipMemcpyDeviceToHost), "hipMemcpy counters");
        check(hipMemcpy(scalars, tracer->scalars.ptr, sizeof(scalars), hipMemcpyDeviceToHost), "hipMemcpy scalars");
        std::memset(out, 0, sizeof(*out));
        out->photons = dc.photons;
        out->casts = dc.casts;
        out->bounces = dc.bounces;
        out->deposits = dc.deposits;
        out->kept = dc.kept;
        out->culled = dc.culled;
        out->store_attempted = scalars[1];
        out->store_size = std::min<uint64_t>(scalars[1], tracer->storeCapacity);
        out->store_dropped = out->store_attempted - out->store_size;
        out->splat_total = dc.splatTotal;
        out->splat_radius_clamped = dc.splatRadiusClamped;
        out->splat_luminance_clamped = dc.splatLuminanceClamped;
        out->density_cells = dc.densityCells;
        out->density_deposits = dc.densityDeposits - dc.densityOverflow;
        out->density_dropped = dc.densityOverflow;
        for (int k = 0; k < 3; ++k)
        {
            out->deposited_power[k] = dc.power[k];
        }
    });
}

extern "C++"
{
namespace
{

// RCCL, loaded when a collective is first asked for (the library itself does not link it: a tracer on
// one GPU needs none of it).  The three entry points used, by their C signatures (rccl.h).
struct Rccl
{
    typedef int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
    typedef int (*CommInitAll)(void**, int, const int*);
    typedef int (*CommDestroy)(void*);
    typedef const char* (*ErrorString)(int);
    AllReduce allReduce = nullptr;
    CommInitAll commInitAll = nullptr;
    CommDestroy commDestroy = nullptr;
    ErrorString errorString = nullptr;
    static constexpr int kUint64 = 5, kFloat32 = 7, kFloat64 = 8, kSum = 0;  // ncclDataType_t, ncclRedOp_t
};

const Rccl& rccl()
{
    static Rccl lib;
    static std::once_flag once;
    std::call_once(once, []() {
        void* handle = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (handle == nullptr) handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (handle == nullptr) return;
        lib.allReduce = reinterpret_cast<Rccl::AllReduce>(dlsym(handle, "ncclAllReduce"));
        lib.commInitAll = reinterpret_cast<Rccl::CommInitAll>(dlsym(handle, "ncclCommInitAll"));
        lib.commDestroy = reinterpret_cast<Rccl::CommDestroy>(dlsym(handle, "ncclCommDestroy"));
        lib.errorString = reinterpret_cast<Rccl::ErrorString>(dlsym(handle, "ncclGetErrorString"));
    });
    if (lib.allReduce == nullptr || lib.commInitAll == nullptr || lib.commDestroy == nullptr)
    {
        throw ApiError(RT_ERR_STATE, "RCCL (librccl.so) cannot be loaded");
    }
    return lib;
}

void checkRccl(int code, const char* what)
{
    if (code != 0)
    {
        const Rccl& lib = rccl();
        throw ApiError(RT_ERR_DEVICE, std::string(what) + ": " + (lib.errorString ? lib.errorString(code) : "RCCL error"));
    }
}

}  // namespace
}  // extern "C++"

int32_t rt_allreduce(rt_tracer* tracer, void* rccl_comm, rt_counters* totals)
{
    return guarded([&]() {
        require(tracer != nullptr && rccl_comm != nullptr, "null argument");
        const Rccl& lib = rccl();
        activate(tracer);
        for (const rtd::Camera& camera : tracer->cameraHost)
        {
            const size_t floats = static_cast<size_t>(camera.width) * static_cast<size_t>(camera.height) * 3;
            if (floats == 0 || camera.buffer == nullptr) continue;
            checkRccl(lib.allReduce(camera.buffer, camera.buffer, floats, Rccl::kFloat32, Rccl::kSum, rccl_comm, tracer->stream),
                      "ncclAllReduce camera buffer");
        }
        // The counters as rt_read_counters reports them, summed: fifteen integers and three doubles.  They
        // are reduced whether or not the caller wants them: the ranks of a communicator must issue the same
        // collectives in the same order, whatever each of them passed for `totals`.
        rt_counters mine{};
        const int32_t code = rt_read_counters(tracer, &mine);
        if (code != RT_OK) throw ApiError(code, rt_last_error());
        constexpr size_t kIntegers = offsetof(rt_counters, deposited_power) / sizeof(uint64_t);
        static_assert(sizeof(rt_counters) == kIntegers * sizeof(uint64_t) + 3 * sizeof(double), "rt_counters layout");
        unsigned char* staged = tracer->countersStaged.ptr;
        check(hipMemcpyAsync(staged, &mine, sizeof(mine), hipMemcpyHostToDevice, tracer->stream), "hipMemcpyAsync counters");
        checkRccl(lib.allReduce(staged, staged, kIntegers, Rccl::kUint64, Rccl::kSum, rccl_comm, tracer->stream),
                  "ncclAllReduce counters");
        unsigned char* powers = staged + kIntegers * sizeof(uint64_t);
        checkRccl(lib.allReduce(powers, powers, 3, Rccl::kFloat64, Rccl::kSum, rccl_comm, tracer->stream),
                  "ncclAllReduce deposited power");
        rt_counters summed{};
        check(hipMemcpyAsync(&summed, staged, sizeof(summed), hipMemcpyDeviceToHost, tracer->stream), "hipMemcpyAsync totals");
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        if (totals != nullptr) *totals = summed;
    });
}

int32_t rt_comm_init_all(int32_t n, const int32_t* devices, void** comms)
{
    return guarded([&]() {
        require(n >= 1 && devices != nullptr && comms != nullptr, "bad arguments");
        const Rccl& lib = rccl();
        std::vector<int> list(devices, devices + n);
        checkRccl(lib.commInitAll(comms, n, list.data()), "ncclCommInitAll");
    });
}

int32_t rt_comm_destroy(void* rccl_comm)
{
    return guarded([&]() {
        require(rccl_comm != nullptr, "null argument");
        checkRccl(rccl().commDestroy(rccl_comm), "ncclCommDestroy");
    });
}

int32_t rt_read_camera_buffer(rt_tracer* tracer, int32_t camera, float* rgb)
{
    return guarded([&]() {
        require(tracer != nullptr && rgb != nullptr, "null argument");
        require(camera >= 0 && static_cast<size_t>(camera) < tracer->cameraHost.size(), "camera index out of range");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        const rtd::Camera& c = tracer->cameraHost[camera];
        check(hipMemcpy(rgb, c.buffer, static_cast<size_t>(c.width) * c.height * 3 * sizeof(float), hipMemcpyDeviceToHost),
              "hipMemcpy camera buffer");
    });
}

int32_t rt_read_store(rt_tracer* tracer, float* records, uint64_t max_records, uint64_t* count)
{
    return guarded([&]() {
        require(tracer != nullptr && count != nullptr, "null argument");
        activate(tracer);
        settleStore(tracer);
        const uint64_t cursor = storeCursor(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        const uint64_t size = std::min<uint64_t>(cursor, tracer->storeCapacity);
        *count = size;
        const uint64_t n = std::min(size, max_records);
        if (n > 0)
        {
            require(records != nullptr, "records is null");
            check(hipMemcpy(records, tracer->storeRecords.ptr, n * 13 * sizeof(float), hipMemcpyDeviceToHost),
                  "hipMemcpy store");
        }
    });
}

int32_t rt_read_density(rt_tracer* tracer, int64_t* keys, float* power, uint64_t* counts, uint64_t max_cells,
                        uint64_t* count)
{
    return guarded([&]() {
        require(tracer != nullptr && count != nullptr, "null argument");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        *count = 0;
        const size_t slots = tracer->densityKeys.count;
        if (slots == 0)
        {
            return;
        }
        std::vector<unsigned long long> hKeys(slots), hCount(slots);
        std::vector<float> hPower(3 * slots);
        check(hipMemcpy(hKeys.data(), tracer->densityKeys.ptr, slots * sizeof(unsigned long long), hipMemcpyDeviceToHost),
              "hipMemcpy density keys");
        check(hipMemcpy(hCount.data(), tracer->densityCount.ptr, slots * sizeof(unsigned long long), hipMemcpyDeviceToHost),
              "hipMemcpy density counts");
        check(hipMemcpy(hPower.data(), tracer->densityPower.ptr, 3 * slots * sizeof(float), hipMemcpyDeviceToHost),
              "hipMemcpy density power");
        uint64_t n = 0;
        const int64_t lim = 1 << 20;
        for (size_t s = 0; s < slots; ++s)
        {
            if (hKeys[s] == rtd::kEmptyKey)
            {
                continue;
            }
            if (n < max_cells && keys && power && counts)
            {
                keys[3 * n + 0] = static_cast<int64_t>((hKeys[s] >> 42) & 0x1FFFFF) - lim;
                keys[3 * n + 1] = static_cast<int64_t>((hKeys[s] >> 21) & 0x1FFFFF) - lim;
                keys[3 * n + 2] = static_cast<int64_t>(hKeys[s] & 0x1FFFFF) - lim;
                power[3 * n + 0] = hPower[3 * s + 0];
                power[3 * n + 1] = hPower[3 * s + 1];
                power[3 * n + 2] = hPower[3 * s + 2];
                counts[n] = hCount[s];
            }
            ++n;
        }
        *count = n;
    });
}

int32_t rt_read_records(rt_tracer* tracer, uint64_t* recorded, double* emitted, int32_t* casts_valid,
                        double* bounce_f64, int32_t* bounce_i32, float* bounce_power)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        activate(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        if (recorded)
        {
            *recorded = tracer->recordPhotons;
        }
        auto fetch = [&](void* dst, const void* src, size_t bytes) {
            if (dst && bytes > 0)
            {
                check(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost), "hipMemcpy records");
            }
        };
        fetch(emitted, tracer->recEmitted.ptr, tracer->recEmitted.count * sizeof(double));
        fetch(casts_valid, tracer->recCasts.ptr, tracer->recCasts.count * sizeof(int32_t));
        fetch(bounce_f64, tracer->recF64.ptr, tracer->recF64.count * sizeof(double));
        fetch(bounce_i32, tracer->recI32.ptr, tracer->recI32.count * sizeof(int32_t));
        fetch(bounce_power, tracer->recPower.ptr, tracer->recPower.count * sizeof(float));
    });
}

int32_t rt_collect_gather_points(rt_tracer* tracer, int32_t camera, double* rows19, uint64_t capacity,
                                 uint64_t* count, uint64_t* stats3)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        require(camera >= 0 && camera < static_cast<int32_t>(tracer->cameraHost.size()), "camera index out of range");
        require(count != nullptr, "count is null");
        activate(tracer);
        rt_tracer::GatherCache& cache = ensureGatherPoints(tracer, camera);
        *count = cache.total;
        if (stats3 != nullptr)
        {
            std::memcpy(stats3, cache.stats, sizeof(cache.stats));
        }
        if (rows19 != nullptr)
        {
            const std::vector<rtd::GatherPoint>& points = hostGatherPoints(tracer, cache);
            const uint64_t n = std::min<uint64_t>(capacity, points.size());
            for (uint64_t i = 0; i < n; ++i)
            {
                const rtd::GatherPoint& g = points[i];
                double* r = rows19 + 19 * i;
                r[0] = g.px;
                r[1] = g.py;
                for (int k = 0; k < 3; ++k)
                {
                    r[2 + k] = g.position[k];
                    r[5 + k] = g.normal[k];
                    r[8 + k] = g.viewDir[k];
                    r[12 + k] = g.throughput[k];
                }
                r[11] = g.material;
                r[15] = g.unfoldedPathLength;
                r[16] = g.footprintRadius;
                r[17] = g.sampleTime;
                r[18] = g.sampleWeight;
            }
        }
    });
}

int32_t rt_store_append(rt_tracer* tracer, const float* records13, uint64_t n)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        tracer->storeVersion += 1;
        require(n == 0 || records13 != nullptr, "records is null");
        if (!tracer->probesSet)
        {
            throw ApiError(RT_ERR_STATE, "rt_set_probes must be called before rt_store_append");
        }
        if (n == 0)
        {
            return;
        }
        activate(tracer);
        // BounceStore::append n times (reference src/BounceStore.cpp:11-20): the cursor always
        // advances, records beyond the capacity are dropped
        settleStore(tracer);
        unsigned long long cursor = storeCursor(tracer);
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        const uint64_t room = cursor < tracer->storeCapacity ? tracer->storeCapacity - cursor : 0;
        const uint64_t fit = std::min<uint64_t>(room, n);
        if (fit > 0)
        {
            check(hipMemcpy(tracer->storeRecords.ptr + 13 * cursor, records13, fit * 13 * sizeof(float),
                            hipMemcpyHostToDevice),
                  "hipMemcpy store append");
        }
        cursor += n;
        check(hipMemcpy(tracer->scalars.ptr + 1, &cursor, sizeof(cursor), hipMemcpyHostToDevice), "hipMemcpy cursor");
        tracer->storeCursorHost = cursor;
        tracer->storeCursorKnown = true;
    });
}

int32_t rt_gather(rt_tracer* tracer, int32_t camera, double gather_cell_size, double min_gather_radius,
                  uint64_t* stats4)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        require(camera >= 0 && camera < static_cast<int32_t>(tracer->cameraHost.size()), "camera index out of range");
        if (!tracer->probesSet)
        {
            throw ApiError(RT_ERR_STATE, "rt_set_probes must be called before rt_gather");
        }
        activate(tracer);
        rt_tracer::GatherCache& cache = ensureGatherPoints(tracer, camera);
        settleStore(tracer);
        const uint64_t cursor = storeCursor(tracer);
        const uint64_t n = std::min<uint64_t>(cursor, tracer->storeCapacity);
        require(n < (1ull << 31), "bounce store too large for the gather index");
        const double cellSize = gather_cell_size > 0.0 ? gather_cell_size : 1.0;  // reference src/BounceStore.cpp:44

        // index: key every record by its cell, sort (key, record) pairs.  The work space is kept
        // between calls (it only grows): a frame gathers once per camera, a sequence once per frame.
        rt_tracer::GatherSpace& g = tracer->gatherSpace;
        auto& keysIn = g.keysIn;
        auto& keysOut = g.keysOut;
        auto& orderIn = g.orderIn;
        auto& orderOut = g.orderOut;
        auto& scratch = g.scratch;
        auto& sortedStore = g.sortedStore;
        auto& stats = g.stats;
        if (stats.count == 0) stats.allocate(4);  // [0..1] the launch's tallies, [2] records that carry a key
        uint64_t indexed = 0;
        const bool reuse = g.indexedVersion == tracer->storeVersion && g.indexedRecords == n && g.indexedCellSize == cellSize;
        if (reuse)
        {
            indexed = g.indexed;  // another camera of the same frame has built this index already
        }
        else if (n > 0)
        {
            if (keysIn.count < n)
            {
                keysIn.allocate(n);
                keysOut.allocate(n);
                orderIn.allocate(n);
                orderOut.allocate(n);
            }
            // the box of the records' cells first: the keys are relative to it and as wide as it needs
            if (g.cellBox.count == 0) g.cellBox.allocate(6);
            const int32_t emptyBox[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
            check(hipMemcpy(g.cellBox.ptr, emptyBox, sizeof(emptyBox), hipMemcpyHostToDevice), "hipMemcpy cell box");
            rtd::launchStoreBox(tracer->storeRecords.ptr, n, 1.0 / cellSize, g.cellBox.ptr, tracer->stream);
            check(hipGetLastError(), "store box launch");
            check(hipMemcpyAsync(g.box, g.cellBox.ptr, sizeof(g.box), hipMemcpyDeviceToHost, tracer->stream), "hipMemcpyAsync cell box");
            check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
            g.layout = rtd::keyLayoutFor(g.box);
            rtd::launchStoreKeys(tracer->storeRecords.ptr, n, 1.0 / cellSize, g.layout, keysIn.ptr, orderIn.ptr, tracer->stream);
            check(hipGetLastError(), "store keys launch");
            size_t bytes = 0;
            check(rtd::sortStoreKeys(nullptr, &bytes, keysIn.ptr, keysOut.ptr, orderIn.ptr, orderOut.ptr, n, g.layout.bits,
                                     tracer->stream),
                  "sort size query");
            if (scratch.count < bytes) scratch.allocate(bytes > 0 ? bytes : 1);
            check(rtd::sortStoreKeys(scratch.ptr, &bytes, keysIn.ptr, keysOut.ptr, orderIn.ptr, orderOut.ptr, n, g.layout.bits,
                                     tracer->stream),
                  "store sort");
            check(hipGetLastError(), "store sort");
            // records whose cell could not be packed sort to the end; find where they start
            rtd::launchCountIndexed(keysOut.ptr, n, stats.ptr + 2, tracer->stream);
            unsigned long long hostIndexed = 0;
            check(hipMemcpyAsync(&hostIndexed, stats.ptr + 2, sizeof(hostIndexed), hipMemcpyDeviceToHost, tracer->stream),
                  "hipMemcpyAsync indexed");
            check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
            indexed = hostIndexed;
            g.directoryKind = rtd::kDirectoryNone;
            if (indexed > 0)
            {
                if (sortedStore.count < 13 * indexed) sortedStore.allocate(13 * n);
                rtd::launchReorderStore(tracer->storeRecords.ptr, orderOut.ptr, indexed, sortedStore.ptr, tracer->stream);
                check(hipGetLastError(), "store reorder launch");
                // Directory of the sorted keys: where every cell of the box around the records starts.
                // The gather then reads a row's range instead of searching for it.  A box of more cells
                // than the limit (small cells in a large scene) gets a directory of its rows of cells
                // instead, within which the gather searches; one with more rows than that goes without.
                // (RT_AMD_GATHER_DIRECTORY=0 turns it off, any other number is the limit in entries)
                const int32_t* box = g.box;
                uint64_t limit = 1ull << 27;
                if (const char* v = std::getenv("RT_AMD_GATHER_DIRECTORY")) limit = std::strtoull(v, nullptr, 10);
                uint64_t cells = 0;
                bool fits = false;
                g.directoryKind = rtd::kDirectoryNone;
                if (limit > 0 && box[0] <= box[3] && box[1] <= box[4] && box[2] <= box[5])
                {
                    uint64_t dim[3];
                    for (int k = 0; k < 3; ++k)
                    {
                        g.cellMin[k] = box[k];
                        dim[k] = static_cast<uint64_t>(static_cast<int64_t>(box[3 + k]) - box[k] + 1);
                    }
                    if (dim[1] * dim[2] <= limit)  // (each below 2^21: no overflow)
                    {
                        fits = true;
                        g.directoryKind = dim[0] * dim[1] * dim[2] <= limit ? rtd::kDirectoryCells : rtd::kDirectoryRows;
                        if (g.directoryKind == rtd::kDirectoryRows)
                        {
                            dim[0] = 1;
                            g.cellMin[0] = INT32_MIN;
                        }
                        cells = dim[0] * dim[1] * dim[2];
                        for (int k = 0; k < 3; ++k)
                        {
                            g.cellDim[k] = static_cast<int32_t>(dim[k]);
                        }
                    }
                }
                if (fits)
                {
                    if (g.cellStart.count < cells + 1) g.cellStart.allocate(cells + 1);
                    size_t directoryBytes = 0;
                    check(rtd::buildCellDirectory(nullptr, &directoryBytes, keysOut.ptr, indexed, g.layout, g.cellMin, g.cellDim,
                                                  g.cellStart.ptr, tracer->stream),
                          "directory size query");
                    if (g.directoryScratch.count < directoryBytes) g.directoryScratch.allocate(directoryBytes > 0 ? directoryBytes : 1);
                    check(rtd::buildCellDirectory(g.directoryScratch.ptr, &directoryBytes, keysOut.ptr, indexed, g.layout, g.cellMin,
                                                  g.cellDim, g.cellStart.ptr, tracer->stream),
                          "cell directory");
                    check(hipGetLastError(), "cell directory");
                }
            }
        }
        g.indexedVersion = tracer->storeVersion;
        g.indexedRecords = n;
        g.indexedCellSize = cellSize;
        g.indexed = indexed;
        check(hipMemsetAsync(stats.ptr, 0, 2 * sizeof(unsigned long long), tracer->stream), "hipMemsetAsync gather stats");

        rtd::GatherArgs args{};
        args.scene = tracer->view;
        args.store = sortedStore.ptr;
        args.keys = keysOut.ptr;
        args.layout = g.layout;
        args.indexed = indexed;
        args.invCellSize = 1.0 / cellSize;
        args.minGatherRadius = min_gather_radius;
        args.timeHalfWindow = std::max(0.0f, static_cast<float>(tracer->settings.shutter_time));
        args.camera = camera;
        args.pixels = cache.pixels;
        args.counts = cache.counts.ptr;
        args.offsets = cache.byPixel ? nullptr : cache.offsets.ptr;
        args.points = cache.byPixel ? cache.slots.ptr : cache.devicePoints.ptr;
        args.stats = stats.ptr;
        args.directoryKind = indexed > 0 ? g.directoryKind : rtd::kDirectoryNone;
        args.cellStart = args.directoryKind != rtd::kDirectoryNone ? g.cellStart.ptr : nullptr;
        for (int k = 0; k < 3; ++k)
        {
            args.cellMin[k] = g.cellMin[k];
            args.cellDim[k] = g.cellDim[k];
        }
        if (cache.total > 0)
        {
            // One lane per sampled pixel looks up the pixel's rows; a pixel with more records in them than a
            // limit is handed to two waves of its own in a second launch.  Up to a few million gather points
            // the limit is one record — the first launch only sorts out the pixels with nothing to add, and
            // waves do the gathering, 64 records at a time —; a 4K frame has more pixels than that pays for,
            // and each of its pixels few records: there a lane keeps its pixel unless it looks at a crowded
            // spot (an emitter's deposits, a caustic).
            // (RT_AMD_GATHER_HEAVY=<records> sets the limit, 0 turns the hand-over off;
            // RT_AMD_GATHER_LANES=1|8|64 forces one launch with that many lanes per pixel, handing over
            // under the limit when it is 1: for measurements and tests.)
            int lanes = 1;
            uint32_t heavy = cache.total <= (4u << 20) ? 1u : 256u;
            if (const char* v = std::getenv("RT_AMD_GATHER_LANES")) lanes = std::atoi(v);
            if (const char* v = std::getenv("RT_AMD_GATHER_HEAVY")) heavy = static_cast<uint32_t>(std::strtoul(v, nullptr, 10));
            if (lanes < 8 && heavy != 0 && cache.pixels < (1ll << 32))
            {
                if (g.heavyPixels.count < static_cast<size_t>(cache.pixels)) g.heavyPixels.allocate(static_cast<size_t>(cache.pixels));
                check(hipMemsetAsync(stats.ptr + 3, 0, sizeof(unsigned long long), tracer->stream), "hipMemsetAsync heavy count");
                args.heavyThreshold = heavy;
                // (RT_AMD_GATHER_WAVES=1|2|4|8: the waves that share a pixel of the second launch)
                args.heavyWaves = 2;
                if (const char* v = std::getenv("RT_AMD_GATHER_WAVES")) args.heavyWaves = static_cast<uint32_t>(std::max(1, std::atoi(v)));
                args.heavyPixels = g.heavyPixels.ptr;
                args.heavyCount = stats.ptr + 3;
            }
            rtd::launchGather(args, lanes, tracer->computeUnits, tracer->stream);
            check(hipGetLastError(), "gather launch");
            if (args.heavyThreshold != 0 && std::getenv("RT_AMD_VERBOSE") != nullptr)
            {
                unsigned long long handedOver = 0;
                check(hipMemcpy(&handedOver, stats.ptr + 3, sizeof(handedOver), hipMemcpyDeviceToHost), "hipMemcpy heavy count");
                std::fprintf(stderr, "[rt_amd] gather: directory kind %d (%d x %d x %d), %llu of %lld pixels handed over to waves of their own\n",
                             args.directoryKind, args.cellDim[0], args.cellDim[1], args.cellDim[2], handedOver,
                             static_cast<long long>(cache.pixels));
            }
        }
        unsigned long long hostStats[2] = {0, 0};
        check(hipMemcpyAsync(hostStats, stats.ptr, sizeof(hostStats), hipMemcpyDeviceToHost, tracer->stream),
              "hipMemcpyAsync gather stats");
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        if (stats4 != nullptr)
        {
            stats4[0] = cache.total;  // pixelsHit: every record reached a non-delta surface
            stats4[1] = hostStats[0];
            stats4[2] = hostStats[1];
            stats4[3] = n - indexed;
        }
    });
}

int32_t rt_probe_keep_test(rt_tracer* tracer, int64_t n, const double* points, uint8_t* keep)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        require(n >= 0 && (n == 0 || (points != nullptr && keep != nullptr)), "bad arguments");
        if (!tracer->probesSet)
        {
            throw ApiError(RT_ERR_STATE, "rt_set_probes must be called before rt_probe_keep_test");
        }
        if (n == 0)
        {
            return;
        }
        activate(tracer);
        DeviceArray<double> in;
        DeviceArray<uint8_t> out;
        in.allocate(static_cast<size_t>(3 * n));
        out.allocate(static_cast<size_t>(n));
        check(hipMemcpyAsync(in.ptr, points, static_cast<size_t>(3 * n) * sizeof(double), hipMemcpyHostToDevice,
                             tracer->stream),
              "hipMemcpyAsync points");
        rtd::launchProbeTest(tracer->probeView, n, in.ptr, out.ptr, tracer->stream);
        check(hipGetLastError(), "probe test launch");
        check(hipMemcpyAsync(keep, out.ptr, static_cast<size_t>(n), hipMemcpyDeviceToHost, tracer->stream),
              "hipMemcpyAsync keep");
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
    });
}

int32_t rt_cast_rays(rt_tracer* tracer, int64_t n, const double* rays, double self_hit, double* out_f64,
                     int32_t* out_i32)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        require(n >= 0, "negative ray count");
        if (n == 0)
        {
            return;
        }
        require(rays && out_f64 && out_i32, "null buffer");
        activate(tracer);
        DeviceArray<double> dRays, dF;
        DeviceArray<int32_t> dI;
        dRays.allocate(6 * n);
        dF.allocate(7 * n);
        dI.allocate(4 * n);
        check(hipMemcpy(dRays.ptr, rays, 6 * n * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy rays");
        rtd::launchCastRays(tracer->view, n, dRays.ptr, self_hit, dF.ptr, dI.ptr, tracer->stream);
        check(hipGetLastError(), "cast launch");
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        check(hipMemcpy(out_f64, dF.ptr, 7 * n * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy hits");
        check(hipMemcpy(out_i32, dI.ptr, 4 * n * sizeof(int32_t), hipMemcpyDeviceToHost), "hipMemcpy ids");
    });
}

int32_t rt_emit(rt_tracer* tracer, int32_t light, int64_t n, const double* draws5, uint64_t total_photons,
                double* rays, float* colors)
{
    return guarded([&]() {
        require(tracer != nullptr, "tracer is null");
        require(light >= 0 && light < tracer->lightCount, "light index out of range");
        require(n >= 0 && total_photons > 0, "bad photon counts");
        if (n == 0)
        {
            return;
        }
        require(draws5 && rays && colors, "null buffer");
        activate(tracer);
        DeviceArray<double> dDraws, dRays;
        DeviceArray<float> dColors;
        dDraws.allocate(5 * n);
        dRays.allocate(6 * n);
        dColors.allocate(3 * n);
        check(hipMemcpy(dDraws.ptr, draws5, 5 * n * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy draws");
        rtd::launchEmit(tracer->view, light, n, dDraws.ptr, total_photons, dRays.ptr, dColors.ptr, tracer->stream);
        check(hipGetLastError(), "emit launch");
        check(hipStreamSynchronize(tracer->stream), "hipStreamSynchronize");
        check(hipMemcpy(rays, dRays.ptr, 6 * n * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy rays");
        check(hipMemcpy(colors, dColors.ptr, 3 * n * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy colors");
    });
}

#define RT_ROWS(name, kind, inCols, outType, outCols)                                                          \
    int32_t name(int32_t device, int64_t n, const double* in, outType* out)                                     \
    {                                                                                                           \
        return guarded([&]() {                                                                                  \
            runRows(device, n, in, sizeof(double) * (inCols) * n, out, sizeof(outType) * (outCols) * n,         \
                    [&](unsigned char* dIn, unsigned char* dOut) {                                              \
                        rtd::launchIntersect(kind, n, reinterpret_cast<const double*>(dIn), dOut, nullptr);     \
                    });                                                                                         \
        });                                                                                                     \
    }

RT_ROWS(rt_intersect_triangles, 0, 24, double, 8)
RT_ROWS(rt_intersect_bounds, 1, 12, int32_t, 1)
RT_ROWS(rt_intersect_spheres, 2, 10, double, 8)
RT_ROWS(rt_intersect_quads, 3, 15, double, 20)
RT_ROWS(rt_intersect_planes, 4, 10, double, 8)
#undef RT_ROWS

int32_t rt_bsdf_sample(int32_t device, int64_t n, const double* in14, double* out10)
{
    return guarded([&]() {
        runRows(device, n, in14, sizeof(double) * 14 * n, out10, sizeof(double) * 10 * n,
                [&](unsigned char* dIn, unsigned char* dOut) {
                    rtd::launchBsdfSample(n, reinterpret_cast<const double*>(dIn), reinterpret_cast<double*>(dOut), nullptr);
                });
    });
}

int32_t rt_bsdf_evaluate(int32_t device, int64_t n, const double* in14, double* out4)
{
    return guarded([&]() {
        runRows(device, n, in14, sizeof(double) * 14 * n, out4, sizeof(double) * 4 * n,
                [&](unsigned char* dIn, unsigned char* dOut) {
                    rtd::launchBsdfEvaluate(n, reinterpret_cast<const double*>(dIn), reinterpret_cast<double*>(dOut), nullptr);
                });
    });
}

}  // extern "C"
