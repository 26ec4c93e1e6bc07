/*
 * rt_amd.h — C-ABI of the MI355X photon tracer (librt_amd.so).
 *
 * The reference has no FFI layer: its photon pass is the C++ `Worker` thread loop
 * wired up by `Renderer::renderFrame` (reference src/Renderer.cpp:421-464) and read
 * back through `WorkerDebug::*` counters, `Buffer` and `BounceStore`.  This header is
 * the boundary a replacement for that loop exposes: plain pointers and sizes, integer
 * status codes, no exceptions and no C++/torch types.  Each entry point names the
 * reference interface it stands in for.
 *
 * Conventions
 *   - every function returns RT_OK (0) or an RT_ERR_* code; rt_last_error() returns
 *     the message of the calling thread's last failure.
 *   - the caller owns every host pointer it passes; a handle owns its device memory
 *     unless a buffer was bound with rt_bind_*.
 *   - a handle is bound to one GPU and is not thread-safe; use one host thread (or
 *     one process) per GPU.
 *   - all kernels are launched on the handle's stream (rt_set_stream; default: the
 *     null stream); rt_trace returns once its last launch is queued — at once for a pass
 *     below 4 Mi photons, after the launches of the earlier bounce levels have finished for
 *     a larger one (see rt_trace) —, every rt_read_* synchronises first.
 *   - there is NO CPU fallback: without a usable GPU every call that computes fails
 *     with RT_ERR_DEVICE.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

enum
{
    RT_OK = 0,
    RT_ERR_ARGUMENT = 1, /* bad pointer / size / index / unsupported scene feature */
    RT_ERR_DEVICE = 2,   /* HIP runtime failure (message carries hipGetErrorString) */
    RT_ERR_STATE = 3     /* call made in the wrong order (e.g. trace before sink set-up) */
};

enum rt_sink_mode
{
    RT_SINK_NONE = 0,  /* transport only (counters + records) */
    RT_SINK_PROBE = 1, /* default mode: probe keep-test + raw bounce store
                          (reference src/Worker.cpp:578-607) */
    RT_SINK_SPLAT = 2  /* legacy mode: density grid + direct camera splat
                          (reference src/Worker.cpp:608-625) */
};

typedef struct rt_tracer rt_tracer;

/* Counters of one trace, the WorkerDebug::* set (reference include/Worker.h:220-246)
 * plus the transport totals the benchmark metric is made of. */
typedef struct rt_counters
{
    uint64_t photons;                 /* photons emitted and traced */
    uint64_t casts;                   /* scene casts, including the one that escapes */
    uint64_t bounces;                 /* casts that produced a valid hit (> 1e-4)     */
    uint64_t deposits;                /* non-delta bounces                            */
    uint64_t kept;                    /* WorkerDebug::bounceKept()                    */
    uint64_t culled;                  /* WorkerDebug::bounceCulled()                  */
    uint64_t store_attempted;         /* BounceStore::attemptedCount()                */
    uint64_t store_size;              /* BounceStore::size()                          */
    uint64_t store_dropped;           /* BounceStore::droppedCount()                  */
    uint64_t splat_total;             /* WorkerDebug::splatTotal()                    */
    uint64_t splat_radius_clamped;    /* WorkerDebug::splatRadiusClamped()            */
    uint64_t splat_luminance_clamped; /* WorkerDebug::splatLuminanceClamped()         */
    uint64_t density_cells;           /* DensityGrid::cellCount()                     */
    uint64_t density_deposits;        /* DensityGrid::depositCount()                  */
    uint64_t density_dropped;         /* deposits the grid could not take (table full, or a cell beyond +-2^20
                                         cells from the origin); the reference's grid has no limit, so this is 0 there */
    double deposited_power[3];        /* sum of photon power over all non-delta bounces */
} rt_counters;

/* rt_trace_params::flags.  RT_TRACE_REUSE_ORDER: the order in which the last pass took this very
 * range of photons (a schedule computed from the seed and the ids, see rt_trace) may be used
 * again instead of being worked out anew: for the frames of a sequence.  Results are the same. */
#define RT_TRACE_REUSE_ORDER 1

typedef struct rt_trace_params
{
    int32_t sink_mode;       /* rt_sink_mode */
    int32_t flags;           /* RT_TRACE_... or 0 */
    uint64_t first_photon;   /* per light: trace photon indices [first, first+count)   */
    uint64_t photon_count;   /* 0 = settings.photons_per_light - first_photon           */
    uint64_t record_photons; /* keep per-bounce records of the first M photons traced   */
} rt_trace_params;

const char* rt_last_error(void);

/* sizeof() of the ABI structs as this library was compiled, for binding checks:
 * which = 0 rt_scene, 1 rt_volume, 2 rt_mesh, 3 rt_material, 4 rt_light, 5 rt_camera,
 * 6 rt_settings, 7 rt_counters, 8 rt_trace_params. */
uint64_t rt_abi_struct_size(int32_t which);

/* Number of usable GPUs (0 when there is none). */
int32_t rt_device_count(void);

/* Build a tracer for `scene` on GPU `device`: builds each mesh's tree with the
 * reference's median-split algorithm (reference src/Tree.cpp:289-362) to learn every
 * triangle's visiting rank and node box, then a surface-area tree over the same
 * triangles that returns the same closest hit (see rt_check_tree), flattens it, and
 * uploads volumes, materials, lights and cameras.  Stands in for the Worker wiring
 * block of Renderer::renderFrame (reference src/Renderer.cpp:421-459).
 * Environment: RT_AMD_LEAF=<1..16> triangles per leaf (default 2); RT_AMD_VERBOSE=1
 * prints each mesh's node count and depth. */
int32_t rt_create(const rt_scene* scene, int32_t device, rt_tracer** out);
void rt_destroy(rt_tracer* tracer);

/* Host only (no GPU needed): builds both trees of one mesh (tri_pos: 9 doubles per triangle,
 * `count` triangles; Tree's constructor, reference src/Tree.cpp:289-362, and the surface-area
 * tree with `leaf_size` triangles per leaf and boxes grown by `pad`) and verifies what the
 * device walk relies on: both hold every triangle exactly once; child ranges and skip links
 * are consistent; every surface-area box contains its triangles' vertices with `pad` to
 * spare and its children's boxes; ranks are a permutation of the reference's visiting order;
 * every triangle's reference box is the box of the reference node whose page holds it and
 * contains the triangle; the two-box nodes the device walks reach every triangle exactly once
 * and their single-precision boxes contain the tree's.  stats = {reference nodes, reference depth, surface-area nodes,
 * surface-area depth, leaves, largest leaf}.  RT_OK, or RT_ERR_* with rt_last_error(). */
int32_t rt_check_tree(const double* tri_pos, int64_t count, int32_t leaf_size, double pad, int64_t stats[6]);

/* Launch on this HIP stream (a hipStream_t passed as void*; NULL = null stream). */
int32_t rt_set_stream(rt_tracer* tracer, void* hip_stream);

/* Sink A set-up: the ProbeIndex (reference src/ProbeIndex.cpp:5-25) over `probes`
 * [n][3] with cubic cells of `cell_size`, the keep radius, and a BounceStore of
 * `store_capacity` records (reference src/BounceStore.cpp:5-9). */
int32_t rt_set_probes(rt_tracer* tracer, const double* probes, uint64_t n, double cell_size,
                      double keep_radius, uint64_t store_capacity);

/* Sink A set up without leaving the device: the probes are the positions of the probe-pass
 * records of every camera that takes part in the gather (those without a bounce or light
 * filter), as Renderer::renderFrame collects them (reference src/Renderer.cpp:297-371), and
 * the store holds min(max_store_capacity, records_per_probe x probes) records (:382-390; all
 * of max_store_capacity when there is no probe).  Same result as rt_collect_gather_points for
 * each camera followed by rt_set_probes, without the round trip of the records through the host. */
int32_t rt_set_probes_from_cameras(rt_tracer* tracer, double cell_size, double keep_radius, uint64_t records_per_probe,
                                   uint64_t max_store_capacity, uint64_t* probe_count, uint64_t* store_capacity);

/* One rank's share of the BounceStore when a frame's photons are spread over several GPUs: from
 * now on the store takes `max_records` records in all (0, or more than the capacity given to
 * rt_set_probes*: the whole capacity; rt_set_probes* sets it back to the whole).  The reference
 * has ONE store per frame whose capacity is a setting of the scene (reference
 * src/Renderer.cpp:382-390, src/BounceStore.cpp:21-33: an append beyond it is dropped and
 * counted); ranks that each kept a whole capacity would keep up to N times the records of the
 * one-GPU frame, and a frame that overflows would come out that much brighter.  A sharded frame
 * therefore gives rank r the emitter deposits (rank 0 only) plus 1/N of the rest of the capacity.
 * Call with a store that holds no more than `max_records` records. */
int32_t rt_set_store_limit(rt_tracer* tracer, uint64_t max_records);

/* ProbeGather::depositEmitters (reference src/ProbeGather.cpp:464-550) on the device: the lit
 * face of every area light is tiled with cell-centred deposits `deposit_spacing` apart, those
 * within the keep radius of a probe are appended to the store as timeless records carrying the
 * patch's power share.  Call after rt_set_probes* and rt_reset and before rt_trace, so that a
 * store overflow can never drop them (reference src/Renderer.cpp:394-418).  *samples = deposits
 * generated, *kept = deposits that passed the keep test (either may be NULL). */
int32_t rt_deposit_emitters(rt_tracer* tracer, double deposit_spacing, uint64_t* samples, uint64_t* kept);

/* Sink B set-up: Worker::setMinSplatRadius / setSplatTargets / the DensityGrid
 * (reference src/Renderer.cpp:196-274,445-457).  density_cell_size <= 0 disables the
 * grid; density_capacity is the number of hash slots reserved for occupied cells. */
int32_t rt_set_splat(rt_tracer* tracer, double min_splat_radius, double density_cell_size,
                     uint64_t density_capacity);

/* Use caller-owned DEVICE memory for camera `camera`'s splat buffer ([h][w][3]
 * floats) so that a framework can all-reduce it in place.  NULL returns to the
 * internal buffer. */
int32_t rt_bind_camera_buffer(rt_tracer* tracer, int32_t camera, float* device_rgb);

/* Zero the camera buffers, the bounce store cursor, the density grid, the counters
 * (Buffer::clear, reference src/Buffer.cpp:14-20; WorkerDebug::reset*Counters). */
int32_t rt_reset(rt_tracer* tracer);

/* The photon pass: emit and trace the photon range of every light (reference
 * Worker::exec / processLights / processPhotons, src/Worker.cpp:116-174,429-680).
 * Runs on the handle's stream and accumulates into sinks and counters until rt_reset.
 * The photon ids of the pass are first put in emission order (a schedule only: a photon's
 * path depends on its id, not on when it is traced).  A pass of 4 Mi photons or more is
 * traced one bounce level per launch and the host waits between launches for the count of
 * photons that go on, so rt_trace returns when the last launch has been queued; a smaller
 * pass is one launch and rt_trace returns at once.  rt_synchronize waits for either. */
int32_t rt_trace(rt_tracer* tracer, const rt_trace_params* params);
int32_t rt_synchronize(rt_tracer* tracer);

/* rt_trace for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  It has no counterpart in the reference.  Same
 * pass, same sinks and counters as rt_trace, restricted to what needs no allocation: at most
 * rt_trace_static_max_photons() photon ids per call (photon_count x lights; 1 Mi) — a larger range
 * is refused with RT_ERR_ARGUMENT, split it over several calls —, record_photons == 0, one launch
 * (no bounce-level queues) and no stage timings (rt_read_timings keeps failing with RT_ERR_STATE
 * after it).  The emission-order list and the radix sort's work space live in a 32 MiB array that
 * is part of the library's device code; every GPU the library runs on holds its copy until the
 * process ends.  All tracers of one GPU share that array: calls that overlap on one GPU (two
 * streams, two host threads) race, calls in sequence on one stream do not.  Sinks must have been
 * set up (rt_set_probes / rt_set_splat: those allocate) and reset (rt_reset: does not) before. */
int32_t rt_trace_static(rt_tracer* tracer, const rt_trace_params* params);
uint64_t rt_trace_static_max_photons(void);
/* Device allocations (hipMalloc) and releases (hipFree) the library has made in this process so far;
 * every one of them goes through one wrapper, which counts.  No counterpart in the reference: it is
 * how a caller (and tests/test_zz_no_alloc.py) checks that a call allocated nothing. */
void rt_debug_device_allocations(uint64_t* allocations, uint64_t* releases);
/* hipMemGetInfo of GPU `device`: bytes free and in all (either may be NULL).  Also for that check. */
int32_t rt_debug_device_memory(int32_t device, uint64_t* free_bytes, uint64_t* total_bytes);

/* Device time of the last rt_trace, from HIP events recorded on the tracer's stream around
 * its two stages: the emission-order sort of the photon ids (two digits of the emission cell placed
 * tile by tile and the low bits run by run for a pass of 1 Mi to 256 Mi ids, else keys + radix sort; 0
 * when the pass has 2^31 ids or more and is traced unsorted) and the trace kernel.  The reference reports one wall time for the whole pass
 * (RenderResult::photonPassSeconds, include/Renderer.h:51-103); the split is what a
 * per-kernel roofline needs.  Waits for the pass to finish. */
int32_t rt_read_timings(rt_tracer* tracer, float* sort_ms, float* trace_ms);

/* Device time of the frame's camera-side set-up, which the reference's photonPassSeconds includes
 * (ProbeGather::collectGatherPoints and the ProbeIndex, reference src/Renderer.cpp:297-419): the probe pass
 * of the cameras collected so far (rt_collect_gather_points / rt_set_probes_from_cameras), the build of
 * the probe index by the last rt_set_probes* — the probes keyed by cell, sorted, and the dense array of
 * cells (index_build_ms), then the table of the bricks of cells the keep test looks up (filter_build_ms) —;
 * 0 for what has not run, or ran on the host (the index of probes whose cells do not pack).  From HIP
 * events on the tracer's stream; waits for the stream.  Any pointer may be NULL. */
int32_t rt_read_setup_timings(rt_tracer* tracer, float* probe_pass_ms, float* index_build_ms, float* filter_build_ms);
/* Forget the probe passes kept per camera (rt_collect_gather_points caches them): the next collect runs
 * them again.  The reference runs them once per frame; a tracer serves one frame's scene, so this is
 * for measuring a frame's set-up more than once. */
int32_t rt_forget_gather_points(rt_tracer* tracer);

/* The emission order the last rt_trace worked out (or reused): photon ids of the pass, numbered light
 * by light from 0, in the order the waves claim them.  The reference has no such list (its workers take
 * photons in index order, src/Worker.cpp:429-505); it is exposed so that the schedule can be checked to
 * be a permutation of the pass and ordered by emission cell.  *count receives the length of the list
 * (0: the last pass was traced unsorted); at most `capacity` ids are copied to `ids` (host memory).
 * Waits for the stream. */
int32_t rt_read_emission_order(rt_tracer* tracer, uint32_t* ids, uint64_t capacity, uint64_t* count);

int32_t rt_read_counters(rt_tracer* tracer, rt_counters* out);

/* ---- more than one GPU ----
 * Photons shard by index range (rt_trace_params first_photon / photon_count), one tracer per GPU; what
 * the reference's workers merge by writing into shared buffers (reference src/Renderer.cpp:421-582,
 * src/Buffer.cpp:22-37) is merged here by an RCCL all-reduce.  rt_allreduce sums every camera buffer
 * of the tracer over the ranks of `rccl_comm` (an ncclComm_t passed as void*; the caller made it, with
 * ncclCommInitRank in a process per GPU or with rt_comm_init_all in one process), in place, on the
 * tracer's stream, and waits; `totals` receives the counters summed over the ranks (deposited_power
 * too; store_size is the sum of the ranks' stores) and may be NULL on any rank: the counters are reduced
 * either way, so the sequence of collectives is the same on every rank whatever each passes (cameras in
 * index order, then the fifteen integer counters, then the three powers).  Every rank calls it once per
 * frame, after its gather, with tracers of the same scene.  RCCL (librccl.so) is loaded when first
 * needed: RT_ERR_STATE if it cannot be. */
int32_t rt_allreduce(rt_tracer* tracer, void* rccl_comm, rt_counters* totals);
/* One process, n GPUs: ncclCommInitAll over `devices`; comms[i] belongs to devices[i] and is to be used
 * from its own thread.  rt_comm_destroy releases one of them. */
int32_t rt_comm_init_all(int32_t n, const int32_t* devices, void** comms);
int32_t rt_comm_destroy(void* rccl_comm);
/* Buffer::fetchColor over the whole image: rgb [h][w][3] (reference src/Buffer.cpp:39-53). */
int32_t rt_read_camera_buffer(rt_tracer* tracer, int32_t camera, float* rgb);
/* BounceStore records, 13 floats each (reference include/BounceStore.h:55-113):
 * px py pz ix iy iz nx ny nz time power_r power_g power_b. */
int32_t rt_read_store(rt_tracer* tracer, float* records, uint64_t max_records, uint64_t* count);
/* DensityGrid cells: keys [n][3] int64, power [n][3] float, counts [n] uint64. */
int32_t rt_read_density(rt_tracer* tracer, int64_t* keys, float* power, uint64_t* counts,
                        uint64_t max_cells, uint64_t* count);
/* Per-bounce records of the last trace, M recorded photons, B = bounce_threshold + 1 slots each:
 * emitted [M][6] (origin, direction), casts_valid [M][2] (casts, valid hits),
 * f64 [M][B][10] (hit position 3, normal 3, distance, incoming direction 3),
 * i32 [M][B][8] (photon index, bounce, volume, triangle, material, 0, delta flag, light),
 * power [M][B][3] (the photon's colour at the hit).  Any pointer may be NULL. */
int32_t rt_read_records(rt_tracer* tracer, uint64_t* recorded, double* emitted, int32_t* casts_valid,
                        double* bounce_f64, int32_t* bounce_i32, float* bounce_power);

/* ---- function-level entry points (device-executed; host pointers in and out) ----
 * Same row layouts as the reference-generated vectors under tests/golden/. */

/* The probe pass of one camera: ProbeGather::collectGatherPoints (reference
 * include/ProbeGather.h:96-104, src/ProbeGather.cpp:239-460).  Every `probe_sub_sample`-th
 * pixel is sampled on the device (one thread per pixel; the records come back in the
 * reference's pixel-major order).  *count receives the number of gather points; when
 * rows19 is not NULL the first min(capacity, *count) records are written, 19 doubles each:
 * px, py, position[3], normal[3], viewDir[3], material (-1: emitter face), throughput[3],
 * unfoldedPathLength, footprintRadius, sampleTime, sampleWeight.  stats3 (optional)
 * receives cameraRays, deltaExtensions, misses (reference include/ProbeGather.h:53-64).
 * The result is cached per camera: a size query followed by a fetch runs the pass once. */
int32_t rt_collect_gather_points(rt_tracer* tracer, int32_t camera, double* rows19, uint64_t capacity,
                                 uint64_t* count, uint64_t* stats3);

/* BounceStore::append for n host records of 13 floats (position, incoming direction, normal,
 * time, power; reference include/BounceStore.h:55-113, src/BounceStore.cpp:11-20): the cursor
 * advances by n, records beyond the capacity are dropped and counted.  Called after rt_reset
 * and before rt_trace it places the emitter deposits ahead of the photon deposits, as
 * Renderer::renderFrame does (reference src/Renderer.cpp:394-418). */
int32_t rt_store_append(rt_tracer* tracer, const float* records13, uint64_t n);

/* BounceStore::buildIndex + ProbeGather::run for one camera (reference src/BounceStore.cpp:42-98,
 * src/ProbeGather.cpp:596-728,865-994; call site src/Renderer.cpp:624-641,698-723): indexes the
 * store of the last pass by cells of gather_cell_size, then adds every gather point's density
 * estimate (footprint floored at min_gather_radius, temporal window = the scene's shutter) into the
 * camera buffer.  stats4 (optional) receives pixelsHit, pixelsGathered, depositsAccum
 * (reference include/ProbeGather.h:106-125) and the number of records left out of the index
 * because their cell exceeds +-2^20. */
int32_t rt_gather(rt_tracer* tracer, int32_t camera, double gather_cell_size, double min_gather_radius,
                  uint64_t* stats4);

/* ProbeIndex::anyWithinKeepRadius (reference src/ProbeIndex.cpp:48-83) for n host points
 * (xyz doubles) against the probes given to rt_set_probes; keep[i] = 1 when a probe lies
 * within the keep radius.  Used for the emitter deposits of src/ProbeGather.cpp:520-545. */
int32_t rt_probe_keep_test(rt_tracer* tracer, int64_t n, const double* points, uint8_t* keep);

/* Closest valid hit of world rays against the whole scene (reference
 * src/Worker.cpp:538-573): rays [n][6] -> f64 [n][7] position normal distance,
 * i32 [n][4] hit volume triangle material. */
int32_t rt_cast_rays(rt_tracer* tracer, int64_t n, const double* rays, double self_hit,
                     double* out_f64, int32_t* out_i32);
/* Light::emit from explicit draws, 5 per photon (reference src/OmniLight.cpp:23-44,
 * src/SpotLight.cpp:50-70, src/ParallelLight.cpp:26-47, src/AreaLight.cpp:106-155). */
int32_t rt_emit(rt_tracer* tracer, int32_t light, int64_t n, const double* draws5,
                uint64_t total_photons, double* rays, float* colors);
/* rayIntersectsTriangle / Bounds / Sphere / Quad / Plane (reference src/Ray.cpp:16-375). */
int32_t rt_intersect_triangles(int32_t device, int64_t n, const double* in24, double* out8);
int32_t rt_intersect_bounds(int32_t device, int64_t n, const double* in12, int32_t* out);
int32_t rt_intersect_spheres(int32_t device, int64_t n, const double* in10, double* out8);
int32_t rt_intersect_quads(int32_t device, int64_t n, const double* in15, double* out20);
int32_t rt_intersect_planes(int32_t device, int64_t n, const double* in10, double* out8);
/* Material::sample with explicit draws / Material::evaluate + pdf (reference
 * src/LambertianMaterial.cpp, MirrorMaterial.cpp, DielectricMaterial.cpp,
 * MicrofacetMaterial.cpp): in [n][14] -> out [n][10] / [n][4]. */
int32_t rt_bsdf_sample(int32_t device, int64_t n, const double* in14, double* out10);
int32_t rt_bsdf_evaluate(int32_t device, int64_t n, const double* in14, double* out4);

#ifdef __cplusplus
}
#endif

#endif /* RT_AMD_H */
