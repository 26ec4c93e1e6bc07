/*
 * oracle.h — C interface of the CPU oracle (TEST INFRASTRUCTURE, not product).
 *
 * The oracle is a plain CPU restatement of the reference's photon pass.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product library (librt_amd.so) never does.  Its parity with the reference is
 * pinned by tests/test_oracle_golden.py against the fixtures in tests/golden/,
 * which oracle/ref_harness/refdump produced by calling the reference's own code.
 *
 * Two switches select which arithmetic a run uses:
 *   rng_mode  ORC_RNG_MT      one sequential std::mt19937 in the reference's draw
 *                             order (bit-comparable with the reference, 1 worker)
 *             ORC_RNG_COUNTER the per-photon Philox stream of rt_portable.h
 *                             (bit-comparable with the HIP path, any thread count)
 *   math_mode ORC_MATH_LIBM / ORC_MATH_PORTABLE   std::sin... vs rt_portable.h
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_RNG_MT = 0, ORC_RNG_COUNTER = 1 };
enum { ORC_MATH_LIBM = 0, ORC_MATH_PORTABLE = 1 };
enum { ORC_SINK_NONE = 0, ORC_SINK_PROBE = 1, ORC_SINK_SPLAT = 2 };

typedef struct orc_context orc_context;

/* Per-bounce record layout shared with the HIP path's debug records:
 *   f64[10] = position(3) normal(3) distance incoming_direction(3)
 *   i32[8]  = photon, bounce, volume, triangle(-1 if not a mesh), material,
 *             reserved, is_delta, light
 *   f32[3]  = photon power arriving at the bounce                                 */
typedef struct orc_trace_config
{
    int32_t rng_mode;
    int32_t math_mode;
    int32_t sink_mode;
    int32_t threads;
    uint64_t first_photon;   /* per light: trace photon indices [first, first+count) */
    uint64_t photon_count;   /* 0 = settings.photons_per_light - first_photon */
    uint64_t record_photons; /* record emitted rays + bounces of the first M photons traced */
    /* sink A inputs */
    const double* probes;    /* 3 * probe_count */
    uint64_t probe_count;
    double probe_cell_size;
    double keep_radius;
    uint64_t store_capacity;
    /* sink B inputs */
    double min_splat_radius;
    double density_cell_size; /* <= 0: no density grid */
    /* records appended to the store before the photon pass (emitter deposits,
     * reference src/Renderer.cpp:394-418); 13 floats each */
    const float* preload;
    uint64_t preload_count;
} orc_trace_config;

typedef struct orc_trace_result
{
    uint64_t photons;
    uint64_t casts;          /* scene casts, including the one that escapes */
    uint64_t bounces;        /* casts that produced a valid hit */
    uint64_t deposits;       /* non-delta bounces */
    uint64_t kept, culled;   /* sink A keep-test */
    uint64_t store_attempted, store_size, store_dropped;
    uint64_t splat_total, splat_radius_clamped, splat_luminance_clamped;
    uint64_t density_cells, density_deposits;
    double deposited_power[3]; /* sum over all non-delta bounces (ledger) */
    double seconds;
    /* arrays owned by the context, valid until the next trace / destroy */
    uint64_t recorded_photons;
    int32_t max_bounces_per_photon;  /* = bounce_threshold + 1 */
    const double* emitted;       /* [recorded][6] */
    const int32_t* casts_valid;  /* [recorded][2] */
    const double* bounce_f64;    /* [recorded][max][10] */
    const int32_t* bounce_i32;   /* [recorded][max][8] */
    const float* bounce_power;   /* [recorded][max][3] */
    const float* store;          /* [store_size][13] RawBounce records */
    const float* const* splat;   /* per camera, [h][w][3] */
} orc_trace_result;

orc_context* orc_create(const rt_scene* scene);
void orc_destroy(orc_context* ctx);
const char* orc_last_error(void);

int orc_trace(orc_context* ctx, const orc_trace_config* config, orc_trace_result* result);

/* Derived frame quantities exactly as Renderer::renderFrame computes them
 * (src/Renderer.cpp:222-274,303-310): out[0] sceneDepthFootprint, [1] cellSize,
 * [2] minSplatRadius, [3] keepRadius. */
void orc_derive(const orc_context* ctx, double out[4]);

/* Closest hit of world rays against the whole scene (src/Worker.cpp:538-573):
 * rays [n][6]; out_f64 [n][7] = position normal distance; out_i32 [n][4] =
 * hit(0/1), volume, triangle, material.  `self_hit` is the distance floor. */
void orc_cast_rays(const orc_context* ctx, int64_t n, const double* rays, double self_hit,
                   double* out_f64, int32_t* out_i32);

/* Function-level entry points, same array layouts as the refdump `unit` vectors. */
void orc_intersect_triangles(int64_t n, const double* in24, double* out8);
void orc_intersect_bounds(int64_t n, const double* in12, int32_t* out);
void orc_intersect_spheres(int64_t n, const double* in10, double* out8);
void orc_intersect_quads(int64_t n, const double* in15, double* out20);
void orc_intersect_planes(int64_t n, const double* in10, double* out8);
void orc_quaternions(int64_t n, const double* in6, double* out10);
void orc_bsdf_sample(int64_t n, const double* in14, double* out10, int32_t math_mode);
void orc_bsdf_evaluate(int64_t n, const double* in14, double* out4);
/* Emit n photons of light `light` from explicit draws (5 per photon, unused ones
 * ignored): rays [n][6], colors [n][3]. */
void orc_emit(const orc_context* ctx, int32_t light, int64_t n, const double* draws5,
              uint64_t total_photons, int32_t math_mode, double* rays, float* colors);
/* Same with the sequential mt19937 stream seeded with `seed` (reference order). */
void orc_emit_mt(const orc_context* ctx, int32_t light, int64_t n, uint32_t seed,
                 uint64_t total_photons, double* rays, float* colors);

/* Test aid: the camera-side passes (orc_collect_gather_points, orc_gather) visit only the pixels
 * x0 <= x < x1, y0 <= y < y1 of every camera — a crop of a frame too large to restate whole.  In
 * counter mode a pixel's records do not depend on the window (one stream per pixel); in mt19937 mode
 * they do (one sequential stream), so the window is for counter mode. */
void orc_set_pixel_window(orc_context* ctx, int64_t x0, int64_t y0, int64_t x1, int64_t y1);

/* Probe pass (ProbeGather::collectGatherPoints, src/ProbeGather.cpp:239-460) for one
 * camera; rows of 19 doubles as refdump's `probes` mode.  Returns the row count and
 * keeps the array owned by the context. */
int64_t orc_collect_gather_points(orc_context* ctx, int32_t camera, int32_t rng_mode,
                                  int32_t math_mode, const double** rows);

/* ProbeGather::depositEmitters (reference src/ProbeGather.cpp:464-550): the deposits of
 * every emitter patch that pass the probe keep-test, in the reference's order, 13 floats
 * each, at most `capacity`.  stats3 receives patches, generated, kept.  Returns kept. */
int64_t orc_deposit_emitters(const orc_context* ctx, const double* probes, uint64_t probe_count, double probe_cell_size,
                             double keep_radius, double spacing, float* records, uint64_t capacity, uint64_t* stats3);

/* BounceStore::buildIndex + ProbeGather::run (reference src/BounceStore.cpp:42-98,
 * src/ProbeGather.cpp:596-728,865-994) for one camera over the given store records, single
 * threaded: the camera's gather points are produced with the given draw / math modes, then
 * gathered into buffer ([h][w][3], must be zeroed by the caller).  stats4 receives
 * pixelsHit, pixelsGathered, depositsAccum, (unused). */
int32_t orc_gather(orc_context* ctx, int32_t camera, int32_t rng_mode, int32_t math_mode, const float* store,
                   uint64_t store_count, double gather_cell_size, double min_gather_radius, float* buffer,
                   uint64_t* stats4);

/* Renderer::tonemapBufferToImage (reference src/Renderer.cpp:40-76): [h][w][3] float -> [h][w][3] u16, flipped. */
void orc_tonemap(const float* buffer, int32_t width, int32_t height, double saturation_luminance, uint16_t* image);

#ifdef __cplusplus
}
#endif

#endif /* ORACLE_H */
