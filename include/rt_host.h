/*
 * rt_host.h — C entry points of the host-side library (librt_host.so): the scene
 * loader and frame renderer that sit above the tracer's C-ABI (rt_amd.h).
 *
 * These wrap the C++ interfaces that mirror the reference's
 * SceneLoader::loadFromFile (reference include/SceneLoader.h:60) and
 * Renderer::renderFrame (reference include/Renderer.h:143-144) so that a non-C++
 * host (the Python tests and bench.py) can drive them.  Every function returns 0
 * on success and a non-zero code on failure; rth_last_error() gives the text of
 * the exception the C++ layer threw.  No exception crosses this boundary.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

const char* rth_last_error(void);

/* Load a scene JSON (reference key set) and flatten it. */
int rth_scene_load(const char* json_path, rth_scene** out);
/* The flattened description; valid until rth_scene_free. */
const rt_scene* rth_scene_get(const rth_scene* scene);
/* Mutable access to the settings block (tests shrink photon counts etc.). */
rt_settings* rth_scene_settings(rth_scene* scene);
int rth_scene_write(const rth_scene* scene, const char* path);
void rth_scene_free(rth_scene* scene);

/* Frame quantities Renderer::renderFrame derives before the photon pass (reference
 * src/Renderer.cpp:222-274,303-310): out[0] sceneDepthFootprint, [1] density cell
 * size, [2] minimum splat radius, [3] probe keep radius. */
int rth_frame_derive(const rt_scene* scene, double out[4]);

/* Material::sample / evaluate / pdf / isDelta of material `material` of the scene's library
 * (reference include/Material.h:56-101), run through the host classes' virtuals — which evaluate the
 * device's BSDFs (needs a GPU).  sample: in6 = incident, normal; the draws come from a
 * RandomGenerator(seed) (mt19937, reference src/RandomGenerator.cpp:3-17); out9 = valid,
 * direction[3], weight[3], pdf, number of draws consumed.  evaluate: in9 = wi, wo, normal;
 * out5 = f[3], pdf, isDelta. */
int rth_material_sample(const rth_scene* scene, int32_t material, const double in6[6], uint32_t seed, double out9[9]);
int rth_material_evaluate(const rth_scene* scene, int32_t material, const double in9[9], double out5[5]);

/* The CLI's image writer (reference src/PngWriter.cpp, src/PngWriteSession.cpp:118-173):
 * rgb[height][width][3] as a 16-bit RGB PNG with a "Title" text chunk when title is not empty. */
int rth_write_png16(const char* path, const uint16_t* rgb, int32_t width, int32_t height, const char* title);

/* The control flow of a frame whose photons are spread over several GPUs of one process
 * (Renderer::renderFrame under RT_AMD_DEVICES; host/RankFrames.h), run with stand-in shares and a
 * stand-in collective, no GPU: `world` ranks each trace eight pieces of a share, rank `failing_rank`
 * (-1: none) then throws, rank 0 gives the frame up after `cancel_after_pieces` of its pieces (-1: never);
 * the ranks meet; the collective — which nobody leaves before all have entered — is entered only if no
 * share failed.  Receives the ranks that entered the collective, the pieces traced over all ranks and the
 * frame's failure text (empty: none).  Returns 0, or 2 when the collective was entered by some ranks
 * only, i.e. a real one would have hung.  The reference rethrows a worker's exception after stopping the
 * other workers (reference src/Renderer.cpp:511-523,584-595); this is how that is tested here. */
int rth_test_rank_frames(int32_t world, int32_t failing_rank, int32_t cancel_after_pieces, int32_t* entered_collective,
                         int32_t* pieces_traced, char* text, int32_t text_capacity);

#ifdef __cplusplus
}
#endif

#endif /* RT_HOST_H */
