/*
 * rt_scene.h — plain-data description of a flattened scene.
 *
 * This is the one data format every consumer of a scene shares: the C-ABI
 * library (rt_amd.h), the CPU oracle under oracle/, and the dump harness that
 * links the reference's own objects.  It holds no pointers into C++ objects,
 * only counted arrays of PODs, so it can be built from any host language and
 * written to / read from a file byte-for-byte (rt_scene_io.h).
 *
 * Everything is already resolved to WORLD space: the scene graph
 * (parent/child transform composition, reference src/Object.cpp:21-43) is
 * applied by whoever builds the description, and volume scale is ignored as in
 * the reference (src/Volume.cpp:57-63).  Array order is the reference's object
 * order (key-sorted JSON iteration, src/SceneLoader.cpp:736-743,993-997),
 * because that order decides tie-breaks between volumes (src/Worker.cpp:556-567)
 * and which light emits first (src/Worker.cpp:435-503).
 */
#ifndef RT_SCENE_H
#define RT_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum rt_volume_type
{
    RT_VOLUME_MESH = 0,   /* MeshVolume  -> Tree<Triangle>      (src/MeshVolume.cpp:26-29)  */
    RT_VOLUME_SPHERE = 1, /* SphereVolume -> rayIntersectsSphere (src/SphereVolume.cpp:44-47) */
    RT_VOLUME_QUAD = 2,   /* QuadVolume  -> rayIntersectsQuad   (src/QuadVolume.cpp:67-70)  */
    RT_VOLUME_PLANE = 3   /* PlaneVolume -> rayIntersectsPlane  (src/PlaneVolume.cpp:17-20) */
};

enum rt_material_type
{
    RT_MATERIAL_LAMBERTIAN = 0, /* src/LambertianMaterial.cpp */
    RT_MATERIAL_MIRROR = 1,     /* src/MirrorMaterial.cpp     */
    RT_MATERIAL_DIELECTRIC = 2, /* src/DielectricMaterial.cpp */
    RT_MATERIAL_MICROFACET = 3  /* src/MicrofacetMaterial.cpp */
};

enum rt_light_type
{
    RT_LIGHT_OMNI = 0,     /* src/OmniLight.cpp     */
    RT_LIGHT_SPOT = 1,     /* src/SpotLight.cpp     */
    RT_LIGHT_PARALLEL = 2, /* src/ParallelLight.cpp */
    RT_LIGHT_AREA = 3      /* src/AreaLight.cpp     */
};

enum rt_projection
{
    RT_PROJECTION_PERSPECTIVE = 0,
    RT_PROJECTION_ORTHOGRAPHIC = 1,
    RT_PROJECTION_REALLENS = 2
};

/* One Volume.  `position`/`rotation` are the resolved world transform
 * (quaternion stored x,y,z,w).  `param` depends on `type`:
 *   SPHERE: [0..2] center, [3] radius                      (include/Sphere.h)
 *   QUAD  : [0..2] origin, [3..5] edgeU, [6..8] edgeV,
 *           [9..11] normal, [12..14] invU, [15..17] invV   (src/Quad.cpp:21-48)
 *   PLANE : [0..2] normal, [3] dot                         (include/Plane.h)
 *   MESH  : unused; `mesh` indexes rt_scene.meshes.
 */
typedef struct rt_volume
{
    int32_t type;
    int32_t material;
    int32_t mesh;
    int32_t animation; /* 0 = static; otherwise 1 + index into rt_scene.animations */
    double position[3];
    double rotation[4];
    double param[18];
} rt_volume;

/* A mesh is a contiguous run of triangles in load order.  Triangle t of the
 * scene has vertices tri_pos[9*t .. 9*t+8] = (a, b, c) and per-vertex normals
 * tri_nrm[9*t .. 9*t+8] = (aN, bN, cN)  (include/Triangle.h).  The triangle id
 * reported by the tracer is the index WITHIN its mesh. */
typedef struct rt_mesh
{
    int64_t first_triangle;
    int64_t triangle_count;
    char name[64];
} rt_mesh;

typedef struct rt_material
{
    int32_t type;
    int32_t reserved;
    float color[3]; /* albedo / F0 / tint */
    float reserved2;
    double alpha;   /* microfacet roughness, already floored at 1e-3 (src/MicrofacetMaterial.cpp:132-137) */
    double ior;     /* dielectric index of refraction */
    char name[64];
} rt_material;

typedef struct rt_light
{
    int32_t type;
    int32_t shape;          /* area light: 0 square, 1 disc */
    double position[3];
    double rotation[4];
    float color[3];
    float reserved;
    double luminous_flux;   /* Phi = Light::luminousFlux()  (src/Light.cpp:41-46, src/AreaLight.cpp:67-78) */
    double intensity_candela; /* raw "$brightness"/"$intensityCandela" the flux was derived from */
    double flux_override;   /* area light "$luminousFlux" (0 = none) */
    double angle;           /* spot: clamped full cone angle in radians (src/SpotLight.cpp:23) */
    double inner_radius;    /* omni / spot */
    double max_angle;       /* spot: AngleGenerator::maxAngle = angle / 2 (src/SpotLight.cpp:43) */
    double linearity;       /* AngleGenerator::linearity (always 1.0 from the loader) */
    double radius;          /* parallel light / disc area light */
    double width;           /* square area light */
    double height;
    char name[64];
} rt_light;

typedef struct rt_camera
{
    double position[3];
    double rotation[4];
    double vertical_fov_degrees;
    int32_t width;
    int32_t height;
    int32_t projection;
    int32_t bounce_filter; /* -1 = all (include/Camera.h) */
    int32_t light_filter;  /* -1 = all */
    int32_t animation;     /* 0 = static; otherwise 1 + index into rt_scene.animations */
    double orthographic_height;
    double aperture_radius; /* Camera::effectiveApertureRadius() */
    double focus_distance;
    double f_number;
    double shutter_time;
    double iso;
    char name[64];
} rt_camera;

/* RenderSettings (include/RenderSettings.h) — the keys the photon pass reads. */
typedef struct rt_settings
{
    uint64_t photons_per_light;
    uint64_t bounce_store_capacity;
    uint64_t fetch_size;
    uint64_t worker_count;
    uint64_t probe_sub_sample;
    int32_t image_width;
    int32_t image_height;
    int32_t bounce_threshold;
    int32_t use_probe_gather;
    int32_t deterministic;
    int32_t seeded;            /* 0: reference would seed from random_device */
    uint32_t seed;
    int32_t camera_time_samples;
    double termination_threshold;
    double hash_grid_cell_size;
    double density_cell_scale;
    double splat_min_radius_scale;
    double splat_luminance_clamp;
    double probe_keep_radius_scale;
    double frame_time;
    double shutter_time;
} rt_settings;

/* One keyframe of an animated property (reference include/Property.h:37-51).  Scalar
 * tracks (rotation angle, radians) use component [0]. */
typedef struct rt_keyframe
{
    double time;
    double value[3];
    double in_tangent[3];
    double out_tangent[3];
    int32_t use_auto_tangent; /* Catmull-Rom tangent from the neighbours (Property.h:161-186) */
    int32_t reserved;
} rt_keyframe;

/* A keyframed object: KeyframedAnimationQuery::AnimatedObject (reference
 * include/AnimationQuery.h:107-163).  At time t the object's WORLD transform is
 *   position = position track at t (or constant_position when the track is empty)
 *   rotation = has_rotation ? fromAxisAngle(rotation_axis, angle track at t) * base_rotation
 *                           : base_rotation
 * and replaces the scene-load transform (reference src/Volume.cpp:65-81,
 * src/Camera.cpp:388-402).  Tracks are cubic Hermite splines, held outside their range
 * (reference include/Property.h:95-156). */
typedef struct rt_animation
{
    int32_t has_rotation;
    int32_t reserved;
    double rotation_axis[3];
    double base_rotation[4];
    double constant_position[3];
    int64_t first_position_key, position_key_count; /* into rt_scene.keyframes */
    int64_t first_rotation_key, rotation_key_count;
} rt_animation;

typedef struct rt_scene
{
    int64_t volume_count;
    int64_t mesh_count;
    int64_t triangle_count;
    int64_t material_count;
    int64_t light_count;
    int64_t camera_count;
    rt_volume* volumes;
    rt_mesh* meshes;
    double* tri_pos;  /* 9 * triangle_count */
    double* tri_nrm;  /* 9 * triangle_count */
    rt_material* materials;
    rt_light* lights;
    rt_camera* cameras;
    rt_settings settings;
    int64_t animation_count;
    int64_t keyframe_count;
    rt_animation* animations;
    rt_keyframe* keyframes;
} rt_scene;

#ifdef __cplusplus
}
#endif

#endif /* RT_SCENE_H */
