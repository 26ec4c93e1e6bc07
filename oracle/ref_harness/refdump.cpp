// refdump — dump tool that links the REFERENCE's own compiled objects.
//
// It exists to pin the CPU oracle (oracle/oracle.cpp) and the HIP path to the
// reference: it feeds the reference's public functions (rayIntersects*, Material::
// sample/evaluate, Light::emit, Volume::castRayAt, Worker::splatToCamera,
// ProbeGather::collectGatherPoints, Renderer::renderFrame) with inputs built from
// an rt_scene file or from its own seeded generators, and writes inputs + outputs
// as .npy arrays.  None of the reference's code is restated here; this file only
// calls it.  Outputs become the fixtures under tests/golden/ (see oracle/make_golden.py).
//
// Modes:
//   refdump unit   <outdir>                     function-level vectors
//   refdump emit   <scene> <outdir> <n> <seed>  Light::emit per light
//   refdump walk   <scene> <outdir> [record]    seeded 1-worker photon walk, per-bounce data
//   refdump render <scene> <outdir>             Renderer::renderFrame, store/buffers/counters
//   refdump probes <scene> <outdir>             ProbeGather::collectGatherPoints per camera
//   refdump bench  <scene> [workers]            time renderFrame, print one JSON line

#include "rt_scene.h"
#include "rt_scene_io.h"
#include "npy.h"

#include "AreaLight.h"
#include "BounceStore.h"
#include "Camera.h"
#include "DielectricMaterial.h"
#include "LambertianMaterial.h"
#include "LightQueue.h"
#include "MeshVolume.h"
#include "MicrofacetMaterial.h"
#include "MirrorMaterial.h"
#include "OmniLight.h"
#include "ParallelLight.h"
#include "PlaneVolume.h"
#include "ProbeGather.h"
#include "ProbeIndex.h"
#include "QuadVolume.h"
#include "Ray.h"
#include "Renderer.h"
#include "SphereVolume.h"
#include "SpotLight.h"
#include "Utility.h"
#include "Worker.h"

#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <random>

// MeshLibrary.cpp references ObjReader::loadMeshes; the harness never loads OBJ
// files (meshes come from the rt_scene file), so this loader loads nothing.
namespace ObjReader
{
std::vector<std::shared_ptr<Mesh>> loadMeshes(const std::filesystem::path&)
{
    return {};
}
}

namespace
{

Vector vec(const double* p) { return Vector{p[0], p[1], p[2]}; }
Quaternion quat(const double* p) { return Quaternion{p[0], p[1], p[2], p[3]}; }
void push3(std::vector<double>& out, const Vector& v)
{
    out.push_back(v.x);
    out.push_back(v.y);
    out.push_back(v.z);
}

struct BuiltScene
{
    LoadedScene scene;
    std::vector<std::shared_ptr<Volume>> volumes;
    std::vector<std::shared_ptr<Light>> lights;
    std::vector<std::vector<Triangle>> meshTriangles;  // per rt_mesh, load order
    std::vector<int> volumeMesh;                       // rt_volume.mesh per volume
};

std::shared_ptr<Material> makeMaterial(const rt_material& m)
{
    const Color color{m.color[0], m.color[1], m.color[2]};
    switch (m.type)
    {
        case RT_MATERIAL_LAMBERTIAN: return std::make_shared<LambertianMaterial>(m.name, color);
        case RT_MATERIAL_MIRROR: return std::make_shared<MirrorMaterial>(m.name, color);
        case RT_MATERIAL_DIELECTRIC: return std::make_shared<DielectricMaterial>(m.name, m.ior, color);
        case RT_MATERIAL_MICROFACET: return std::make_shared<MicrofacetMaterial>(m.name, color, m.alpha);
        default: throw std::runtime_error("bad material type");
    }
}

BuiltScene build(const rt_scene& in)
{
    BuiltScene out;
    LoadedScene& scene = out.scene;
    const rt_settings& s = in.settings;

    RenderSettings& rs = scene.settings;
    rs.photonsPerLight = s.photons_per_light;
    rs.bounceStoreCapacity = s.bounce_store_capacity;
    rs.fetchSize = s.fetch_size;
    rs.workerCount = s.worker_count;
    rs.probeSubSample = s.probe_sub_sample;
    rs.imageWidth = s.image_width;
    rs.imageHeight = s.image_height;
    rs.bounceThreshold = s.bounce_threshold;
    rs.useProbeGather = s.use_probe_gather != 0;
    rs.deterministic = s.deterministic != 0;
    rs.seed = s.seeded ? s.seed : RenderSettings::kUnseeded;
    rs.cameraTimeSamples = s.camera_time_samples;
    rs.terminationThreshold = s.termination_threshold;
    rs.hashGridCellSize = s.hash_grid_cell_size;
    rs.densityCellScale = s.density_cell_scale;
    rs.splatMinRadiusScale = s.splat_min_radius_scale;
    rs.splatLuminanceClamp = s.splat_luminance_clamp;
    rs.probeKeepRadiusScale = s.probe_keep_radius_scale;
    rs.frameTime = s.frame_time;
    rs.shutterTime = s.shutter_time;
    rs.photonQueueSize = std::max<size_t>(20000000, 4 * s.fetch_size + 1000);

    // Materials: the library's constructor adds the nine built-ins; the scene file
    // lists them first as well, then the scene's own.
    scene.materialLibrary = std::make_shared<MaterialLibrary>();
    const size_t builtins = scene.materialLibrary->size();
    if (static_cast<size_t>(in.material_count) < builtins)
    {
        throw std::runtime_error("scene must list the built-in materials first");
    }
    for (size_t i = 0; i < builtins; ++i)
    {
        if (scene.materialLibrary->fetchByIndex(i)->name() != in.materials[i].name)
        {
            throw std::runtime_error("built-in material order mismatch");
        }
    }
    for (int64_t i = static_cast<int64_t>(builtins); i < in.material_count; ++i)
    {
        scene.materialLibrary->add(makeMaterial(in.materials[i]));
    }

    scene.meshLibrary = std::make_shared<MeshLibrary>();
    for (int64_t m = 0; m < in.mesh_count; ++m)
    {
        std::vector<Triangle> triangles;
        for (int64_t t = 0; t < in.meshes[m].triangle_count; ++t)
        {
            const double* p = in.tri_pos + 9 * (in.meshes[m].first_triangle + t);
            const double* n = in.tri_nrm + 9 * (in.meshes[m].first_triangle + t);
            Triangle tri(vec(p), vec(p + 3), vec(p + 6));
            tri.aNormal = vec(n);
            tri.bNormal = vec(n + 3);
            tri.cNormal = vec(n + 6);
            triangles.push_back(tri);
        }
        scene.meshLibrary->add(std::make_shared<Mesh>(in.meshes[m].name, triangles));
        out.meshTriangles.push_back(triangles);
    }

    for (int64_t c = 0; c < in.camera_count; ++c)
    {
        const rt_camera& rc = in.cameras[c];
        auto cam = std::make_shared<Camera>();
        cam->name(rc.name);
        cam->transform.position = vec(rc.position);
        cam->transform.rotation = quat(rc.rotation);
        cam->verticalFieldOfView(rc.vertical_fov_degrees);
        cam->setResolution(rc.width, rc.height);
        cam->projection(static_cast<Camera::Projection>(rc.projection));
        cam->orthographicHeight(rc.orthographic_height);
        cam->apertureRadius(rc.aperture_radius);
        cam->focusDistance(rc.focus_distance);
        cam->fNumber(rc.f_number);
        cam->shutterTime(rc.shutter_time);
        cam->iso(rc.iso);
        cam->bounceFilter(rc.bounce_filter);
        cam->lightFilter(rc.light_filter);
        scene.objects.push_back(cam);
        scene.cameras.push_back(cam);
    }
    if (!scene.cameras.empty())
    {
        scene.camera = scene.cameras.front();
    }

    for (int64_t l = 0; l < in.light_count; ++l)
    {
        const rt_light& rl = in.lights[l];
        std::shared_ptr<Light> light;
        switch (rl.type)
        {
            case RT_LIGHT_OMNI:
            {
                auto o = std::make_shared<OmniLight>();
                o->innerRadius(rl.inner_radius);
                light = o;
                break;
            }
            case RT_LIGHT_SPOT:
            {
                auto o = std::make_shared<SpotLight>();
                o->innerRadius(rl.inner_radius);
                o->angle(rl.angle);
                light = o;
                break;
            }
            case RT_LIGHT_PARALLEL:
            {
                auto o = std::make_shared<ParallelLight>();
                o->radius(rl.radius);
                light = o;
                break;
            }
            case RT_LIGHT_AREA:
            {
                auto o = std::make_shared<AreaLight>();
                o->shape(rl.shape == 1 ? AreaLight::Shape::Disc : AreaLight::Shape::Square);
                o->width(rl.width);
                o->height(rl.height);
                o->radius(rl.radius);
                o->luminousFluxOverride(rl.flux_override);
                light = o;
                break;
            }
            default: throw std::runtime_error("bad light type");
        }
        light->name(rl.name);
        light->transform.position = vec(rl.position);
        light->transform.rotation = quat(rl.rotation);
        light->color(Color{rl.color[0], rl.color[1], rl.color[2]});
        light->intensityCandela(rl.intensity_candela);
        if (light->luminousFlux() != rl.luminous_flux)
        {
            std::cerr << "WARNING: light " << rl.name << " flux mismatch: reference "
                      << light->luminousFlux() << " vs scene " << rl.luminous_flux << std::endl;
        }
        scene.objects.push_back(light);
        out.lights.push_back(light);
    }

    for (int64_t v = 0; v < in.volume_count; ++v)
    {
        const rt_volume& rv = in.volumes[v];
        std::shared_ptr<Volume> volume;
        switch (rv.type)
        {
            case RT_VOLUME_MESH:
                volume = std::make_shared<MeshVolume>(rv.material, scene.meshLibrary->fetchByIndex(rv.mesh));
                break;
            case RT_VOLUME_SPHERE:
                volume = std::make_shared<SphereVolume>(rv.material, vec(rv.param), rv.param[3]);
                break;
            case RT_VOLUME_QUAD:
                volume = std::make_shared<QuadVolume>(rv.material, Quad(vec(rv.param), vec(rv.param + 3), vec(rv.param + 6)));
                break;
            case RT_VOLUME_PLANE:
                volume = std::make_shared<PlaneVolume>(rv.material);
                break;
            default: throw std::runtime_error("bad volume type");
        }
        volume->name("volume_" + std::to_string(v));
        volume->transform.position = vec(rv.position);
        volume->transform.rotation = quat(rv.rotation);
        scene.objects.push_back(volume);
        out.volumes.push_back(volume);
        out.volumeMesh.push_back(rv.type == RT_VOLUME_MESH ? rv.mesh : -1);
    }

    // keyframed objects -> the reference's KeyframedAnimationQuery, keyed by object name
    auto animatedObject = [&](const rt_animation& ra) {
        KeyframedAnimationQuery::AnimatedObject a;
        a.hasPosition = true;
        a.position = Property<Vector>(vec(ra.constant_position));
        for (int64_t k = 0; k < ra.position_key_count; ++k)
        {
            const rt_keyframe& rk = in.keyframes[ra.first_position_key + k];
            Keyframe<Vector> key;
            key.time = rk.time;
            key.value = vec(rk.value);
            key.inTangent = vec(rk.in_tangent);
            key.outTangent = vec(rk.out_tangent);
            key.useAutoTangent = rk.use_auto_tangent != 0;
            a.position.addKeyframe(key);
        }
        a.hasRotation = ra.has_rotation != 0;
        a.rotationAxis = vec(ra.rotation_axis);
        for (int64_t k = 0; k < ra.rotation_key_count; ++k)
        {
            const rt_keyframe& rk = in.keyframes[ra.first_rotation_key + k];
            Keyframe<double> key;
            key.time = rk.time;
            key.value = rk.value[0];
            key.inTangent = rk.in_tangent[0];
            key.outTangent = rk.out_tangent[0];
            key.useAutoTangent = rk.use_auto_tangent != 0;
            a.rotationAngle.addKeyframe(key);
        }
        a.baseRotation = quat(ra.base_rotation);
        return a;
    };
    scene.animation = std::make_shared<KeyframedAnimationQuery>();
    for (int64_t v = 0; v < in.volume_count; ++v)
    {
        if (in.volumes[v].animation > 0)
        {
            scene.animation->setObject("volume_" + std::to_string(v), animatedObject(in.animations[in.volumes[v].animation - 1]));
        }
    }
    for (int64_t c = 0; c < in.camera_count; ++c)
    {
        if (in.cameras[c].animation > 0)
        {
            scene.animation->setObject(in.cameras[c].name, animatedObject(in.animations[in.cameras[c].animation - 1]));
        }
    }
    return out;
}

// The animation oracle a pass over `built` must use: the keyframed one when any object is
// animated, else the static one (as Renderer::renderFrame chooses, src/Renderer.cpp:116-130).
const AnimationQuery* animationOf(const BuiltScene& built)
{
    static StaticAnimationQuery fixed;
    if (built.scene.animation && !built.scene.animation->empty())
    {
        return built.scene.animation.get();
    }
    return &fixed;
}

// ---------------------------------------------------------------- unit vectors

struct Rng
{
    std::mt19937_64 gen;
    explicit Rng(uint64_t seed) : gen(seed) {}
    double uni(double lo, double hi) { return std::uniform_real_distribution<double>(lo, hi)(gen); }
    Vector v(double lo, double hi) { return Vector{uni(lo, hi), uni(lo, hi), uni(lo, hi)}; }
    Vector unit()
    {
        for (;;)
        {
            Vector p = v(-1, 1);
            const double m = p.magnitude();
            if (m > 1e-3 && m <= 1.0)
            {
                return p / m;
            }
        }
    }
};

void pushHit(std::vector<double>& out, const std::optional<Hit>& hit)
{
    out.push_back(hit ? 1.0 : 0.0);
    if (hit)
    {
        push3(out, hit->position);
        push3(out, hit->normal);
        out.push_back(hit->distance);
    }
    else
    {
        for (int i = 0; i < 7; ++i) out.push_back(0.0);
    }
}

void dumpTriangles(const std::string& dir)
{
    Rng rng(101);
    std::vector<double> in, out;
    size_t n = 0;
    auto emit = [&](const Ray& ray, const Triangle& tri) {
        push3(in, ray.origin);
        push3(in, ray.direction);
        push3(in, tri.a);
        push3(in, tri.b);
        push3(in, tri.c);
        push3(in, tri.aNormal);
        push3(in, tri.bNormal);
        push3(in, tri.cNormal);
        pushHit(out, rayIntersectsTriangle(ray, tri));
        ++n;
    };
    auto flat = [](const Vector& a, const Vector& b, const Vector& c) {
        Triangle t{a, b, c};
        t.aNormal = t.normal;
        t.bNormal = t.normal;
        t.cNormal = t.normal;
        return t;
    };
    // (1) the shared-diagonal sweeps of tests/test_TriangleWatertight.cpp:52-75,97-128
    const Vector v0{-1, -1, 0}, v1{1, -1, 0}, v2{1, 1, 0}, v3{-1, 1, 0};
    const Triangle t1 = flat(v0, v1, v2), t2 = flat(v0, v2, v3);
    for (int i = 0; i <= 200; ++i)
    {
        const double x = -1.0 + 2.0 * (static_cast<double>(i) / 200.0);
        const Ray axis{{x, x, 10.0}, {0, 0, -1}};
        emit(axis, t1);
        emit(axis, t2);
        const Vector origin{0.3, -0.7, 12.0};
        const Ray oblique{origin, Vector{x, x, 0.0} - origin};
        emit(oblique, t1);
        emit(oblique, t2);
    }
    emit(Ray{{0.2, -0.3, -10.0}, {0, 0, 1}}, t1);  // back face (:130-139)
    emit(Ray{{0.2, -0.3, -10.0}, {0, 0, 1}}, t2);
    // (2) axis-aligned wall split on a diagonal, rays aimed at edges / vertices / interior
    const Vector w0{-150, 0, 150}, w1{150, 0, 150}, w2{150, 300, 150}, w3{-150, 300, 150};
    const Triangle wa = flat(w0, w2, w1), wb = flat(w0, w3, w2);  // face -z
    for (int i = 0; i < 400; ++i)
    {
        const Vector origin = rng.v(-100, 100) + Vector{0, 150, -200};
        Vector target;
        const int kind = i % 4;
        const double s = rng.uni(0, 1);
        if (kind == 0) target = w0 + (w2 - w0) * s;            // on the diagonal
        else if (kind == 1) target = w0 + (w1 - w0) * s;       // on an outer edge
        else if (kind == 2) target = (i % 8 < 4) ? w0 : w2;    // a vertex
        else target = Vector{rng.uni(-160, 160), rng.uni(-10, 310), 150};
        Vector d = target - origin;
        if (i % 2) d.normalize();
        emit(Ray{origin, d}, wa);
        emit(Ray{origin, d}, wb);
    }
    // (3) random triangles with smooth normals, rays aimed near them
    for (int i = 0; i < 4000; ++i)
    {
        const Vector a = rng.v(-100, 100), b = a + rng.v(-30, 30), c = a + rng.v(-30, 30);
        Triangle tri{a, b, c};
        tri.aNormal = (tri.normal + rng.v(-0.3, 0.3)).normalized();
        tri.bNormal = (tri.normal + rng.v(-0.3, 0.3)).normalized();
        tri.cNormal = (tri.normal + rng.v(-0.3, 0.3)).normalized();
        const double u = rng.uni(-0.3, 1.3), v = rng.uni(-0.3, 1.3);
        const Vector target = a + (b - a) * u + (c - a) * v;
        const Vector origin = rng.v(-200, 200);
        Vector d = target - origin;
        if (i % 3 == 0) d.normalize();
        if (i % 7 == 0) d = d * -1.0;
        emit(Ray{origin, d}, tri);
    }
    npy::writeF64(dir + "/tri_in.npy", in, {n, 24});
    npy::writeF64(dir + "/tri_out.npy", out, {n, 8});
}

void dumpBounds(const std::string& dir)
{
    Rng rng(202);
    std::vector<double> in;
    std::vector<int32_t> out;
    size_t n = 0;
    for (int i = 0; i < 6000; ++i)
    {
        Vector lo = rng.v(-100, 100);
        Vector hi = lo + rng.v(0, 60);
        if (i % 5 == 0) hi.x = lo.x;  // flat box (axis-aligned wall)
        if (i % 11 == 0) hi.y = lo.y;
        const Bounds bounds{lo, hi};
        const Vector origin = rng.v(-150, 150);
        const Vector target = lo + Vector{(hi.x - lo.x) * rng.uni(-0.2, 1.2), (hi.y - lo.y) * rng.uni(-0.2, 1.2),
                                          (hi.z - lo.z) * rng.uni(-0.2, 1.2)};
        Vector d = target - origin;
        if (i % 2) d.normalize();
        if (i % 13 == 0) d.x = 0.0;  // axis-parallel branch
        if (i % 17 == 0) d.z = 0.0;
        if (i % 19 == 0) d = d * -1.0;
        const Ray ray{origin, d};
        push3(in, origin);
        push3(in, d);
        push3(in, lo);
        push3(in, hi);
        out.push_back(rayIntersectsBounds(ray, bounds) ? 1 : 0);
        ++n;
    }
    npy::writeF64(dir + "/bounds_in.npy", in, {n, 12});
    npy::writeI32(dir + "/bounds_out.npy", out, {n});
}

void dumpAnalytic(const std::string& dir)
{
    Rng rng(303);
    {
        std::vector<double> in, out;
        size_t n = 0;
        for (int i = 0; i < 4000; ++i)
        {
            const Vector center = rng.v(-50, 50);
            const double radius = rng.uni(0.5, 60);
            Vector origin = rng.v(-150, 150);
            if (i % 4 == 0) origin = center + rng.unit() * (radius * rng.uni(0, 0.95));  // inside
            if (i % 9 == 0) origin = center + rng.unit() * radius;                        // on the surface
            const Vector target = center + rng.unit() * (radius * rng.uni(0, 1.3));
            Vector d = target - origin;
            if (i % 2) d.normalize();
            const Ray ray{origin, d};
            push3(in, origin);
            push3(in, d);
            push3(in, center);
            in.push_back(radius);
            pushHit(out, rayIntersectsSphere(ray, Sphere{center, radius}));
            ++n;
        }
        npy::writeF64(dir + "/sphere_in.npy", in, {n, 10});
        npy::writeF64(dir + "/sphere_out.npy", out, {n, 8});
    }
    {
        std::vector<double> in, out;
        size_t n = 0;
        for (int i = 0; i < 4000; ++i)
        {
            const Vector o = rng.v(-100, 100);
            Vector eu = rng.v(-80, 80), ev = rng.v(-80, 80);
            if (i % 3 == 0)
            {
                eu = Vector{rng.uni(10, 200), 0, 0};
                ev = Vector{0, rng.uni(10, 200), 0};
            }
            const Quad quad{o, eu, ev};
            const Vector target = o + eu * rng.uni(-0.2, 1.2) + ev * rng.uni(-0.2, 1.2);
            const Vector origin = rng.v(-200, 200);
            Vector d = target - origin;
            if (i % 2) d.normalize();
            if (i % 10 == 0) d = d * -1.0;
            const Ray ray{origin, d};
            push3(in, origin);
            push3(in, d);
            push3(in, o);
            push3(in, eu);
            push3(in, ev);
            const std::optional<Hit> hit = rayIntersectsQuad(ray, quad);
            pushHit(out, hit);
            push3(out, hit ? hit->uv : Vector{});
            push3(out, quad.normal);
            push3(out, quad.invU);
            push3(out, quad.invV);
            ++n;
        }
        npy::writeF64(dir + "/quad_in.npy", in, {n, 15});
        npy::writeF64(dir + "/quad_out.npy", out, {n, 20});
    }
    {
        std::vector<double> in, out;
        size_t n = 0;
        for (int i = 0; i < 2000; ++i)
        {
            Plane plane;
            plane.normal = (i % 2) ? rng.unit() : Vector{0, 1, 0};
            plane.dot = (i % 2) ? rng.uni(-50, 50) : 0.0;
            const Vector origin = rng.v(-100, 100);
            Vector d = rng.unit();
            if (i % 5 == 0) d = d * rng.uni(0.1, 5);
            const Ray ray{origin, d};
            push3(in, origin);
            push3(in, d);
            push3(in, plane.normal);
            in.push_back(plane.dot);
            pushHit(out, rayIntersectsPlane(ray, plane));
            ++n;
        }
        npy::writeF64(dir + "/plane_in.npy", in, {n, 10});
        npy::writeF64(dir + "/plane_out.npy", out, {n, 8});
    }
    {
        // Quaternion rotate / inverse-rotate and pitch-yaw-roll construction
        std::vector<double> in, out;
        size_t n = 0;
        for (int i = 0; i < 2000; ++i)
        {
            const double pitch = rng.uni(-3.2, 3.2), yaw = rng.uni(-3.2, 3.2), roll = rng.uni(-3.2, 3.2);
            const Quaternion q = Quaternion::fromPitchYawRoll(pitch, yaw, roll);
            const Vector v = rng.v(-100, 100);
            in.insert(in.end(), {pitch, yaw, roll});
            push3(in, v);
            out.insert(out.end(), {q.x, q.y, q.z, q.w});
            push3(out, q * v);
            push3(out, q.inverse() * v);
            ++n;
        }
        npy::writeF64(dir + "/quat_in.npy", in, {n, 6});
        npy::writeF64(dir + "/quat_out.npy", out, {n, 10});
    }
}

void dumpBsdf(const std::string& dir)
{
    Rng rng(404);
    std::vector<double> in, out, evalIn, evalOut;
    size_t n = 0, ne = 0;
    const double roughness[] = {0.0005, 0.05, 0.2, 0.3, 0.5, 1.0};
    for (int i = 0; i < 8000; ++i)
    {
        rt_material m{};
        m.type = i % 4;
        m.color[0] = static_cast<float>(rng.uni(0.05, 1.0));
        m.color[1] = static_cast<float>(rng.uni(0.05, 1.0));
        m.color[2] = static_cast<float>(rng.uni(0.05, 1.0));
        m.alpha = roughness[(i / 4) % 6];
        m.ior = (i % 8 < 4) ? 1.5 : rng.uni(1.05, 2.4);
        std::snprintf(m.name, sizeof(m.name), "m%d", i);
        const std::shared_ptr<Material> material = makeMaterial(m);

        Vector normal = rng.unit();
        if (i % 50 == 0) normal = Vector{0, 0, -1};  // Frisvad singular branch
        if (i % 50 == 1) normal = Vector{0, 0, 1};
        Vector incident = rng.unit();
        if (i % 3 != 0 && Vector::dot(incident, normal) > 0.0) incident = incident * -1.0;  // mostly arriving from above
        if (i % 29 == 0)
        {
            // grazing incidence
            Vector t = Vector::cross(normal, rng.unit()).normalized();
            incident = (t - normal * rng.uni(1e-4, 2e-2)).normalized();
        }
        const uint32_t seed = 1000u + static_cast<uint32_t>(i);
        RandomGenerator replay(seed);
        const double u1 = replay.value();
        const double u2 = replay.value();
        RandomGenerator generator(seed);
        const UnitVector unitNormal = UnitVector::alreadyNormalized(normal);
        const BSDFSample s = material->sample(incident, unitNormal, generator);

        in.push_back(m.type);
        in.insert(in.end(), {m.color[0], m.color[1], m.color[2]});
        // the alpha the material actually uses (floored), so consumers need no floor logic
        in.push_back(std::max(1e-3, m.alpha));
        in.push_back(m.ior);
        push3(in, incident);
        push3(in, normal);
        in.push_back(u1);
        in.push_back(u2);
        out.push_back(s.valid ? 1.0 : 0.0);
        push3(out, s.direction);
        out.insert(out.end(), {s.weight.red, s.weight.green, s.weight.blue});
        out.push_back(s.pdf);
        out.push_back(s.isDelta ? 1.0 : 0.0);
        out.push_back(material->isDelta() ? 1.0 : 0.0);
        ++n;

        // evaluate()/pdf() for a pair of directions (sink B / gather use evaluate)
        const Vector wi = (i % 5 == 0) ? rng.unit() : incident * -1.0;
        Vector wo = rng.unit();
        if (i % 4 != 3 && Vector::dot(wo, normal) < 0.0) wo = wo * -1.0;
        const Color f = material->evaluate(wi, wo, unitNormal);
        const double p = material->pdf(wi, wo, unitNormal);
        evalIn.push_back(m.type);
        evalIn.insert(evalIn.end(), {m.color[0], m.color[1], m.color[2]});
        evalIn.push_back(std::max(1e-3, m.alpha));
        push3(evalIn, wi);
        push3(evalIn, wo);
        push3(evalIn, normal);
        evalOut.insert(evalOut.end(), {f.red, f.green, f.blue, p});
        ++ne;
    }
    npy::writeF64(dir + "/bsdf_in.npy", in, {n, 14});
    npy::writeF64(dir + "/bsdf_out.npy", out, {n, 10});
    npy::writeF64(dir + "/eval_in.npy", evalIn, {ne, 14});
    npy::writeF64(dir + "/eval_out.npy", evalOut, {ne, 4});
}

// ---------------------------------------------------------------- scene modes

rt_scene loadScene(const std::string& path)
{
    rt_scene s;
    if (rt_scene_read(&s, path.c_str()) != 0)
    {
        throw std::runtime_error("cannot read scene " + path);
    }
    return s;
}

int modeEmit(const std::string& scenePath, const std::string& dir, size_t n, uint32_t seed)
{
    rt_scene in = loadScene(scenePath);
    BuiltScene built = build(in);
    for (size_t l = 0; l < built.lights.size(); ++l)
    {
        std::vector<Photon> storage(n);
        WorkQueue<Photon>::Block block(0, n, storage);
        RandomGenerator generator(seed);
        const double photonFlux = built.lights[l]->luminousFlux() / static_cast<double>(n);
        built.lights[l]->emit(block, photonFlux, generator);
        std::vector<double> rays;
        std::vector<float> colors;
        for (const Photon& p : storage)
        {
            push3(rays, p.ray.origin);
            push3(rays, p.ray.direction);
            colors.insert(colors.end(), {p.color.red, p.color.green, p.color.blue});
        }
        std::vector<double> draws;
        RandomGenerator replay(seed);
        for (size_t i = 0; i < n * 5; ++i)
        {
            draws.push_back(replay.value());
        }
        const std::string tag = dir + "/emit_L" + std::to_string(l);
        npy::writeF64(tag + "_rays.npy", rays, {n, 6});
        npy::writeF32(tag + "_color.npy", colors, {n, 3});
        npy::writeF64(tag + "_draws.npy", draws, {n * 5});
    }
    return 0;
}

// Scene-depth pixel footprint exactly as Renderer::renderFrame derives it
// (this calls nothing but public accessors; the arithmetic is the harness's own
// restatement and is validated by comparing `walk` deposits with `render`'s store).
struct Derived
{
    double sceneDepthFootprint = 0.0;
    double cellSize = 0.0;
    double minSplatRadius = 0.0;
    double keepRadius = 0.0;
};

Derived derive(const BuiltScene& built)
{
    const RenderSettings& s = built.scene.settings;
    Derived d;
    d.cellSize = s.hashGridCellSize;
    const std::shared_ptr<Camera> cam = built.scene.cameras.empty() ? nullptr : built.scene.cameras.front();
    if (cam && s.imageHeight > 0 && !built.volumes.empty())
    {
        Vector centroid{0, 0, 0};
        for (const auto& v : built.volumes)
        {
            centroid = centroid + v->position();
        }
        centroid = centroid / static_cast<double>(built.volumes.size());
        const double depth = (centroid - cam->position()).magnitude();
        const double pixelHalfAngle = 0.5 * Utility::radians(cam->verticalFieldOfView()) /
                                      static_cast<double>(cam->height() > 0 ? cam->height() : s.imageHeight);
        const double r = depth * std::tan(pixelHalfAngle);
        if (r > 0.0)
        {
            d.cellSize = r;
            d.sceneDepthFootprint = r;
        }
    }
    d.minSplatRadius = (d.sceneDepthFootprint > 0.0 && s.splatMinRadiusScale > 0.0)
                           ? s.splatMinRadiusScale * d.sceneDepthFootprint
                           : 0.0;
    d.cellSize *= std::max(0.05, s.densityCellScale);
    const double footprint = (d.sceneDepthFootprint > 0.0) ? d.sceneDepthFootprint : d.cellSize;
    d.keepRadius = std::max(footprint, s.probeKeepRadiusScale * footprint);
    return d;
}

int identifyTriangle(const BuiltScene& built, size_t volumeIndex, const Ray& worldRay, float time, const Hit& hit,
                     int& matches)
{
    matches = 0;
    const int mesh = built.volumeMesh[volumeIndex];
    if (mesh < 0)
    {
        return -1;
    }
    const auto& volume = built.volumes[volumeIndex];
    // the pose Volume::castRayAt used for this ray (src/Volume.cpp:65-81): the animation's when there is one
    Vector position = volume->position();
    Quaternion rotation = volume->rotation();
    if (const std::optional<Transform> moved = animationOf(built)->transformAt(volume->name(), time))
    {
        position = moved->position;
        rotation = moved->rotation;
    }
    const Quaternion inv = rotation.inverse();
    const Ray local{inv * (worldRay.origin - position), inv * worldRay.direction};
    int found = -1;
    const auto& triangles = built.meshTriangles[mesh];
    for (size_t t = 0; t < triangles.size(); ++t)
    {
        const std::optional<Hit> h = rayIntersectsTriangle(local, triangles[t]);
        if (h && h->distance == hit.distance)
        {
            if (found < 0) found = static_cast<int>(t);
            ++matches;
        }
    }
    return found;
}

struct WalkOutput
{
    std::vector<double> emitted;     // [photons, 6]
    std::vector<double> bounceF64;   // [bounces, 10]: pos3 nrm3 dist dirIn3
    std::vector<int32_t> bounceI32;  // [bounces, 8]: photonLo, bounce, volume, triangle, material, matches, delta, light
    std::vector<float> bouncePower;  // [bounces, 3]
    std::vector<int32_t> casts;      // [photons, 2]: casts, valid bounces
    std::vector<RawBounce> deposits; // every non-delta bounce, in order (all photons)
    std::vector<Vector> depositPositions;  // the same bounces' full-precision positions
    uint64_t totalBounces = 0;
    uint64_t totalCasts = 0;
    uint64_t photons = 0;
};

// One seeded worker, walking photons in the order Worker::exec does: emit a batch
// of fetchSize photons from the first light that still owes some, stamp them, trace
// the whole batch to completion, repeat (src/Worker.cpp:116-174,429-506,508-680).
// Every geometric / material step below is a call INTO the reference.
WalkOutput walk(BuiltScene& built, size_t recordPhotons, Worker* splatWorker)
{
    WalkOutput out;
    const RenderSettings& s = built.scene.settings;
    const std::uint32_t seed = (s.seed != RenderSettings::kUnseeded) ? s.seed : 0u;
    RandomGenerator generator(seed);
    std::vector<Hit> castBuffer;
    const Flux termination{std::max(0.0, s.terminationThreshold)};
    const double selfHit = WorkerDebug::selfHitThreshold();
    const Camera::ExposureWindow window = built.scene.camera ? built.scene.camera->globalExposureWindow()
                                                             : Camera::ExposureWindow{};

    size_t photonCounter = 0;
    for (size_t l = 0; l < built.lights.size(); ++l)
    {
        size_t remaining = s.photonsPerLight;
        const double photonFlux =
            s.photonsPerLight > 0 ? built.lights[l]->luminousFlux() / static_cast<double>(s.photonsPerLight) : 0.0;
        while (remaining > 0)
        {
            const size_t n = std::min(remaining, s.fetchSize);
            remaining -= n;
            std::vector<Photon> batch(n);
            WorkQueue<Photon>::Block block(0, n, batch);
            built.lights[l]->emit(block, photonFlux, generator);
            for (Photon& p : batch)
            {
                p.lightId = static_cast<int>(l);
            }
            if (std::isfinite(window.start))
            {
                const bool finiteSpan = std::isfinite(window.end) && window.end > window.start;
                const float span = finiteSpan ? (window.end - window.start) : 0.0f;
                for (Photon& p : batch)
                {
                    p.time = finiteSpan ? window.start + static_cast<float>(generator.value(span)) : window.start;
                }
            }
            std::vector<Photon> slot(1);
            for (const Photon& source : batch)
            {
                const bool record = photonCounter < recordPhotons;
                if (record)
                {
                    push3(out.emitted, source.ray.origin);
                    push3(out.emitted, source.ray.direction);
                }
                Photon photon = source;
                int32_t casts = 0, valid = 0;
                while (true)
                {
                    if (photon.color.brightness() < std::numeric_limits<double>::epsilon())
                    {
                        break;
                    }
                    ++casts;
                    double minDistance = std::numeric_limits<double>::max();
                    bool validHit = false;
                    Hit best;
                    size_t bestVolume = 0;
                    for (size_t v = 0; v < built.volumes.size(); ++v)
                    {
                        std::optional<Hit> hit = built.volumes[v]->castRayAt(photon.ray, castBuffer, photon.time, animationOf(built));
                        if (hit && hit->distance < minDistance && hit->distance > selfHit)
                        {
                            validHit = true;
                            minDistance = hit->distance;
                            best = *hit;
                            bestVolume = v;
                        }
                    }
                    if (!validHit)
                    {
                        break;
                    }
                    ++valid;
                    std::shared_ptr<Material> material = built.scene.materialLibrary->fetchByIndex(best.material);
                    const bool delta = material->isDelta();
                    if (record)
                    {
                        int matches = 0;
                        const int tri = identifyTriangle(built, bestVolume, photon.ray, photon.time, best, matches);
                        push3(out.bounceF64, best.position);
                        push3(out.bounceF64, best.normal);
                        out.bounceF64.push_back(best.distance);
                        push3(out.bounceF64, photon.ray.direction);
                        out.bounceI32.insert(out.bounceI32.end(),
                                             {static_cast<int32_t>(photonCounter), photon.bounces,
                                              static_cast<int32_t>(bestVolume), tri, static_cast<int32_t>(best.material),
                                              matches, delta ? 1 : 0, photon.lightId});
                        out.bouncePower.insert(out.bouncePower.end(),
                                               {photon.color.red, photon.color.green, photon.color.blue});
                    }
                    if (!delta)
                    {
                        out.deposits.emplace_back(best.position, photon.ray.direction, best.normal, photon.time,
                                                  photon.color);
                        out.depositPositions.push_back(best.position);
                        if (splatWorker)
                        {
                            splatWorker->splatToCamera(PhotonHit{photon, best}, material);
                        }
                    }
                    if (!photonDecayAlive(photon, termination) || photon.bounces >= static_cast<int>(s.bounceThreshold))
                    {
                        break;
                    }
                    WorkQueue<Photon>::Block one(0, 1, slot);
                    material->generateDaughters(one, 0, 0, 1, 1, photon.ray.direction, best.normal, best.position,
                                                photon.color, photon.time, photon.bounces, photon.lightId, generator);
                    photon = slot[0];
                }
                if (record)
                {
                    out.casts.push_back(casts);
                    out.casts.push_back(valid);
                }
                out.totalBounces += static_cast<uint64_t>(valid);
                out.totalCasts += static_cast<uint64_t>(casts);
                ++photonCounter;
            }
        }
    }
    out.photons = photonCounter;
    return out;
}

void setExposureWindows(BuiltScene& built)
{
    // Same window Renderer::renderFrame installs (src/Renderer.cpp:145-179).
    const RenderSettings& s = built.scene.settings;
    Camera::ExposureWindow window;
    const float start = static_cast<float>(s.frameTime);
    const float shutter = static_cast<float>(s.shutterTime);
    window.start = start;
    window.end = (shutter > 0.0f) ? start + shutter : std::numeric_limits<float>::infinity();
    for (auto& cam : built.scene.cameras)
    {
        cam->setGlobalExposureWindow(window);
    }
}

void writeDeposits(const std::string& path, const std::vector<RawBounce>& deposits)
{
    std::vector<float> flat;
    flat.reserve(deposits.size() * 13);
    for (const RawBounce& r : deposits)
    {
        flat.insert(flat.end(), {r.px, r.py, r.pz, r.ix, r.iy, r.iz, r.nx, r.ny, r.nz, r.time, r.power.red,
                                 r.power.green, r.power.blue});
    }
    npy::writeF32(path, flat, {deposits.size(), 13});
}

int modeWalk(const std::string& scenePath, const std::string& dir, size_t recordPhotons)
{
    rt_scene in = loadScene(scenePath);
    BuiltScene built = build(in);
    setExposureWindows(built);
    const Derived d = derive(built);

    // Splat targets: one buffer per camera, wired into a Worker that is never started;
    // only its public splatToCamera is called (src/Worker.cpp:221-427).
    Worker worker(0, built.scene.settings.fetchSize);
    worker.camera = built.scene.camera;
    worker.objects = built.scene.objects;
    worker.materialLibrary = built.scene.materialLibrary;
    if (built.scene.animation && !built.scene.animation->empty())
    {
        worker.animationQuery = built.scene.animation;  // the splat's occlusion cast sees the photon's time
    }
    worker.setPhotonsPerLight(static_cast<double>(built.scene.settings.photonsPerLight));
    worker.setMinSplatRadius(d.minSplatRadius);
    worker.setSplatLuminanceClamp(built.scene.settings.splatLuminanceClamp);
    std::vector<std::shared_ptr<Buffer>> buffers;
    std::vector<Worker::SplatTarget> targets;
    for (const auto& cam : built.scene.cameras)
    {
        auto buffer = std::make_shared<Buffer>(cam->width(), cam->height());
        buffer->clear();
        buffers.push_back(buffer);
        targets.push_back(Worker::SplatTarget{cam, buffer, cam->bounceFilter(), cam->lightFilter()});
    }
    worker.setSplatTargets(targets);
    WorkerDebug::resetSplatCounters();

    WalkOutput out = walk(built, recordPhotons, &worker);

    const size_t photons = out.casts.size() / 2;
    const size_t bounces = out.bounceI32.size() / 8;
    npy::writeF64(dir + "/walk_emitted.npy", out.emitted, {photons, 6});
    npy::writeF64(dir + "/walk_bounce_f64.npy", out.bounceF64, {bounces, 10});
    npy::writeI32(dir + "/walk_bounce_i32.npy", out.bounceI32, {bounces, 8});
    npy::writeF32(dir + "/walk_bounce_power.npy", out.bouncePower, {bounces, 3});
    npy::writeI32(dir + "/walk_casts.npy", out.casts, {photons, 2});
    const size_t keep = std::min<size_t>(out.deposits.size(), 65536);
    writeDeposits(dir + "/walk_deposits_head.npy",
                  std::vector<RawBounce>(out.deposits.begin(), out.deposits.begin() + keep));
    double sumPower[3] = {0, 0, 0};
    for (const RawBounce& r : out.deposits)
    {
        sumPower[0] += r.power.red;
        sumPower[1] += r.power.green;
        sumPower[2] += r.power.blue;
    }
    std::vector<double> totals = {static_cast<double>(out.photons),
                                  static_cast<double>(out.totalCasts),
                                  static_cast<double>(out.totalBounces),
                                  static_cast<double>(out.deposits.size()),
                                  sumPower[0], sumPower[1], sumPower[2],
                                  static_cast<double>(WorkerDebug::splatTotal()),
                                  static_cast<double>(WorkerDebug::splatRadiusClamped()),
                                  static_cast<double>(WorkerDebug::splatLuminanceClamped()),
                                  d.sceneDepthFootprint, d.cellSize, d.minSplatRadius, d.keepRadius};
    npy::writeF64(dir + "/walk_totals.npy", totals, {totals.size()});
    for (size_t c = 0; c < buffers.size(); ++c)
    {
        const auto& cam = built.scene.cameras[c];
        std::vector<float> pixels;
        for (size_t y = 0; y < cam->height(); ++y)
        {
            for (size_t x = 0; x < cam->width(); ++x)
            {
                const Color col = buffers[c]->fetchColor({x, y});
                pixels.insert(pixels.end(), {col.red, col.green, col.blue});
            }
        }
        npy::writeF32(dir + "/walk_splat_c" + std::to_string(c) + ".npy", pixels, {cam->height(), cam->width(), 3});
    }
    std::cout << "walk: photons " << out.photons << " casts " << out.totalCasts << " bounces " << out.totalBounces
              << " deposits " << out.deposits.size() << std::endl;
    return 0;
}

std::vector<double> flattenGatherPoints(const std::vector<ProbeGather::GatherPoint>& points)
{
    std::vector<double> flat;
    for (const auto& gp : points)
    {
        flat.push_back(static_cast<double>(gp.pixel.x));
        flat.push_back(static_cast<double>(gp.pixel.y));
        push3(flat, gp.position);
        push3(flat, gp.normal);
        push3(flat, gp.viewDir);
        flat.push_back(gp.materialIndex == ProbeGather::testing::kEmitterMaterial
                           ? -1.0
                           : static_cast<double>(gp.materialIndex));
        flat.insert(flat.end(), {gp.specularThroughput.red, gp.specularThroughput.green, gp.specularThroughput.blue});
        flat.push_back(gp.unfoldedPathLength);
        flat.push_back(gp.footprintRadius);
        flat.push_back(gp.sampleTime);
        flat.push_back(gp.sampleWeight);
    }
    return flat;
}

long long probeSeedFor(const RenderSettings& s)
{
    const bool seedActive = s.deterministic || (s.seed != RenderSettings::kUnseeded);
    const std::uint32_t baseSeed = (s.seed != RenderSettings::kUnseeded) ? s.seed : 0u;
    return seedActive ? static_cast<long long>(baseSeed) + 0x9E3779B9LL : -1;
}

int modeProbes(const std::string& scenePath, const std::string& dir)
{
    rt_scene in = loadScene(scenePath);
    BuiltScene built = build(in);
    setExposureWindows(built);
    const RenderSettings& s = built.scene.settings;
    for (size_t c = 0; c < built.scene.cameras.size(); ++c)
    {
        const auto t0 = std::chrono::steady_clock::now();
        ProbeGather::ProbeResult r = ProbeGather::collectGatherPoints(
            built.scene.objects, *built.scene.cameras[c], *built.scene.materialLibrary, animationOf(built),
            static_cast<float>(s.frameTime), static_cast<float>(s.shutterTime), s.cameraTimeSamples,
            s.probeSubSample, probeSeedFor(s));
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const std::vector<double> flat = flattenGatherPoints(r.points);
        npy::writeF64(dir + "/probes_c" + std::to_string(c) + ".npy", flat, {r.points.size(), 19});
        std::vector<double> counts = {static_cast<double>(r.cameraRays), static_cast<double>(r.deltaExtensions),
                                      static_cast<double>(r.misses), secs};
        npy::writeF64(dir + "/probes_counts_c" + std::to_string(c) + ".npy", counts, {counts.size()});
        std::cout << "probes: camera " << c << " points " << r.points.size() << " rays " << r.cameraRays
                  << " misses " << r.misses << " seconds " << secs << std::endl;
    }
    return 0;
}

int modeRender(const std::string& scenePath, const std::string& dir)
{
    rt_scene in = loadScene(scenePath);
    BuiltScene built = build(in);
    WorkerDebug::resetSplatCounters();
    WorkerDebug::resetBounceCounters();
    RenderResult result = Renderer::renderFrame(built.scene);

    std::vector<double> counters = {
        static_cast<double>(WorkerDebug::bounceKept()),
        static_cast<double>(WorkerDebug::bounceCulled()),
        static_cast<double>(result.bounceStoreDropped),
        static_cast<double>(result.bounceStore ? result.bounceStore->attemptedCount() : 0),
        static_cast<double>(result.bounceStore ? result.bounceStore->size() : 0),
        static_cast<double>(result.bounceStore ? result.bounceStore->capacity() : 0),
        static_cast<double>(result.emitterDepositsKept),
        static_cast<double>(WorkerDebug::splatTotal()),
        static_cast<double>(WorkerDebug::splatRadiusClamped()),
        static_cast<double>(WorkerDebug::splatLuminanceClamped()),
        result.photonPassSeconds,
        static_cast<double>(result.densityGrid ? result.densityGrid->cellCount() : 0),
        static_cast<double>(result.densityGrid ? result.densityGrid->depositCount() : 0),
    };
    npy::writeF64(dir + "/render_counters.npy", counters, {counters.size()});

    // per-camera composited float buffers, tonemapped images and gather diagnostics
    for (size_t c = 0; c < result.cameras.size(); ++c)
    {
        const CameraRender& cr = result.cameras[c];
        if (!cr.buffer || !cr.image || !cr.camera) continue;
        const size_t w = cr.camera->width(), h = cr.camera->height();
        std::vector<float> pixels;
        std::vector<int32_t> image;
        pixels.reserve(w * h * 3);
        image.reserve(w * h * 3);
        for (size_t y = 0; y < h; ++y)
        {
            for (size_t x = 0; x < w; ++x)
            {
                const Color col = cr.buffer->fetchColor({x, y});
                pixels.insert(pixels.end(), {col.red, col.green, col.blue});
                const Pixel& px = cr.image->getPixel(x, y);
                image.insert(image.end(), {static_cast<int32_t>(px.red), static_cast<int32_t>(px.green),
                                           static_cast<int32_t>(px.blue)});
            }
        }
        npy::writeF32(dir + "/render_buffer_c" + std::to_string(c) + ".npy", pixels, {h, w, 3});
        npy::writeI32(dir + "/render_image_c" + std::to_string(c) + ".npy", image, {h, w, 3});
        std::vector<double> probe = {static_cast<double>(cr.probe.pixelsHit), static_cast<double>(cr.probe.pixelsGathered),
                                     static_cast<double>(cr.probe.depositsAccum), cr.camera->saturationLuminance()};
        npy::writeF64(dir + "/render_gather_c" + std::to_string(c) + ".npy", probe, {probe.size()});
    }
    if (result.bounceStore && result.bounceStore->size() <= (1u << 20))
    {
        std::vector<RawBounce> all;
        for (size_t i = 0; i < result.bounceStore->size(); ++i) all.push_back((*result.bounceStore)[i]);
        writeDeposits(dir + "/render_store.npy", all);
    }

    if (result.bounceStore)
    {
        const size_t n = result.bounceStore->size();
        std::vector<RawBounce> records;
        const size_t keep = std::min<size_t>(n, 65536);
        double sum[3] = {0, 0, 0};
        for (size_t i = 0; i < n; ++i)
        {
            const RawBounce& r = (*result.bounceStore)[i];
            if (i < keep) records.push_back(r);
            sum[0] += r.power.red;
            sum[1] += r.power.green;
            sum[2] += r.power.blue;
        }
        writeDeposits(dir + "/render_store_head.npy", records);
        std::vector<double> ledger = {static_cast<double>(n), sum[0], sum[1], sum[2]};
        npy::writeF64(dir + "/render_ledger.npy", ledger, {ledger.size()});

        // Cross-check: the harness walk (same seed, one worker) filtered by the same
        // keep-test must reproduce the store record for record.  Only meaningful in
        // deterministic mode.
        if (built.scene.settings.deterministic)
        {
            BuiltScene again = build(in);
            setExposureWindows(again);
            const Derived d = derive(again);
            std::vector<Vector> probes;
            for (const auto& cam : again.scene.cameras)
            {
                if (cam->bounceFilter() >= 0 || cam->lightFilter() >= 0) continue;
                const RenderSettings& s = again.scene.settings;
                ProbeGather::ProbeResult r = ProbeGather::collectGatherPoints(
                    again.scene.objects, *cam, *again.scene.materialLibrary, animationOf(again),
                    static_cast<float>(s.frameTime), static_cast<float>(s.shutterTime), s.cameraTimeSamples,
                    s.probeSubSample, probeSeedFor(s));
                for (const auto& gp : r.points) probes.push_back(gp.position);
            }
            ProbeIndex index(probes, d.keepRadius, d.keepRadius);
            WalkOutput w = walk(again, 0, nullptr);
            size_t cursor = result.emitterDepositsKept;
            size_t mismatches = 0, kept = 0;
            for (size_t i = 0; i < w.deposits.size(); ++i)
            {
                const RawBounce& r = w.deposits[i];
                if (!index.anyWithinKeepRadius(w.depositPositions[i])) continue;
                ++kept;
                if (cursor < n)
                {
                    const RawBounce& s = (*result.bounceStore)[cursor];
                    if (std::memcmp(&s, &r, sizeof(RawBounce)) != 0) ++mismatches;
                }
                ++cursor;
            }
            std::cout << "render: store " << n << " (emitter " << result.emitterDepositsKept << "), walk kept " << kept
                      << ", record mismatches " << mismatches << std::endl;
            std::vector<double> cross = {static_cast<double>(kept), static_cast<double>(mismatches)};
            npy::writeF64(dir + "/render_crosscheck.npy", cross, {cross.size()});
        }
    }

    for (size_t c = 0; c < result.cameras.size(); ++c)
    {
        const CameraRender& cr = result.cameras[c];
        if (!cr.buffer) continue;
        const size_t w = cr.camera->width(), h = cr.camera->height();
        std::vector<float> pixels;
        for (size_t y = 0; y < h; ++y)
        {
            for (size_t x = 0; x < w; ++x)
            {
                const Color col = cr.buffer->fetchColor({x, y});
                pixels.insert(pixels.end(), {col.red, col.green, col.blue});
            }
        }
        npy::writeF32(dir + "/render_buffer_c" + std::to_string(c) + ".npy", pixels, {h, w, 3});
        std::vector<int32_t> image;
        for (size_t y = 0; y < h; ++y)
        {
            for (size_t x = 0; x < w; ++x)
            {
                const Pixel& p = cr.image->getPixel(x, y);
                image.insert(image.end(), {p.red, p.green, p.blue});
            }
        }
        npy::writeI32(dir + "/render_image_c" + std::to_string(c) + ".npy", image, {h, w, 3});
        std::cout << "render: camera " << c << " mean luminance " << cr.meanLuminance << " gather s " << cr.gatherSeconds
                  << std::endl;
    }
    std::cout << "render: photonPassSeconds " << result.photonPassSeconds << std::endl;
    return 0;
}

int modeBench(const std::string& scenePath, int workers)
{
    rt_scene in = loadScene(scenePath);
    BuiltScene built = build(in);
    if (workers > 0)
    {
        built.scene.settings.workerCount = static_cast<size_t>(workers);
        built.scene.settings.deterministic = false;
    }
    WorkerDebug::resetBounceCounters();
    WorkerDebug::resetSplatCounters();
    const auto t0 = std::chrono::steady_clock::now();
    RenderResult result = Renderer::renderFrame(built.scene);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "{\"photon_pass_seconds\": " << result.photonPassSeconds << ", \"wall_seconds\": " << wall
              << ", \"workers\": " << built.scene.settings.workerCount
              << ", \"photons_per_light\": " << built.scene.settings.photonsPerLight
              << ", \"lights\": " << built.lights.size()
              << ", \"non_delta_bounces\": " << (WorkerDebug::bounceKept() + WorkerDebug::bounceCulled())
              << ", \"splats\": " << WorkerDebug::splatTotal() << "}" << std::endl;
    return 0;
}

}  // namespace

int main(int argc, char** argv)
{
    try
    {
        const std::string mode = argc > 1 ? argv[1] : "";
        if (mode == "unit" && argc >= 3)
        {
            dumpTriangles(argv[2]);
            dumpBounds(argv[2]);
            dumpAnalytic(argv[2]);
            dumpBsdf(argv[2]);
            return 0;
        }
        if (mode == "emit" && argc >= 6)
        {
            return modeEmit(argv[2], argv[3], std::stoul(argv[4]), static_cast<uint32_t>(std::stoul(argv[5])));
        }
        if (mode == "walk" && argc >= 4)
        {
            return modeWalk(argv[2], argv[3], argc >= 5 ? std::stoul(argv[4]) : 4096);
        }
        if (mode == "render" && argc >= 4)
        {
            return modeRender(argv[2], argv[3]);
        }
        if (mode == "probes" && argc >= 4)
        {
            return modeProbes(argv[2], argv[3]);
        }
        if (mode == "bench" && argc >= 3)
        {
            return modeBench(argv[2], argc >= 4 ? std::stoi(argv[3]) : 0);
        }
        std::cerr << "usage: refdump unit|emit|walk|render|probes|bench ..." << std::endl;
        return 2;
    }
    catch (const std::exception& e)
    {
        std::cerr << "refdump: " << e.what() << std::endl;
        return 1;
    }
}
