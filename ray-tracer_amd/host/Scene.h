// Host-side scene model: the object graph, materials, meshes and render settings
// a scene file loads into.  Class and member names follow the reference's
// (include/Object.h, Camera.h, Volume.h, Light.h, Material.h, RenderSettings.h,
// SceneLoader.h) so code written against the reference's loader reads the same
// here; what differs is that these objects only DESCRIBE the scene — intersection,
// emission and scattering run on the GPU from the flattened description
// (Flatten.h), never on these classes.
#pragma once

#include "Math.h"

#include <algorithm>
#include <cstdint>
#include <filesystem>
#include <limits>
#include <map>
#include <random>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

// ------------------------------------------------------------------ objects
class Object
{
public:
    virtual ~Object() = default;

    void name(const std::string& name) { m_name = name; }
    const std::string& name() const { return m_name; }

    // World transform: parent chain composition (reference src/Object.cpp:21-43).
    Vector position() const
    {
        if (auto parent = m_parent.lock())
        {
            return parent->position() + (parent->rotation() * transform.position);
        }
        return transform.position;
    }
    Quaternion rotation() const
    {
        if (auto parent = m_parent.lock())
        {
            return parent->rotation() * transform.rotation;
        }
        return transform.rotation;
    }
    Vector forward() const { return rotation() * Vector(0, 0, 1); }

    static void setParent(const std::shared_ptr<Object>& child, const std::shared_ptr<Object>& parent)
    {
        if (parent)
        {
            parent->m_children.push_back(child);
            child->m_parent = parent;
        }
    }

    Transform transform;

private:
    std::string m_name;
    std::weak_ptr<Object> m_parent;
    std::vector<std::shared_ptr<Object>> m_children;
};

class Camera : public Object
{
public:
    enum class Projection
    {
        Perspective,
        Orthographic,
        RealLens
    };
    struct ExposureWindow
    {
        float start = -std::numeric_limits<float>::infinity();
        float end = std::numeric_limits<float>::infinity();
        bool contains(float t) const { return t >= start && t < end; }
    };
    static constexpr double kMeterCalibration = 12.5;

    size_t width() const { return m_width; }
    size_t height() const { return m_height; }
    double aspectRatio() const { return m_aspectRatio; }
    void verticalFieldOfView(double v) { m_verticalFieldOfView = v; }
    double verticalFieldOfView() const { return m_verticalFieldOfView; }
    void fNumber(double v) { m_fNumber = v; }
    double fNumber() const { return m_fNumber; }
    void shutterTime(double v) { m_shutterTime = v; }
    double shutterTime() const { return m_shutterTime; }
    void iso(double v) { m_iso = v; }
    double iso() const { return m_iso; }
    void projection(Projection p) { m_projection = p; }
    Projection projection() const { return m_projection; }
    void orthographicHeight(double v) { m_orthographicHeight = v; }
    double orthographicHeight() const { return m_orthographicHeight; }
    void apertureRadius(double v) { m_apertureRadius = v; }
    double apertureRadius() const { return m_apertureRadius; }
    void focusDistance(double v) { m_focusDistance = v; }
    double focusDistance() const { return m_focusDistance; }
    void focalLength(double v) { m_focalLength = v; }
    double focalLength() const { return m_focalLength; }
    void outputName(const std::string& n) { m_outputName = n; }
    const std::string& outputName() const { return m_outputName; }
    void bounceFilter(int b) { m_bounceFilter = b; }
    int bounceFilter() const { return m_bounceFilter; }
    void lightFilter(int l) { m_lightFilter = l; }
    int lightFilter() const { return m_lightFilter; }
    bool hasResolutionOverride() const { return m_hasResolutionOverride; }
    void setGlobalExposureWindow(const ExposureWindow& w) { m_globalExposureWindow = w; }
    ExposureWindow globalExposureWindow() const { return m_globalExposureWindow; }

    // reference src/Camera.cpp:143-156
    double effectiveApertureRadius() const
    {
        if (m_apertureRadius > 0.0) return m_apertureRadius;
        if (m_fNumber <= 0.0) return 0.0;
        return m_focalLength / (2.0 * m_fNumber);
    }
    // reference src/Camera.cpp:158-167
    double saturationLuminance() const
    {
        const double denom = m_shutterTime * m_iso;
        if (denom <= 0.0) return 0.0;
        return (m_fNumber * m_fNumber * kMeterCalibration) / denom;
    }
    // reference src/Camera.cpp:169-180
    void setFromRenderConfiguration(size_t width, size_t height)
    {
        if (width == 0 || height == 0)
        {
            throw std::runtime_error("Cannot configure Camera with 0 width or 0 height");
        }
        m_width = width;
        m_height = height;
        m_aspectRatio = static_cast<double>(m_width) / static_cast<double>(m_height);
    }
    void setResolution(size_t width, size_t height)
    {
        setFromRenderConfiguration(width, height);
        m_hasResolutionOverride = true;
    }

private:
    size_t m_width = 0, m_height = 0;
    double m_aspectRatio = 0.0;
    double m_verticalFieldOfView = 0.0;
    ExposureWindow m_globalExposureWindow;
    double m_fNumber = 8.0, m_shutterTime = 0.01, m_iso = 100.0;
    Projection m_projection = Projection::Perspective;
    double m_orthographicHeight = 100.0, m_apertureRadius = 0.0, m_focusDistance = 100.0, m_focalLength = 50.0;
    std::string m_outputName;
    bool m_hasResolutionOverride = false;
    int m_bounceFilter = -1, m_lightFilter = -1;
};

// ------------------------------------------------------------------ meshes
struct Triangle
{
    Vector a, b, c;
    Vector aNormal, bNormal, cNormal;
};

class Mesh
{
public:
    Mesh(const std::string& name, const std::vector<Triangle>& triangles) : m_name(name), m_triangles(triangles) {}
    void name(const std::string& n) { m_name = n; }
    const std::string& name() const { return m_name; }
    const std::vector<Triangle>& triangles() const { return m_triangles; }

private:
    std::string m_name;
    std::vector<Triangle> m_triangles;
};

// reference include/Library.h, src/Library.cpp
template <typename T>
class Library
{
public:
    void add(std::shared_ptr<T> data)
    {
        if (m_indexMap.count(data->name()) > 0)
        {
            throw std::runtime_error("Library already contains data named " + data->name());
        }
        m_indexMap.insert_or_assign(data->name(), m_contents.size());
        m_contents.push_back(data);
    }
    size_t indexForName(const std::string& name) const
    {
        if (m_indexMap.count(name) == 0)
        {
            throw std::runtime_error("Library does not contain item named " + name);
        }
        return m_indexMap.at(name);
    }
    std::shared_ptr<T> fetch(const std::string& name) const { return m_contents[indexForName(name)]; }
    std::shared_ptr<T> fetchByIndex(size_t index) const
    {
        if (index >= m_contents.size())
        {
            throw std::runtime_error("Library does not contain item with index " + std::to_string(index));
        }
        return m_contents[index];
    }
    size_t size() const { return m_contents.size(); }

private:
    std::vector<std::shared_ptr<T>> m_contents;
    std::unordered_map<std::string, size_t> m_indexMap;
};

class MeshLibrary : public Library<Mesh>
{
public:
    void addFromFile(const std::filesystem::path& path);  // ObjReader.cpp
};

// ------------------------------------------------------------------ materials
// reference include/RandomGenerator.h:6-22, src/RandomGenerator.cpp:3-17: the sequential draw source
// of the reference's host-side interfaces (the photon pass itself draws from per-photon counter streams).
class RandomGenerator
{
public:
    RandomGenerator() : m_generator(std::random_device{}()) {}
    explicit RandomGenerator(std::uint32_t seed) : m_generator(seed) {}
    double value(double scale = 1.0) { return m_distribution(m_generator) * scale; }

private:
    std::mt19937 m_generator;
    std::uniform_real_distribution<double> m_distribution{0.0, 1.0};
};

// The BSDF interface of reference include/Material.h:34-128: the virtuals a scene's materials are
// used through.  The photon pass runs on the device, so a material is, to the pass, the record
// deviceDescription() returns: the id of a device BSDF (RT_MATERIAL_* of include/rt_scene.h) and its
// parameters.  The host-side sample / evaluate / pdf evaluate that same record with the device
// library's function-level entry points (rt_bsdf_sample, rt_bsdf_evaluate: one-element batches), so
// that the interface and the pass share ONE implementation of every BSDF and there is no host BSDF
// code to fall out of step.  A new material that is a re-parameterisation of an existing BSDF
// overrides deviceDescription() only; a new BSDF takes a new id and its three device functions
// (sampleMaterial / evaluateMaterial / pdfMaterial in csrc/rt_device.h) as well.
struct BSDFSample
{
    Vector direction;
    Color weight{1.0f};
    double pdf = 0.0;
    bool isDelta = false;
    bool valid = false;
};

struct DeviceMaterial
{
    int bsdf = 0;        // RT_MATERIAL_LAMBERTIAN / _MIRROR / _DIELECTRIC / _MICROFACET
    Color color{1.0f};   // albedo or tint
    double alpha = 0.3;  // GGX roughness (microfacet)
    double ior = 1.5;    // index of refraction (dielectric)
};

class Material
{
public:
    enum class Kind
    {
        Lambertian = 0,
        Mirror = 1,
        Dielectric = 2,
        Microfacet = 3
    };
    Material(const std::string& name, Kind kind, const Color& color, double alpha = 0.3, double ior = 1.5)
        : m_name(name), m_kind(kind), m_color(color), m_alpha(alpha), m_ior(ior)
    {
    }
    virtual ~Material() = default;

    const std::string& name() const { return m_name; }
    Kind kind() const { return m_kind; }
    const Color& color() const { return m_color; }
    double alpha() const { return m_alpha; }
    double ior() const { return m_ior; }

    // What the device runs for this material (Flatten.cpp copies it into rt_material).
    virtual DeviceMaterial deviceDescription() const
    {
        return DeviceMaterial{static_cast<int>(m_kind), m_color, m_alpha, m_ior};
    }

    // reference include/Material.h:56-101.  sample() draws from `generator` what the BSDF consumes,
    // in the reference's order: Lambertian two draws, Mirror none, Dielectric one, Microfacet two
    // once the incident direction is known to arrive from the front (src/LambertianMaterial.cpp:57-58,
    // src/DielectricMaterial.cpp:118, src/MicrofacetMaterial.cpp:149-163).
    virtual BSDFSample sample(const Vector& incident, const Vector& normal, RandomGenerator& generator) const;
    virtual BSDFSample sampleMode(const Vector& incident, const Vector& normal, RandomGenerator& generator) const
    {
        return sample(incident, normal, generator);
    }
    virtual Color evaluate(const Vector& wi, const Vector& wo, const Vector& normal) const;
    virtual double pdf(const Vector& wi, const Vector& wo, const Vector& normal) const;
    virtual bool isDelta() const
    {
        const int bsdf = deviceDescription().bsdf;
        return bsdf == static_cast<int>(Kind::Mirror) || bsdf == static_cast<int>(Kind::Dielectric);
    }
    // one daughter per bounce: the live path of the reference (src/Worker.cpp:655-669)
    virtual size_t daughterPhotonCount() const { return 1; }

    // The same sample from explicit draws (u1, u2 as far as the BSDF consumes them).
    BSDFSample sample(const Vector& incident, const Vector& normal, double u1, double u2) const;

private:
    std::string m_name;
    Kind m_kind;
    Color m_color;
    double m_alpha;
    double m_ior;
};

class LambertianMaterial : public Material
{
public:
    LambertianMaterial(const std::string& name, const Color& albedo = {1.0f, 1.0f, 1.0f})
        : Material(name, Kind::Lambertian, albedo)
    {
    }
};
class MirrorMaterial : public Material
{
public:
    MirrorMaterial(const std::string& name, const Color& albedo = {1.0f, 1.0f, 1.0f})
        : Material(name, Kind::Mirror, albedo)
    {
    }
};
class DielectricMaterial : public Material
{
public:
    DielectricMaterial(const std::string& name, double ior = 1.5, const Color& tint = {1.0f, 1.0f, 1.0f})
        : Material(name, Kind::Dielectric, tint, 0.3, ior)
    {
    }
};
class MicrofacetMaterial : public Material
{
public:
    static constexpr double kDefaultRoughness = 0.3;
    // alpha is floored at 1e-3 (reference src/MicrofacetMaterial.cpp:132-137)
    MicrofacetMaterial(const std::string& name, const Color& albedo = {1.0f, 1.0f, 1.0f},
                       double roughness = kDefaultRoughness)
        : Material(name, Kind::Microfacet, albedo, roughness > 1e-3 ? roughness : 1e-3)
    {
    }
};

// reference src/MaterialLibrary.cpp:6-18 — nine built-ins occupy indices 0..8.
class MaterialLibrary : public Library<Material>
{
public:
    MaterialLibrary()
    {
        add(std::make_shared<LambertianMaterial>("Default"));
        add(std::make_shared<LambertianMaterial>("White", Color(1.0f, 1.0f, 1.0f)));
        add(std::make_shared<LambertianMaterial>("Black", Color(0.0f, 0.0f, 0.0f)));
        add(std::make_shared<LambertianMaterial>("Red", Color(1.0f, 0.0f, 0.0f)));
        add(std::make_shared<LambertianMaterial>("Yellow", Color(1.0f, 1.0f, 0.0f)));
        add(std::make_shared<LambertianMaterial>("Green", Color(0.0f, 1.0f, 0.0f)));
        add(std::make_shared<LambertianMaterial>("Cyan", Color(0.0f, 1.0f, 1.0f)));
        add(std::make_shared<LambertianMaterial>("Blue", Color(0.0f, 0.0f, 1.0f)));
        add(std::make_shared<LambertianMaterial>("Magenta", Color(1.0f, 0.0f, 1.0f)));
    }
};

// ------------------------------------------------------------------ volumes
class Volume : public Object
{
public:
    void materialIndex(size_t i) { m_materialIndex = i; }
    size_t materialIndex() const { return m_materialIndex; }

private:
    size_t m_materialIndex = static_cast<size_t>(-1);
};

class MeshVolume : public Volume
{
public:
    void mesh(std::shared_ptr<Mesh> m) { m_mesh = std::move(m); }
    std::shared_ptr<Mesh> mesh() const { return m_mesh; }

private:
    std::shared_ptr<Mesh> m_mesh;
};

class SphereVolume : public Volume
{
public:
    void center(const Vector& c) { m_center = c; }
    Vector center() const { return m_center; }
    void radius(double r) { m_radius = r; }
    double radius() const { return m_radius; }

private:
    Vector m_center{0, 0, 0};
    double m_radius = 1.0;
};

// reference include/Quad.h, src/Quad.cpp:21-48
struct Quad
{
    Vector origin{0, 0, 0}, edgeU{1, 0, 0}, edgeV{0, 1, 0};
    Vector normal, invU, invV;
    Quad() { update(); }
    Quad(const Vector& o, const Vector& u, const Vector& v) : origin(o), edgeU(u), edgeV(v) { update(); }
    static Quad fromCorners(const Vector& c0, const Vector& c1, const Vector&, const Vector& c3)
    {
        return Quad(c0, c1 - c0, c3 - c0);
    }
    void update()
    {
        normal = Vector::cross(edgeU, edgeV).normalized();
        const double g11 = Vector::dot(edgeU, edgeU), g22 = Vector::dot(edgeV, edgeV), g12 = Vector::dot(edgeU, edgeV);
        const double det = g11 * g22 - g12 * g12;
        if (det == 0.0)
        {
            invU = Vector{};
            invV = Vector{};
            return;
        }
        const double invDet = 1.0 / det;
        invU = (g22 * edgeU - g12 * edgeV) * invDet;
        invV = (g11 * edgeV - g12 * edgeU) * invDet;
    }
};

class QuadVolume : public Volume
{
public:
    void quad(const Quad& q) { m_quad = q; }
    const Quad& quad() const { return m_quad; }

private:
    Quad m_quad;
};

// reference src/PlaneVolume.cpp:3-15: Plane({0,0,1},{1,0,0}) through the origin,
// i.e. normal = normalize(cross((0,0,1),(1,0,0))) = +Y and dot = 0.
class PlaneVolume : public Volume
{
public:
    Vector normal() const { return Vector::cross(Vector(0, 0, 1), Vector(1, 0, 0)).normalized(); }
    double planeDot() const { return 0.0; }
};

// ------------------------------------------------------------------ lights
class Light : public Object
{
public:
    void color(const Color& c) { m_color = c; }
    Color color() const { return m_color; }
    void intensityCandela(double i) { m_intensityCandela = i; }
    double intensityCandela() const { return m_intensityCandela; }
    void brightness(double b) { m_intensityCandela = b; }
    double brightness() const { return m_intensityCandela; }
    // Phi = I * Omega (reference src/Light.cpp:41-46)
    virtual double luminousFlux() const { return m_intensityCandela * m_emissionSolidAngle; }

protected:
    Color m_color;
    double m_intensityCandela = 0.0;
    double m_emissionSolidAngle = 0.0;
};

class OmniLight : public Light
{
public:
    OmniLight() { m_emissionSolidAngle = Utility::pi * 4.0; }
    void innerRadius(double r) { m_innerRadius = r; }
    double innerRadius() const { return m_innerRadius; }

private:
    double m_innerRadius = 0.0;
};

class SpotLight : public Light
{
public:
    void innerRadius(double r) { m_innerRadius = r; }
    double innerRadius() const { return m_innerRadius; }
    // reference src/SpotLight.cpp:21-44
    void angle(double angle)
    {
        m_angle = std::min(std::max(0.0, angle), Utility::pi2);
        double coneAngle = m_angle;
        if (m_angle > Utility::pi)
        {
            coneAngle = Utility::pi2 - m_angle;
        }
        const double coneHeight = std::sin(coneAngle / 2.0);
        const double capHeight = 1.0 - coneHeight;
        m_emissionSolidAngle = Utility::pi2 * capHeight;
        if (m_angle > Utility::pi)
        {
            m_emissionSolidAngle = Utility::pi4 - m_emissionSolidAngle;
        }
        m_maxAngle = m_angle / 2.0;
    }
    double angle() const { return m_angle; }
    double maxAngle() const { return m_maxAngle; }

private:
    double m_innerRadius = 0.0, m_angle = 0.0, m_maxAngle = 0.0;
};

class ParallelLight : public Light
{
public:
    // reference src/ParallelLight.cpp:13-19
    void radius(double r)
    {
        m_radius = r;
        m_emissionSolidAngle = Utility::pi * m_radius * m_radius;
    }
    double radius() const { return m_radius; }

private:
    double m_radius = 0.0;
};

class AreaLight : public Light
{
public:
    enum class Shape
    {
        Square,
        Disc
    };
    AreaLight() { m_emissionSolidAngle = Utility::pi; }
    void shape(Shape s) { m_shape = s; }
    Shape shape() const { return m_shape; }
    void width(double w) { m_width = w; }
    double width() const { return m_width; }
    void height(double h) { m_height = h; }
    double height() const { return m_height; }
    void radius(double r) { m_radius = r; }
    double radius() const { return m_radius; }
    void luminousFluxOverride(double f) { m_luminousFluxOverride = f; }
    double luminousFluxOverride() const { return m_luminousFluxOverride; }
    // reference src/AreaLight.cpp:67-78
    double luminousFlux() const override
    {
        if (m_luminousFluxOverride > 0.0) return m_luminousFluxOverride;
        return Light::luminousFlux();
    }

private:
    Shape m_shape = Shape::Square;
    double m_width = 0.0, m_height = 0.0, m_radius = 0.0, m_luminousFluxOverride = 0.0;
};

// ------------------------------------------------------------------ settings
// reference include/RenderSettings.h:15-193 (same names, same defaults)
struct RenderSettings
{
    static constexpr size_t kMillion = 1000000;
    size_t photonQueueSize = 20 * kMillion;
    size_t photonsPerLight = 20 * kMillion;
    size_t workerCount = 32;
    size_t fetchSize = 100000;
    size_t imageWidth = 1080;
    size_t imageHeight = 1080;
    size_t startFrame = 0;
    size_t endFrame = 0;
    size_t bounceThreshold = 1;
    double frameRate = 24.0;
    double shutterTime = 0.0;
    double frameOffset = 0.0;
    double frameTime = 0.0;
    double terminationThreshold = 1.0;
    double hashGridCellSize = 1.0;
    double densityCellScale = 1.0;
    double splatMinRadiusScale = 0.5;
    double splatLuminanceClamp = 0.0;
    bool useProbeGather = true;
    size_t bounceStoreCapacity = 40 * kMillion;
    double probeKeepRadiusScale = 1.5;
    size_t probeSubSample = 1;
    int probeTimeSlices = 5;
    int cameraTimeSamples = 16;
    static constexpr std::uint32_t kUnseeded = 0xFFFFFFFFu;
    std::uint32_t seed = kUnseeded;
    bool deterministic = false;
};

// ------------------------------------------------------------------ animation
// reference include/Property.h:37-51 and include/AnimationQuery.h:107-163: keyframed world
// position and spin angle of an object, keyed by object name.
template <typename T>
struct Keyframe
{
    double time = 0.0;
    T value{};
    T inTangent{};
    T outTangent{};
    bool useAutoTangent = true;
};

struct AnimatedObject
{
    std::vector<Keyframe<Vector>> position;   // empty: hold constantPosition
    Vector constantPosition;                  // the object's scene-load (local) position
    bool hasRotation = false;
    Vector rotationAxis{0.0, 1.0, 0.0};
    std::vector<Keyframe<double>> rotationAngle;  // radians about rotationAxis
    Quaternion baseRotation;                  // the object's scene-load (local) rotation
};

struct LoadedScene
{
    std::vector<std::shared_ptr<Object>> objects;
    std::shared_ptr<Camera> camera;
    std::vector<std::shared_ptr<Camera>> cameras;
    std::shared_ptr<MaterialLibrary> materialLibrary;
    std::shared_ptr<MeshLibrary> meshLibrary;
    RenderSettings settings;
    std::filesystem::path renderPath;
    std::string renderName;
    std::map<std::string, AnimatedObject> animation;  // by object name
};
