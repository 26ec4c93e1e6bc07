// Scene JSON -> LoadedScene.  Accepts exactly the key set the reference's loader
// reads (reference src/SceneLoader.cpp:781-869 settings, :873-944 materials,
// :957-971 meshes, :357-710 object types) with the same defaults and the same
// errors; keys the reference accepts-and-ignores are ignored here too.

#include "SceneLoader.h"

#include "Json.h"

#include <iostream>

namespace
{

using Json::Value;

template <typename T>
T convert(const Value& v);
template <>
double convert<double>(const Value& v) { return v.asDouble(); }
template <>
size_t convert<size_t>(const Value& v) { return v.asSize(); }
template <>
int convert<int>(const Value& v) { return static_cast<int>(v.asInteger()); }
template <>
std::uint32_t convert<std::uint32_t>(const Value& v) { return static_cast<std::uint32_t>(v.asInteger()); }
template <>
bool convert<bool>(const Value& v) { return v.asBool(); }
template <>
std::string convert<std::string>(const Value& v) { return v.asString(); }

template <typename T>
void setFromJsonIfPresent(T& output, const Value& container, const std::string& key, bool log = false)
{
    if (container.contains(key))
    {
        output = convert<T>(container.at(key));
        if (log)
        {
            std::cout << "  Setting " << key << " to " << output << std::endl;
        }
    }
}

Vector parseVectorFromJson(const Value& v)
{
    const char* message = "Vector object must be an array with 3 or 4 number elements";
    if (!v.isArray() || v.size() < 3 || v.size() > 4)
    {
        throw std::runtime_error(message);
    }
    for (const Value& e : v.elements())
    {
        if (!e.isNumber())
        {
            throw std::runtime_error(message);
        }
    }
    return Vector{v.at(0).asDouble(), v.at(1).asDouble(), v.at(2).asDouble()};
}

void setVectorFromJsonIfPresent(Vector& output, const Value& container, const std::string& key)
{
    if (container.contains(key))
    {
        output = parseVectorFromJson(container.at(key));
    }
}

Quaternion parseRotationFromJson(const Value& v)
{
    const char* message = "Rotation object must contain a $type string and a Vector $value object";
    if (!v.isObject() || !v.contains("$type") || !v.contains("$value") || !v.at("$type").isString())
    {
        throw std::runtime_error(message);
    }
    const std::string type = v.at("$type").asString();
    const Vector value = parseVectorFromJson(v.at("$value"));
    if (type == "PitchYawRollDegrees")
    {
        return Quaternion::fromPitchYawRoll(Utility::radians(value.x), Utility::radians(value.y),
                                            Utility::radians(value.z));
    }
    if (type == "PitchYawRollRadians")
    {
        return Quaternion::fromPitchYawRoll(value.x, value.y, value.z);
    }
    throw std::runtime_error(std::string("Unsupported rotation type: ") + type);
}

Color parseColorFromJson(const Value& v)
{
    const char* message = "Color object must be an array with 1 or 3 number elements";
    if (!v.isArray() || (v.size() != 1 && v.size() != 3))
    {
        throw std::runtime_error(message);
    }
    for (const Value& e : v.elements())
    {
        if (!e.isNumber())
        {
            throw std::runtime_error(message);
        }
    }
    if (v.size() == 1)
    {
        const float grey = v.at(0).asFloat();
        return {grey, grey, grey};
    }
    return {v.at(0).asFloat(), v.at(1).asFloat(), v.at(2).asFloat()};
}

class ObjectFactory
{
public:
    ObjectFactory(std::shared_ptr<MaterialLibrary> materials, std::shared_ptr<MeshLibrary> meshes)
        : m_materialLibrary(std::move(materials)), m_meshLibrary(std::move(meshes))
    {
    }

    std::shared_ptr<Object> createObjectFromJson(const Value& j)
    {
        std::string type = "Object";
        setFromJsonIfPresent(type, j, "$type");
        if (type == "Object")
        {
            auto o = std::make_shared<Object>();
            setObject(*o, j);
            return o;
        }
        if (type == "Camera")
        {
            auto o = std::make_shared<Camera>();
            setCamera(*o, j);
            return o;
        }
        if (type == "MeshVolume")
        {
            auto o = std::make_shared<MeshVolume>();
            setVolume(*o, j);
            std::string mesh = "Default";
            setFromJsonIfPresent(mesh, j, "$mesh");
            o->mesh(m_meshLibrary->fetch(mesh));
            return o;
        }
        if (type == "PlaneVolume")
        {
            auto o = std::make_shared<PlaneVolume>();
            setVolume(*o, j);
            return o;
        }
        if (type == "SphereVolume")
        {
            auto o = std::make_shared<SphereVolume>();
            setVolume(*o, j);
            Vector center = o->center();
            setVectorFromJsonIfPresent(center, j, "$center");
            o->center(center);
            double radius = o->radius();
            setFromJsonIfPresent(radius, j, "$radius");
            o->radius(radius);
            return o;
        }
        if (type == "Quad" || type == "QuadVolume")
        {
            auto o = std::make_shared<QuadVolume>();
            setVolume(*o, j);
            if (j.contains("$corners"))
            {
                const Value& corners = j.at("$corners");
                if (!corners.isArray() || corners.size() != 4)
                {
                    throw std::runtime_error("Quad \"$corners\" must be an array of exactly 4 vectors");
                }
                o->quad(Quad::fromCorners(parseVectorFromJson(corners.at(0)), parseVectorFromJson(corners.at(1)),
                                          parseVectorFromJson(corners.at(2)), parseVectorFromJson(corners.at(3))));
            }
            else
            {
                Vector origin = o->quad().origin, edgeU = o->quad().edgeU, edgeV = o->quad().edgeV;
                setVectorFromJsonIfPresent(origin, j, "$origin");
                setVectorFromJsonIfPresent(edgeU, j, "$edgeU");
                setVectorFromJsonIfPresent(edgeV, j, "$edgeV");
                o->quad(Quad(origin, edgeU, edgeV));
            }
            return o;
        }
        if (type == "OmniLight")
        {
            auto o = std::make_shared<OmniLight>();
            setLight(*o, j);
            double innerRadius = 0.0;
            setFromJsonIfPresent(innerRadius, j, "$innerRadius");
            o->innerRadius(innerRadius);
            return o;
        }
        if (type == "ParallelLight")
        {
            auto o = std::make_shared<ParallelLight>();
            setLight(*o, j);
            double radius = 0.0;
            setFromJsonIfPresent(radius, j, "$radius");
            o->radius(radius);
            return o;
        }
        if (type == "SpotLight")
        {
            auto o = std::make_shared<SpotLight>();
            setLight(*o, j);
            double innerRadius = 0.0;
            setFromJsonIfPresent(innerRadius, j, "$innerRadius");
            o->innerRadius(innerRadius);
            double angle = 0.0;
            setFromJsonIfPresent(angle, j, "$angle");
            o->angle(angle);
            return o;
        }
        if (type == "AreaLight")
        {
            auto o = std::make_shared<AreaLight>();
            setLight(*o, j);
            std::string shape = "Square";
            setFromJsonIfPresent(shape, j, "$shape");
            o->shape(shape == "Disc" ? AreaLight::Shape::Disc : AreaLight::Shape::Square);
            double size = 0.0;
            setFromJsonIfPresent(size, j, "$size");
            double width = size, height = size;
            setFromJsonIfPresent(width, j, "$width");
            setFromJsonIfPresent(height, j, "$height");
            o->width(width);
            o->height(height);
            double radius = 0.0;
            setFromJsonIfPresent(radius, j, "$radius");
            o->radius(radius);
            double flux = 0.0;
            setFromJsonIfPresent(flux, j, "$luminousFlux");
            o->luminousFluxOverride(flux);
            return o;
        }
        throw std::runtime_error(std::string("Unrecognized object type: ") + type);
    }

private:
    void setObject(Object& o, const Value& j)
    {
        setVectorFromJsonIfPresent(o.transform.position, j, "$position");
        if (j.contains("$rotation"))
        {
            o.transform.rotation = parseRotationFromJson(j.at("$rotation"));
        }
    }

    void setCamera(Camera& o, const Value& j)
    {
        setObject(o, j);
        double vfov = 0;
        setFromJsonIfPresent(vfov, j, "$verticalFieldOfView");
        o.verticalFieldOfView(vfov);

        std::string projection;
        setFromJsonIfPresent(projection, j, "$projection");
        if (projection == "orthographic") o.projection(Camera::Projection::Orthographic);
        else if (projection == "reallens") o.projection(Camera::Projection::RealLens);
        else if (projection.empty() || projection == "perspective") o.projection(Camera::Projection::Perspective);
        else
        {
            throw std::runtime_error("Unknown $projection \"" + projection +
                                     "\" (expected perspective | orthographic | reallens)");
        }
        double v = o.orthographicHeight();
        setFromJsonIfPresent(v, j, "$orthoHeight");
        o.orthographicHeight(v);
        v = o.apertureRadius();
        setFromJsonIfPresent(v, j, "$apertureRadius");
        o.apertureRadius(v);
        v = o.focusDistance();
        setFromJsonIfPresent(v, j, "$focusDistance");
        o.focusDistance(v);
        v = o.focalLength();
        setFromJsonIfPresent(v, j, "$focalLength");
        o.focalLength(v);
        v = o.fNumber();
        setFromJsonIfPresent(v, j, "$fNumber");
        o.fNumber(v);
        v = o.shutterTime();
        setFromJsonIfPresent(v, j, "$shutterTime");
        o.shutterTime(v);
        v = o.iso();
        setFromJsonIfPresent(v, j, "$iso");
        o.iso(v);
        if (j.contains("$exposureWindow"))
        {
            const Value& window = j.at("$exposureWindow");
            if (window.isArray() && window.size() == 2)
            {
                Camera::ExposureWindow w;
                w.start = window.at(0).asFloat();
                w.end = window.at(1).asFloat();
                o.setGlobalExposureWindow(w);
            }
        }
        std::string outputName;
        setFromJsonIfPresent(outputName, j, "$outputName");
        o.outputName(outputName);
        if (j.contains("$width") && j.contains("$height"))
        {
            o.setResolution(j.at("$width").asSize(), j.at("$height").asSize());
        }
        int filter = -1;
        setFromJsonIfPresent(filter, j, "$bounceFilter");
        o.bounceFilter(filter);
        filter = -1;
        setFromJsonIfPresent(filter, j, "$lightFilter");
        o.lightFilter(filter);
    }

    void setVolume(Volume& o, const Value& j)
    {
        setObject(o, j);
        std::string material = "Default";
        setFromJsonIfPresent(material, j, "$material");
        o.materialIndex(m_materialLibrary->indexForName(material));
    }

    void setLight(Light& o, const Value& j)
    {
        setObject(o, j);
        Color color;
        if (j.contains("$color"))
        {
            color = parseColorFromJson(j.at("$color"));
        }
        o.color(color);
        double intensity = 0.0;
        setFromJsonIfPresent(intensity, j, "$brightness");
        setFromJsonIfPresent(intensity, j, "$intensityCandela");
        o.intensityCandela(intensity);
    }

    std::shared_ptr<MaterialLibrary> m_materialLibrary;
    std::shared_ptr<MeshLibrary> m_meshLibrary;
};

// reference src/SceneLoader.cpp:717-746: parent first, then children in key order.
// reference src/SceneLoader.cpp:202-254: {"t": seconds, "value": number | [x, y, z]} entries,
// scalar tracks optionally with explicit per-second tangents.  Keyframes stay sorted by time
// (a new key goes after the keys of equal time, reference include/Property.h:73-79).
template <typename T>
void insertKeyframe(std::vector<Keyframe<T>>& keys, const Keyframe<T>& key)
{
    auto at = std::upper_bound(keys.begin(), keys.end(), key,
                               [](const Keyframe<T>& a, const Keyframe<T>& b) { return a.time < b.time; });
    keys.insert(at, key);
}

void requireKeyframe(const Value& entry)
{
    if (!entry.isObject() || !entry.contains("t") || !entry.contains("value"))
    {
        throw std::runtime_error("Each keyframe must be an object with 't' and 'value'");
    }
}

std::vector<Keyframe<double>> parseScalarKeyframes(const Value& list)
{
    if (!list.isArray())
    {
        throw std::runtime_error("$animation keyframe list must be an array");
    }
    std::vector<Keyframe<double>> keys;
    for (const Value& entry : list.elements())
    {
        requireKeyframe(entry);
        Keyframe<double> key;
        key.time = entry.at("t").asDouble();
        key.value = entry.at("value").asDouble();
        if (entry.contains("inTangent") || entry.contains("outTangent"))
        {
            key.useAutoTangent = false;
            key.inTangent = entry.contains("inTangent") ? entry.at("inTangent").asDouble() : 0.0;
            key.outTangent = entry.contains("outTangent") ? entry.at("outTangent").asDouble() : 0.0;
        }
        insertKeyframe(keys, key);
    }
    return keys;
}

std::vector<Keyframe<Vector>> parseVectorKeyframes(const Value& list)
{
    if (!list.isArray())
    {
        throw std::runtime_error("$animation keyframe list must be an array");
    }
    std::vector<Keyframe<Vector>> keys;
    for (const Value& entry : list.elements())
    {
        requireKeyframe(entry);
        Keyframe<Vector> key;
        key.time = entry.at("t").asDouble();
        key.value = parseVectorFromJson(entry.at("value"));
        insertKeyframe(keys, key);
    }
    return keys;
}

// reference src/SceneLoader.cpp:257-303
AnimatedObject parseAnimationBlock(const Value& j, const Transform& loadTransform)
{
    AnimatedObject animated;
    animated.baseRotation = loadTransform.rotation;
    animated.constantPosition = loadTransform.position;
    if (j.contains("$rotationAxis"))
    {
        animated.rotationAxis = parseVectorFromJson(j.at("$rotationAxis")).normalized();
    }
    if (j.contains("$rotation"))
    {
        animated.rotationAngle = parseScalarKeyframes(j.at("$rotation"));
        animated.hasRotation = true;
    }
    if (j.contains("$position"))
    {
        animated.position = parseVectorKeyframes(j.at("$position"));
    }
    return animated;
}

std::vector<std::shared_ptr<Object>> parseObjectFromJson(const std::string& name, const Value& j,
                                                         const std::shared_ptr<Object>& parent, ObjectFactory& factory,
                                                         std::map<std::string, AnimatedObject>& animation)
{
    std::vector<std::shared_ptr<Object>> parsed;
    std::shared_ptr<Object> object = factory.createObjectFromJson(j);
    parsed.push_back(object);
    object->name(name);
    Object::setParent(object, parent);
    if (j.contains("$animation"))
    {
        animation[name] = parseAnimationBlock(j.at("$animation"), object->transform);
    }
    for (const auto& [childName, child] : j.items())
    {
        if (!childName.empty() && childName[0] != '$')
        {
            auto children = parseObjectFromJson(childName, child, object, factory, animation);
            parsed.insert(parsed.end(), children.begin(), children.end());
        }
    }
    return parsed;
}

std::filesystem::path resolvePath(const std::filesystem::path& path, const std::filesystem::path& base)
{
    return path.is_relative() ? std::filesystem::absolute(base / path) : path;
}

// reference src/SceneLoader.cpp:57-77 states these as contracts that are inert in a
// release build; a scene that violates them cannot be traced, so they throw here.
void validateRenderSettings(const RenderSettings& s)
{
    auto require = [](bool ok, const char* what) {
        if (!ok)
        {
            throw std::runtime_error(std::string("Invalid render configuration: ") + what);
        }
    };
    require(s.imageWidth > 0, "$width must be > 0");
    require(s.imageHeight > 0, "$height must be > 0");
    require(s.photonsPerLight > 0, "$photonsPerLight must be > 0");
    require(s.workerCount > 0, "$workerCount must be >= 1");
    require(s.fetchSize > 0, "$fetchSize must be > 0");
    require(s.photonQueueSize > 0, "$photonQueueSize must be > 0");
    require(s.bounceThreshold >= 1, "$bounceThreshold must be >= 1");
    require(s.terminationThreshold >= 0.0, "$terminationThreshold must be >= 0");
    require(s.densityCellScale > 0.0, "$densityCellScale must be > 0");
    require(s.splatMinRadiusScale >= 0.0, "$splatMinRadiusScale must be >= 0");
    require(s.splatLuminanceClamp >= 0.0, "$splatLuminanceClamp must be >= 0");
}

}  // namespace

namespace SceneLoader
{

LoadedScene loadFromFile(const std::filesystem::path& scenePath, bool logToStdout)
{
    const std::filesystem::path absoluteScenePath = std::filesystem::absolute(scenePath);
    if (logToStdout)
    {
        std::cout << "Loading project file " << absoluteScenePath.generic_string() << std::endl;
    }
    const Value json = Value::parseFile(absoluteScenePath.string());
    if (!json.isObject())
    {
        throw std::runtime_error("Scene file must contain a JSON object");
    }

    LoadedScene scene;
    RenderSettings& settings = scene.settings;

    if (json.contains("$workerConfiguration"))
    {
        const Value& w = json.at("$workerConfiguration");
        setFromJsonIfPresent(settings.workerCount, w, "$workerCount", logToStdout);
        setFromJsonIfPresent(settings.fetchSize, w, "$fetchSize", logToStdout);
        setFromJsonIfPresent(settings.photonQueueSize, w, "$photonQueueSize", logToStdout);
        // $hitQueueSize / $emittingQueueSize / $finalQueueSize: accepted and ignored.
    }

    if (json.contains("$renderConfiguration"))
    {
        const Value& r = json.at("$renderConfiguration");
        setFromJsonIfPresent(settings.imageWidth, r, "$width", logToStdout);
        setFromJsonIfPresent(settings.imageHeight, r, "$height", logToStdout);
        setFromJsonIfPresent(settings.photonsPerLight, r, "$photonsPerLight", logToStdout);
        setFromJsonIfPresent(settings.startFrame, r, "$startFrame", logToStdout);
        setFromJsonIfPresent(settings.endFrame, r, "$endFrame", logToStdout);
        setFromJsonIfPresent(settings.bounceThreshold, r, "$bounceThreshold", logToStdout);
        setFromJsonIfPresent(settings.frameRate, r, "$frameRate", logToStdout);
        setFromJsonIfPresent(settings.shutterTime, r, "$shutterTime", logToStdout);
        setFromJsonIfPresent(settings.frameOffset, r, "$frameOffset", logToStdout);
        setFromJsonIfPresent(settings.terminationThreshold, r, "$terminationThreshold", logToStdout);
        setFromJsonIfPresent(settings.hashGridCellSize, r, "$hashGridCellSize", logToStdout);
        setFromJsonIfPresent(settings.densityCellScale, r, "$densityCellScale", logToStdout);
        setFromJsonIfPresent(settings.splatMinRadiusScale, r, "$splatMinRadiusScale", logToStdout);
        setFromJsonIfPresent(settings.splatLuminanceClamp, r, "$splatLuminanceClamp", logToStdout);
        setFromJsonIfPresent(settings.useProbeGather, r, "$probeGather", logToStdout);
        setFromJsonIfPresent(settings.bounceStoreCapacity, r, "$bounceStoreCapacity", logToStdout);
        setFromJsonIfPresent(settings.probeKeepRadiusScale, r, "$probeKeepRadiusScale", logToStdout);
        setFromJsonIfPresent(settings.probeSubSample, r, "$probeSubSample", logToStdout);
        setFromJsonIfPresent(settings.probeTimeSlices, r, "$probeTimeSlices", logToStdout);
        setFromJsonIfPresent(settings.cameraTimeSamples, r, "$cameraTimeSamples", logToStdout);
        setFromJsonIfPresent(settings.seed, r, "$seed", logToStdout);
        setFromJsonIfPresent(settings.deterministic, r, "$deterministic", logToStdout);
        if (r.contains("$renderPath"))
        {
            scene.renderPath = resolvePath(r.at("$renderPath").asString(), absoluteScenePath.parent_path());
        }
        if (r.contains("$renderName"))
        {
            scene.renderName = r.at("$renderName").asString();
        }
    }

    scene.materialLibrary = std::make_shared<MaterialLibrary>();
    if (json.contains("$materials"))
    {
        for (const auto& [name, material] : json.at("$materials").items())
        {
            if (!material.contains("$type"))
            {
                throw std::runtime_error(std::string("Material \"") + name + std::string("\" requires $type"));
            }
            const std::string type = material.at("$type").asString();
            Color color;
            if (material.contains("$color"))
            {
                const Value& c = material.at("$color");
                if (c.size() == 1) color = Color(c.at(0).asFloat());
                else if (c.size() == 3) color = Color(c.at(0).asFloat(), c.at(1).asFloat(), c.at(2).asFloat());
            }
            if (type == "Diffuse" || type == "Lambertian")
            {
                scene.materialLibrary->add(std::make_shared<LambertianMaterial>(name, color));
            }
            else if (type == "Mirror")
            {
                scene.materialLibrary->add(std::make_shared<MirrorMaterial>(name, color));
            }
            else if (type == "Microfacet" || type == "GGX")
            {
                double roughness = MicrofacetMaterial::kDefaultRoughness;
                if (material.contains("$roughness")) roughness = material.at("$roughness").asDouble();
                scene.materialLibrary->add(std::make_shared<MicrofacetMaterial>(name, color, roughness));
            }
            else if (type == "Glass" || type == "Dielectric")
            {
                double ior = 1.5;
                if (material.contains("$ior")) ior = material.at("$ior").asDouble();
                const Color tint = material.contains("$color") ? color : Color{1.0f, 1.0f, 1.0f};
                scene.materialLibrary->add(std::make_shared<DielectricMaterial>(name, ior, tint));
            }
            else if (logToStdout)
            {
                std::cout << "WARNING: Unsupported material type \"" << type << "\"" << std::endl;
            }
        }
    }

    scene.meshLibrary = std::make_shared<MeshLibrary>();
    if (json.contains("$meshes"))
    {
        for (const Value& meshFilename : json.at("$meshes").elements())
        {
            scene.meshLibrary->addFromFile(resolvePath(meshFilename.asString(), absoluteScenePath.parent_path()));
        }
    }

    ObjectFactory factory{scene.materialLibrary, scene.meshLibrary};
    if (json.contains("$scene"))
    {
        for (const auto& [name, object] : json.at("$scene").items())
        {
            auto parsed = parseObjectFromJson(name, object, nullptr, factory, scene.animation);
            scene.objects.insert(scene.objects.end(), parsed.begin(), parsed.end());
        }
    }

    for (auto& object : scene.objects)
    {
        if (auto camera = std::dynamic_pointer_cast<Camera>(object))
        {
            if (!camera->hasResolutionOverride())
            {
                camera->setFromRenderConfiguration(settings.imageWidth, settings.imageHeight);
            }
            scene.cameras.push_back(camera);
        }
    }
    if (!scene.cameras.empty())
    {
        scene.camera = scene.cameras.front();
    }
    if (logToStdout)
    {
        std::cout << "Parsed " << scene.objects.size() << " object(s) from json" << std::endl;
    }
    validateRenderSettings(settings);
    return scene;
}

}  // namespace SceneLoader
