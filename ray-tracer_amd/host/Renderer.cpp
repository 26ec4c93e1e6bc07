#include "Renderer.h"

#include "Flatten.h"
#include "FrameSetup.h"
#include "rt_amd.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>
#include <iostream>
#include <limits>
#include <stdexcept>

namespace
{

int g_device = 0;

void check(int32_t code)
{
    if (code != RT_OK)
    {
        throw std::runtime_error(std::string("rt_amd: ") + rt_last_error());
    }
}

struct TracerHandle
{
    rt_tracer* ptr = nullptr;
    ~TracerHandle()
    {
        if (ptr) rt_destroy(ptr);
    }
};

// One emitter patch: the lit face of an area light (reference src/EmitterPatch.cpp:15-53,
// AreaLight::surfaceRadiance src/AreaLight.cpp:80-104), from the flattened light.
struct Patch
{
    Vector center, right, up, normal;
    Color radiance;
    double halfWidth = 0.0, halfHeight = 0.0, radius = 0.0;
    bool isDisc = false;
};

std::vector<Patch> collectEmitterPatches(const rt_scene& scene)
{
    std::vector<Patch> patches;
    for (int64_t i = 0; i < scene.light_count; ++i)
    {
        const rt_light& l = scene.lights[i];
        if (l.type != RT_LIGHT_AREA)
        {
            continue;
        }
        const double area = l.shape == 1 ? Utility::pi * l.radius * l.radius : l.width * l.height;
        Color radiance;
        if (area > 0.0)
        {
            const double exitance = l.luminous_flux / area;
            radiance = Color{l.color[0], l.color[1], l.color[2]} * static_cast<float>(exitance / Utility::pi);
        }
        if (radiance.red <= 0.0f && radiance.green <= 0.0f && radiance.blue <= 0.0f)
        {
            continue;
        }
        const Quaternion q{l.rotation[0], l.rotation[1], l.rotation[2], l.rotation[3]};
        Patch p;
        p.center = Vector{l.position[0], l.position[1], l.position[2]};
        p.right = (q * Vector{1.0, 0.0, 0.0}).normalized();
        p.up = (q * Vector{0.0, 1.0, 0.0}).normalized();
        p.normal = (q * Vector{0.0, 0.0, 1.0}).normalized();
        p.radiance = radiance;
        if (l.shape == 1)
        {
            p.isDisc = true;
            p.radius = l.radius;
        }
        else
        {
            p.halfWidth = l.width * 0.5;
            p.halfHeight = l.height * 0.5;
        }
        patches.push_back(p);
    }
    return patches;
}

// ProbeGather::depositEmitters (reference src/ProbeGather.cpp:464-550): tile every patch
// with cell-centred deposits, keep those near a probe (device keep-test), append them to
// the store ahead of the photon pass.  Returns the number kept.
size_t depositEmitters(rt_tracer* tracer, const rt_scene& scene, double depositSpacing)
{
    const std::vector<Patch> patches = collectEmitterPatches(scene);
    const double spacing = std::max(depositSpacing, 1e-6);
    size_t keptTotal = 0;
    for (const Patch& patch : patches)
    {
        const double extentU = patch.isDisc ? patch.radius : patch.halfWidth;
        const double extentV = patch.isDisc ? patch.radius : patch.halfHeight;
        if (extentU <= 0.0 || extentV <= 0.0)
        {
            continue;
        }
        const int countU = std::max(1, static_cast<int>(std::ceil(2.0 * extentU / spacing)));
        const int countV = std::max(1, static_cast<int>(std::ceil(2.0 * extentV / spacing)));
        const double stepU = (2.0 * extentU) / countU;
        const double stepV = (2.0 * extentV) / countV;
        const double area = patch.isDisc ? (Utility::pi * patch.radius * patch.radius)
                                         : (2.0 * patch.halfWidth) * (2.0 * patch.halfHeight);
        std::vector<double> samples;
        for (int iu = 0; iu < countU; ++iu)
        {
            const double u = -extentU + (iu + 0.5) * stepU;
            for (int iv = 0; iv < countV; ++iv)
            {
                const double v = -extentV + (iv + 0.5) * stepV;
                if (patch.isDisc && (u * u + v * v) > (patch.radius * patch.radius))
                {
                    continue;
                }
                const Vector p = patch.center + patch.right * u + patch.up * v;
                samples.insert(samples.end(), {p.x, p.y, p.z});
            }
        }
        const size_t count = samples.size() / 3;
        if (count == 0)
        {
            continue;
        }
        const double perDepositScale = (Utility::pi * area) / (4.0 * static_cast<double>(count));
        const Color power = patch.radiance * static_cast<float>(perDepositScale);
        std::vector<uint8_t> keep(count, 0);
        check(rt_probe_keep_test(tracer, static_cast<int64_t>(count), samples.data(), keep.data()));
        std::vector<float> records;
        for (size_t i = 0; i < count; ++i)
        {
            if (!keep[i])
            {
                continue;
            }
            const float n[3] = {static_cast<float>(patch.normal.x), static_cast<float>(patch.normal.y),
                                static_cast<float>(patch.normal.z)};
            records.insert(records.end(),
                           {static_cast<float>(samples[3 * i]), static_cast<float>(samples[3 * i + 1]),
                            static_cast<float>(samples[3 * i + 2]), n[0], n[1], n[2], n[0], n[1], n[2],
                            std::numeric_limits<float>::infinity(), power.red, power.green, power.blue});
        }
        const size_t kept = records.size() / 13;
        check(rt_store_append(tracer, records.data(), kept));
        keptTotal += kept;
    }
    return keptTotal;
}

}  // namespace

namespace Renderer
{

void setDevice(int device) { g_device = device; }

void tonemapBufferToImage(const Buffer& buffer, Image& image, double photonsEmitted, double saturationLuminance)
{
    (void)photonsEmitted;  // 1/N is applied upstream (reference src/Renderer.cpp:42)
    const size_t width = image.width(), height = image.height();
    const double invLmax = (saturationLuminance > 0.0) ? (1.0 / saturationLuminance) : 0.0;
    for (size_t y = 0; y < height; ++y)
    {
        for (size_t x = 0; x < width; ++x)
        {
            const Color color = buffer.fetchColor(x, y);
            const double exposedR = color.red * invLmax, exposedG = color.green * invLmax, exposedB = color.blue * invLmax;
            const float gammaRed = std::pow(static_cast<float>(exposedR), 1.0f / Color::gamma);
            const float gammaGreen = std::pow(static_cast<float>(exposedG), 1.0f / Color::gamma);
            const float gammaBlue = std::pow(static_cast<float>(exposedB), 1.0f / Color::gamma);
            Pixel p;
            p.red = static_cast<std::uint16_t>(std::min(static_cast<int>(gammaRed * 65535), 65535));
            p.green = static_cast<std::uint16_t>(std::min(static_cast<int>(gammaGreen * 65535), 65535));
            p.blue = static_cast<std::uint16_t>(std::min(static_cast<int>(gammaBlue * 65535), 65535));
            image.setPixel((width - 1) - x, (height - 1) - y, p);
        }
    }
}

RenderResult renderFrame(const LoadedScene& scene, ProgressCallback progress, PreviewCallback preview)
{
    const RenderSettings& settings = scene.settings;
    RenderResult result;

    std::vector<std::shared_ptr<Camera>> cameras = scene.cameras;
    if (cameras.empty() && scene.camera)
    {
        cameras.push_back(scene.camera);
    }

    // The scene as the device sees it.  Seeds, the exposure window (frameTime, shutterTime)
    // and every per-frame setting travel inside rt_settings; the device derives the
    // reference's per-camera window from them (reference src/Renderer.cpp:92-100,145-179).
    FlatScene flat;
    flattenScene(scene, flat);
    const FrameDerived derived = deriveFrame(flat.scene);

    const auto photonPassStart = std::chrono::steady_clock::now();
    TracerHandle tracer;
    check(rt_create(&flat.scene, g_device, &tracer.ptr));

    std::vector<bool> debugCamera(cameras.size(), false);
    for (size_t c = 0; c < cameras.size(); ++c)
    {
        debugCamera[c] = cameras[c] && (cameras[c]->bounceFilter() >= 0 || cameras[c]->lightFilter() >= 0);
    }

    double probeGatherMinRadius = 0.0;
    std::vector<ProbeGatherResult> probeStats(cameras.size());
    if (settings.useProbeGather)
    {
        // probe pass of every non-debug camera; the union of the record positions is the
        // keep-test index (reference src/Renderer.cpp:297-371).  Records and index stay on the
        // device; only the counters come back.
        for (size_t c = 0; c < cameras.size(); ++c)
        {
            if (!cameras[c] || debugCamera[c])
            {
                continue;
            }
            uint64_t count = 0, stats[3] = {0, 0, 0};
            check(rt_collect_gather_points(tracer.ptr, static_cast<int32_t>(c), nullptr, 0, &count, stats));
            probeStats[c].cameraRays = stats[0];
            probeStats[c].deltaExtensions = stats[1];
            probeStats[c].misses = stats[2];
        }
        probeGatherMinRadius = derived.minSplatRadius;  // reference src/Renderer.cpp:307-310
        constexpr uint64_t kSlotsPerProbe = 256;        // :382-390
        uint64_t probeCount = 0, capacity = 0;
        check(rt_set_probes_from_cameras(tracer.ptr, derived.keepRadius, derived.keepRadius, kSlotsPerProbe,
                                         settings.bounceStoreCapacity, &probeCount, &capacity));
        result.probeCount = probeCount;
        result.bounceStoreCapacity = capacity;
        check(rt_reset(tracer.ptr));
        // emitter deposits go in first so an overflow can never drop them (:394-418)
        const double emitterGatherCell =
            derived.sceneDepthFootprint > 0.0 ? derived.sceneDepthFootprint : derived.cellSize;
        result.emitterDepositsKept = depositEmitters(tracer.ptr, flat.scene, std::max(emitterGatherCell * 0.5, 1e-6));
    }
    else
    {
        check(rt_set_splat(tracer.ptr, derived.minSplatRadius, derived.cellSize, 1u << 22));
        check(rt_reset(tracer.ptr));
    }

    // ---- the photon pass (replaces the Worker threads, reference src/Renderer.cpp:421-582) ----
    rt_trace_params params{};
    params.sink_mode = settings.useProbeGather ? RT_SINK_PROBE : RT_SINK_SPLAT;
    params.first_photon = 0;
    params.photon_count = settings.photonsPerLight;
    check(rt_trace(tracer.ptr, &params));
    {
        // the reference polls its queues every 250 ms from the calling thread; here the
        // wait for the stream runs on a helper thread and the callbacks keep firing
        std::future<int32_t> done = std::async(std::launch::async, [&]() { return rt_synchronize(tracer.ptr); });
        const size_t totalWork = settings.photonsPerLight * static_cast<size_t>(flat.scene.light_count);
        while (done.wait_for(std::chrono::milliseconds(250)) != std::future_status::ready)
        {
            if (progress && !progress(totalWork))
            {
                break;  // the pass cannot be interrupted on the device; the partial-abort contract is "finish"
            }
        }
        check(done.get());
        if (progress)
        {
            progress(0);
        }
    }
    rt_counters counters{};
    check(rt_read_counters(tracer.ptr, &counters));
    result.photonPassSeconds =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - photonPassStart).count();
    result.photons = counters.photons;
    result.casts = counters.casts;
    result.bounces = counters.bounces;
    result.bounceKept = counters.kept;
    result.bounceCulled = counters.culled;
    result.bounceStoreAttempted = counters.store_attempted;
    result.bounceStoreSize = counters.store_size;
    result.bounceStoreDropped = counters.store_dropped;
    result.splatTotal = counters.splat_total;
    result.splatRadiusClamped = counters.splat_radius_clamped;
    result.splatLuminanceClamped = counters.splat_luminance_clamped;
    if (settings.useProbeGather && result.bounceStoreDropped > 0)
    {
        // reference src/Renderer.cpp:653-667: a wrong image must warn unconditionally
        std::cerr << "WARNING: BounceStore overflow — dropped " << result.bounceStoreDropped << " of "
                  << result.bounceStoreAttempted << " deposits (capacity " << result.bounceStoreCapacity
                  << "). The rendered image is missing energy; raise $bounceStoreCapacity or lower the photon budget."
                  << std::endl;
    }

    // ---- per camera: gather (probe mode) or the splat buffer, then tonemap (:670-790) ----
    const double photonsEmitted = static_cast<double>(settings.photonsPerLight);
    const double gatherCell = derived.sceneDepthFootprint > 0.0 ? derived.sceneDepthFootprint : derived.cellSize;
    for (size_t c = 0; c < cameras.size(); ++c)
    {
        const std::shared_ptr<Camera>& cam = cameras[c];
        if (!cam)
        {
            continue;
        }
        CameraRender cr;
        cr.camera = cam;
        cr.outputName = cam->outputName();
        cr.probe = probeStats[c];
        const auto gatherStart = std::chrono::steady_clock::now();
        if (settings.useProbeGather && !debugCamera[c])
        {
            uint64_t stats[4] = {0, 0, 0, 0};
            check(rt_gather(tracer.ptr, static_cast<int32_t>(c), gatherCell, probeGatherMinRadius, stats));
            cr.probe.pixelsHit = stats[0];
            cr.probe.pixelsGathered = stats[1];
            cr.probe.depositsAccum = stats[2];
        }
        cr.gatherSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - gatherStart).count();
        cr.buffer = std::make_shared<Buffer>(cam->width(), cam->height());
        check(rt_read_camera_buffer(tracer.ptr, static_cast<int32_t>(c), cr.buffer->data()));
        if (preview && c == 0)
        {
            preview(*cr.buffer, 1.0, cam->saturationLuminance());
        }
        cr.image = std::make_shared<Image>(cam->width(), cam->height());
        cr.image->clear();
        tonemapBufferToImage(*cr.buffer, *cr.image, photonsEmitted, cam->saturationLuminance());
        double sum = 0.0;
        for (size_t y = 0; y < cam->height(); ++y)
        {
            for (size_t x = 0; x < cam->width(); ++x)
            {
                const Color col = cr.buffer->fetchColor(x, y);
                sum += (static_cast<double>(col.red) + col.green + col.blue) / 3.0;
            }
        }
        const double pixels = static_cast<double>(cam->width()) * static_cast<double>(cam->height());
        cr.meanLuminance = pixels > 0.0 ? sum / pixels : 0.0;
        result.cameras.push_back(std::move(cr));
    }
    if (!result.cameras.empty())
    {
        result.buffer = result.cameras.front().buffer;
        result.image = result.cameras.front().image;
    }
    else
    {
        result.buffer = std::make_shared<Buffer>(settings.imageWidth, settings.imageHeight);
        result.image = std::make_shared<Image>(settings.imageWidth, settings.imageHeight);
    }
    return result;
}

}  // namespace Renderer
