#include "Renderer.h"

#include "Flatten.h"
#include "FrameSetup.h"
#include "RankFrames.h"
#include "rt_amd.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <functional>
#include <future>
#include <iostream>
#include <limits>
#include <stdexcept>
#include <thread>
#include <vector>

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
    TracerHandle() = default;
    TracerHandle(const TracerHandle&) = delete;
    TracerHandle& operator=(const TracerHandle&) = delete;
    ~TracerHandle()
    {
        if (ptr) rt_destroy(ptr);
    }
};

// The frame's photons over several GPUs of this machine: RT_AMD_DEVICES=<n> (at most the GPUs there
// are; default 1).  RT_AMD_COLLECTIVE=1 takes the same path — communicator, all-reduce — with one GPU.
int shardWorld()
{
    int wanted = 1;
    if (const char* v = std::getenv("RT_AMD_DEVICES")) wanted = std::atoi(v);
    return std::max(1, std::min(wanted, rt_device_count()));
}

struct Communicators
{
    std::vector<void*> comms;
    ~Communicators()
    {
        for (void* c : comms)
        {
            if (c) rt_comm_destroy(c);
        }
    }
};

// Records rank `rank` among `world` may keep of the frame's one bounce store (reference
// src/Renderer.cpp:382-390): the emitter deposits, which are rank 0's, plus an equal part of the rest,
// so that a frame whose store overflows keeps as many records on several GPUs as on one.
uint64_t storeShareOf(uint64_t capacity, uint64_t emitterRecords, int rank, int world)
{
    const uint64_t emitters = std::min(emitterRecords, capacity);
    const uint64_t rest = capacity - emitters;
    const uint64_t ranks = static_cast<uint64_t>(std::max(1, world));
    return std::max<uint64_t>(1, (rank == 0 ? emitters : 0) + (rest + ranks - 1) / ranks);
}

// photon indices [first, first + count) of rank `rank` among `world`
void shardOf(size_t perLight, int rank, int world, size_t* first, size_t* count)
{
    *first = perLight * static_cast<size_t>(rank) / static_cast<size_t>(world);
    *count = perLight * (static_cast<size_t>(rank) + 1) / static_cast<size_t>(world) - *first;
}

// RT_AMD_FAIL_RANK=<rank>: that rank's share of the frame fails once it is done (a test switch for the
// path a failing GPU takes: the frame must end with the rank's message, not hang).
void failIfAskedTo(int rank)
{
    if (const char* v = std::getenv("RT_AMD_FAIL_RANK"))
    {
        if (std::atoi(v) == rank)
        {
            throw std::runtime_error("injected failure of rank " + std::to_string(rank) + " (RT_AMD_FAIL_RANK)");
        }
    }
}

// What every rank but the first does for a frame before the ranks meet, on its own GPU and thread: the
// same probe pass and probe index (they are replicated), its share of the photons into its own store — in
// `pieces` index ranges, none started once the frame is being given up —, its own gather into its own camera
// buffers.  Emitter deposits are the first rank's alone (a gather is linear in the records, so the sum of
// the ranks' images is the image of the union of their stores).  Throws what went wrong.
void otherRankShare(TracerHandle& tracer, const rt_scene* scene, const FrameDerived& derived, int device, int rank, int world,
                    const std::vector<bool>& gathers, size_t pieces, const RankFrames& frames)
{
    check(rt_create(scene, device, &tracer.ptr));
    const bool probeGather = scene->settings.use_probe_gather != 0;
    if (probeGather)
    {
        for (size_t c = 0; c < gathers.size(); ++c)
        {
            if (!gathers[c]) continue;
            uint64_t count = 0;
            check(rt_collect_gather_points(tracer.ptr, static_cast<int32_t>(c), nullptr, 0, &count, nullptr));
        }
        uint64_t probeCount = 0, capacity = 0;
        check(rt_set_probes_from_cameras(tracer.ptr, derived.keepRadius, derived.keepRadius, 256,
                                         scene->settings.bounce_store_capacity, &probeCount, &capacity));
        // the emitter deposits are laid down once to learn their number (rank 0 keeps its own), then
        // the store is held to this rank's share
        check(rt_reset(tracer.ptr));
        const double emitterGatherCell = derived.sceneDepthFootprint > 0.0 ? derived.sceneDepthFootprint : derived.cellSize;
        uint64_t emitterKept = 0;
        check(rt_deposit_emitters(tracer.ptr, std::max(emitterGatherCell * 0.5, 1e-6), nullptr, &emitterKept));
        check(rt_reset(tracer.ptr));
        check(rt_set_store_limit(tracer.ptr, storeShareOf(capacity, emitterKept, rank, world)));
    }
    else
    {
        const uint64_t depositBound = static_cast<uint64_t>(scene->settings.photons_per_light) *
                                      static_cast<uint64_t>(std::max<int64_t>(1, scene->light_count)) *
                                      (static_cast<uint64_t>(std::max(0, scene->settings.bounce_threshold)) + 1);
        check(rt_set_splat(tracer.ptr, derived.minSplatRadius, derived.cellSize,
                           std::min<uint64_t>(1ull << 27, std::max<uint64_t>(1ull << 22, depositBound))));
    }
    check(rt_reset(tracer.ptr));
    size_t first = 0, count = 0;
    shardOf(static_cast<size_t>(scene->settings.photons_per_light), rank, world, &first, &count);
    pieces = std::max<size_t>(1, pieces);
    for (size_t piece = 0; piece < pieces && count > 0 && !frames.abandoned(); ++piece)
    {
        rt_trace_params params{};
        params.sink_mode = probeGather ? RT_SINK_PROBE : RT_SINK_SPLAT;
        params.first_photon = first + count * piece / pieces;
        params.photon_count = first + count * (piece + 1) / pieces - params.first_photon;
        if (params.photon_count == 0) continue;
        check(rt_trace(tracer.ptr, &params));
        check(rt_synchronize(tracer.ptr));
    }
    if (probeGather)
    {
        const double gatherCell = derived.sceneDepthFootprint > 0.0 ? derived.sceneDepthFootprint : derived.cellSize;
        for (size_t c = 0; c < gathers.size(); ++c)
        {
            if (!gathers[c]) continue;
            check(rt_gather(tracer.ptr, static_cast<int32_t>(c), gatherCell, derived.minSplatRadius, nullptr));
        }
    }
    failIfAskedTo(rank);
}

}  // namespace

namespace Renderer
{

void setDevice(int device) { g_device = device; }

// Rows [begin, end) of a frame, split over the host's threads (at most 16); f(firstRow, lastRow).
template <class F>
void forRowBands(size_t height, F&& f)
{
    const size_t workers = std::max<size_t>(1, std::min<size_t>({16, std::thread::hardware_concurrency(), height}));
    const size_t band = (height + workers - 1) / workers;
    std::vector<std::thread> threads;
    for (size_t w = 1; w < workers; ++w)
    {
        const size_t first = w * band, last = std::min(height, first + band);
        if (first < last)
        {
            threads.emplace_back([&f, first, last]() { f(first, last); });
        }
    }
    f(0, std::min(height, band));
    for (std::thread& t : threads)
    {
        t.join();
    }
}

void tonemapBufferToImage(const Buffer& buffer, Image& image, double photonsEmitted, double saturationLuminance)
{
    (void)photonsEmitted;  // 1/N is applied upstream (reference src/Renderer.cpp:42)
    const size_t width = image.width(), height = image.height();
    const double invLmax = (saturationLuminance > 0.0) ? (1.0 / saturationLuminance) : 0.0;
    // every pixel is independent: the reference's loop (src/Renderer.cpp:36-68), rows shared out over threads
    forRowBands(height, [&](size_t firstRow, size_t lastRow) {
        for (size_t y = firstRow; y < lastRow; ++y)
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
    });
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
    // RT_AMD_VERBOSE=1: wall time of every stage of the frame on stderr
    const bool verbose = std::getenv("RT_AMD_VERBOSE") != nullptr;
    auto stageStart = photonPassStart;
    auto stage = [&](const char* name) {
        if (verbose)
        {
            const auto now = std::chrono::steady_clock::now();
            std::fprintf(stderr, "[rt_amd] %-28s %8.1f ms\n", name,
                         std::chrono::duration<double, std::milli>(now - stageStart).count());
            stageStart = now;
        }
    };
    std::vector<bool> debugCamera(cameras.size(), false);
    for (size_t c = 0; c < cameras.size(); ++c)
    {
        debugCamera[c] = cameras[c] && (cameras[c]->bounceFilter() >= 0 || cameras[c]->lightFilter() >= 0);
    }

    // More than one GPU: this thread is rank 0 on the configured device, every other rank gets a
    // thread and a GPU of its own (devices 0 .. n-1 but the configured one last-swapped in).  Each rank
    // does its share, the ranks meet, and only if every share came through do they enter the all-reduce
    // (RankFrames.h): a rank that fails ends the frame with its message instead of hanging the others.
    const int world = shardWorld();
    const bool collective = world > 1 || (std::getenv("RT_AMD_COLLECTIVE") != nullptr && std::atoi(std::getenv("RT_AMD_COLLECTIVE")) != 0);
    Communicators communicators;
    std::vector<int32_t> devices(static_cast<size_t>(world), g_device);
    std::vector<bool> gathers(cameras.size(), false);
    if (collective)
    {
        for (int r = 0; r < world; ++r)
        {
            devices[static_cast<size_t>(r)] = r;
        }
        if (g_device > 0 && g_device < world) std::swap(devices[0], devices[static_cast<size_t>(g_device)]);
        else devices[0] = g_device;
        communicators.comms.assign(static_cast<size_t>(world), nullptr);
        check(rt_comm_init_all(world, devices.data(), communicators.comms.data()));
        for (size_t c = 0; c < cameras.size(); ++c)
        {
            gathers[c] = cameras[c] && !debugCamera[c];
        }
        if (verbose) std::fprintf(stderr, "[rt_amd] %d GPU(s), frames summed by RCCL\n", world);
    }
    RankFrames frames(world);
    std::vector<TracerHandle> rankTracers(static_cast<size_t>(world));
    TracerHandle& tracer = rankTracers[0];

    // Without a progress callback a rank's photons are one pass.  With one, they go out in up to eight
    // index ranges (the sinks add up over ranges) and the callback is asked between them, so that
    // returning false stops the frame early as it does in the reference's poll loop — on every rank; a
    // range that has been launched runs to its end.
    size_t shardFirst = 0, perLight = 0;  // this rank's share of the photons of every light
    shardOf(settings.photonsPerLight, 0, world, &shardFirst, &perLight);
    const size_t pieces = progress ? std::max<size_t>(1, std::min<size_t>(8, perLight / 1000000)) : 1;

    // what the first rank's share leaves for the rest of the frame
    double probeGatherMinRadius = 0.0;
    std::vector<ProbeGatherResult> probeStats(cameras.size());
    rt_counters counters{};
    const double gatherCell = derived.sceneDepthFootprint > 0.0 ? derived.sceneDepthFootprint : derived.cellSize;
    std::vector<std::array<uint64_t, 4>> gatherStats(cameras.size(), std::array<uint64_t, 4>{0, 0, 0, 0});
    std::vector<double> gatherSeconds(cameras.size(), 0.0);
    std::vector<bool> gathered(cameras.size(), false);

    auto firstRankShare = [&]() {
    check(rt_create(&flat.scene, g_device, &tracer.ptr));
    stage("rt_create");

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
        stage("probe pass");
        probeGatherMinRadius = derived.minSplatRadius;  // reference src/Renderer.cpp:307-310
        constexpr uint64_t kSlotsPerProbe = 256;        // :382-390
        uint64_t probeCount = 0, capacity = 0;
        check(rt_set_probes_from_cameras(tracer.ptr, derived.keepRadius, derived.keepRadius, kSlotsPerProbe,
                                         settings.bounceStoreCapacity, &probeCount, &capacity));
        result.probeCount = probeCount;
        result.bounceStoreCapacity = capacity;
        stage("probe index and store");
        check(rt_reset(tracer.ptr));
        // emitter deposits go in first so an overflow can never drop them (:394-418)
        const double emitterGatherCell =
            derived.sceneDepthFootprint > 0.0 ? derived.sceneDepthFootprint : derived.cellSize;
        uint64_t emitterSamples = 0, emitterKept = 0;
        check(rt_deposit_emitters(tracer.ptr, std::max(emitterGatherCell * 0.5, 1e-6), &emitterSamples, &emitterKept));
        result.emitterDepositsKept = emitterKept;
        if (world > 1)
        {
            check(rt_set_store_limit(tracer.ptr, storeShareOf(capacity, emitterKept, 0, world)));
        }
        stage("emitter deposits");
    }
    else
    {
        // the reference's DensityGrid grows without bound; here it is sized for the most cells the
        // pass can touch (one per deposit), between 4 Mi and 128 Mi cells
        const uint64_t depositBound = static_cast<uint64_t>(settings.photonsPerLight) *
                                      static_cast<uint64_t>(std::max<int64_t>(1, flat.scene.light_count)) *
                                      (static_cast<uint64_t>(std::max(0, flat.scene.settings.bounce_threshold)) + 1);
        const uint64_t densityCells = std::min<uint64_t>(1ull << 27, std::max<uint64_t>(1ull << 22, depositBound));
        check(rt_set_splat(tracer.ptr, derived.minSplatRadius, derived.cellSize, densityCells));
        check(rt_reset(tracer.ptr));
    }

    // ---- the photon pass (replaces the Worker threads, reference src/Renderer.cpp:421-582) ----
    const size_t totalWork = perLight * static_cast<size_t>(flat.scene.light_count);
    size_t traced = 0;
    bool goOn = true;
    for (size_t piece = 0; piece < pieces && goOn && perLight > 0 && !frames.abandoned(); ++piece)
    {
        rt_trace_params params{};
        params.sink_mode = settings.useProbeGather ? RT_SINK_PROBE : RT_SINK_SPLAT;
        params.first_photon = shardFirst + perLight * piece / pieces;
        params.photon_count = shardFirst + perLight * (piece + 1) / pieces - params.first_photon;
        if (params.photon_count == 0)
        {
            continue;
        }
        check(rt_trace(tracer.ptr, &params));
        // the reference polls its queues every 250 ms from the calling thread; here the wait for the
        // stream runs on a helper thread and the callbacks keep firing.  The library keeps its error
        // text per thread, so the helper hands it back together with the code.
        std::future<std::pair<int32_t, std::string>> done = std::async(std::launch::async, [&]() {
            const int32_t code = rt_synchronize(tracer.ptr);
            return std::make_pair(code, code == RT_OK ? std::string() : std::string(rt_last_error()));
        });
        while (done.wait_for(std::chrono::milliseconds(250)) != std::future_status::ready)
        {
            if (progress && !progress(totalWork - traced))
            {
                goOn = false;  // takes effect once this range has finished
            }
        }
        const std::pair<int32_t, std::string> waited = done.get();
        if (waited.first != RT_OK)
        {
            throw std::runtime_error("rt_amd: " + waited.second);
        }
        traced += static_cast<size_t>(params.photon_count) * static_cast<size_t>(flat.scene.light_count);
        if (progress && goOn && piece + 1 < pieces)
        {
            goOn = progress(totalWork - traced);
        }
        if (!goOn)
        {
            frames.cancel();  // the other ranks start no further range either
        }
    }
    if (progress)
    {
        progress(0);
    }
    check(rt_read_counters(tracer.ptr, &counters));
    stage("photon pass (sort + trace)");
    result.photonPassSeconds =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - photonPassStart).count();
    // Several GPUs: every rank gathers its own store into its own buffers first, then the buffers and
    // the counters are summed over the ranks; from there on this rank holds the whole frame.
    if (collective)
    {
        for (size_t c = 0; c < cameras.size(); ++c)
        {
            if (cameras[c] && settings.useProbeGather && !debugCamera[c])
            {
                const auto gatherStart = std::chrono::steady_clock::now();
                check(rt_gather(tracer.ptr, static_cast<int32_t>(c), gatherCell, probeGatherMinRadius, gatherStats[c].data()));
                gatherSeconds[c] = std::chrono::duration<double>(std::chrono::steady_clock::now() - gatherStart).count();
                gathered[c] = true;
            }
        }
        stage("gather");
        failIfAskedTo(0);
    }
    };

    if (collective)
    {
        const std::string failure = frames.run(
            [&](int rank) {
                if (rank == 0) firstRankShare();
                else otherRankShare(rankTracers[static_cast<size_t>(rank)], &flat.scene, derived, devices[static_cast<size_t>(rank)], rank,
                                    world, gathers, pieces, frames);
            },
            [&](int rank) {
                check(rt_allreduce(rankTracers[static_cast<size_t>(rank)].ptr, communicators.comms[static_cast<size_t>(rank)],
                                   rank == 0 ? &counters : nullptr));
            });
        if (!failure.empty())
        {
            throw std::runtime_error(failure);
        }
        stage("all-reduce");
    }
    else
    {
        firstRankShare();
    }
    if (counters.density_dropped > 0)
    {
        std::cout << "Density grid: " << counters.density_dropped
                  << " deposits not recorded (grid full or cell out of range); the image is not affected" << std::endl;
    }
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
            if (!gathered[c])
            {
                check(rt_gather(tracer.ptr, static_cast<int32_t>(c), gatherCell, probeGatherMinRadius, gatherStats[c].data()));
            }
            cr.probe.pixelsHit = gatherStats[c][0];
            cr.probe.pixelsGathered = gatherStats[c][1];
            cr.probe.depositsAccum = gatherStats[c][2];
        }
        cr.gatherSeconds = gatherSeconds[c] + std::chrono::duration<double>(std::chrono::steady_clock::now() - gatherStart).count();
        stage("gather");
        cr.buffer = std::make_shared<Buffer>(cam->width(), cam->height());
        check(rt_read_camera_buffer(tracer.ptr, static_cast<int32_t>(c), cr.buffer->data()));
        stage("camera buffer to the host");
        if (preview && c == 0)
        {
            preview(*cr.buffer, 1.0, cam->saturationLuminance());
        }
        cr.image = std::make_shared<Image>(cam->width(), cam->height());
        cr.image->clear();
        tonemapBufferToImage(*cr.buffer, *cr.image, photonsEmitted, cam->saturationLuminance());
        // row sums in parallel, added in row order: the same value whatever the thread count
        std::vector<double> rowSums(cam->height(), 0.0);
        forRowBands(cam->height(), [&](size_t firstRow, size_t lastRow) {
            for (size_t y = firstRow; y < lastRow; ++y)
            {
                double rowSum = 0.0;
                for (size_t x = 0; x < cam->width(); ++x)
                {
                    const Color col = cr.buffer->fetchColor(x, y);
                    rowSum += (static_cast<double>(col.red) + col.green + col.blue) / 3.0;
                }
                rowSums[y] = rowSum;
            }
        });
        double sum = 0.0;
        for (const double rowSum : rowSums)
        {
            sum += rowSum;
        }
        const double pixels = static_cast<double>(cam->width()) * static_cast<double>(cam->height());
        cr.meanLuminance = pixels > 0.0 ? sum / pixels : 0.0;
        stage("tonemap and mean luminance");
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
